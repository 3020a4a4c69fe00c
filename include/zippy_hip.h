/*
 * zippy_hip.h -- C ABI of the MI355X-native batched DEFLATE engine.
 *
 * This is the drop-in boundary for guzba/zippy's compress()/uncompress() path.
 * zippy has no FFI/plugin interface of its own: the boundary is its exported
 * Nim procs, so every entry point below cites the Nim proc it stands behind
 * (file:line under the reference tree).  A Nim shim that binds these symbols
 * and re-exposes zippy's exact signatures is in INTEGRATION.md.
 *
 * Plain C: pointers, sizes, status codes.  No torch types, no C++ types.
 * All work runs on one GPU through hand-written gfx950 kernels; there is no
 * CPU fallback (calls fail with ZH_ERR_DEVICE if no GPU is usable).
 */
#ifndef ZIPPY_HIP_H
#define ZIPPY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CompressedDataFormat, src/zippy/common.nim:4-5 (same ordinals) */
enum { ZH_DF_DETECT = 0, ZH_DF_ZLIB = 1, ZH_DF_GZIP = 2, ZH_DF_DEFLATE = 3 };

/* Levels, src/zippy/common.nim:7-12; valid range -2..9 (deflate.nim:208-209) */
enum {
  ZH_NO_COMPRESSION = 0,
  ZH_BEST_SPEED = 1,
  ZH_BEST_COMPRESSION = 9,
  ZH_DEFAULT_COMPRESSION = -1,
  ZH_HUFFMAN_ONLY = -2
};

/* Status codes.  1..18 map one-to-one onto the ZippyError raise sites of the
 * reference; zh_strerror() returns the reference's message. */
enum {
  ZH_OK = 0,
  ZH_ERR_INVALID_LEVEL = 1,      /* deflate.nim:208-209 */
  ZH_ERR_INVALID_FORMAT = 2,     /* zippy.nim:83-84 */
  ZH_ERR_DETECT = 3,             /* zippy.nim:125 */
  ZH_ERR_UNSUPPORTED_METHOD = 4, /* zippy.nim:141, gzip.nim:26 */
  ZH_ERR_COMPRESSION_INFO = 5,   /* zippy.nim:144 */
  ZH_ERR_INVALID_HEADER = 6,     /* zippy.nim:147 */
  ZH_ERR_PRESET_DICT = 7,        /* zippy.nim:150 */
  ZH_ERR_CHECKSUM = 8,           /* zippy.nim:162, gzip.nim:81 */
  ZH_ERR_SIZE = 9,               /* gzip.nim:85,88 */
  ZH_ERR_GZIP_ID = 10,           /* gzip.nim:23 */
  ZH_ERR_RESERVED_FLAGS = 11,    /* gzip.nim:29 */
  ZH_ERR_UNSUPPORTED_FLAGS = 12, /* gzip.nim:41 */
  ZH_ERR_INVALID_BUFFER = 13,    /* internal.nim:191-192 */
  ZH_ERR_COMPRESS_INTERNAL = 14, /* internal.nim:194-195 */
  ZH_ERR_END_OF_BUFFER = 15,     /* bitstreams.nim:16-17 */
  ZH_ERR_BYTE_BOUNDARY = 16,     /* bitstreams.nim:66,113 */
  ZH_ERR_BLOCK_HEADER = 17,      /* inflate.nim:289 */
  ZH_ERR_INVALID_SYMBOL = 18,    /* inflate.nim:165 */
  ZH_ERR_NOMEM = 19,             /* host or device allocation failed */
  ZH_ERR_DEVICE = 20,            /* no usable GPU / HIP runtime error (zh_last_error) */
  ZH_ERR_DST_TOO_SMALL = 21,     /* device API: output slot capacity exceeded */
  ZH_ERR_ARGUMENT = 22,          /* NULL pointer, bad plan, ... */
  /* archive layer (zh_zip_*): the ZippyError raise sites of ziparchives.nim */
  ZH_ERR_ARCHIVE_EOF = 23,        /* internal.nim:197-198 failArchiveEOF */
  ZH_ERR_ZIP_FILE_HEADER = 24,    /* ziparchives.nim:58-59 */
  ZH_ERR_ZIP_METHOD = 25,         /* ziparchives.nim:87-88,296-297 */
  ZH_ERR_ZIP_NO_RECORD = 26,      /* ziparchives.nim:43-52,89-90 */
  ZH_ERR_ZIP_CRC = 27,            /* ziparchives.nim:91-92 */
  ZH_ERR_ZIP_UNSUPPORTED = 28,    /* ziparchives.nim:214-218,248-255 disk / record numbers */
  ZH_ERR_ZIP_CENTRAL_HEADER = 29, /* ziparchives.nim:224-225,279-280 */
  ZH_ERR_ZIP_DISK_NUMBER = 30,    /* ziparchives.nim:299-300 */
  ZH_ERR_ZIP_DUPLICATE = 31,      /* ziparchives.nim:314-315 */
  ZH_ERR_ZIP_CENTRAL_SIZE = 32,   /* ziparchives.nim:343-344 */
  ZH_ERR_ZIP_NAME = 33,           /* ziparchives.nim:506-511 empty / absolute / over-long path */
  ZH_ERR_TAR_HEADER_TYPE = 34,    /* tarballs.nim:119 */
  ZH_ERR_UNSAFE_PATH = 35,        /* internal.nim:294-302 verifyPathIsSafeToExtract */
  ZH_ERR_TAR_NUMBER = 36          /* tarballs.nim:17-23 (octal field that is not octal) */
};

/* Engine context: one GPU, one HIP stream, reusable scratch. Thread-compatible
 * (one thread at a time per context; separate contexts are independent), like
 * the reference's re-entrant pure procs (SURVEY.md 8b "Threading"). */
typedef struct zh_ctx zh_ctx;

/* device < 0: current device.  stream: a hipStream_t to enqueue on, or NULL to
 * let the context create its own. */
int zh_create(int device, void *stream, zh_ctx **out);
/* The chain levels' (-1, 2..9) link kernels take the ORDER of their results from a property of the LDS
 * unit -- the lanes of one returning atomic are served in ascending lane order -- which the ISA manual does
 * not promise (lz77.nim:69-71's `chain[windowPos] = head[hash]; head[hash] = windowPos`, 64 positions a
 * step).  zh_create asks the device itself, once per device and process (a known-answer launch on the
 * context's stream, which it waits for); a device that answers otherwise gets the in-order link kernels
 * -- same bytes, slower -- and zh_last_error says so.  -> 1: the device passed, 0: the in-order kernels run
 * (also under ZH_CHAIN_PREV=serial). */
int zh_chain_links_parallel(zh_ctx *ctx);
void zh_destroy(zh_ctx *ctx);
const char *zh_strerror(int status);
const char *zh_last_error(zh_ctx *ctx); /* detail of the last ZH_ERR_DEVICE */
void *zh_stream(zh_ctx *ctx);           /* the hipStream_t work is enqueued on */

/* gzip FNAME length: the reference inserts 0..25 letters chosen at random per
 * call (zippy.nim:26-42).  k < 0 (default): random per buffer like the
 * reference; 0..25: fixed (deterministic output for tests). */
void zh_set_gzip_fname_len(zh_ctx *ctx, int k);

/* Which inflate kernels later uncompress calls of this context use (no reference counterpart;
 * same results either way): 0 = a stream's Huffman codes decoded in parallel, then one writer
 * per stream (csrc/zh_inflate_split.hip); 1 = the two-wave serial decoder (csrc/zh_inflate.hip);
 * < 0 = the default (0, or the ZH_INFLATE=serial environment variable). */
void zh_set_inflate_mode(zh_ctx *ctx, int mode);

/* Which BestSpeed (level 1) match finder later compress calls of this context use.
 * 0 = the reference's parse (snappy.nim:12-136 replayed decision for decision: the streams are
 * byte-identical to zippy's own); 1 = the parallel parse of csrc/zh_l1p_match.hip: every position
 * enters the hash table and the greedy selection runs chunk-parallel, so the token stream differs
 * from zippy's while meeting the encoder contract -- a valid RFC 1951/1950/1952 stream that zippy's
 * uncompress() decodes to the input bit for bit, compressed size within 2 % of zippy's at level 1
 * (tests/test_gpu_parity.py: test_gpu_parallel_parse_*).  < 0 = the default (0, or the
 * ZH_L1_PARSE=parallel environment variable).  All other levels are unaffected. */
void zh_set_l1_parse(zh_ctx *ctx, int mode);

/* Host-buffer compress calls of at least min_batch_bytes of input run as pipelined groups of
 * about group_bytes each: one group's kernels overlap the neighbours' transfers (no reference
 * counterpart; the results are the same bytes either way).  0 = default (1 GiB / 512 MiB, or the
 * ZH_PIPE_MIN / ZH_PIPE_GROUP environment variables). */
void zh_set_host_pipeline(zh_ctx *ctx, size_t min_batch_bytes, size_t group_bytes);

/* Upper bound of compress() output for len input bytes: stored form
 * len + 5*ceil(len/65535) (deflate.nim:179-205) + container + slack. */
size_t zh_compress_bound(size_t len, int data_format);

/* ------------------------------------------------------------------ *
 * Host-buffer API: what the Nim shim binds.                           *
 * Inputs are borrowed read-only for the call; outputs are freshly     *
 * allocated by the library (release with zh_free) -- the ownership    *
 * model of `compress*(...): string` (zippy.nim:11-16).               *
 * ------------------------------------------------------------------ */

/* compress*(src: pointer, len, level, dataFormat): string -- zippy.nim:11-84,
 * for n independent buffers at once.  statuses[i] is per buffer; the return
 * value is ZH_OK unless the call as a whole could not run. */
int zh_compress_batch(zh_ctx *ctx, const void *const *srcs, const size_t *lens, size_t n,
                      int level, int data_format, void **dsts, size_t *dst_lens,
                      int32_t *statuses);

/* uncompress*(src: pointer, len, dataFormat): string -- zippy.nim:100-165,
 * gzip.nim:3-88, for n independent streams at once.  A bad stream only fails
 * its own slot. */
int zh_uncompress_batch(zh_ctx *ctx, const void *const *srcs, const size_t *lens, size_t n,
                        int data_format, void **dsts, size_t *dst_lens, int32_t *statuses);

/* The two calls above with the results in buffers of the CALLER'S: dsts[i] (caps[i] bytes) on
 * entry.  No reference counterpart -- zippy returns fresh strings (zippy.nim:11-18,100-104) -- but
 * what a binding that owns its strings wants: the shim allocates `newString(zh_compress_bound(n))`
 * (or a string of ISIZE bytes), the library fills it, and the only copy left is the one from the
 * pinned staging chunk.  A result that does not fit: statuses[i] = ZH_ERR_DST_TOO_SMALL and
 * dst_lens[i] = the size it needs.  On return dsts[i] is the caller's pointer for every buffer that
 * was filled and NULL otherwise; nothing here goes to zh_free.  For streams without a size field
 * (zlib, raw deflate) caps[i] is also how much is decoded at most before the expansion bound is
 * tried. */
int zh_compress_batch_into(zh_ctx *ctx, const void *const *srcs, const size_t *lens, size_t n,
                           int level, int data_format, void **dsts, const size_t *caps,
                           size_t *dst_lens, int32_t *statuses);
int zh_uncompress_batch_into(zh_ctx *ctx, const void *const *srcs, const size_t *lens, size_t n,
                             int data_format, void **dsts, const size_t *caps, size_t *dst_lens,
                             int32_t *statuses);

/* One batch over several GPUs of a node.  zippy's compress()/uncompress() are pure functions of
 * one buffer (zippy.nim:11-16,100-104), so a batch shards by contiguous index ranges with no
 * exchange step: context r (one per device, made with zh_create(r, NULL, &ctx[r])) takes range r
 * of the batch -- the first n % n_ctx ranges hold one buffer more -- on its own host thread, and
 * every result lands in the caller's arrays at its own index.  Same results as the
 * single-context calls.  Contexts must be distinct; two contexts on ONE device are allowed
 * (they share the GPU).  Returns the first failing shard's call status, else ZH_OK. */
int zh_device_count(void);
int zh_compress_batch_multi(zh_ctx *const *ctxs, size_t n_ctx, const void *const *srcs,
                            const size_t *lens, size_t n, int level, int data_format, void **dsts,
                            size_t *dst_lens, int32_t *statuses);
int zh_uncompress_batch_multi(zh_ctx *const *ctxs, size_t n_ctx, const void *const *srcs,
                              const size_t *lens, size_t n, int data_format, void **dsts,
                              size_t *dst_lens, int32_t *statuses);

/* Single-buffer forms (batch of 1); return the buffer's status. */
int zh_compress(zh_ctx *ctx, const void *src, size_t len, int level, int data_format,
                void **dst, size_t *dst_len);
int zh_uncompress(zh_ctx *ctx, const void *src, size_t len, int data_format, void **dst,
                  size_t *dst_len);

/* crc32*(src: pointer, len): uint32 -- crc.nim:53-72 ; adler32* -- adler32.nim:6 */
int zh_crc32(zh_ctx *ctx, const void *src, size_t len, uint32_t *out);
int zh_adler32(zh_ctx *ctx, const void *src, size_t len, uint32_t *out);

/* n checksums in one launch pair: crc32* applied to every buffer (crc.nim:53-72). */
int zh_crc32_batch(zh_ctx *ctx, const void *const *srcs, const size_t *lens, size_t n, uint32_t *out);

/* zh_compress_batch that also returns crc32(srcs[i]) -- what createZipArchive needs per entry
 * (ziparchives.nim:526-530: crc32(contents); compress(contents, BestSpeed, dfDeflate)). */
int zh_compress_batch_crc32(zh_ctx *ctx, const void *const *srcs, const size_t *lens, size_t n,
                            int level, int data_format, void **dsts, size_t *dst_lens,
                            int32_t *statuses, uint32_t *crcs);
/* zh_uncompress_batch for callers that know the output sizes up front (ZIP central directory,
 * ziparchives.nim:85-93; gzip.nim:72-76 trustSize): size_hints[i] replaces the guess (4x the
 * stream, then a sizing pass if that was too little) of zlib/raw streams (a wrong hint only costs a retry), crcs[i] (optional) = crc32 of output i. */
int zh_uncompress_batch_sized(zh_ctx *ctx, const void *const *srcs, const size_t *lens, size_t n,
                              int data_format, const uint64_t *size_hints, void **dsts,
                              size_t *dst_lens, int32_t *statuses, uint32_t *crcs);

void zh_free(void *p);

/* ------------------------------------------------------------------ *
 * Device-resident API: buffers already in HBM (pipelines, bench.py).  *
 * A plan owns the device-side descriptors and scratch for one batch   *
 * geometry; running it only enqueues kernels on the context's stream  *
 * (no host synchronisation), so it can be timed with HIP events or    *
 * captured in a hipGraph.                                             *
 * ------------------------------------------------------------------ */
typedef struct zh_plan zh_plan;

/* Device memory for a caller without a HIP binding of its own (a Nim / cgo / JNI shim that drives the plans; callers
 * that have one -- hipMalloc, a torch tensor's data_ptr -- pass their own pointers): blocks of the context's device,
 * copies on the context's stream and waited for.  zh_device_free waits for the stream first. */
int zh_device_malloc(zh_ctx *ctx, size_t bytes, void **d_out);
void zh_device_free(zh_ctx *ctx, void *d);
int zh_device_upload(zh_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int zh_device_download(zh_ctx *ctx, void *dst, const void *d_src, size_t bytes);

/* Buffer i is d_src[src_off[i] .. +src_len[i]); its output slot is
 * d_dst[dst_off[i] .. +dst_cap[i]).  Offsets are in bytes. */
int zh_plan_compress(zh_ctx *ctx, size_t n, const uint64_t *src_off, const uint64_t *src_len,
                     const uint64_t *dst_off, const uint64_t *dst_cap, int level,
                     int data_format, zh_plan **out);
/* An uncompress plan of a handful of streams (its longest stream weighs more than 1/40 of the batch
 * + 8 MiB) decodes every stream of >= 128 KiB with many workgroups (block starts searched for,
 * decoded from at once, proved afterwards; same bytes and statuses as one decoder per stream, which
 * remains the fallback stream by stream).  Its scratch -- about 45 bytes per compressed byte --
 * is allocated at the plan's first run; ZH_SEG=0 in the environment turns this off. */
int zh_plan_uncompress(zh_ctx *ctx, size_t n, const uint64_t *src_off, const uint64_t *src_len,
                       const uint64_t *dst_off, const uint64_t *dst_cap, int data_format,
                       zh_plan **out);
/* Enqueue the plan. d_src/d_dst are device pointers.
 * Preconditions and side effects (compress plans):
 *  - d_dst must be 4-byte aligned (the encoder addresses the output as 32-bit words; slot
 *    offsets themselves may be any byte); a misaligned d_dst returns ZH_ERR_ARGUMENT;
 *  - nothing is cleared beforehand and nothing but a stream's own bytes is written: of slot i only
 *    [dst_off, dst_off + out_len) changes -- whatever the slot held before --, the rest of the slot and
 *    the bytes of d_dst between slots keep their contents (until round 6 the whole slot was zeroed first);
 *  - slots must not overlap.
 * Both directions read whole aligned 32-bit words around a source buffer, i.e. up to 3 bytes
 * before src_off and after src_off + src_len: those bytes must be mapped (true inside any
 * hipMalloc allocation, which is 256-byte aligned and padded); their values are ignored.
 * Uncompress plans: on ZH_ERR_DST_TOO_SMALL out_len is the number of leading bytes of the slot
 * that hold valid output (out_len <= dst_cap; 0 for a large stream decoded segment-wise, which
 * writes nothing once it knows the slot is too small), not the required size; callers that need
 * the size use the host-buffer calls (which size and retry) or a larger slot. */
int zh_plan_run(zh_plan *plan, const void *d_src, void *d_dst);
/* Wait for the stream and fetch per-buffer output lengths and statuses. */
int zh_plan_results(zh_plan *plan, uint64_t *out_lens, int32_t *statuses);
/* Device arrays of the same (uint64 lens[n], int32 statuses[n]); valid after run. */
const uint64_t *zh_plan_device_lens(zh_plan *plan);
const int32_t *zh_plan_device_statuses(zh_plan *plan);
/* Re-point an uncompress plan at new per-stream compressed lengths (same
 * offsets/capacities), e.g. after a compress plan produced them on device. */
int zh_plan_set_src_lens_device(zh_plan *plan, const uint64_t *d_lens);
/* Whole buffers on their way between GPUs (zippy.nim:11-18: a buffer is compressed / uncompressed by itself, so a
 * batch is sharded by handing whole buffers around and nothing else ever travels).  Output slots have the worst-case
 * size; what goes over a link is the streams alone:
 *  - zh_plan_pack, after zh_plan_run: result i of the plan (d_slots + dst_off[i], zh_plan_device_lens()[i] bytes; 0
 *    bytes where the status is not ZH_OK) is copied to d_packed + d_offsets[i], back to back: d_offsets[0] = 0,
 *    d_offsets[i + 1] = d_offsets[i] + length i -- n + 1 device uint64, written by the call.  Nothing is written at
 *    or beyond d_packed + packed_cap (a caller that sized d_packed too small sees d_offsets[n] > packed_cap).
 *  - zh_plan_unpack, before zh_plan_run of an UNCOMPRESS plan: stream i (d_packed + d_offsets[i], d_offsets[i + 1] -
 *    d_offsets[i] bytes, at most the src_len[i] the plan was made with: the slot's size) is copied to d_slots +
 *    src_off[i], and the plan decodes streams of these lengths from now on (as after zh_plan_set_src_lens_device --
 *    and, like there, for good: the plan's segment geometry was laid over the lengths it was made with, so from the
 *    first unpack on it decodes every stream with one workgroup, large ones included; make a new plan to get the
 *    segment-wise decode of large streams back).  ZH_ERR_ARGUMENT (more than 2^31 copy workgroups) is returned before
 *    anything is launched or written, for both calls.
 * Device pointers throughout, kernel launches on the context's stream only, no host synchronisation: a pack may
 * follow a run, a run an unpack, at once. */
int zh_plan_pack(zh_plan *plan, const void *d_slots, void *d_packed, uint64_t packed_cap, uint64_t *d_offsets);
int zh_plan_unpack(zh_plan *plan, const void *d_packed, const uint64_t *d_offsets, void *d_slots);
void zh_plan_destroy(zh_plan *plan);

/* CRC-32 of the uncompressed side of every buffer (sources of a compress plan, outputs of an
 * uncompress plan) whatever the container: request before zh_plan_run, read after it. */
int zh_plan_request_crc32(zh_plan *plan, int on);
int zh_plan_crc32(zh_plan *plan, uint32_t *crcs);

/* Per-kernel timing of the LAST zh_plan_run when profiling is on (HIP events on
 * the context's stream around every launch).  names[i] are static strings. */
void zh_plan_set_profiling(zh_plan *plan, int on);
int zh_plan_kernel_times(zh_plan *plan, const char **names, float *ms, int max_entries);

/* ------------------------------------------------------------------ *
 * Block-parallel form of ONE large buffer (BASELINE.json config 5).   *
 * The reference cuts a buffer into deflate blocks of 4 MiB            *
 * (deflate.nim:228, internal.nim:16) and calls the matcher once per   *
 * block (deflate.nim:243-272), so no match ever reaches across a      *
 * block start: a block can be decoded by itself once its bit position *
 * is known.  These entry points expose that: the same encoder with    *
 * the block size as a parameter (block_bytes: a multiple of 32768,    *
 * 32768 .. 4194304; 4194304 reproduces zh_compress byte for byte),    *
 * and the index of block starts that lets the decoder run one wave    *
 * pair per block instead of one per stream.  The stream stays plain   *
 * RFC 1951/1950/1952: zippy's uncompress() (and zh_uncompress) decode *
 * it without the index.                                               *
 * ------------------------------------------------------------------ */
typedef struct zh_block_entry {
  uint64_t bit_off; /* block's BFINAL bit, in bits from the start of the compressed buffer */
  uint64_t out_off; /* offset of the block's first byte in the uncompressed data */
} zh_block_entry;

/* index: library-allocated (zh_free), n_entries = deflate blocks + 1; the closing entry holds
 * the bit just past the last block and the uncompressed length.  A stored logical block longer
 * than 65535 bytes contributes one entry per stored chunk (deflate.nim:179-205). */
int zh_compress_blocks(zh_ctx *ctx, const void *src, size_t len, int level, int data_format,
                       size_t block_bytes, void **dst, size_t *dst_len, zh_block_entry **index,
                       size_t *n_entries);
/* Decode with one decoder per index entry.  Fails with ZH_ERR_INVALID_BUFFER when a block does
 * not produce exactly the bytes its entry promises (or reaches back before its own start);
 * container checks and checksum as in zh_uncompress. */
int zh_uncompress_indexed(zh_ctx *ctx, const void *src, size_t len, int data_format,
                          const zh_block_entry *index, size_t n_entries, void **dst,
                          size_t *dst_len);

/* Device-resident forms.  zh_plan_compress_blocks = zh_plan_compress with the block size;
 * after zh_plan_run + zh_plan_results, zh_plan_block_index returns buffer `buf`'s index
 * (library-allocated, zh_free).  zh_plan_uncompress_indexed plans ONE stream
 * d_src[src_off .. +src_len) -> d_dst[dst_off .. +dst_cap) with its index (host array). */
int zh_plan_compress_blocks(zh_ctx *ctx, size_t n, const uint64_t *src_off, const uint64_t *src_len,
                            const uint64_t *dst_off, const uint64_t *dst_cap, int level,
                            int data_format, size_t block_bytes, zh_plan **out);
int zh_plan_block_index(zh_plan *plan, size_t buf, zh_block_entry **index, size_t *n_entries);
int zh_plan_uncompress_indexed(zh_ctx *ctx, uint64_t src_off, uint64_t src_len, uint64_t dst_off,
                               uint64_t dst_cap, int data_format, const zh_block_entry *index,
                               size_t n_entries, zh_plan **out);

/* ------------------------------------------------------------------ *
 * ZIP archives as batch clients of the codec. *
 * Record parsing / assembly of src/zippy/ziparchives.nim on the host, *
 * every entry's deflate stream and CRC-32 in ONE GPU batch.  The file *
 * system side of extractAll (ziparchives.nim:374-453) stays with the  *
 * caller.                                                             *
 * ------------------------------------------------------------------ */
typedef struct zh_zip_reader zh_zip_reader;
typedef struct zh_zip_entry {
  const char *path;           /* UTF-8, not NUL-terminated; CP437 names converted (ziparchives.nim:108-160) */
  size_t path_len;
  int is_directory;           /* ziparchives.nim:352-359 */
  uint64_t header_offset;     /* of the local file header in the image */
  uint64_t compressed_size, uncompressed_size;
  uint32_t crc32;
  uint32_t unix_mode;         /* external attributes >> 16 (parseFilePermissions' input) */
} zh_zip_entry;

/* openZipArchive(zipPath) -- ziparchives.nim:183-372 -- on a memory image, which stays borrowed
 * until zh_zip_close.  Entries keep central-directory order. */
int zh_zip_open(const void *archive, size_t len, zh_zip_reader **out);
void zh_zip_close(zh_zip_reader *reader);
size_t zh_zip_num_entries(const zh_zip_reader *reader);
int zh_zip_entry_at(const zh_zip_reader *reader, size_t i, zh_zip_entry *out);
int zh_zip_find(const zh_zip_reader *reader, const char *path, size_t path_len, size_t *index);
/* extractFile(reader, path) -- ziparchives.nim:39-93 -- for n records at once; statuses[k] is
 * the outcome of record indices[k], dsts[k] library-allocated (zh_free). */
int zh_zip_extract_batch(zh_ctx *ctx, const zh_zip_reader *reader, const size_t *indices, size_t n,
                         void **dsts, size_t *dst_lens, int32_t *statuses);
/* createZipArchive(entries: OrderedTable[string, string]) -- ziparchives.nim:455-634.  Entries in
 * insertion order (the archive lists them last to first, as the reference does); dos_time /
 * dos_date = msdos(getTime()) (ziparchives.nim:475-493), taken from the caller so that the
 * call is a pure function. */
int zh_zip_create(zh_ctx *ctx, const char *const *paths, const size_t *path_lens,
                  const void *const *contents, const size_t *content_lens, size_t n,
                  uint16_t dos_time, uint16_t dos_date, void **archive, size_t *archive_len);

/* ------------------------------------------------------------------ *
 * Tarballs: extractAll of src/zippy/tarballs.nim *
 * without its file-system half.  A .tar.gz is ONE foreign gzip member *
 * -- one decoder, no parallelism to offer; the GPU still does the     *
 * inflate + CRC-32, with ISIZE as the output size (trustSize).        *
 * ------------------------------------------------------------------ */
typedef struct zh_tar_reader zh_tar_reader;
typedef struct zh_tar_entry {
  const char *path;      /* prefix / name, or the preceding 'L' block's long name */
  size_t path_len;
  const char *linkname;  /* symlinks (typeflag '2') */
  size_t linkname_len;
  char typeflag;         /* '0' or '\0' file, '5' directory, '2' symlink */
  uint32_t mode;
  int64_t mtime;
  uint64_t offset, size; /* the entry's bytes inside zh_tar_data() */
} zh_tar_entry;

/* image: the bytes of a .tar.gz (decoded here; ctx required) or of a .tar (borrowed until close;
 * ctx may be NULL).  Header walk and checks: tarballs.nim:61-124. */
int zh_tar_open(zh_ctx *ctx, const void *image, size_t len, zh_tar_reader **out);
void zh_tar_close(zh_tar_reader *reader);
size_t zh_tar_num_entries(const zh_tar_reader *reader);
int zh_tar_entry_at(const zh_tar_reader *reader, size_t i, zh_tar_entry *out);
const void *zh_tar_data(const zh_tar_reader *reader, size_t *len);

/* ------------------------------------------------------------------ *
 * Introspection for parity tests (not part of the drop-in surface).   *
 * ------------------------------------------------------------------ */
/* Level-1 parse of one buffer as the u16 token stream of SURVEY.md 8a row a4
 * (snappy.nim:33-64 format), block by block, fragment by fragment: lets tests
 * compare the device matcher with the oracle token-for-token.  tokens is
 * library-allocated (zh_free). */
int zh_debug_tokens(zh_ctx *ctx, const void *src, size_t len, int level, uint16_t **tokens,
                    size_t *num_tokens);
/* Debug hook: ONE prefix code from a histogram of num_freq <= 288 symbols (deflate.nim:13-151 huffmanCodes:
 * min_codes as the reference's minCodes, limit <= 15 bits).  contract 0: the replay of the reference (its heap
 * order, its length limiting) -- the tests hold it against the oracle symbol for symbol; 1: contract mode's
 * builder (zh_set_l1_parse(ctx, 1)) -- other tie-breaks; optimal unless the length limit binds (then repaired the
 * way zlib / miniz do: valid, not always the minimum).  codes (bit-reversed, as they
 * go into the stream) and lens hold num_freq + 2 entries; *num_codes is the reference's numCodes (min_codes <=
 * num_freq, as in every call of the reference's: anything else is ZH_ERR_ARGUMENT). */
int zh_debug_huffman(zh_ctx *ctx, const uint32_t *freq, int num_freq, int min_codes, int limit, int contract,
                     uint16_t *codes, uint8_t *lens, int *num_codes);
/* Large streams are decoded by many workgroups each (segment-wise) when the chain of their segments holds, by one
 * workgroup otherwise -- same bytes and statuses either way, so only a count can tell the two apart: since the
 * context was made, *cut = streams that uncompress runs cut into segments, *held = those of them whose chain held.
 * Counted on the device by EVERY run of a segmented plan -- a sizing (count-only) pass and a re-run of the same plan
 * count again --, so compare differences around the runs of interest.  A stream that is damaged, or has fewer than
 * four block starts and sub-starts, legitimately does not hold. */
int zh_debug_segment_stats(zh_ctx *ctx, uint64_t *cut, uint64_t *held);

#ifdef __cplusplus
}
#endif
#endif
