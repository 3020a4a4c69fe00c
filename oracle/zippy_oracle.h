/*
 * zippy_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11) of guzba/zippy v0.10.18's codec hot path, used as
 * the parity checker for the MI355X HIP engine in this repository.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (zippy_amd/, include/zippy_hip.h) never does.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Nim is not available in the build container, so the
 * reference itself cannot be compiled; the restatement is pinned by
 *   - the reference's own decode fixtures (tests/test.nim:41-60,
 *     tests/test_known_bad.nim:3 -> tests/golden/ in this repo),
 *   - round trips at all levels/formats (tests/test.nim:62-85,
 *     tests/test_levels.nim:18-25),
 *   - cross-decoding both ways against system zlib (tests/validate.nim).
 * Compressed BYTES are not pinned by the reference (no encode goldens; Huffman
 * tie-breaking lives in Nim's std/heapqueue, which is a port of CPython's
 * heapq and is restated here as such) -- see DESIGN.md "parity status".
 */
#ifndef ZIPPY_ORACLE_H
#define ZIPPY_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/zippy/common.nim:4-12 */
enum { ZO_DF_DETECT = 0, ZO_DF_ZLIB = 1, ZO_DF_GZIP = 2, ZO_DF_DEFLATE = 3 };
enum {
  ZO_NO_COMPRESSION = 0,
  ZO_BEST_SPEED = 1,
  ZO_BEST_COMPRESSION = 9,
  ZO_DEFAULT_COMPRESSION = -1,
  ZO_HUFFMAN_ONLY = -2
};

/* Status codes: one per ZippyError raise site category. 0 = ok. */
enum {
  ZO_OK = 0,
  ZO_ERR_INVALID_LEVEL = 1,        /* deflate.nim:208-209 */
  ZO_ERR_INVALID_FORMAT = 2,       /* zippy.nim:83-84 */
  ZO_ERR_DETECT = 3,               /* zippy.nim:125 */
  ZO_ERR_UNSUPPORTED_METHOD = 4,   /* zippy.nim:141, gzip.nim:26 */
  ZO_ERR_COMPRESSION_INFO = 5,     /* zippy.nim:144 */
  ZO_ERR_INVALID_HEADER = 6,       /* zippy.nim:147 */
  ZO_ERR_PRESET_DICT = 7,          /* zippy.nim:150 */
  ZO_ERR_CHECKSUM = 8,             /* zippy.nim:162, gzip.nim:81 */
  ZO_ERR_SIZE = 9,                 /* gzip.nim:85,88 */
  ZO_ERR_GZIP_ID = 10,             /* gzip.nim:23 */
  ZO_ERR_RESERVED_FLAGS = 11,      /* gzip.nim:29 */
  ZO_ERR_UNSUPPORTED_FLAGS = 12,   /* gzip.nim:41 */
  ZO_ERR_INVALID_BUFFER = 13,      /* internal.nim:191-192 failUncompress */
  ZO_ERR_COMPRESS_INTERNAL = 14,   /* internal.nim:194-195 failCompress */
  ZO_ERR_END_OF_BUFFER = 15,       /* bitstreams.nim:16-17 */
  ZO_ERR_BYTE_BOUNDARY = 16,       /* bitstreams.nim:66,113 */
  ZO_ERR_BLOCK_HEADER = 17,        /* inflate.nim:289 */
  ZO_ERR_INVALID_SYMBOL = 18,      /* inflate.nim:165 */
  ZO_ERR_NOMEM = 19
};

const char *zo_strerror(int status);

/* Growable output owned by the oracle; release with zo_free(). */
typedef struct {
  uint8_t *data;
  size_t len;
  size_t cap;
} zo_buf;

void zo_free(void *p);

/* src/zippy.nim:11-84.  gzip FNAME length is random in the reference (0..25
 * letters, zippy.nim:28-42); here it is an explicit argument (fname_len < 0
 * means "pick at random like the reference"). */
int zo_compress(const uint8_t *src, size_t len, int level, int data_format,
                int fname_len, zo_buf *out);

/* Block-parallel variant (BASELINE.json config 5; no counterpart in the reference): zo_compress
 * with deflate.nim:228's 4 MiB block size replaced by block_bytes (a multiple of 32768, <= 4 MiB)
 * plus the index of deflate block starts.  Entry k: bit offset of block k's BFINAL bit from the
 * start of the compressed buffer, and the offset of its first output byte; one closing entry
 * holds the bit just past the last block and the total length.  The stream itself is an
 * ordinary RFC 1951 stream that zo_uncompress / zlib decode. */
typedef struct {
  uint64_t bit_off;
  uint64_t out_off;
} zo_block_entry;
int zo_compress_blocks(const uint8_t *src, size_t len, int level, int data_format, int fname_len,
                       size_t block_bytes, zo_buf *out, zo_block_entry **index, size_t *n_entries);

/* src/zippy.nim:100-165, src/zippy/gzip.nim:3-88 */
int zo_uncompress(const uint8_t *src, size_t len, int data_format, zo_buf *out);

/* src/zippy/deflate.nim:207-467: appends the raw deflate stream to out. */
int zo_deflate(const uint8_t *src, size_t len, int level, zo_buf *out);

/* src/zippy/inflate.nim:268-291: decodes starting at byte pos of src. */
int zo_inflate(const uint8_t *src, size_t len, size_t pos, zo_buf *out);

/* src/zippy/crc.nim:29-72 (table path; PCLMUL path yields identical values) */
uint32_t zo_crc32(const uint8_t *src, size_t len);
/* src/zippy/adler32.nim:19-63 */
uint32_t zo_adler32(const uint8_t *src, size_t len);

/* ---- introspection used by kernel-level parity tests ---- */

/* internal.nim:128-131 */
typedef struct {
  uint32_t litlen_freq[286];
  uint32_t distance_freq[30];
  int64_t num_literals;
} zo_block_metadata;

/* Token stream of one <=4 MiB block: snappy.nim:138-163
 * (level 1), lz77.nim:10-130 (levels -1, 2..9), deflate.nim:153-177 (-2).
 * tokens is malloc'ed (zo_free). */
int zo_encode_block_tokens(const uint8_t *src, size_t block_start,
                           size_t block_len, int level, uint16_t **tokens,
                           size_t *num_tokens, zo_block_metadata *meta);

/* deflate.nim:13-151.  codes/lens must hold max(highest,min_codes)+1 entries
 * (<= num_freq).  Returns the number of codes. */
int zo_huffman_codes(const uint32_t *freq, int num_freq, int min_codes,
                     int code_length_limit, uint16_t *codes, uint8_t *lens);

/* bench.py's cpu_baseline: a batch on `threads` pinned worker threads (dir 0 compress, 1
 * uncompress), the wall time of the parallel region in *seconds; results in outs[] if given. */
int zo_batch_mt(const uint8_t *const *srcs, const size_t *lens, size_t n, int dir, int level, int fmt,
                int threads, zo_buf *outs, double *seconds);

#ifdef __cplusplus
}
#endif
#endif
