## MI355X backend for zippy's compress()/uncompress() hot path -- the binding a maintainer drops into the
## reference tree as src/zippy/hip.nim (build with -d:zippyHip; see INTEGRATION.md, which quotes this file).
## Binds libzippy_hip.so (include/zippy_hip.h).  NOT compiled anywhere in this repository: the build image has no
## Nim.  tests/test_abi.py checks that every `importc` proc declared here names a symbol the header declares and
## the library exports, with the same number of parameters; the C99 consumer (tests/native/c_consumer.c) drives the
## same calls end to end.
import common

const zhLib = "libzippy_hip.so"

type
  ZhCtx = pointer

proc zh_create(device: cint, stream: pointer, ctx: ptr ZhCtx): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_destroy(ctx: ZhCtx) {.importc, cdecl, dynlib: zhLib.}
proc zh_strerror(status: cint): cstring {.importc, cdecl, dynlib: zhLib.}
proc zh_free(p: pointer) {.importc, cdecl, dynlib: zhLib.}
proc zh_compress(ctx: ZhCtx, src: pointer, len: csize_t, level, dataFormat: cint,
                 dst: ptr pointer, dstLen: ptr csize_t): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_uncompress(ctx: ZhCtx, src: pointer, len: csize_t, dataFormat: cint,
                   dst: ptr pointer, dstLen: ptr csize_t): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_compress_batch(ctx: ZhCtx, srcs: ptr pointer, lens: ptr csize_t, n: csize_t,
                       level, dataFormat: cint, dsts: ptr pointer, dstLens: ptr csize_t,
                       statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_uncompress_batch(ctx: ZhCtx, srcs: ptr pointer, lens: ptr csize_t, n: csize_t,
                         dataFormat: cint, dsts: ptr pointer, dstLens: ptr csize_t,
                         statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_crc32(ctx: ZhCtx, src: pointer, len: csize_t, res: ptr uint32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_adler32(ctx: ZhCtx, src: pointer, len: csize_t, res: ptr uint32): cint {.importc, cdecl, dynlib: zhLib.}

var ctx {.threadvar.}: ZhCtx   # one context per thread: contexts are thread-compatible

proc engine(): ZhCtx =
  if ctx.isNil:
    let st = zh_create(-1, nil, ctx.addr)
    if st != 0: raise newException(ZippyError, $zh_strerror(st))
  ctx

proc take(p: pointer, len: csize_t, st: cint): string =
  ## status -> ZippyError with the reference's message; result -> GC-owned string
  if st != 0:
    if p != nil: zh_free(p)
    raise newException(ZippyError, $zh_strerror(st))
  result = newString(len.int)
  if len > 0: copyMem(result[0].addr, p, len.int)
  zh_free(p)

proc hipCompress*(src: pointer, len: int, level = DefaultCompression,
                  dataFormat = dfGzip): string {.raises: [ZippyError].} =
  var dst: pointer
  var dstLen: csize_t
  let st = zh_compress(engine(), src, len.csize_t, level.cint, ord(dataFormat).cint,
                       dst.addr, dstLen.addr)
  take(dst, dstLen, st)

proc hipUncompress*(src: pointer, len: int,
                    dataFormat = dfDetect): string {.raises: [ZippyError].} =
  var dst: pointer
  var dstLen: csize_t
  let st = zh_uncompress(engine(), src, len.csize_t, ord(dataFormat).cint,
                         dst.addr, dstLen.addr)
  take(dst, dstLen, st)

proc hipCompressBatch*(srcs: openArray[string], level = DefaultCompression,
                       dataFormat = dfGzip): seq[string] {.raises: [ZippyError].} =
  ## n independent compress() calls in one launch sequence (the shape the GPU wants)
  let n = srcs.len
  var
    ptrs = newSeq[pointer](n)
    lens = newSeq[csize_t](n)
    dsts = newSeq[pointer](n)
    dlens = newSeq[csize_t](n)
    sts = newSeq[int32](n)
  for i, s in srcs:
    ptrs[i] = if s.len > 0: s[0].unsafeAddr else: nil
    lens[i] = s.len.csize_t
  let rc = zh_compress_batch(engine(), ptrs[0].addr, lens[0].addr, n.csize_t, level.cint,
                             ord(dataFormat).cint, dsts[0].addr, dlens[0].addr, sts[0].addr)
  if rc != 0: raise newException(ZippyError, $zh_strerror(rc))
  for i in 0 ..< n: result.add take(dsts[i], dlens[i], sts[i].cint)

# ---- results into Nim strings the shim owns (zh_*_batch_into); contract mode for BestSpeed ----
proc zh_compress_bound(len: csize_t, dataFormat: cint): csize_t {.importc, cdecl, dynlib: zhLib.}
proc zh_compress_batch_into(ctx: ZhCtx, srcs: ptr pointer, lens: ptr csize_t, n: csize_t,
                            level, dataFormat: cint, dsts: ptr pointer, caps: ptr csize_t,
                            dstLens: ptr csize_t, statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_uncompress_batch_into(ctx: ZhCtx, srcs: ptr pointer, lens: ptr csize_t, n: csize_t,
                              dataFormat: cint, dsts: ptr pointer, caps: ptr csize_t,
                              dstLens: ptr csize_t, statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_set_l1_parse(ctx: ZhCtx, mode: cint) {.importc, cdecl, dynlib: zhLib.}

proc hipCompressBatchInto*(srcs: openArray[string], level = DefaultCompression,
                           dataFormat = dfGzip): seq[string] {.raises: [ZippyError].} =
  ## as hipCompressBatch, without the library's malloc + the copy out of it: the results land in
  ## strings this proc allocated (newString(bound), then setLen to what was written)
  let n = srcs.len
  result = newSeq[string](n)
  var
    ptrs = newSeq[pointer](n)
    lens = newSeq[csize_t](n)
    dsts = newSeq[pointer](n)
    caps = newSeq[csize_t](n)
    dlens = newSeq[csize_t](n)
    sts = newSeq[int32](n)
  for i, s in srcs:
    ptrs[i] = if s.len > 0: s[0].unsafeAddr else: nil
    lens[i] = s.len.csize_t
    caps[i] = zh_compress_bound(lens[i], ord(dataFormat).cint)
    result[i] = newString(caps[i].int)
    dsts[i] = result[i][0].addr
  let rc = zh_compress_batch_into(engine(), ptrs[0].addr, lens[0].addr, n.csize_t, level.cint,
                                  ord(dataFormat).cint, dsts[0].addr, caps[0].addr, dlens[0].addr, sts[0].addr)
  if rc != 0: raise newException(ZippyError, $zh_strerror(rc))
  for i in 0 ..< n:
    if sts[i] != 0: raise newException(ZippyError, $zh_strerror(sts[i].cint))
    result[i].setLen(dlens[i].int)

proc useParallelBestSpeedParse*(on = true) =
  ## BestSpeed only.  OFF (default): zippy's own parse, the streams are byte for byte zippy's.
  ## ON: a different token stream of about the same size (measured: 1-4 % smaller) that zippy's
  ## uncompress() decodes to the same bytes -- 2.4 x faster match finding on the device.
  zh_set_l1_parse(engine(), if on: 1 else: 0)

# ---- more than one GPU ----
proc zh_device_count(): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_compress_batch_multi(ctxs: ptr ZhCtx, nCtx: csize_t, srcs: ptr pointer, lens: ptr csize_t,
                             n: csize_t, level, dataFormat: cint, dsts: ptr pointer,
                             dstLens: ptr csize_t, statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_uncompress_batch_multi(ctxs: ptr ZhCtx, nCtx: csize_t, srcs: ptr pointer, lens: ptr csize_t,
                               n: csize_t, dataFormat: cint, dsts: ptr pointer, dstLens: ptr csize_t,
                               statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}

# ---- device-resident batches (one process a GPU: buffers and slots already in HBM, e.g. on their way between GPUs) ----
type ZhPlan = pointer
proc zh_plan_compress(ctx: ZhCtx, n: csize_t, srcOff, srcLen, dstOff, dstCap: ptr uint64, level, dataFormat: cint,
                      plan: ptr ZhPlan): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_plan_uncompress(ctx: ZhCtx, n: csize_t, srcOff, srcLen, dstOff, dstCap: ptr uint64, dataFormat: cint,
                        plan: ptr ZhPlan): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_plan_run(plan: ZhPlan, dSrc, dDst: pointer): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_plan_results(plan: ZhPlan, outLens: ptr uint64, statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_plan_pack(plan: ZhPlan, dSlots, dPacked: pointer, packedCap: uint64,
                  dOffsets: ptr uint64): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_plan_unpack(plan: ZhPlan, dPacked: pointer, dOffsets: ptr uint64,
                    dSlots: pointer): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_plan_destroy(plan: ZhPlan) {.importc, cdecl, dynlib: zhLib.}
proc zh_device_malloc(ctx: ZhCtx, bytes: csize_t, dOut: ptr pointer): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_device_free(ctx: ZhCtx, d: pointer) {.importc, cdecl, dynlib: zhLib.}
proc zh_device_upload(ctx: ZhCtx, dDst, src: pointer, bytes: csize_t): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_device_download(ctx: ZhCtx, dst, dSrc: pointer, bytes: csize_t): cint {.importc, cdecl, dynlib: zhLib.}

# ---- the archive layer (src/zippy/ziparchives.nim) ----
type
  ZhZipEntry {.bycopy.} = object
    path: cstring                       # not NUL-terminated: use pathLen
    pathLen: csize_t
    isDirectory: cint
    headerOffset, compressedSize, uncompressedSize: uint64
    crc32, unixMode: uint32

proc zh_zip_open(archive: pointer, len: csize_t, reader: ptr pointer): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_zip_close(reader: pointer) {.importc, cdecl, dynlib: zhLib.}
proc zh_zip_num_entries(reader: pointer): csize_t {.importc, cdecl, dynlib: zhLib.}
proc zh_zip_entry_at(reader: pointer, i: csize_t, e: ptr ZhZipEntry): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_zip_extract_batch(ctx: ZhCtx, reader: pointer, indices: ptr csize_t, n: csize_t,
                          dsts: ptr pointer, lens: ptr csize_t, statuses: ptr int32): cint {.importc, cdecl, dynlib: zhLib.}
proc zh_zip_create(ctx: ZhCtx, paths: ptr cstring, pathLens: ptr csize_t, contents: ptr pointer,
                   contentLens: ptr csize_t, n: csize_t, dosTime, dosDate: uint16,
                   archive: ptr pointer, archiveLen: ptr csize_t): cint {.importc, cdecl, dynlib: zhLib.}

proc hipCreateZipArchive*(entries: OrderedTable[string, string]): string {.raises: [ZippyError].} =
  ## createZipArchiveImpl (ziparchives.nim:455-623) with the per-entry codec work in one batch
  var paths: seq[cstring]; var pathLens, contentLens: seq[csize_t]; var contents: seq[pointer]
  for k, v in entries:                 # insertion order; the library lists them last to first
    paths.add k.cstring; pathLens.add k.len.csize_t
    contents.add (if v.len > 0: v[0].unsafeAddr else: nil); contentLens.add v.len.csize_t
  let (t, d) = msdos(getTime())        # ziparchives.nim:475-493, unchanged
  var p: pointer; var n: csize_t
  let st = zh_zip_create(engine(), paths[0].addr, pathLens[0].addr, contents[0].addr,
                         contentLens[0].addr, paths.len.csize_t, t, d, p.addr, n.addr)
  take(p, n, st)
