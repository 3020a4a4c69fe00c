// ZIP archives as batch clients of the codec.
//
// Host-side record handling of src/zippy/ziparchives.nim on top of the public C ABI only:
//   zh_zip_open           openZipArchive    ziparchives.nim:183-372  (on a memory image)
//   zh_zip_extract_batch  extractFile       ziparchives.nim:39-93    for many records: ONE batched
//                                            raw-deflate decode (sizes from the central directory)
//                                            and CRC-32 of every result on the GPU
//   zh_zip_create         createZipArchive  ziparchives.nim:455-634  (OrderedTable form): ONE batched
//                                            compress(BestSpeed, dfDeflate) + CRC-32 on the GPU,
//                                            then record assembly
// File-system work (memfiles, createDir, permissions, mtimes: ziparchives.nim:374-453) stays with
// the caller.  No codec work happens on the CPU here.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/zippy_hip.h"

namespace {

constexpr uint32_t kFileHeaderSig = 0x04034b50u, kCentralSig = 0x02014b50u, kEocdSig = 0x06054b50u,
                   kZip64EocdSig = 0x06064b50u, kZip64LocatorSig = 0x07064b50u;
constexpr uint16_t kZip64ExtraId = 1;
constexpr int64_t kFileHeaderLen = 30;

// Code page 437, bytes 0x80..0xff -> Unicode (the mapping utf8ify uses, ziparchives.nim:108-160)
const uint16_t kCp437High[128] = {
    0x00c7, 0x00fc, 0x00e9, 0x00e2, 0x00e4, 0x00e0, 0x00e5, 0x00e7,
    0x00ea, 0x00eb, 0x00e8, 0x00ef, 0x00ee, 0x00ec, 0x00c4, 0x00c5,
    0x00c9, 0x00e6, 0x00c6, 0x00f4, 0x00f6, 0x00f2, 0x00fb, 0x00f9,
    0x00ff, 0x00d6, 0x00dc, 0x00a2, 0x00a3, 0x00a5, 0x20a7, 0x0192,
    0x00e1, 0x00ed, 0x00f3, 0x00fa, 0x00f1, 0x00d1, 0x00aa, 0x00ba,
    0x00bf, 0x2310, 0x00ac, 0x00bd, 0x00bc, 0x00a1, 0x00ab, 0x00bb,
    0x2591, 0x2592, 0x2593, 0x2502, 0x2524, 0x2561, 0x2562, 0x2556,
    0x2555, 0x2563, 0x2551, 0x2557, 0x255d, 0x255c, 0x255b, 0x2510,
    0x2514, 0x2534, 0x252c, 0x251c, 0x2500, 0x253c, 0x255e, 0x255f,
    0x255a, 0x2554, 0x2569, 0x2566, 0x2560, 0x2550, 0x256c, 0x2567,
    0x2568, 0x2564, 0x2565, 0x2559, 0x2558, 0x2552, 0x2553, 0x256b,
    0x256a, 0x2518, 0x250c, 0x2588, 0x2584, 0x258c, 0x2590, 0x2580,
    0x03b1, 0x00df, 0x0393, 0x03c0, 0x03a3, 0x03c3, 0x00b5, 0x03c4,
    0x03a6, 0x0398, 0x03a9, 0x03b4, 0x221e, 0x03c6, 0x03b5, 0x2229,
    0x2261, 0x00b1, 0x2265, 0x2264, 0x2320, 0x2321, 0x00f7, 0x2248,
    0x00b0, 0x2219, 0x00b7, 0x221a, 0x207f, 0x00b2, 0x25a0, 0x00a0,
};

struct Image {
  const uint8_t* p;
  int64_t size;
  uint16_t u16(int64_t at) const { return (uint16_t)(p[at] | (p[at + 1] << 8)); }
  uint32_t u32(int64_t at) const {
    return (uint32_t)p[at] | ((uint32_t)p[at + 1] << 8) | ((uint32_t)p[at + 2] << 16) | ((uint32_t)p[at + 3] << 24);
  }
  uint64_t u64(int64_t at) const { return (uint64_t)u32(at) | ((uint64_t)u32(at + 4) << 32); }
  // [at, at + n) lies inside the image.  Written without additions: `at` and `n` come from
  // 64-bit fields of the (untrusted) archive and may be anywhere up to INT64_MAX or negative.
  bool has(int64_t at, int64_t n) const { return at >= 0 && n >= 0 && n <= size && at <= size - n; }
};

struct Record {
  bool directory = false;
  int64_t header_offset = 0;
  std::string path;
  uint32_t crc = 0;
  int64_t compressed_size = 0, uncompressed_size = 0;
  uint32_t unix_mode = 0;
};

// Nim 1.6+ std/unicode validateUtf8 (not under /root/reference): structural check of lead and
// continuation bytes, rejecting 0xc0/0xc1 leads; -1 when the whole string passes.
int64_t validate_utf8(const std::string& s) {
  const int64_t n = (int64_t)s.size();
  int64_t i = 0;
  auto cont = [&](int64_t k) { return k < n && ((uint8_t)s[k] >> 6) == 2; };
  while (i < n) {
    const uint8_t c = (uint8_t)s[i];
    if (c <= 127) {
      i += 1;
    } else if ((c >> 5) == 6) {
      if (c < 0xc2 || !cont(i + 1)) return i;
      i += 2;
    } else if ((c >> 4) == 14) {
      if (!cont(i + 1) || !cont(i + 2)) return i;
      i += 3;
    } else if ((c >> 3) == 30) {
      if (!cont(i + 1) || !cont(i + 2) || !cont(i + 3)) return i;
      i += 4;
    } else {
      return i;
    }
  }
  return -1;
}

std::string utf8ify(const std::string& name) {  // ziparchives.nim:108-160
  if (validate_utf8(name) == -1) return name;
  std::string out;
  for (unsigned char c : name) {
    const uint32_t cp = c > 0x7f ? kCp437High[c - 0x80] : c;
    if (cp < 0x80) {
      out.push_back((char)cp);
    } else if (cp < 0x800) {
      out.push_back((char)(0xc0 | (cp >> 6)));
      out.push_back((char)(0x80 | (cp & 0x3f)));
    } else {
      out.push_back((char)(0xe0 | (cp >> 12)));
      out.push_back((char)(0x80 | ((cp >> 6) & 0x3f)));
      out.push_back((char)(0x80 | (cp & 0x3f)));
    }
  }
  return out;
}

bool ends_with_slash(const std::string& s) { return !s.empty() && s.back() == '/'; }

}  // namespace

struct zh_zip_reader {
  Image img;
  std::vector<Record> records;  // central directory order (an OrderedTable in the reference)
};

// openZipArchive, ziparchives.nim:183-372
extern "C" int zh_zip_open(const void* archive, size_t len, zh_zip_reader** out) {
  if (!out || (len && !archive)) return ZH_ERR_ARGUMENT;
  *out = nullptr;
  const Image im{(const uint8_t*)archive, (int64_t)len};

  // :162-173 the end-of-central-directory record, searched backwards from the shortest possible
  int64_t eocd = im.size - 22;
  for (;; eocd--) {
    if (eocd < 0) return ZH_ERR_ARCHIVE_EOF;
    if (im.u32(eocd) == kEocdSig) break;
  }
  const bool zip64 = eocd - 20 >= 0 && im.u32(eocd - 20) == kZip64LocatorSig;  // :200-203

  int64_t disk_number, start_disk, records_on_disk, num_records, cd_size, cd_start;
  if (zip64) {  // :208-238
    if (im.u32(eocd - 20 + 4) != 0) return ZH_ERR_ZIP_UNSUPPORTED;   // disk of the zip64 EOCD
    const int64_t pos = (int64_t)im.u64(eocd - 20 + 8);
    if (im.u32(eocd - 20 + 16) != 1) return ZH_ERR_ZIP_UNSUPPORTED;  // number of disks
    if (!im.has(pos, 64)) return ZH_ERR_ARCHIVE_EOF;
    if (im.u32(pos) != kZip64EocdSig) return ZH_ERR_ZIP_CENTRAL_HEADER;
    disk_number = im.u32(pos + 16);
    start_disk = im.u32(pos + 20);
    records_on_disk = (int64_t)im.u64(pos + 24);
    num_records = (int64_t)im.u64(pos + 32);
    cd_size = (int64_t)im.u64(pos + 40);
    cd_start = (int64_t)im.u64(pos + 48);
  } else {  // :239-246
    disk_number = im.u16(eocd + 4);
    start_disk = im.u16(eocd + 6);
    records_on_disk = im.u16(eocd + 8);
    num_records = im.u16(eocd + 10);
    cd_size = im.u32(eocd + 12);
    cd_start = im.u32(eocd + 16);
  }
  if (disk_number != 0 || start_disk != 0 || records_on_disk != num_records) return ZH_ERR_ZIP_UNSUPPORTED;
  // zip64 fields above INT64_MAX arrive here negative; nothing in an archive can lie or reach
  // beyond the image, and a record takes at least 46 bytes (the reference runs overflow-checked
  // and ends up raising on such values: same outcome, no wild read)
  if (cd_start < 0 || cd_start > im.size || cd_size < 0 || cd_size > im.size || num_records < 0 ||
      num_records > im.size / 46)
    return ZH_ERR_ARCHIVE_EOF;

  // :257-268 an archive may sit at the end of another file: find the first central header by
  // counting signatures backwards from the EOCD; any failure keeps the recorded start
  int64_t socd = cd_start;
  {
    int64_t at = eocd, found = 0;
    for (; at >= 0; at--) {
      if (im.u32(at) == kCentralSig && ++found == num_records) break;
    }
    if (at >= 0) socd = at;
  }
  const int64_t socd_offset = socd - cd_start;
  int64_t pos = socd_offset + cd_start;

  zh_zip_reader* r = new zh_zip_reader;
  r->img = im;
  std::unordered_set<std::string> seen;
  int status = ZH_OK;
  for (int64_t k = 0; k < num_records && status == ZH_OK; k++) {  // :275-361
    if (!im.has(pos, 46)) { status = ZH_ERR_ARCHIVE_EOF; break; }
    if (im.u32(pos) != kCentralSig) { status = ZH_ERR_ZIP_CENTRAL_HEADER; break; }
    const uint16_t flags = im.u16(pos + 8), method = im.u16(pos + 10);
    const uint32_t crc = im.u32(pos + 16);
    const int64_t name_len = im.u16(pos + 28), extra_len = im.u16(pos + 30), comment_len = im.u16(pos + 32);
    const uint16_t file_disk = im.u16(pos + 34);
    const uint32_t external = im.u32(pos + 38);
    if (method != 0 && method != 8) { status = ZH_ERR_ZIP_METHOD; break; }
    if (file_disk != 0) { status = ZH_ERR_ZIP_DISK_NUMBER; break; }
    int64_t csize = im.u32(pos + 20), usize = im.u32(pos + 24), hoff = im.u32(pos + 42);
    pos += 46;
    if (!im.has(pos, name_len)) { status = ZH_ERR_ARCHIVE_EOF; break; }
    const std::string raw((const char*)im.p + pos, (size_t)name_len);
    if (seen.count(raw)) { status = ZH_ERR_ZIP_DUPLICATE; break; }
    pos += name_len;
    {
      // :303-341 zip64 sizes.  The reference reads each field header at `pos` (the FIRST extra
      // field) while its cursor walks on, so only a zip64 field that comes first is honoured --
      // kept as is: every common writer (and createZipArchive) puts it first.
      int64_t cursor = pos;
      while (cursor < pos + extra_len) {
        if (!im.has(pos, 4)) { status = ZH_ERR_ARCHIVE_EOF; break; }
        const uint16_t id = im.u16(pos);
        const int64_t flen = im.u16(pos + 2);
        cursor += 4;
        if (id != kZip64ExtraId) {
          cursor += flen;
          continue;
        }
        int64_t at = cursor;
        const int64_t fend = cursor + flen;
        auto take64 = [&](int64_t& v) {
          if (at > fend - 8 || !im.has(at, 8)) { status = ZH_ERR_ARCHIVE_EOF; return; }
          v = (int64_t)im.u64(at);
          at += 8;
        };
        if (usize == 0xffffffffll) take64(usize);
        if (status == ZH_OK && csize == 0xffffffffll) take64(csize);
        if (status == ZH_OK && hoff == 0xffffffffll) take64(hoff);
        break;
      }
      if (status != ZH_OK) break;
    }
    pos += extra_len + comment_len;  // (pos <= size + 3 * 65535: no overflow)
    if (pos > socd_offset + cd_start + cd_size) { status = ZH_ERR_ZIP_CENTRAL_SIZE; break; }
    // sizes and offsets no image can hold (or above INT64_MAX): the entry is kept, as the
    // reference keeps it, but can only fail with "end of archive" when it is extracted
    if (hoff < 0 || hoff > im.size) hoff = -1;
    if (csize < 0 || csize > im.size) csize = -1;
    if (usize < 0) usize = -1;

    Record rec;
    rec.path = (flags & 0x0800) ? raw : utf8ify(raw);  // :345-350 language-encoding flag
    rec.directory = (external & 0x10u) != 0 || (external & (0x4000u << 16)) != 0 || ends_with_slash(rec.path);
    rec.header_offset = hoff < 0 ? -1 : hoff + socd_offset;
    rec.crc = crc;
    rec.compressed_size = csize;
    rec.uncompressed_size = usize;
    rec.unix_mode = external >> 16;
    seen.insert(rec.path);
    r->records.push_back(std::move(rec));
  }
  if (status != ZH_OK) {
    delete r;
    return status;
  }
  *out = r;
  return ZH_OK;
}

extern "C" void zh_zip_close(zh_zip_reader* r) { delete r; }
extern "C" size_t zh_zip_num_entries(const zh_zip_reader* r) { return r ? r->records.size() : 0; }

extern "C" int zh_zip_entry_at(const zh_zip_reader* r, size_t i, zh_zip_entry* out) {
  if (!r || !out || i >= r->records.size()) return ZH_ERR_ARGUMENT;
  const Record& rec = r->records[i];
  out->path = rec.path.data();
  out->path_len = rec.path.size();
  out->is_directory = rec.directory ? 1 : 0;
  out->header_offset = (uint64_t)rec.header_offset;
  out->compressed_size = (uint64_t)rec.compressed_size;
  out->uncompressed_size = (uint64_t)rec.uncompressed_size;
  out->crc32 = rec.crc;
  out->unix_mode = rec.unix_mode;
  return ZH_OK;
}

extern "C" int zh_zip_find(const zh_zip_reader* r, const char* path, size_t path_len, size_t* index) {
  if (!r || !index || (path_len && !path)) return ZH_ERR_ARGUMENT;
  for (size_t i = 0; i < r->records.size(); i++)
    if (r->records[i].path.size() == path_len && memcmp(r->records[i].path.data(), path, path_len) == 0) {
      *index = i;
      return ZH_OK;
    }
  return ZH_ERR_ZIP_NO_RECORD;  // ziparchives.nim:43-52
}

// extractFile (ziparchives.nim:39-93) for records indices[0..n): stored entries are copied,
// deflated ones go through one batched decode; every result's CRC-32 comes from the GPU.
extern "C" int zh_zip_extract_batch(zh_ctx* ctx, const zh_zip_reader* r, const size_t* indices, size_t n,
                                    void** dsts, size_t* dst_lens, int32_t* statuses) {
  if (!ctx || !r || (n && (!indices || !dsts || !dst_lens || !statuses))) return ZH_ERR_ARGUMENT;
  const Image& im = r->img;
  std::vector<size_t> deflated, stored;  // positions in the request
  std::vector<const void*> d_src, s_src;
  std::vector<size_t> d_len, s_len;
  std::vector<uint64_t> d_hint;
  std::vector<uint32_t> want_crc(n, 0);
  for (size_t k = 0; k < n; k++) {
    dsts[k] = nullptr;
    dst_lens[k] = 0;
    statuses[k] = ZH_OK;
    if (indices[k] >= r->records.size()) { statuses[k] = ZH_ERR_ZIP_NO_RECORD; continue; }
    const Record& rec = r->records[indices[k]];
    int64_t pos = rec.header_offset;
    if (!im.has(pos, kFileHeaderLen)) { statuses[k] = ZH_ERR_ARCHIVE_EOF; continue; }
    if (im.u32(pos) != kFileHeaderSig) { statuses[k] = ZH_ERR_ZIP_FILE_HEADER; continue; }
    const uint16_t method = im.u16(pos + 8);  // the LOCAL header's method decides (:62)
    pos += kFileHeaderLen + im.u16(pos + 26) + im.u16(pos + 28);
    if (rec.uncompressed_size < 0 || !im.has(pos, rec.compressed_size)) { statuses[k] = ZH_ERR_ARCHIVE_EOF; continue; }
    if (rec.directory) { statuses[k] = ZH_ERR_ZIP_NO_RECORD; continue; }
    want_crc[k] = rec.crc;
    if (method == 0) {
      stored.push_back(k);
      s_src.push_back(im.p + pos);
      s_len.push_back((size_t)rec.compressed_size);
    } else if (method == 8) {
      deflated.push_back(k);
      d_src.push_back(im.p + pos);
      d_len.push_back((size_t)rec.compressed_size);
      // (deflate cannot expand beyond 1032:1; a hint above that is a damaged directory and only
      // costs the sized retry)
      d_hint.push_back((uint64_t)std::min<int64_t>(rec.uncompressed_size, rec.compressed_size * 1032 + 1024));
    } else {
      statuses[k] = ZH_ERR_ZIP_METHOD;
    }
  }
  if (!deflated.empty()) {
    const size_t m = deflated.size();
    std::vector<void*> out(m);
    std::vector<size_t> out_len(m);
    std::vector<int32_t> st(m);
    std::vector<uint32_t> crc(m);
    int rc = zh_uncompress_batch_sized(ctx, d_src.data(), d_len.data(), m, ZH_DF_DEFLATE, d_hint.data(),
                                       out.data(), out_len.data(), st.data(), crc.data());
    if (rc) {
      for (size_t j = 0; j < m; j++) zh_free(out[j]);
      return rc;
    }
    for (size_t j = 0; j < m; j++) {
      const size_t k = deflated[j];
      statuses[k] = st[j];
      if (st[j] == ZH_OK && crc[j] != want_crc[k]) statuses[k] = ZH_ERR_ZIP_CRC;  // :91-92
      if (statuses[k] != ZH_OK) {
        zh_free(out[j]);
        continue;
      }
      dsts[k] = out[j];
      dst_lens[k] = out_len[j];
    }
  }
  if (!stored.empty()) {
    const size_t m = stored.size();
    std::vector<uint32_t> crc(m);
    int rc = zh_crc32_batch(ctx, s_src.data(), s_len.data(), m, crc.data());
    if (rc) return rc;
    for (size_t j = 0; j < m; j++) {
      const size_t k = stored[j];
      if (crc[j] != want_crc[k]) { statuses[k] = ZH_ERR_ZIP_CRC; continue; }
      dsts[k] = malloc(s_len[j] ? s_len[j] : 1);
      if (!dsts[k]) { statuses[k] = ZH_ERR_NOMEM; continue; }
      if (s_len[j]) memcpy(dsts[k], s_src[j], s_len[j]);
      dst_lens[k] = s_len[j];
    }
  }
  return ZH_OK;
}

// createZipArchive(entries: OrderedTable[string, string]), ziparchives.nim:455-634.  Entries are
// given in insertion order; like the reference (which pops keys off the end, :503-505) the
// archive lists them last to first.
extern "C" int zh_zip_create(zh_ctx* ctx, const char* const* paths, const size_t* path_lens,
                             const void* const* contents, const size_t* content_lens, size_t n,
                             uint16_t dos_time, uint16_t dos_date, void** archive, size_t* archive_len) {
  if (!ctx || !archive || !archive_len || (n && (!paths || !path_lens || !contents || !content_lens)))
    return ZH_ERR_ARGUMENT;
  *archive = nullptr;
  *archive_len = 0;
  std::unordered_set<std::string> names;
  for (size_t i = n; i-- > 0;) {  // :506-511, in processing order
    if (path_lens[i] == 0 || !paths[i]) return ZH_ERR_ZIP_NAME;   // "Invalid empty file name"
    if (paths[i][0] == '/') return ZH_ERR_ZIP_NAME;               // "File paths must be relative"
    if (path_lens[i] > 0xffffu) return ZH_ERR_ZIP_NAME;           // "File name len > uint16.high"
    if (content_lens[i] && !contents[i]) return ZH_ERR_ARGUMENT;
    if (!names.insert(std::string(paths[i], path_lens[i])).second) return ZH_ERR_ZIP_DUPLICATE;  // a table key is unique
  }
  // one batch: compress(contents, BestSpeed, dfDeflate) + crc32(contents) of the non-empty entries (:519-530)
  std::vector<size_t> order;  // non-empty entries, processing order
  std::vector<const void*> srcs;
  std::vector<size_t> lens;
  for (size_t i = n; i-- > 0;)
    if (content_lens[i]) {
      order.push_back(i);
      srcs.push_back(contents[i]);
      lens.push_back(content_lens[i]);
    }
  const size_t m = order.size();
  std::vector<void*> comp(m, nullptr);
  std::vector<size_t> comp_len(m, 0);
  std::vector<int32_t> st(m, ZH_OK);
  std::vector<uint32_t> crc(m, 0);
  struct Freer {
    std::vector<void*>& v;
    ~Freer() { for (void* p : v) zh_free(p); }
  } freer{comp};
  if (m) {
    int rc = zh_compress_batch_crc32(ctx, srcs.data(), lens.data(), m, ZH_BEST_SPEED, ZH_DF_DEFLATE, comp.data(),
                                     comp_len.data(), st.data(), crc.data());
    if (rc) return rc;
    for (size_t j = 0; j < m; j++)
      if (st[j] != ZH_OK) return st[j];
  }

  std::string z;
  auto add16 = [&](uint32_t v) { z.push_back((char)v); z.push_back((char)(v >> 8)); };
  auto add32 = [&](uint32_t v) { add16(v & 0xffffu); add16(v >> 16); };
  auto add64 = [&](uint64_t v) { add32((uint32_t)v); add32((uint32_t)(v >> 32)); };
  struct Entry {
    size_t src;
    uint64_t header_offset, ulen, clen;
    uint16_t method;
    uint32_t crc;
  };
  std::vector<Entry> entries;
  size_t next = 0;  // next compressed result
  for (size_t i = n; i-- > 0;) {
    Entry e{i, z.size(), content_lens[i], 0, 0, 0};
    const void* data = nullptr;
    if (content_lens[i]) {
      e.method = 8;
      e.clen = comp_len[next];
      e.crc = crc[next];
      data = comp[next];
      next++;
    }
    add32(kFileHeaderSig);  // :540-553
    add16(45);
    add16(1u << 11);
    add16(e.method);
    add16(dos_time);
    add16(dos_date);
    add32(e.crc);
    add32(0xffffffffu);
    add32(0xffffffffu);
    add16((uint32_t)path_lens[i]);
    add16(20);
    z.append(paths[i], path_lens[i]);
    add16(kZip64ExtraId);
    add16(16);
    add64(e.ulen);
    add64(e.clen);
    if (e.clen) z.append((const char*)data, e.clen);
    entries.push_back(e);
  }
  const uint64_t cd_start = z.size();
  for (const Entry& e : entries) {  // :570-596
    add32(kCentralSig);
    add16(45);
    add16(45);
    add16(1u << 11);
    add16(e.method);
    add16(dos_time);
    add16(dos_date);
    add32(e.crc);
    add32(0xffffffffu);
    add32(0xffffffffu);
    add16((uint32_t)path_lens[e.src]);
    add16(28);
    add16(0);
    add16(0);
    add16(0);
    add32(0);
    add32(0xffffffffu);
    z.append(paths[e.src], path_lens[e.src]);
    add16(kZip64ExtraId);
    add16(24);
    add64(e.ulen);
    add64(e.clen);
    add64(e.header_offset);
  }
  const uint64_t cd_end = z.size();
  add32(kZip64EocdSig);  // :600-609
  add64(44);
  add16(45);
  add16(45);
  add32(0);
  add32(0);
  add64(entries.size());
  add64(entries.size());
  add64(cd_end - cd_start);
  add64(cd_start);
  add32(kZip64LocatorSig);  // :611-614
  add32(0);
  add64(cd_end);
  add32(1);
  add32(kEocdSig);  // :616-623
  add16(0);
  add16(0);
  add16(0xffffu);
  add16(0xffffu);
  add32(0xffffffffu);
  add32(0xffffffffu);
  add16(0);

  *archive = malloc(z.size() ? z.size() : 1);
  if (!*archive) return ZH_ERR_NOMEM;
  memcpy(*archive, z.data(), z.size());
  *archive_len = z.size();
  return ZH_OK;
}
