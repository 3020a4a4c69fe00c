// Tarballs as a client of the codec: src/zippy/tarballs.nim:26-141 without
// its file-system half.  zh_tar_open takes the bytes of a .tar.gz or .tar: a gzip member is
// decoded on the GPU with its ISIZE as the output size (uncompressGzip(..., trustSize = true),
// tarballs.nim:47-50, gzip.nim:72-76), then the ustar headers are walked on the host exactly
// like tarballs.nim:61-124 and reported as entries pointing into the uncompressed image.
// createDir / writeFile / permissions / mtimes (tarballs.nim:98-131) stay with the caller.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/zippy_hip.h"

namespace {

struct Entry {
  std::string path, linkname;
  char typeflag;
  uint32_t mode;
  int64_t mtime;
  uint64_t offset, size;
};

// tarballs.nim:5-23 parseTarOctInt: the first run of decimal digits, read as octal (a digit 8 or
// 9 is Nim's ValueError -> ZippyError)
bool tar_octal(const uint8_t* s, size_t n, int64_t* out) {
  size_t start = 0;
  while (start < n && !(s[start] >= '0' && s[start] <= '9')) start++;
  size_t len = 0;
  while (start + len < n && s[start + len] >= '0' && s[start + len] <= '9') len++;
  int64_t v = 0;
  for (size_t i = 0; i < len; i++) {
    if (s[start + i] > '7') return false;
    v = v * 8 + (s[start + i] - '0');
  }
  *out = v;
  return true;
}

std::string field(const uint8_t* p, size_t n) {  // $(slice).cstring: up to the first NUL
  size_t k = 0;
  while (k < n && p[k]) k++;
  return std::string((const char*)p, k);
}

std::string join_path(const std::string& head, const std::string& tail) {  // std/os `/`
  if (head.empty()) return tail;
  const bool hs = head.back() == '/', ts = !tail.empty() && tail[0] == '/';
  if (hs && ts) return head + tail.substr(1);
  if (hs || ts) return head + tail;
  return head + "/" + tail;
}

bool starts_with(const std::string& s, const char* p) { return s.compare(0, strlen(p), p) == 0; }

// internal.nim:294-302 verifyPathIsSafeToExtract
bool safe_path(const std::string& path) {
  if (!path.empty() && path[0] == '/') return false;
  if (starts_with(path, "../") || starts_with(path, "..\\")) return false;
  if (path.find("/../") != std::string::npos || path.find("\\..\\") != std::string::npos) return false;
  return true;
}

}  // namespace

struct zh_tar_reader {
  void* owned = nullptr;         // the uncompressed tarball when it came out of a gzip member
  const uint8_t* data = nullptr;
  size_t len = 0;
  std::vector<Entry> entries;
};

extern "C" void zh_tar_close(zh_tar_reader* r) {
  if (!r) return;
  zh_free(r->owned);
  delete r;
}

extern "C" int zh_tar_open(zh_ctx* ctx, const void* image, size_t len, zh_tar_reader** out) {
  if (!out || (len && !image)) return ZH_ERR_ARGUMENT;
  *out = nullptr;
  const uint8_t* src = (const uint8_t*)image;
  if (len < 2) return ZH_ERR_INVALID_BUFFER;  // tarballs.nim:43-44
  zh_tar_reader* r = new zh_tar_reader;
  if (src[0] == 31 && src[1] == 139) {  // tarballs.nim:47-50
    if (!ctx) {
      delete r;
      return ZH_ERR_ARGUMENT;
    }
    if (len < 18) {  // gzip.nim:10-11
      delete r;
      return ZH_ERR_INVALID_BUFFER;
    }
    const void* srcs[1] = {image};
    size_t lens[1] = {len}, out_len = 0;
    const uint8_t* t = src + len - 4;
    uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
    void* dst = nullptr;
    int32_t st = ZH_OK;
    int rc = zh_uncompress_batch_sized(ctx, srcs, lens, 1, ZH_DF_GZIP, &isize, &dst, &out_len, &st, nullptr);
    if (rc || st) {
      zh_free(dst);
      delete r;
      return rc ? rc : st;
    }
    r->owned = dst;
    r->data = (const uint8_t*)dst;
    r->len = out_len;
  } else {  // tarballs.nim:51-54: an uncompressed tarball, borrowed
    r->data = src;
    r->len = len;
  }

  const uint8_t* u = r->data;
  const uint64_t n = r->len;
  std::string long_name;  // set by 'L' blocks for the entry that follows (tarballs.nim:59)
  uint64_t pos = 0;
  int status = ZH_OK;
  while (pos < n) {  // tarballs.nim:61-124
    if (pos + 512 > n) { status = ZH_ERR_ARCHIVE_EOF; break; }
    const std::string name = field(u + pos, 100);
    int64_t mode = 0, size = 0, mtime = 0;
    if (!tar_octal(u + pos + 100, 7, &mode) || !tar_octal(u + pos + 124, 11, &size) ||
        !tar_octal(u + pos + 136, 11, &mtime)) {
      status = ZH_ERR_TAR_NUMBER;
      break;
    }
    const char typeflag = (char)u[pos + 156];
    const std::string linkname = field(u + pos + 157, 100);
    const std::string prefix = field(u + pos + 257, 6) == "ustar" ? field(u + pos + 345, 155) : std::string();
    pos += 512;
    if (pos + (uint64_t)size > n) { status = ZH_ERR_ARCHIVE_EOF; break; }
    if (!name.empty() || !long_name.empty()) {
      std::string path;
      if (!long_name.empty()) {
        path = long_name;
        long_name.clear();
      } else {
        path = join_path(prefix, name);
      }
      if (!safe_path(path)) { status = ZH_ERR_UNSAFE_PATH; break; }
      if (typeflag == '0' || typeflag == '\0' || typeflag == '5' || typeflag == '2') {
        r->entries.push_back(Entry{path, linkname, typeflag, (uint32_t)mode, mtime, pos, (uint64_t)size});
      } else if (typeflag == 'L') {
        long_name.assign((const char*)u + pos, (size_t)size);
      } else if (typeflag == 'g' || typeflag == 'x' || (typeflag >= 'A' && typeflag <= 'Z')) {
        // extended headers and vendor types are skipped
      } else {
        status = ZH_ERR_TAR_HEADER_TYPE;
        break;
      }
    }
    pos += ((uint64_t)size + 511u) & ~(uint64_t)511;
  }
  if (status != ZH_OK) {
    zh_tar_close(r);
    return status;
  }
  *out = r;
  return ZH_OK;
}

extern "C" size_t zh_tar_num_entries(const zh_tar_reader* r) { return r ? r->entries.size() : 0; }
extern "C" const void* zh_tar_data(const zh_tar_reader* r, size_t* len) {
  if (len) *len = r ? r->len : 0;
  return r ? r->data : nullptr;
}
extern "C" int zh_tar_entry_at(const zh_tar_reader* r, size_t i, zh_tar_entry* out) {
  if (!r || !out || i >= r->entries.size()) return ZH_ERR_ARGUMENT;
  const Entry& e = r->entries[i];
  out->path = e.path.data();
  out->path_len = e.path.size();
  out->linkname = e.linkname.data();
  out->linkname_len = e.linkname.size();
  out->typeflag = e.typeflag;
  out->mode = e.mode;
  out->mtime = e.mtime;
  out->offset = e.offset;
  out->size = e.size;
  return ZH_OK;
}
