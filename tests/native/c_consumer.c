/* A plain C99 consumer of include/zippy_hip.h: what a cgo / Nim importc / JNI shim compiles
 * against.  Built by tests/test_c_consumer.py with gcc -std=c99 -pedantic -Werror and linked
 * with the library under test (the emulator build on the CPU box, the HIP build on a GPU box).
 * Test infrastructure only. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zippy_hip.h"

#define N 5

static int fail(const char *what, int code) {
  fprintf(stderr, "c_consumer: %s -> %d (%s)\n", what, code, zh_strerror(code));
  return 1;
}

int main(void) {
  zh_ctx *ctx = NULL;
  int rc = zh_create(-1, NULL, &ctx);
  if (rc != ZH_OK) return fail("zh_create", rc);
  zh_set_gzip_fname_len(ctx, 0);

  static unsigned char text[N][70000];
  const void *srcs[N];
  size_t lens[N];
  const size_t want[N] = {0, 1, 4097, 70000, 33333};
  size_t i, k;
  uint32_t seed = 12345u;
  for (i = 0; i < N; i++) {
    for (k = 0; k < want[i]; k++) {
      seed = seed * 1664525u + 1013904223u;
      text[i][k] = (unsigned char)("the quick brown fox "[(k + (seed >> 29)) % 20]);
    }
    srcs[i] = text[i];
    lens[i] = want[i];
  }

  void *comp[N];
  size_t comp_len[N];
  int32_t st[N];
  rc = zh_compress_batch(ctx, srcs, lens, N, 1 /* BestSpeed */, ZH_DF_GZIP, comp, comp_len, st);
  if (rc != ZH_OK) return fail("zh_compress_batch", rc);
  for (i = 0; i < N; i++)
    if (st[i] != ZH_OK) return fail("compress status", st[i]);
  if (comp_len[3] >= want[3]) return fail("text did not shrink", -1);

  void *back[N];
  size_t back_len[N];
  rc = zh_uncompress_batch(ctx, (const void *const *)comp, comp_len, N, ZH_DF_DETECT, back, back_len, st);
  if (rc != ZH_OK) return fail("zh_uncompress_batch", rc);
  for (i = 0; i < N; i++) {
    if (st[i] != ZH_OK) return fail("uncompress status", st[i]);
    if (back_len[i] != lens[i] || memcmp(back[i], text[i], lens[i]) != 0) return fail("round trip differs", -1);
  }

  /* the same two calls with the results in buffers of this program's (zh_*_batch_into), and with
   * the parallel BestSpeed parse switched on: other bytes, the same round trip */
  {
    static unsigned char cbuf[N][2 * 70000 + 8192], ubuf[N][70000];
    void *cd[N], *ud[N];
    size_t ccap[N], ucap[N], clen[N], ulen[N];
    int pass;
    for (pass = 0; pass < 2; pass++) {
      zh_set_l1_parse(ctx, pass);
      for (i = 0; i < N; i++) {
        cd[i] = cbuf[i];
        ccap[i] = zh_compress_bound(lens[i], ZH_DF_GZIP);
        if (ccap[i] > sizeof cbuf[i]) return fail("zh_compress_bound", -1);
        ud[i] = ubuf[i];
        ucap[i] = lens[i];
      }
      rc = zh_compress_batch_into(ctx, srcs, lens, N, 1, ZH_DF_GZIP, cd, ccap, clen, st);
      if (rc != ZH_OK) return fail("zh_compress_batch_into", rc);
      for (i = 0; i < N; i++) {
        if (st[i] != ZH_OK || cd[i] != (void *)cbuf[i]) return fail("compress_into status", st[i]);
        if (pass == 0 && (clen[i] != comp_len[i] || memcmp(cbuf[i], comp[i], clen[i]) != 0))
          return fail("compress_into differs from compress", -1);
      }
      rc = zh_uncompress_batch_into(ctx, (const void *const *)cd, clen, N, ZH_DF_GZIP, ud, ucap, ulen, st);
      if (rc != ZH_OK) return fail("zh_uncompress_batch_into", rc);
      for (i = 0; i < N; i++)
        if (st[i] != ZH_OK || ulen[i] != lens[i] || memcmp(ubuf[i], text[i], lens[i]) != 0)
          return fail("into round trip differs", st[i]);
    }
    zh_set_l1_parse(ctx, -1);
    /* a buffer that is too small only fails its own slot and learns its size */
    for (i = 0; i < N; i++) cd[i] = cbuf[i];
    ccap[3] = 10;
    rc = zh_compress_batch_into(ctx, srcs, lens, N, 1, ZH_DF_GZIP, cd, ccap, clen, st);
    if (rc != ZH_OK || st[3] != ZH_ERR_DST_TOO_SMALL || cd[3] != NULL || clen[3] != comp_len[3])
      return fail("compress_into with a small buffer", rc ? rc : st[3]);
    if (st[4] != ZH_OK) return fail("neighbour of a small buffer", st[4]);
    /* ... and so does uncompress: the status a binding grows its buffer and retries on, not a checksum error */
    for (i = 0; i < N; i++) {
      cd[i] = comp[i];
      ud[i] = ubuf[i];
      ucap[i] = lens[i];
    }
    ucap[3] = lens[3] - 1;
    rc = zh_uncompress_batch_into(ctx, (const void *const *)cd, comp_len, N, ZH_DF_GZIP, ud, ucap, ulen, st);
    if (rc != ZH_OK || st[3] != ZH_ERR_DST_TOO_SMALL || ud[3] != NULL || ulen[3] != lens[3])
      return fail("uncompress_into with a small buffer", rc ? rc : st[3]);
    if (st[4] != ZH_OK || ulen[4] != lens[4]) return fail("neighbour of a small buffer (uncompress)", st[4]);
  }

  /* a damaged member fails its own slot with the reference's error, the others still decode */
  ((unsigned char *)comp[3])[comp_len[3] / 2] ^= 0x10;
  for (i = 0; i < N; i++) zh_free(back[i]);
  rc = zh_uncompress_batch(ctx, (const void *const *)comp, comp_len, N, ZH_DF_GZIP, back, back_len, st);
  if (rc != ZH_OK) return fail("zh_uncompress_batch (damaged)", rc);
  if (st[3] == ZH_OK || back[3] != NULL) return fail("damage went unnoticed", -1);
  if (st[4] != ZH_OK || back_len[4] != lens[4]) return fail("neighbour of a damaged member", st[4]);

  uint32_t crc = 0, adler = 0;
  if ((rc = zh_crc32(ctx, "123456789", 9, &crc)) != ZH_OK) return fail("zh_crc32", rc);
  if ((rc = zh_adler32(ctx, "123456789", 9, &adler)) != ZH_OK) return fail("zh_adler32", rc);
  if (crc != 0xcbf43926u || adler != 0x091e01deu) return fail("checksum check values", -1);

  for (i = 0; i < N; i++) {
    zh_free(comp[i]);
    zh_free(back[i]);
  }

  /* block-parallel form of one buffer: independent 32 KiB deflate blocks + their index */
  {
    void *blk = NULL, *out = NULL;
    size_t blk_len = 0, out_len = 0, n_entries = 0;
    zh_block_entry *index = NULL;
    rc = zh_compress_blocks(ctx, text[3], lens[3], 1, ZH_DF_GZIP, 32768, &blk, &blk_len, &index, &n_entries);
    if (rc != ZH_OK) return fail("zh_compress_blocks", rc);
    if (n_entries != 3 + 1 || index[n_entries - 1].out_off != lens[3]) return fail("block index shape", -1);
    rc = zh_uncompress_indexed(ctx, blk, blk_len, ZH_DF_DETECT, index, n_entries, &out, &out_len);
    if (rc != ZH_OK) return fail("zh_uncompress_indexed", rc);
    if (out_len != lens[3] || memcmp(out, text[3], out_len) != 0) return fail("indexed round trip differs", -1);
    zh_free(out);
    zh_free(index);
    zh_free(blk);
  }

  /* ZIP: create, open, look up, extract (ziparchives.nim) */
  {
    const char *names[3] = {"docs/a.txt", "b.bin", "empty"};
    const size_t name_lens[3] = {10, 5, 5};
    const void *contents[3];
    size_t content_lens[3];
    void *zip = NULL, *got[2];
    size_t zip_len = 0, got_len[2], idx[2], where = 0;
    zh_zip_reader *rd = NULL;
    zh_zip_entry ent;
    contents[0] = text[3];
    content_lens[0] = lens[3];
    contents[1] = text[2];
    content_lens[1] = lens[2];
    contents[2] = text[0];
    content_lens[2] = 0;
    rc = zh_zip_create(ctx, names, name_lens, contents, content_lens, 3, 0, 33, &zip, &zip_len);
    if (rc != ZH_OK) return fail("zh_zip_create", rc);
    if ((rc = zh_zip_open(zip, zip_len, &rd)) != ZH_OK) return fail("zh_zip_open", rc);
    if (zh_zip_num_entries(rd) != 3) return fail("zip entry count", -1);
    if ((rc = zh_zip_find(rd, "b.bin", 5, &where)) != ZH_OK) return fail("zh_zip_find", rc);
    if ((rc = zh_zip_entry_at(rd, where, &ent)) != ZH_OK) return fail("zh_zip_entry_at", rc);
    if (ent.path_len != 5 || memcmp(ent.path, "b.bin", 5) != 0 || ent.is_directory) return fail("zip entry fields", -1);
    idx[0] = where;
    if ((rc = zh_zip_find(rd, names[0], name_lens[0], &idx[1])) != ZH_OK) return fail("zh_zip_find (2)", rc);
    rc = zh_zip_extract_batch(ctx, rd, idx, 2, got, got_len, st);
    if (rc != ZH_OK || st[0] != ZH_OK || st[1] != ZH_OK) return fail("zh_zip_extract_batch", rc ? rc : (st[0] ? st[0] : st[1]));
    if (got_len[0] != lens[2] || memcmp(got[0], text[2], lens[2]) != 0) return fail("zip entry b.bin differs", -1);
    if (got_len[1] != lens[3] || memcmp(got[1], text[3], lens[3]) != 0) return fail("zip entry docs/a.txt differs", -1);
    if (zh_zip_find(rd, "missing", 7, &where) == ZH_OK) return fail("found a missing entry", -1);
    zh_free(got[0]);
    zh_free(got[1]);
    zh_zip_close(rd);
    zh_free(zip);
  }
  /* one batch over two contexts (two GPUs in production; here both on the current device): every
   * result lands at its own index and equals the single-context result */
  {
    zh_ctx *pair[2] = {NULL, NULL};
    void *one[N], *two[N], *rt[N];
    size_t one_len[N], two_len[N], rt_len[N];
    int32_t st2[N];
    if (zh_device_count() < 1) return fail("zh_device_count", -1);
    if ((rc = zh_create(-1, NULL, &pair[0])) != ZH_OK) return fail("zh_create (pair 0)", rc);
    if ((rc = zh_create(-1, NULL, &pair[1])) != ZH_OK) return fail("zh_create (pair 1)", rc);
    zh_set_gzip_fname_len(pair[0], 0);
    zh_set_gzip_fname_len(pair[1], 0);
    rc = zh_compress_batch(ctx, srcs, lens, N, 1, ZH_DF_GZIP, one, one_len, st);
    if (rc != ZH_OK) return fail("zh_compress_batch (reference for multi)", rc);
    rc = zh_compress_batch_multi(pair, 2, srcs, lens, N, 1, ZH_DF_GZIP, two, two_len, st2);
    if (rc != ZH_OK) return fail("zh_compress_batch_multi", rc);
    for (i = 0; i < N; i++) {
      if (st2[i] != ZH_OK) return fail("multi compress status", st2[i]);
      if (two_len[i] != one_len[i] || memcmp(two[i], one[i], one_len[i]) != 0) return fail("multi != single", -1);
    }
    rc = zh_uncompress_batch_multi(pair, 2, (const void *const *)two, two_len, N, ZH_DF_DETECT, rt, rt_len, st2);
    if (rc != ZH_OK) return fail("zh_uncompress_batch_multi", rc);
    for (i = 0; i < N; i++) {
      if (st2[i] != ZH_OK) return fail("multi uncompress status", st2[i]);
      if (rt_len[i] != lens[i] || memcmp(rt[i], text[i], lens[i]) != 0) return fail("multi round trip differs", -1);
      zh_free(one[i]);
      zh_free(two[i]);
      zh_free(rt[i]);
    }
    pair[1] = pair[0];
    if (zh_compress_batch_multi(pair, 2, srcs, lens, N, 1, ZH_DF_GZIP, two, two_len, st2) != ZH_ERR_ARGUMENT)
      return fail("the same context twice must be refused", -1);
    zh_destroy(pair[0]);
  }
  /* device-resident plans without a HIP binding: the batch compressed in device slots, the streams packed back to
   * back (what would go over a link to another GPU), scattered into an uncompress plan's slots and decoded */
  {
    uint64_t src_off[N], src_len[N], dst_off[N], dst_cap[N], offs[N + 1], out_lens[N], back_off[N], back_cap[N];
    uint64_t in_total = 0, slot_total = 0, packed_cap;
    void *d_in = NULL, *d_slots = NULL, *d_packed = NULL, *d_offs = NULL, *d_slots2 = NULL, *d_back = NULL;
    zh_plan *cp = NULL, *up = NULL;
    static unsigned char flat[N * 70000], rt2[N * 70000];
    void *one[N];
    size_t one_len[N];
    for (i = 0; i < N; i++) {
      src_off[i] = in_total;
      src_len[i] = lens[i];
      memcpy(flat + in_total, text[i], lens[i]);
      in_total += lens[i];
      dst_off[i] = slot_total;
      dst_cap[i] = zh_compress_bound(lens[i], ZH_DF_GZIP);
      slot_total += (dst_cap[i] + 255u) & ~(uint64_t)255u;
      back_off[i] = src_off[i];
      back_cap[i] = lens[i];
    }
    packed_cap = slot_total;
    if ((rc = zh_device_malloc(ctx, in_total + 16, &d_in)) != ZH_OK) return fail("zh_device_malloc", rc);
    if ((rc = zh_device_malloc(ctx, slot_total, &d_slots)) != ZH_OK) return fail("zh_device_malloc", rc);
    if ((rc = zh_device_malloc(ctx, packed_cap, &d_packed)) != ZH_OK) return fail("zh_device_malloc", rc);
    if ((rc = zh_device_malloc(ctx, sizeof offs, &d_offs)) != ZH_OK) return fail("zh_device_malloc", rc);
    if ((rc = zh_device_malloc(ctx, slot_total, &d_slots2)) != ZH_OK) return fail("zh_device_malloc", rc);
    if ((rc = zh_device_malloc(ctx, in_total + 16, &d_back)) != ZH_OK) return fail("zh_device_malloc", rc);
    if ((rc = zh_device_upload(ctx, d_in, flat, in_total)) != ZH_OK) return fail("zh_device_upload", rc);
    if ((rc = zh_plan_compress(ctx, N, src_off, src_len, dst_off, dst_cap, 1, ZH_DF_GZIP, &cp)) != ZH_OK)
      return fail("zh_plan_compress", rc);
    if ((rc = zh_plan_run(cp, d_in, d_slots)) != ZH_OK) return fail("zh_plan_run", rc);
    if ((rc = zh_plan_pack(cp, d_slots, d_packed, packed_cap, (uint64_t *)d_offs)) != ZH_OK) return fail("zh_plan_pack", rc);
    if ((rc = zh_plan_results(cp, out_lens, st)) != ZH_OK) return fail("zh_plan_results", rc);
    if ((rc = zh_device_download(ctx, offs, d_offs, sizeof offs)) != ZH_OK) return fail("zh_device_download", rc);
    rc = zh_compress_batch(ctx, srcs, lens, N, 1, ZH_DF_GZIP, one, one_len, st);
    if (rc != ZH_OK) return fail("zh_compress_batch (reference for pack)", rc);
    if (offs[0] != 0) return fail("pack: offsets[0]", -1);
    for (i = 0; i < N; i++) {
      static unsigned char got[2 * 70000 + 8192];
      if (offs[i + 1] - offs[i] != out_lens[i] || out_lens[i] != one_len[i]) return fail("pack: offsets", -1);
      if ((rc = zh_device_download(ctx, got, (unsigned char *)d_packed + offs[i], one_len[i])) != ZH_OK)
        return fail("zh_device_download", rc);
      if (memcmp(got, one[i], one_len[i]) != 0) return fail("pack: stream differs from zh_compress_batch's", -1);
      zh_free(one[i]);
    }
    /* the slots' sizes are all the uncompress plan knows: the lengths arrive with the streams */
    if ((rc = zh_plan_uncompress(ctx, N, dst_off, dst_cap, back_off, back_cap, ZH_DF_GZIP, &up)) != ZH_OK)
      return fail("zh_plan_uncompress", rc);
    if ((rc = zh_plan_unpack(up, d_packed, (const uint64_t *)d_offs, d_slots2)) != ZH_OK) return fail("zh_plan_unpack", rc);
    if ((rc = zh_plan_run(up, d_slots2, d_back)) != ZH_OK) return fail("zh_plan_run (uncompress)", rc);
    if ((rc = zh_plan_results(up, out_lens, st)) != ZH_OK) return fail("zh_plan_results (uncompress)", rc);
    if ((rc = zh_device_download(ctx, rt2, d_back, in_total)) != ZH_OK) return fail("zh_device_download", rc);
    for (i = 0; i < N; i++)
      if (st[i] != ZH_OK || out_lens[i] != lens[i] || memcmp(rt2 + src_off[i], text[i], lens[i]) != 0)
        return fail("pack / unpack round trip differs", st[i]);
    if (zh_plan_unpack(cp, d_packed, (const uint64_t *)d_offs, d_slots2) != ZH_ERR_ARGUMENT)
      return fail("unpack into a compress plan must be refused", -1);
    zh_plan_destroy(cp);
    zh_plan_destroy(up);
    zh_device_free(ctx, d_in);
    zh_device_free(ctx, d_slots);
    zh_device_free(ctx, d_packed);
    zh_device_free(ctx, d_offs);
    zh_device_free(ctx, d_slots2);
    zh_device_free(ctx, d_back);
  }
  zh_destroy(ctx);
  printf("c_consumer ok\n");
  return 0;
}
