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
  zh_destroy(ctx);
  printf("c_consumer ok\n");
  return 0;
}
