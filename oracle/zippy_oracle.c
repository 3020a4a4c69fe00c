#define _GNU_SOURCE /* pthread_setaffinity_np, CPU_SET (zo_batch_mt) */
/*
 * zippy_oracle.c -- TEST INFRASTRUCTURE ONLY (see zippy_oracle.h).
 *
 * CPU restatement of guzba/zippy v0.10.18's DEFLATE path.  Citations are
 * file:line under /root/reference.  Control flow follows the reference so that
 * parse decisions (tokens), block structure, Huffman code lengths and error
 * categories agree; data structures are plain C.
 *
 * Third-party piece restated here: Nim's std/heapqueue (used by
 * deflate.nim:47-75 through `HeapQueue[Node]`), which is a port of CPython's
 * heapq (push = append + sift towards root; pop = move last to root, sink to a
 * leaf along smaller children, then sift back up).  tests/test_oracle.py
 * cross-checks this against Python's own heapq.
 */
#include "zippy_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stddef.h>

/* ------------------------------------------------------------------ */
/* internal.nim:9-24 constants                                          */
/* ------------------------------------------------------------------ */
#define MAX_CODE_LENGTH 15
#define MAX_LITLEN_CODES 286
#define MAX_DISTANCE_CODES 30
#define MAX_FIXED_LITLEN_CODES 288
#define MAX_WINDOW_SIZE 32768
#define MAX_UNCOMPRESSED_BLOCK_SIZE 65535
#define MAX_BLOCK_SIZE 4194304
#define FIRST_LENGTH_CODE_INDEX 257
#define BASE_MATCH_LEN 3
#define MIN_MATCH_LEN 4
#define MAX_MATCH_LEN 258
#define MAX_LITERAL_LENGTH 32767 /* uint16.high shr 1, internal.nim:24 */

/* RFC 1951 3.2.5 tables (internal.nim:26-107) -- generated, not transcribed. */
static uint16_t base_lengths[29];
static uint8_t base_lengths_extra[29];
static uint8_t base_length_indices[256]; /* index by (length - 3), internal.nim:46-73 */
static uint16_t base_distances[30];
static uint8_t base_distance_extra[30];
static uint8_t distance_codes_lut[256]; /* internal.nim:225-242 */
static const uint8_t clcl_order[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                       11, 4,  12, 3, 13, 2, 14, 1, 15};

static uint8_t fixed_litlen_lens[MAX_FIXED_LITLEN_CODES];
static uint16_t fixed_litlen_codes[MAX_FIXED_LITLEN_CODES];
static uint8_t fixed_dist_lens[MAX_DISTANCE_CODES];
static uint16_t fixed_dist_codes[MAX_DISTANCE_CODES];
static uint32_t crc_tables[8][256];

typedef struct {
  int good, lazy, nice, chain;
} compression_config;

/* internal.nim:177-189 (zlib's configuration_table) */
static const compression_config configuration_table[10] = {
    {0, 0, 0, 0},       {4, 4, 8, 4},        {4, 5, 16, 8},      {4, 6, 32, 32},
    {4, 4, 16, 16},     {8, 16, 32, 32},     {8, 16, 128, 128},  {8, 32, 256, 256},
    {32, 128, 258, 1024}, {32, 258, 258, 4096}};

static uint16_t reverse_bits16(uint16_t v) {
  v = (uint16_t)(((v & 0xaaaa) >> 1) | ((v & 0x5555) << 1));
  v = (uint16_t)(((v & 0xcccc) >> 2) | ((v & 0x3333) << 2));
  v = (uint16_t)(((v & 0xf0f0) >> 4) | ((v & 0x0f0f) << 4));
  return (uint16_t)((v >> 8) | (v << 8));
}

/* internal.nim:133-149 makeCodes (canonical, bit-reversed for LSB-first) */
static void make_codes(const uint8_t *lens, int n, uint16_t *codes) {
  unsigned counts[16] = {0}, next[16] = {0};
  for (int i = 0; i < n; i++) counts[lens[i]]++;
  counts[0] = 0;
  for (int i = 1; i <= MAX_CODE_LENGTH; i++) next[i] = (next[i - 1] + counts[i - 1]) << 1;
  for (int i = 0; i < n; i++) {
    codes[i] = 0;
    if (lens[i] != 0) {
      codes[i] = (uint16_t)(reverse_bits16((uint16_t)next[lens[i]]) >> (16 - lens[i]));
      next[lens[i]]++;
    }
  }
}

static int tables_ready = 0;
static void init_tables(void) {
  if (tables_ready) return;
  /* lengths: codes 257..284 in groups of 4 with 0,0(x2 groups),1,2,3,4,5 extra
   * bits; code 285 = 258 with 0 extra bits (RFC 1951 3.2.5). */
  int len = 3;
  for (int i = 0; i < 28; i++) {
    int extra = i < 8 ? 0 : (i - 4) / 4;
    base_lengths[i] = (uint16_t)len;
    base_lengths_extra[i] = (uint8_t)extra;
    len += 1 << extra;
  }
  base_lengths[28] = 258;
  base_lengths_extra[28] = 0;
  for (int l = 3; l <= 258; l++) {
    int idx = 0;
    for (int i = 0; i < 28; i++)
      if (base_lengths[i] <= l) idx = i;
    if (l == 258) idx = 28;
    base_length_indices[l - 3] = (uint8_t)idx;
  }
  int dist = 1;
  for (int i = 0; i < 30; i++) {
    int extra = i < 4 ? 0 : (i - 2) / 2;
    base_distances[i] = (uint16_t)dist;
    base_distance_extra[i] = (uint8_t)extra;
    dist += 1 << extra;
  }
  for (int v = 0; v < 256; v++) {
    int idx = 0;
    for (int i = 0; i < 30; i++)
      if (base_distances[i] <= v + 1) idx = i;
    distance_codes_lut[v] = (uint8_t)idx;
  }
  /* internal.nim:151-175 fixed codes */
  for (int i = 0; i < MAX_FIXED_LITLEN_CODES; i++)
    fixed_litlen_lens[i] = (uint8_t)(i <= 143 ? 8 : i <= 255 ? 9 : i <= 279 ? 7 : 8);
  make_codes(fixed_litlen_lens, MAX_FIXED_LITLEN_CODES, fixed_litlen_codes);
  for (int i = 0; i < MAX_DISTANCE_CODES; i++) fixed_dist_lens[i] = 5;
  make_codes(fixed_dist_lens, MAX_DISTANCE_CODES, fixed_dist_codes);
  /* crc.nim:6-23 */
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int j = 0; j < 8; j++) c = (c >> 1) ^ ((c & 1) * 0xedb88320u);
    crc_tables[0][i] = c;
  }
  for (int i = 0; i < 256; i++)
    for (int t = 1; t < 8; t++)
      crc_tables[t][i] = (crc_tables[t - 1][i] >> 8) ^ crc_tables[0][crc_tables[t - 1][i] & 255];
  tables_ready = 1;
}
__attribute__((constructor)) static void zo_ctor(void) { init_tables(); }

/* internal.nim:224-249 */
static uint16_t distance_code_index(uint16_t value) {
  if (value < 256) return distance_codes_lut[value];
  if ((value >> 7) < 256) return (uint16_t)(distance_codes_lut[value >> 7] + 14);
  return (uint16_t)(distance_codes_lut[value >> 14] + 28);
}

static uint32_t read32(const uint8_t *p, size_t i) {
  uint32_t v;
  memcpy(&v, p + i, 4);
  return v;
}
static uint64_t read64(const uint8_t *p, size_t i) {
  uint64_t v;
  memcpy(&v, p + i, 8);
  return v;
}

/* internal.nim:251-270: length of the common prefix of src[s1..] and src[s2..],
 * s2 bounded by limit. */
static int determine_match_length(const uint8_t *src, size_t s1, size_t s2, size_t limit) {
  int result = 0;
  while (s2 + 8 <= limit) {
    uint64_t x = read64(src, s2) ^ read64(src, s1 + (size_t)result);
    if (x != 0) return result + (__builtin_ctzll(x) >> 3);
    s2 += 8;
    result += 8;
  }
  while (s2 < limit) {
    if (src[s2] != src[s1 + (size_t)result]) return result;
    s2++;
    result++;
  }
  return result;
}

/* ------------------------------------------------------------------ */
/* buffers                                                              */
/* ------------------------------------------------------------------ */
void zo_free(void *p) { free(p); }

/* Grow capacity (zero-filled, like Nim's setLen on a string). */
static int buf_reserve(zo_buf *b, size_t need) {
  if (need <= b->cap) return 0;
  size_t ncap = b->cap ? b->cap : 64;
  while (ncap < need) ncap *= 2;
  uint8_t *p = (uint8_t *)realloc(b->data, ncap);
  if (!p) return -1;
  memset(p + b->cap, 0, ncap - b->cap);
  b->data = p;
  b->cap = ncap;
  return 0;
}

static int buf_append(zo_buf *b, const void *src, size_t n) {
  if (buf_reserve(b, b->len + n + 8)) return -1;
  memcpy(b->data + b->len, src, n);
  b->len += n;
  return 0;
}
static int buf_push(zo_buf *b, uint8_t v) { return buf_append(b, &v, 1); }

/* ------------------------------------------------------------------ */
/* bitstreams.nim:84-123 BitStreamWriter                                */
/* ------------------------------------------------------------------ */
typedef struct {
  zo_buf *dst;
  size_t pos;
  int bit_pos;
  int err;
} bit_writer;

static void bw_inc_pos(bit_writer *b, size_t bits) {
  b->pos += (bits + (size_t)b->bit_pos) >> 3;
  b->bit_pos = (int)((bits + (size_t)b->bit_pos) & 7);
}

/* bitstreams.nim:88-104: OR value<<bitPos into the zero-initialised tail. */
static void bw_add_bits(bit_writer *b, uint32_t value, int bit_len) {
  if (buf_reserve(b->dst, b->pos + 8)) {
    b->err = ZO_ERR_NOMEM;
    return;
  }
  uint64_t v = (uint64_t)value & ((1ull << bit_len) - 1);
  uint64_t cur = (uint64_t)read32(b->dst->data, b->pos) | (v << b->bit_pos);
  memcpy(b->dst->data + b->pos, &cur, 8);
  bw_inc_pos(b, (size_t)bit_len);
}

/* bitstreams.nim:106-119 */
static void bw_add_bytes(bit_writer *b, const uint8_t *src, size_t src_pos, size_t len) {
  if (b->bit_pos != 0) {
    b->err = ZO_ERR_BYTE_BOUNDARY;
    return;
  }
  if (buf_reserve(b->dst, b->pos + len + 8)) {
    b->err = ZO_ERR_NOMEM;
    return;
  }
  memcpy(b->dst->data + b->pos, src + src_pos, len);
  bw_inc_pos(b, len * 8);
}

/* bitstreams.nim:121-123 */
static void bw_skip_remaining_bits(bit_writer *b) {
  if (b->bit_pos > 0) bw_inc_pos(b, (size_t)(8 - b->bit_pos));
}

/* ------------------------------------------------------------------ */
/* token stream                                      */
/* ------------------------------------------------------------------ */
typedef struct {
  uint16_t *data;
  size_t len, cap;
  int err;
} tokvec;

static void tok_push(tokvec *t, uint16_t v) {
  if (t->len == t->cap) {
    size_t ncap = t->cap ? t->cap * 2 : 1024;
    uint16_t *p = (uint16_t *)realloc(t->data, ncap * sizeof(uint16_t));
    if (!p) {
      t->err = ZO_ERR_NOMEM;
      return;
    }
    t->data = p;
    t->cap = ncap;
  }
  t->data[t->len++] = v;
}

/* snappy.nim:33-47 / lz77.nim:19-33 addLiteral */
static void add_literal(tokvec *enc, zo_block_metadata *meta, const uint8_t *src, size_t start,
                        size_t length) {
  for (size_t i = 0; i < length; i++) meta->litlen_freq[src[start + i]]++;
  meta->num_literals += (int64_t)length;
  size_t remaining = length;
  while (remaining > 0) {
    size_t added = remaining < MAX_LITERAL_LENGTH ? remaining : MAX_LITERAL_LENGTH;
    tok_push(enc, (uint16_t)added);
    remaining -= added;
  }
}

/* snappy.nim:49-64 / lz77.nim:35-50 addCopy */
static void add_copy(tokvec *enc, zo_block_metadata *meta, size_t offset, size_t length) {
  uint16_t length_index = base_length_indices[length - BASE_MATCH_LEN];
  uint16_t dist_index = distance_code_index((uint16_t)(offset - 1));
  meta->litlen_freq[length_index + FIRST_LENGTH_CODE_INDEX]++;
  meta->distance_freq[dist_index]++;
  tok_push(enc, (uint16_t)(((length_index << 8) | dist_index) | (1u << 15)));
  tok_push(enc, (uint16_t)offset);
  tok_push(enc, (uint16_t)length);
}

/* ------------------------------------------------------------------ */
/* snappy.nim:12-136 encodeFragment (BestSpeed matcher)                 */
/* ------------------------------------------------------------------ */
#define MAX_COMPRESS_TABLE_SIZE (1 << 14)

static void encode_fragment(tokvec *enc, zo_block_metadata *meta, const uint8_t *src, size_t start,
                            size_t bytes_to_read, uint16_t *table) {
  const size_t ip_end = start + bytes_to_read;
  size_t ip = start, next_emit = start;
  size_t table_size = 256;
  int shift = 24;
  while (table_size < MAX_COMPRESS_TABLE_SIZE && table_size < bytes_to_read) { /* :24-29 */
    table_size <<= 1;
    shift--;
  }
  memset(table, 0, table_size * sizeof(uint16_t)); /* :31 */

#define HASH(v) ((uint32_t)((uint32_t)(v) * 0x1e35a7bdu) >> shift) /* :70-71 */
#define U32_AT(v, off) ((uint32_t)(((v) >> (8 * (off))) & 0xffffffffu))

  if (bytes_to_read >= 15) { /* :76 */
    const size_t ip_limit = start + bytes_to_read - 15;
    ip++;
    uint32_t next_hash = HASH(read32(src, ip));
    for (;;) {
      size_t skip_bytes = 32, next_ip = ip, candidate;
      for (;;) { /* :86-101 probe loop */
        ip = next_ip;
        uint32_t h = next_hash;
        size_t bytes_between = skip_bytes >> 5;
        skip_bytes++;
        next_ip = ip + bytes_between;
        if (next_ip > ip_limit) goto emit_remainder;
        next_hash = HASH(read32(src, next_ip));
        candidate = start + table[h];
        table[h] = (uint16_t)(ip - start);
        if (read32(src, ip) == read32(src, candidate)) break;
      }
      add_literal(enc, meta, src, next_emit, ip - next_emit); /* :103 */

      uint64_t input_bytes;
      for (;;) { /* :108-131 */
        size_t limit = ip_end < ip + MAX_MATCH_LEN ? ip_end : ip + MAX_MATCH_LEN;
        size_t matched = 4 + (size_t)determine_match_length(src, candidate + 4, ip + 4, limit);
        size_t offset = ip - candidate;
        ip += matched;
        add_copy(enc, meta, offset, matched);

        size_t insert_tail = ip - 1;
        next_emit = ip;
        if (ip >= ip_limit) goto emit_remainder;
        input_bytes = read64(src, insert_tail);
        uint32_t prev_hash = HASH(U32_AT(input_bytes, 0));
        uint32_t cur_hash = HASH(U32_AT(input_bytes, 1));
        table[prev_hash] = (uint16_t)(ip - start - 1);
        candidate = start + table[cur_hash];
        uint32_t candidate_bytes = read32(src, candidate);
        table[cur_hash] = (uint16_t)(ip - start);
        if (U32_AT(input_bytes, 1) != candidate_bytes) break;
      }
      next_hash = HASH(U32_AT(input_bytes, 2));
      ip++;
    }
  }
emit_remainder: /* :66-68 */
  if (next_emit < ip_end) add_literal(enc, meta, src, next_emit, ip_end - next_emit);
#undef HASH
#undef U32_AT
}

/* snappy.nim:138-163 */
static void encode_snappy(tokvec *enc, zo_block_metadata *meta, const uint8_t *src,
                          size_t block_start, size_t block_len) {
  meta->litlen_freq[256] = 1;
  uint16_t *table = (uint16_t *)malloc(MAX_COMPRESS_TABLE_SIZE * sizeof(uint16_t));
  if (!table) {
    enc->err = ZO_ERR_NOMEM;
    return;
  }
  size_t pos = block_start;
  while (pos < block_start + block_len) {
    size_t fragment_size = block_start + block_len - pos;
    size_t bytes_to_read = fragment_size < MAX_WINDOW_SIZE ? fragment_size : MAX_WINDOW_SIZE;
    encode_fragment(enc, meta, src, pos, bytes_to_read, table);
    pos += bytes_to_read;
  }
  free(table);
}

/* ------------------------------------------------------------------ */
/* lz77.nim:10-130 encodeLz77 (hash-chain matcher; greedy, no lazy)     */
/* ------------------------------------------------------------------ */
#define HASH_BITS 17
#define HASH_SIZE (1 << HASH_BITS)

static void encode_lz77(tokvec *enc, compression_config config, zo_block_metadata *meta,
                        const uint8_t *src, size_t block_start, size_t block_len) {
  meta->litlen_freq[256] = 1; /* :52 */
  if (MIN_MATCH_LEN >= block_len) { /* :54-56 */
    add_literal(enc, meta, src, block_start, block_len);
    return;
  }
  uint16_t *head = (uint16_t *)calloc(HASH_SIZE, sizeof(uint16_t));       /* :63 */
  uint16_t *chain = (uint16_t *)calloc(MAX_WINDOW_SIZE, sizeof(uint16_t)); /* :64 */
  if (!head || !chain) {
    free(head);
    free(chain);
    enc->err = ZO_ERR_NOMEM;
    return;
  }
  const size_t block_end = block_start + block_len;
  size_t pos = block_start, literal_len = 0;
  uint32_t hash;
  uint16_t window_pos;

#define HASH4(p) ((uint32_t)(read32(src, (p)) * 0x1e35a7bdu) >> (32 - HASH_BITS)) /* :66-67 */
#define UPDATE_CHAIN()             \
  do {                             \
    chain[window_pos] = head[hash]; \
    head[hash] = window_pos;       \
  } while (0) /* :69-71 */

  while (pos < block_end) {
    if (pos + MIN_MATCH_LEN >= block_end) { /* :74-76 */
      add_literal(enc, meta, src, pos - literal_len, block_end - pos + literal_len);
      break;
    }
    window_pos = (uint16_t)((pos - block_start) & (MAX_WINDOW_SIZE - 1)); /* :78 */
    hash = HASH4(pos);
    UPDATE_CHAIN();

    uint16_t hash_pos = chain[window_pos];
    size_t limit = block_end < pos + MAX_MATCH_LEN ? block_end : pos + MAX_MATCH_LEN;
    int tries = config.chain;
    int prev_offset = 0, longest_offset = 0, longest_len = 0;
    while (tries > 0 && hash_pos != 0) { /* :88-112 */
      tries--;
      int offset;
      if (hash_pos <= window_pos)
        offset = (int)window_pos - (int)hash_pos;
      else
        offset = (int)window_pos - (int)hash_pos + MAX_WINDOW_SIZE;
      if (offset <= 0 || offset < prev_offset) break;
      prev_offset = offset;
      int match_len = determine_match_length(src, pos - (size_t)offset, pos, limit);
      if (match_len > longest_len) {
        if (match_len >= config.good) tries >>= 2;
        longest_len = match_len;
        longest_offset = offset;
      }
      if (longest_len >= config.nice || hash_pos == chain[hash_pos]) break;
      hash_pos = chain[hash_pos];
    }

    if (longest_len > MIN_MATCH_LEN) { /* :114 */
      if (literal_len > 0) {
        add_literal(enc, meta, src, pos - literal_len, literal_len);
        literal_len = 0;
      }
      add_copy(enc, meta, (size_t)longest_offset, (size_t)longest_len);
      for (int i = 1; i < longest_len; i++) { /* :121-126 */
        pos++;
        window_pos = (uint16_t)(pos & (MAX_WINDOW_SIZE - 1)); /* absolute pos, as :123 */
        if (pos + MIN_MATCH_LEN < block_end) {
          hash = HASH4(pos);
          UPDATE_CHAIN();
        }
      }
    } else {
      literal_len++;
    }
    pos++;
  }
  free(head);
  free(chain);
#undef HASH4
#undef UPDATE_CHAIN
}

/* deflate.nim:153-177 encodeAllLiterals (HuffmanOnly) */
static void encode_all_literals(tokvec *enc, zo_block_metadata *meta, const uint8_t *src,
                                size_t start, size_t len) {
  for (size_t i = 0; i < len; i++) meta->litlen_freq[src[start + i]]++;
  size_t a = len / MAX_LITERAL_LENGTH, b = len % MAX_LITERAL_LENGTH;
  for (size_t i = 0; i < a; i++) tok_push(enc, MAX_LITERAL_LENGTH);
  if (b > 0) tok_push(enc, (uint16_t)b);
  meta->litlen_freq[256] = 1;
  meta->num_literals = (int64_t)len;
}

static void encode_block(tokvec *enc, zo_block_metadata *meta, const uint8_t *src,
                         size_t block_start, size_t block_len, int level) {
  /* deflate.nim:243-272 */
  if (level == -2)
    encode_all_literals(enc, meta, src, block_start, block_len);
  else if (level == 1)
    encode_snappy(enc, meta, src, block_start, block_len);
  else
    encode_lz77(enc, configuration_table[level == -1 ? 6 : level], meta, src, block_start,
                block_len);
}

int zo_encode_block_tokens(const uint8_t *src, size_t block_start, size_t block_len, int level,
                           uint16_t **tokens, size_t *num_tokens, zo_block_metadata *meta) {
  init_tables();
  if (level < -2 || level > 9 || level == 0) return ZO_ERR_INVALID_LEVEL;
  tokvec enc = {0};
  memset(meta, 0, sizeof(*meta));
  encode_block(&enc, meta, src, block_start, block_len, level);
  if (enc.err) {
    free(enc.data);
    return enc.err;
  }
  *tokens = enc.data;
  *num_tokens = enc.len;
  return ZO_OK;
}

/* ------------------------------------------------------------------ */
/* deflate.nim:13-151 huffmanCodes                                      */
/* ------------------------------------------------------------------ */
typedef struct {
  int symbol; /* -1 for internal nodes */
  int64_t freq; /* re-used for the leaf depth after the tree is built (:71) */
  int left, right;
} hnode;

/* Nim std/heapqueue (port of CPython heapq); `<` compares freq only
 * (deflate.nim:10-11). */
typedef struct {
  int *data;
  int len;
  const hnode *nodes;
} heapq;

static int heap_less(const heapq *h, int a, int b) { return h->nodes[a].freq < h->nodes[b].freq; }

/* heapqueue.nim siftup(heap, startpos, p) == heapq._siftdown */
static void heap_sift_to_root(heapq *h, int startpos, int pos) {
  int newitem = h->data[pos];
  while (pos > startpos) {
    int parentpos = (pos - 1) >> 1;
    int parent = h->data[parentpos];
    if (heap_less(h, newitem, parent)) {
      h->data[pos] = parent;
      pos = parentpos;
    } else {
      break;
    }
  }
  h->data[pos] = newitem;
}

/* heapqueue.nim siftdownToBottom == heapq._siftup */
static void heap_sink_to_bottom(heapq *h, int pos) {
  int endpos = h->len, startpos = pos;
  int newitem = h->data[pos];
  int childpos = 2 * pos + 1;
  while (childpos < endpos) {
    int rightpos = childpos + 1;
    if (rightpos < endpos && !heap_less(h, h->data[childpos], h->data[rightpos]))
      childpos = rightpos;
    h->data[pos] = h->data[childpos];
    pos = childpos;
    childpos = 2 * pos + 1;
  }
  h->data[pos] = newitem;
  heap_sift_to_root(h, startpos, pos);
}

static void heap_push(heapq *h, int item) {
  h->data[h->len++] = item;
  heap_sift_to_root(h, 0, h->len - 1);
}

static int heap_pop(heapq *h) {
  int lastelt = h->data[--h->len];
  if (h->len > 0) {
    int result = h->data[0];
    h->data[0] = lastelt;
    heap_sink_to_bottom(h, 0);
    return result;
  }
  return lastelt;
}

/* deflate.nim:65-73 visit */
static void visit(hnode *nodes, int n, int level, int limit, int *needs_limiting) {
  if (nodes[n].symbol == -1) {
    visit(nodes, nodes[n].left, level + 1, limit, needs_limiting);
    visit(nodes, nodes[n].right, level + 1, limit, needs_limiting);
  } else {
    nodes[n].freq = level;
    if (level > limit) *needs_limiting = 1;
  }
}

/* deflate.nim:103-121 quickSort (not stable: order among equal depths matters
 * for which symbol receives which length, so it is restated exactly). */
static void quick_sort(int *a, const hnode *nodes, int inl, int inr) {
  int r = inr, l = inl;
  int n = r - l + 1;
  if (n < 2) return;
  int64_t p = nodes[a[l + 3 * n / 4]].freq;
  while (l <= r) {
    if (nodes[a[l]].freq < p) {
      l++;
    } else if (nodes[a[r]].freq > p) {
      r--;
    } else {
      int t = a[l];
      a[l] = a[r];
      a[r] = t;
      l++;
      r--;
    }
  }
  quick_sort(a, nodes, inl, r);
  quick_sort(a, nodes, l, inr);
}

int zo_huffman_codes(const uint32_t *freq, int num_freq, int min_codes, int code_length_limit,
                     uint16_t *codes, uint8_t *lens) {
  init_tables();
  int highest = 0, used = 0;
  for (int s = 0; s < num_freq; s++)
    if (freq[s] > 0) {
      highest = s;
      used++;
    }
  int num_codes = (highest > min_codes ? highest : min_codes) + 1; /* :29 */
  memset(codes, 0, (size_t)num_codes * sizeof(uint16_t));
  memset(lens, 0, (size_t)num_codes);

  if (used == 0) { /* :34-36 */
    lens[0] = 1;
    lens[1] = 1;
  } else if (used == 1) { /* :37-45 */
    for (int i = 0; i < num_freq; i++)
      if (freq[i] != 0) {
        lens[i] = 1;
        if (i == 0)
          lens[1] = 1;
        else
          lens[0] = 1;
        break;
      }
  } else {
    hnode nodes[2 * MAX_FIXED_LITLEN_CODES];
    int leaves[MAX_FIXED_LITLEN_CODES];
    int heap_data[MAX_FIXED_LITLEN_CODES];
    int n = 0;
    for (int s = 0; s < num_freq; s++) /* :48-50 */
      if (freq[s] > 0) {
        nodes[n].symbol = s;
        nodes[n].freq = (int64_t)freq[s];
        nodes[n].left = nodes[n].right = -1;
        leaves[n] = n;
        n++;
      }
    int total = n;
    heapq heap = {heap_data, 0, nodes};
    for (int i = 0; i < n; i++) heap_push(&heap, i); /* :54-55 */
    while (heap.len >= 2) {                          /* :57-63 */
      int left = heap_pop(&heap);
      int right = heap_pop(&heap);
      nodes[total].symbol = -1;
      nodes[total].left = left;
      nodes[total].right = right;
      nodes[total].freq = nodes[left].freq + nodes[right].freq;
      heap_push(&heap, total);
      total++;
    }
    int needs_limiting = 0;
    visit(nodes, heap.data[0], 0, code_length_limit, &needs_limiting); /* :75 */

    if (needs_limiting) { /* :78-131 */
      int longest = 0;
      for (int i = 0; i < n; i++)
        if (nodes[i].freq > longest) longest = (int)nodes[i].freq;
      int histogram[2 * MAX_FIXED_LITLEN_CODES];
      memset(histogram, 0, sizeof(histogram));
      for (int i = 0; i < n; i++) histogram[nodes[i].freq]++;
      int i = longest;
      while (i > code_length_limit) { /* :88-101 */
        if (histogram[i] == 0) {
          i--;
          continue;
        }
        int j = i - 2;
        while (j > 0 && histogram[j] == 0) j--;
        histogram[i] -= 2;
        histogram[i - 1]++;
        histogram[j + 1] += 2;
        histogram[j]--;
      }
      quick_sort(leaves, nodes, 0, n - 1); /* :123 */
      int code_len = 1;
      for (int k = 0; k < n; k++) { /* :125-131 */
        while (histogram[code_len] == 0) code_len++;
        nodes[leaves[k]].freq = code_len;
        histogram[code_len]--;
      }
    }
    for (int i = 0; i < n; i++) lens[nodes[i].symbol] = (uint8_t)nodes[i].freq; /* :133-134 */
  }

  /* :136-149 canonical codes, bit-reversed.  NB the reference counts symbols
   * per length in a uint8 array (wraps at 256); wider counters are
   * used here on purpose. */
  unsigned histogram[MAX_CODE_LENGTH + 1] = {0}, next_code[MAX_CODE_LENGTH + 1] = {0};
  for (int i = 0; i < num_codes; i++) histogram[lens[i]]++;
  histogram[0] = 0;
  for (int i = 1; i <= MAX_CODE_LENGTH; i++) next_code[i] = (next_code[i - 1] + histogram[i - 1]) << 1;
  for (int i = 0; i < num_codes; i++)
    if (lens[i] != 0) {
      codes[i] = (uint16_t)(reverse_bits16((uint16_t)next_code[lens[i]]) >> (16 - lens[i]));
      next_code[lens[i]]++;
    }
  return num_codes;
}

/* ------------------------------------------------------------------ */
/* deflate.nim:179-205 addNoCompressionBlock                            */
/* ------------------------------------------------------------------ */
/* Block-parallel variant (BASELINE.json config 5, not in the reference): the same driver with
 * deflate.nim:228's block size as a parameter, and a record of where every deflate block begins.
 * Each block restarts the matcher (deflate.nim:243-272 call it per block), so a block can be
 * decoded on its own from its bit position. */
static _Thread_local size_t opt_block_size = MAX_BLOCK_SIZE;
static _Thread_local struct {
  zo_block_entry *e;
  size_t n, cap;
  int on, err;
} opt_index;

static void index_note(const bit_writer *b, size_t out_off) {
  if (!opt_index.on) return;
  if (opt_index.n == opt_index.cap) {
    size_t cap = opt_index.cap ? opt_index.cap * 2 : 64;
    zo_block_entry *e = (zo_block_entry *)realloc(opt_index.e, cap * sizeof(*e));
    if (!e) {
      opt_index.err = 1;
      return;
    }
    opt_index.e = e;
    opt_index.cap = cap;
  }
  opt_index.e[opt_index.n].bit_off = (uint64_t)b->pos * 8 + (uint64_t)b->bit_pos;
  opt_index.e[opt_index.n].out_off = out_off;
  opt_index.n++;
}

static void add_no_compression_block(bit_writer *b, const uint8_t *src, size_t block_start,
                                     size_t block_len, int final_block) {
  size_t count = (block_len + MAX_UNCOMPRESSED_BLOCK_SIZE - 1) / MAX_UNCOMPRESSED_BLOCK_SIZE;
  if (count < 1) count = 1;
  for (size_t num = 0; num < count; num++) {
    int ufinal = num == count - 1;
    size_t ustart = block_start + num * MAX_UNCOMPRESSED_BLOCK_SIZE;
    size_t ulen = block_start + block_len - ustart;
    if (ulen > MAX_UNCOMPRESSED_BLOCK_SIZE) ulen = MAX_UNCOMPRESSED_BLOCK_SIZE;
    index_note(b, ustart);
    bw_add_bits(b, (final_block && ufinal) ? 1 : 0, 1);
    bw_add_bits(b, 0, 2);
    bw_skip_remaining_bits(b);
    bw_add_bits(b, (uint32_t)ulen, 16);
    bw_add_bits(b, (uint32_t)(MAX_UNCOMPRESSED_BLOCK_SIZE - ulen), 16);
    if (ulen > 0) bw_add_bytes(b, src, ustart, ulen);
  }
}

/* ------------------------------------------------------------------ */
/* deflate.nim:207-467 deflate                                          */
/* ------------------------------------------------------------------ */
int zo_deflate(const uint8_t *src, size_t len, int level, zo_buf *out) {
  init_tables();
  if (level < -2 || level > 9) return ZO_ERR_INVALID_LEVEL;

  bit_writer b = {out, out->len, 0, 0};

  if (level == 0) { /* :214-226 */
    size_t block_count = (len + MAX_UNCOMPRESSED_BLOCK_SIZE - 1) / MAX_UNCOMPRESSED_BLOCK_SIZE;
    if (block_count < 1) block_count = 1;
    for (size_t num = 0; num < block_count; num++) {
      int final_block = num == block_count - 1;
      size_t block_start = num * MAX_UNCOMPRESSED_BLOCK_SIZE;
      size_t block_len = len - block_start;
      if (block_len > MAX_UNCOMPRESSED_BLOCK_SIZE) block_len = MAX_UNCOMPRESSED_BLOCK_SIZE;
      add_no_compression_block(&b, src, block_start, block_len, final_block);
    }
    if (b.err) return b.err;
    index_note(&b, len);
    if (buf_reserve(out, b.pos)) return ZO_ERR_NOMEM;
    out->len = b.pos;
    return ZO_OK;
  }

  const size_t max_block = opt_block_size;
  size_t block_count = (len + max_block - 1) / max_block; /* :228 */
  if (block_count < 1) block_count = 1;

  tokvec enc = {0};
  uint8_t *rle = NULL;
  size_t rle_cap = 0;
  int status = ZO_OK;

  for (size_t num = 0; num < block_count && status == ZO_OK; num++) {
    size_t block_start = num * max_block;
    size_t block_len = len - block_start;
    if (block_len > max_block) block_len = max_block;
    int final_block = num == block_count - 1;

    enc.len = 0;
    zo_block_metadata meta;
    memset(&meta, 0, sizeof(meta));
    encode_block(&enc, &meta, src, block_start, block_len, level);
    if (enc.err) {
      status = enc.err;
      break;
    }

    /* :274-277 "almost all literals" -> stored.  float32 multiply + truncation. */
    if (level != -2 && meta.num_literals >= (int64_t)((float)block_len * 0.98f)) {
      add_no_compression_block(&b, src, block_start, block_len, final_block);
      continue;
    }

    int use_fixed = level <= 6 && block_len <= 2048; /* :280 */
    uint16_t litlen_codes_buf[MAX_FIXED_LITLEN_CODES], dist_codes_buf[MAX_DISTANCE_CODES + 2];
    uint8_t litlen_lens_buf[MAX_FIXED_LITLEN_CODES], dist_lens_buf[MAX_DISTANCE_CODES + 2];
    const uint16_t *litlen_codes, *dist_codes;
    const uint8_t *litlen_lens, *dist_lens;
    int n_litlen, n_dist;
    if (use_fixed) {
      litlen_codes = fixed_litlen_codes;
      litlen_lens = fixed_litlen_lens;
      n_litlen = MAX_FIXED_LITLEN_CODES;
      dist_codes = fixed_dist_codes;
      dist_lens = fixed_dist_lens;
      n_dist = MAX_DISTANCE_CODES;
    } else {
      n_litlen = zo_huffman_codes(meta.litlen_freq, MAX_LITLEN_CODES, 257, MAX_CODE_LENGTH,
                                  litlen_codes_buf, litlen_lens_buf);
      n_dist = zo_huffman_codes(meta.distance_freq, MAX_DISTANCE_CODES, 2, MAX_CODE_LENGTH,
                                dist_codes_buf, dist_lens_buf);
      litlen_codes = litlen_codes_buf;
      litlen_lens = litlen_lens_buf;
      dist_codes = dist_codes_buf;
      dist_lens = dist_lens_buf;
    }

    if (use_fixed) { /* :296-298 */
      index_note(&b, block_start);
      bw_add_bits(&b, final_block ? 1 : 0, 1);
      bw_add_bits(&b, 1, 2);
    } else {
      uint8_t code_lengths[MAX_LITLEN_CODES + MAX_DISTANCE_CODES + 4];
      int num_codes = n_litlen + n_dist;
      memcpy(code_lengths, litlen_lens, (size_t)n_litlen);
      memcpy(code_lengths + n_litlen, dist_lens, (size_t)n_dist);

      /* :313-350 run-length encode the code lengths */
      if (rle_cap < 1024) {
        rle = (uint8_t *)realloc(rle, 1024);
        rle_cap = 1024;
      }
      size_t rle_len = 0;
      {
        int i = 0;
        while (i < num_codes) {
          int repeat = 0;
          while (i + repeat + 1 < num_codes && code_lengths[i + repeat + 1] == code_lengths[i])
            repeat++;
          if (code_lengths[i] == 0 && repeat >= 2) {
            repeat++; /* initial zero */
            if (repeat <= 10) {
              rle[rle_len++] = 17;
              rle[rle_len++] = (uint8_t)(repeat - 3);
            } else {
              if (repeat > 138) repeat = 138;
              rle[rle_len++] = 18;
              rle[rle_len++] = (uint8_t)(repeat - 11);
            }
            i += repeat - 1;
          } else if (repeat >= 3) {
            int a = repeat / 6, bb = repeat % 6;
            rle[rle_len++] = code_lengths[i];
            for (int j = 0; j < a; j++) {
              rle[rle_len++] = 16;
              rle[rle_len++] = 3;
            }
            if (bb >= 3) {
              rle[rle_len++] = 16;
              rle[rle_len++] = (uint8_t)(bb - 3);
            } else {
              repeat -= bb;
            }
            i += repeat;
          } else {
            rle[rle_len++] = code_lengths[i];
          }
          i++;
        }
      }

      uint32_t cl_freq[19] = {0}; /* :352-360 */
      for (size_t i = 0; i < rle_len; i++) {
        cl_freq[rle[i]]++;
        if (rle[i] >= 16) i++;
      }
      uint16_t cl_codes[20];
      uint8_t cl_lens[20];
      zo_huffman_codes(cl_freq, 19, 19, 7, cl_codes, cl_lens); /* :362 */

      uint16_t clcl_ordered[19];
      for (int i = 0; i < 19; i++) clcl_ordered[i] = cl_lens[clcl_order[i]];
      int hclen = 19;
      while (clcl_ordered[hclen - 1] == 0) hclen--; /* :368-370 (2nd condition is constant) */
      hclen -= 4;

      int hlit = n_litlen - FIRST_LENGTH_CODE_INDEX;
      int hdist = n_dist - 1;

      index_note(&b, block_start);
      bw_add_bits(&b, final_block ? 1 : 0, 1); /* :376-383 */
      bw_add_bits(&b, 2, 2);
      bw_add_bits(&b, (uint32_t)hlit, 5);
      bw_add_bits(&b, (uint32_t)hdist, 5);
      bw_add_bits(&b, (uint32_t)hclen, 4);
      for (int i = 0; i < hclen + 4; i++) bw_add_bits(&b, clcl_ordered[i], 3);

      for (size_t i = 0; i < rle_len;) { /* :388-401 */
        uint8_t symbol = rle[i];
        bw_add_bits(&b, cl_codes[symbol], cl_lens[symbol]);
        i++;
        if (symbol == 16)
          bw_add_bits(&b, rle[i++], 2);
        else if (symbol == 17)
          bw_add_bits(&b, rle[i++], 3);
        else if (symbol == 18)
          bw_add_bits(&b, rle[i++], 7);
      }
    }

    { /* :403-466 token -> bits */
      size_t src_pos = block_start, enc_pos = 0;
      while (enc_pos < enc.len) {
        if (enc.data[enc_pos] & 0x8000) {
          uint16_t value = enc.data[enc_pos], offset = enc.data[enc_pos + 1],
                   length = enc.data[enc_pos + 2];
          unsigned length_index = (value >> 8) & 0x7f, distance_index = value & 0xff;
          int length_extra_bits = base_lengths_extra[length_index];
          uint64_t length_extra = (uint64_t)(length - base_lengths[length_index]);
          int distance_extra_bits = base_distance_extra[distance_index];
          uint64_t distance_extra = (uint64_t)(offset - base_distances[distance_index]);
          enc_pos += 3;
          src_pos += length;

          uint64_t buf = litlen_codes[length_index + 257];
          int bit_len = litlen_lens[length_index + 257];
          buf |= length_extra << bit_len;
          bit_len += length_extra_bits;
          buf |= (uint64_t)dist_codes[distance_index] << bit_len;
          bit_len += dist_lens[distance_index];
          buf |= distance_extra << bit_len;
          bit_len += distance_extra_bits;

          int first = bit_len < 32 ? bit_len : 32;
          bw_add_bits(&b, (uint32_t)buf, first);
          buf >>= first;
          bit_len -= first;
          if (bit_len > 0) bw_add_bits(&b, (uint32_t)buf, bit_len);
        } else {
          size_t literals = enc.data[enc_pos++];
          uint32_t buf = 0;
          int bit_len = 0;
          for (size_t k = 0; k < literals; k++) {
            int code_length = litlen_lens[src[src_pos]];
            if (bit_len + code_length > 32) {
              bw_add_bits(&b, buf, bit_len);
              buf = 0;
              bit_len = 0;
            }
            buf |= (uint32_t)litlen_codes[src[src_pos]] << bit_len;
            bit_len += code_length;
            src_pos++;
          }
          if (bit_len > 0) bw_add_bits(&b, buf, bit_len);
        }
      }
      if (enc_pos != enc.len || src_pos != block_start + block_len) {
        status = ZO_ERR_INVALID_BUFFER;
        break;
      }
    }
    if (litlen_lens[256] == 0) {
      status = ZO_ERR_COMPRESS_INTERNAL;
      break;
    }
    bw_add_bits(&b, litlen_codes[256], litlen_lens[256]); /* EOB :471 */
  }

  free(enc.data);
  free(rle);
  if (status != ZO_OK) return status;
  if (b.err) return b.err;
  index_note(&b, len);
  bw_skip_remaining_bits(&b);
  if (buf_reserve(out, b.pos + 8)) return ZO_ERR_NOMEM;
  out->len = b.pos;
  return ZO_OK;
}

/* ------------------------------------------------------------------ */
/* crc.nim / adler32.nim                                                */
/* ------------------------------------------------------------------ */
uint32_t zo_crc32(const uint8_t *src, size_t len) { /* crc.nim:29-51,53-72 */
  init_tables();
  uint32_t r = ~0u;
  size_t i = 0;
  for (size_t k = 0; k < len / 8; k++) {
    uint32_t one = read32(src, i) ^ r, two = read32(src, i + 4);
    r = crc_tables[7][one & 255] ^ crc_tables[6][(one >> 8) & 255] ^
        crc_tables[5][(one >> 16) & 255] ^ crc_tables[4][one >> 24] ^ crc_tables[3][two & 255] ^
        crc_tables[2][(two >> 8) & 255] ^ crc_tables[1][(two >> 16) & 255] ^
        crc_tables[0][two >> 24];
    i += 8;
  }
  for (; i < len; i++) r = crc_tables[0][(r ^ src[i]) & 255] ^ (r >> 8);
  return ~r;
}

uint32_t zo_adler32(const uint8_t *src, size_t len) { /* adler32.nim:19-63 */
  const size_t nmax = 5552;
  uint32_t s1 = 1, s2 = 0;
  size_t l = len, pos = 0;
  while (l >= nmax) {
    l -= nmax;
    for (size_t i = 0; i < nmax; i++) {
      s1 += src[pos++];
      s2 += s1;
    }
    s1 %= 65521;
    s2 %= 65521;
  }
  for (size_t i = 0; i < l; i++) {
    s1 += src[pos++];
    s2 += s1;
  }
  s1 %= 65521;
  s2 %= 65521;
  return (s2 << 16) | s1;
}

/* ------------------------------------------------------------------ */
/* bitstreams.nim:4-82 BitStreamReader                                  */
/* ------------------------------------------------------------------ */
typedef struct {
  const uint8_t *src;
  size_t len, pos;
  uint64_t bit_buffer;
  int bits_buffered; /* may go negative past the end (:62) */
} bit_reader;

/* bitstreams.nim:22-49.  With < 8 bytes left the reference loads the LAST 8
 * bytes of the buffer and shifts; restated as a zero-extended load (identical
 * for in-range bytes, defined for len < 8). */
static void br_fill(bit_reader *b) {
  if (b->bits_buffered < 0 || b->bits_buffered > 56) return; /* nothing can be added */
  size_t needed = (size_t)(64 - b->bits_buffered) / 8;
  size_t available = b->len - b->pos;
  size_t added = needed < available ? needed : available;
  uint64_t v = 0;
  if (available >= 8)
    v = read64(b->src, b->pos);
  else
    for (size_t i = 0; i < available; i++) v |= (uint64_t)b->src[b->pos + i] << (8 * i);
  b->pos += added;
  b->bit_buffer |= v << b->bits_buffered;
  b->bits_buffered += 8 * (int)added;
}

static uint16_t br_read_bits(bit_reader *b, int bits, int fill) { /* :51-62 */
  if (fill) br_fill(b);
  uint16_t r = (uint16_t)(b->bit_buffer & ((1u << bits) - 1));
  b->bit_buffer >>= bits;
  b->bits_buffered -= bits;
  return r;
}

static int br_read_bytes(bit_reader *b, uint8_t *dst, size_t len) { /* :64-76 */
  if (b->bits_buffered % 8 != 0) return ZO_ERR_BYTE_BOUNDARY;
  long offset = b->bits_buffered / 8; /* truncating division like Nim `div` */
  long start = (long)b->pos - offset;
  if (start < 0 || (size_t)start + len > b->len) return ZO_ERR_END_OF_BUFFER;
  memcpy(dst, b->src + start, len);
  b->pos = (size_t)start + len;
  b->bits_buffered = 0;
  b->bit_buffer = 0;
  return ZO_OK;
}

static void br_skip_remaining_bits(bit_reader *b) { /* :78-82 */
  int mod8 = b->bits_buffered % 8;
  if (mod8 != 0) {
    b->bits_buffered -= mod8;
    b->bit_buffer >>= mod8;
  }
}

/* ------------------------------------------------------------------ */
/* inflate.nim                                                          */
/* ------------------------------------------------------------------ */
#define FAST_BITS 9
#define FAST_MASK ((1 << FAST_BITS) - 1)

typedef struct { /* inflate.nim:14-19 */
  uint16_t first_code[16], first_symbol[16];
  uint32_t max_codes[17];
  uint16_t values[288];
  uint16_t fast[1 << FAST_BITS];
} huffman;

/* inflate.nim:24-65 */
static int init_huffman(huffman *h, const uint8_t *code_lengths, int n) {
  memset(h, 0, sizeof(*h));
  uint16_t histogram[17] = {0};
  for (int i = 0; i < n; i++) histogram[code_lengths[i]]++;
  histogram[0] = 0;
  for (int i = 1; i < 16; i++)
    if (histogram[i] > (1u << i)) return ZO_ERR_INVALID_BUFFER;

  uint32_t code = 0, next_code[16] = {0};
  uint16_t k = 0;
  for (int i = 1; i < 16; i++) {
    next_code[i] = code;
    h->first_code[i] = (uint16_t)code;
    h->first_symbol[i] = k;
    code += histogram[i];
    if (histogram[i] > 0 && code - 1 >= (1u << i)) return ZO_ERR_INVALID_BUFFER;
    h->max_codes[i] = code << (16 - i);
    code <<= 1;
    k = (uint16_t)(k + histogram[i]);
  }
  h->max_codes[16] = 1u << 16;

  for (int i = 0; i < n; i++) {
    uint8_t len = code_lengths[i];
    if (len > 0) {
      uint16_t symbol_id = (uint16_t)(next_code[len] - h->first_code[len] + h->first_symbol[len]);
      h->values[symbol_id] = (uint16_t)i;
      if (len <= FAST_BITS) {
        uint16_t fast = (uint16_t)((len << FAST_BITS) | i);
        unsigned kk = reverse_bits16((uint16_t)next_code[len]) >> (16 - len);
        while (kk < (1u << FAST_BITS)) {
          h->fast[kk] = fast;
          kk += 1u << len;
        }
      }
      next_code[len]++;
    }
  }
  return ZO_OK;
}

/* inflate.nim:67-91 */
static uint16_t decode_symbol_slow(bit_reader *b, const huffman *h) {
  uint16_t k = reverse_bits16((uint16_t)b->bit_buffer);
  unsigned code_length = FAST_BITS + 1;
  while (code_length < 17) {
    if ((uint32_t)k < h->max_codes[code_length]) break;
    code_length++;
  }
  if (code_length >= 16) return 0xffff;
  uint16_t symbol_id = (uint16_t)((k >> (16 - code_length)) - h->first_code[code_length] +
                                  h->first_symbol[code_length]);
  if (symbol_id >= 288) return 0xffff; /* defensive; unreachable for canonical codes */
  b->bit_buffer >>= code_length;
  b->bits_buffered -= (int)code_length;
  return h->values[symbol_id];
}

/* inflate.nim:93-102 */
static inline uint16_t decode_symbol(bit_reader *b, const huffman *h) {
  uint16_t fast = h->fast[b->bit_buffer & FAST_MASK];
  if (fast > 0) {
    unsigned code_length = fast >> FAST_BITS;
    b->bit_buffer >>= code_length;
    b->bits_buffered -= (int)code_length;
    return fast & FAST_MASK;
  }
  return decode_symbol_slow(b, h);
}

static int out_reserve(zo_buf *o, size_t need) {
  if (need <= o->cap) return 0;
  size_t ncap = o->cap ? o->cap : 256;
  while (ncap < need) ncap *= 2;
  uint8_t *p = (uint8_t *)realloc(o->data, ncap);
  if (!p) return -1;
  o->data = p;
  o->cap = ncap;
  return 0;
}

/* internal.nim:233-241 copy64: eight bytes, any alignment, source read before the store */
static inline void copy64(uint8_t *d, size_t to, size_t from) {
  uint64_t v;
  memcpy(&v, d + from, 8);
  memcpy(d + to, &v, 8);
}

/* inflate.nim:104-250 */
static int inflate_block(zo_buf *dst, bit_reader *b, size_t *op_io, int fixed_codes) {
  huffman lit, dist;
  int st;
  if (fixed_codes) { /* :111-113 */
    if ((st = init_huffman(&lit, fixed_litlen_lens, MAX_FIXED_LITLEN_CODES))) return st;
    if ((st = init_huffman(&dist, fixed_dist_lens, MAX_DISTANCE_CODES))) return st;
  } else {
    int hlit = br_read_bits(b, 5, 1) + 257;
    int hdist = br_read_bits(b, 5, 1) + 1;
    int hclen = br_read_bits(b, 4, 1) + 4;
    if (hlit > MAX_LITLEN_CODES) return ZO_ERR_INVALID_BUFFER;    /* :120-121 */
    if (hdist > MAX_DISTANCE_CODES) return ZO_ERR_INVALID_BUFFER; /* :123-124 */
    uint8_t clcls[19] = {0};
    for (int i = 0; i < hclen; i++) clcls[clcl_order[i]] = (uint8_t)br_read_bits(b, 3, 1);
    huffman clh;
    if ((st = init_huffman(&clh, clcls, 19))) return st;

    uint8_t unpacked[320] = {0};
    int i = 0;
    while (i != hlit + hdist) { /* :138-168 */
      if (b->bits_buffered < 15) br_fill(b);
      uint16_t symbol = decode_symbol(b, &clh);
      if (b->bits_buffered < 0) return ZO_ERR_END_OF_BUFFER;
      if (symbol <= 15) {
        unpacked[i++] = (uint8_t)symbol;
      } else if (symbol == 16) {
        if (i == 0) return ZO_ERR_INVALID_BUFFER;
        uint8_t prev = unpacked[i - 1];
        int repeat = br_read_bits(b, 2, 1) + 3;
        if (i + repeat > 320) return ZO_ERR_INVALID_BUFFER;
        for (int r = 0; r < repeat; r++) unpacked[i++] = prev;
      } else if (symbol == 17) {
        i += br_read_bits(b, 3, 1) + 3;
      } else if (symbol == 18) {
        i += br_read_bits(b, 7, 1) + 11;
      } else {
        return ZO_ERR_INVALID_SYMBOL;
      }
      if (i > hlit + hdist) return ZO_ERR_INVALID_BUFFER;
    }
    if ((st = init_huffman(&lit, unpacked, hlit))) return st;
    if ((st = init_huffman(&dist, unpacked + hlit, hdist))) return st;
  }

  size_t op = *op_io;
  for (;;) { /* :173-250 */
    if (b->bits_buffered < 15) br_fill(b);
    uint16_t symbol = decode_symbol(b, &lit);
    if (b->bits_buffered < 0) return ZO_ERR_END_OF_BUFFER;
    if (symbol <= 255) {
      if (op >= dst->cap && out_reserve(dst, op * 2 > 2 ? op * 2 : 2)) return ZO_ERR_NOMEM;
      dst->data[op++] = (uint8_t)symbol;
    } else if (symbol == 256) {
      break;
    } else {
      br_fill(b);
      int length_idx = symbol - 257;
      if (length_idx >= 29) return ZO_ERR_INVALID_BUFFER; /* :202-204 */
      size_t copy_length =
          (size_t)base_lengths[length_idx] + br_read_bits(b, base_lengths_extra[length_idx], 0);
      uint16_t distance_idx = decode_symbol(b, &dist);
      if (distance_idx >= 30) return ZO_ERR_INVALID_BUFFER; /* :211-213 */
      size_t distance =
          (size_t)base_distances[distance_idx] + br_read_bits(b, base_distance_extra[distance_idx], 0);
      if (distance > op) return ZO_ERR_INVALID_BUFFER; /* :224-225 */
      if (op + copy_length + 13 > dst->cap && out_reserve(dst, (op + copy_length) * 2 + 10))
        return ZO_ERR_NOMEM;
      /* :231-249, as written there: eight bytes at a time (the 13 bytes of room reserved above
       * are what the last copy64 may overwrite behind the match); a short distance first doubles
       * the pattern until source and destination are eight bytes apart */
      uint8_t *d = dst->data;
      if (copy_length <= 16 && distance >= 8) {
        copy64(d, op, op - distance);
        copy64(d, op + 8, op - distance + 8);
      } else {
        size_t copy_from = op - distance, copy_to = op;
        ptrdiff_t remaining = (ptrdiff_t)copy_length;
        while (copy_to - copy_from < 8) {
          copy64(d, copy_to, copy_from);
          remaining -= (ptrdiff_t)(copy_to - copy_from);
          copy_to += copy_to - copy_from;
        }
        while (remaining > 0) {
          copy64(d, copy_to, copy_from);
          copy_from += 8;
          copy_to += 8;
          remaining -= 8;
        }
      }
      op += copy_length;
    }
  }
  *op_io = op;
  return ZO_OK;
}

/* inflate.nim:252-266 */
static int inflate_no_compression(zo_buf *dst, bit_reader *b, size_t *op_io) {
  br_skip_remaining_bits(b);
  size_t len = br_read_bits(b, 16, 1);
  size_t nlen = br_read_bits(b, 16, 1);
  if (len + nlen != 65535) return ZO_ERR_INVALID_BUFFER;
  if (len > 0) {
    if (out_reserve(dst, *op_io + len)) return ZO_ERR_NOMEM;
    int st = br_read_bytes(b, dst->data + *op_io, len);
    if (st) return st;
  }
  *op_io += len;
  return ZO_OK;
}

/* inflate.nim:268-291 */
int zo_inflate(const uint8_t *src, size_t len, size_t pos, zo_buf *out) {
  init_tables();
  bit_reader b = {src, len, pos, 0, 0};
  size_t op = 0;
  int final_block = 0;
  while (!final_block) {
    uint16_t bfinal = br_read_bits(&b, 1, 1);
    uint16_t btype = br_read_bits(&b, 2, 1);
    if (bfinal != 0) final_block = 1;
    int st;
    switch (btype) {
      case 0: st = inflate_no_compression(out, &b, &op); break;
      case 1: st = inflate_block(out, &b, &op, 1); break;
      case 2: st = inflate_block(out, &b, &op, 0); break;
      default: st = ZO_ERR_BLOCK_HEADER; break;
    }
    if (st) return st;
  }
  out->len = op;
  return ZO_OK;
}

/* ------------------------------------------------------------------ */
/* gzip.nim:3-88 uncompressGzip                                         */
/* ------------------------------------------------------------------ */
static long next_zero_byte(const uint8_t *src, size_t len, size_t start) { /* :43-47 */
  for (size_t i = start; i < len; i++)
    if (src[i] == 0) return (long)i;
  return -1;
}

static int uncompress_gzip(const uint8_t *src, size_t len, zo_buf *out) {
  if (len < 18) return ZO_ERR_INVALID_BUFFER;
  uint8_t id1 = src[0], id2 = src[1], cm = src[2], flg = src[3];
  if (id1 != 31 || id2 != 139) return ZO_ERR_GZIP_ID;
  if (cm != 8) return ZO_ERR_UNSUPPORTED_METHOD;
  if ((flg & 0xe0) > 0) return ZO_ERR_RESERVED_FLAGS;
  int fhcrc = (flg & 2) != 0, fextra = (flg & 4) != 0, fname = (flg & 8) != 0,
      fcomment = (flg & 16) != 0;
  size_t pos = 10;
  if (fextra) return ZO_ERR_UNSUPPORTED_FLAGS;
  if (fname) {
    long z = next_zero_byte(src, len, pos);
    if (z < 0) return ZO_ERR_INVALID_BUFFER;
    pos = (size_t)z + 1;
  }
  if (fcomment) {
    long z = next_zero_byte(src, len, pos);
    if (z < 0) return ZO_ERR_INVALID_BUFFER;
    pos = (size_t)z + 1;
  }
  if (fhcrc) {
    if (pos + 2 >= len) return ZO_ERR_INVALID_BUFFER;
    pos += 2;
  }
  if (pos + 8 >= len) return ZO_ERR_INVALID_BUFFER;
  uint32_t checksum = read32(src, len - 8), isize = read32(src, len - 4);
  int st = zo_inflate(src, len, pos, out); /* whole gzip length, as :78 */
  if (st) return st;
  if (checksum != zo_crc32(out->data, out->len)) return ZO_ERR_CHECKSUM;
  if (isize != (uint32_t)(out->len & 0xffffffffu)) return ZO_ERR_SIZE;
  return ZO_OK;
}

/* zippy.nim:100-165 */
int zo_uncompress(const uint8_t *src, size_t len, int data_format, zo_buf *out) {
  init_tables();
  switch (data_format) {
    case ZO_DF_DETECT:
      if (len > 18 && src[0] == 31 && src[1] == 139 && src[2] == 8 && (src[3] & 0xe0) == 0)
        return zo_uncompress(src, len, ZO_DF_GZIP, out);
      if (len > 6 && (src[0] & 0x0f) == 8 && (src[0] >> 4) <= 7 &&
          (((unsigned)src[0] * 256) + src[1]) % 31 == 0)
        return zo_uncompress(src, len, ZO_DF_ZLIB, out);
      return ZO_ERR_DETECT;
    case ZO_DF_GZIP:
      return uncompress_gzip(src, len, out);
    case ZO_DF_ZLIB: {
      if (len < 6) return ZO_ERR_INVALID_BUFFER;
      uint8_t cmf = src[0], flg = src[1], cm = cmf & 0x0f, cinfo = cmf >> 4;
      if (cm != 8) return ZO_ERR_UNSUPPORTED_METHOD;
      if (cinfo > 7) return ZO_ERR_COMPRESSION_INFO;
      if ((((unsigned)cmf * 256) + flg) % 31 != 0) return ZO_ERR_INVALID_HEADER;
      if (flg & 0x20) return ZO_ERR_PRESET_DICT;
      int st = zo_inflate(src, len, 2, out);
      if (st) return st;
      uint32_t checksum = (uint32_t)src[len - 4] << 24 | (uint32_t)src[len - 3] << 16 |
                          (uint32_t)src[len - 2] << 8 | src[len - 1];
      if (checksum != zo_adler32(out->data, out->len)) return ZO_ERR_CHECKSUM;
      return ZO_OK;
    }
    case ZO_DF_DEFLATE:
      return zo_inflate(src, len, 0, out);
    default:
      return ZO_ERR_INVALID_FORMAT;
  }
}

/* zippy.nim:11-84 */
int zo_compress(const uint8_t *src, size_t len, int level, int data_format, int fname_len,
                zo_buf *out) {
  init_tables();
  int st;
  switch (data_format) {
    case ZO_DF_GZIP: {
      const uint8_t hdr[10] = {31, 139, 8, 1 << 3, 0, 0, 0, 0, 0, 0};
      if (buf_append(out, hdr, 10)) return ZO_ERR_NOMEM;
      int k = fname_len;
      if (k < 0) k = rand() % 26; /* zippy.nim:28-38: urandom mod 26 */
      if (k > 25) k = 25;
      for (int i = 0; i < k; i++)
        if (buf_push(out, (uint8_t)(97 + i))) return ZO_ERR_NOMEM;
      if (buf_push(out, 0)) return ZO_ERR_NOMEM;
      if ((st = zo_deflate(src, len, level, out))) return st;
      uint32_t checksum = zo_crc32(src, len), isize = (uint32_t)(len & 0xffffffffu);
      uint8_t tr[8];
      for (int i = 0; i < 4; i++) {
        tr[i] = (uint8_t)(checksum >> (8 * i));
        tr[4 + i] = (uint8_t)(isize >> (8 * i));
      }
      if (buf_append(out, tr, 8)) return ZO_ERR_NOMEM;
      return ZO_OK;
    }
    case ZO_DF_ZLIB: {
      const uint8_t cmf = (7 << 4) | 8;
      const uint8_t fcheck = (uint8_t)(31 - ((unsigned)cmf * 256) % 31);
      if (buf_push(out, cmf) || buf_push(out, fcheck)) return ZO_ERR_NOMEM;
      if ((st = zo_deflate(src, len, level, out))) return st;
      uint32_t checksum = zo_adler32(src, len);
      uint8_t tr[4] = {(uint8_t)(checksum >> 24), (uint8_t)(checksum >> 16),
                       (uint8_t)(checksum >> 8), (uint8_t)checksum};
      if (buf_append(out, tr, 4)) return ZO_ERR_NOMEM;
      return ZO_OK;
    }
    case ZO_DF_DEFLATE:
      return zo_deflate(src, len, level, out);
    default:
      return ZO_ERR_INVALID_FORMAT;
  }
}

int zo_compress_blocks(const uint8_t *src, size_t len, int level, int data_format, int fname_len,
                       size_t block_bytes, zo_buf *out, zo_block_entry **index, size_t *n_entries) {
  if (block_bytes < MAX_WINDOW_SIZE || block_bytes > MAX_BLOCK_SIZE || block_bytes % MAX_WINDOW_SIZE)
    return ZO_ERR_INVALID_FORMAT;
  if (out->len != 0) return ZO_ERR_INVALID_FORMAT; /* positions are relative to the buffer start */
  opt_block_size = block_bytes;
  memset(&opt_index, 0, sizeof(opt_index));
  opt_index.on = 1;
  int st = zo_compress(src, len, level, data_format, fname_len, out);
  opt_block_size = MAX_BLOCK_SIZE;
  opt_index.on = 0;
  if (st == ZO_OK && opt_index.err) st = ZO_ERR_NOMEM;
  if (st != ZO_OK) {
    free(opt_index.e);
    return st;
  }
  *index = opt_index.e;
  *n_entries = opt_index.n;
  return ZO_OK;
}

const char *zo_strerror(int status) {
  switch (status) {
    case ZO_OK: return "ok";
    case ZO_ERR_INVALID_LEVEL: return "Invalid compression level";
    case ZO_ERR_INVALID_FORMAT: return "Invalid data format";
    case ZO_ERR_DETECT: return "Unable to detect compressed data format";
    case ZO_ERR_UNSUPPORTED_METHOD: return "Unsupported compression method";
    case ZO_ERR_COMPRESSION_INFO: return "Invalid compression info";
    case ZO_ERR_INVALID_HEADER: return "Invalid header";
    case ZO_ERR_PRESET_DICT: return "Preset dictionary is not yet supported";
    case ZO_ERR_CHECKSUM: return "Checksum verification failed";
    case ZO_ERR_SIZE: return "Size verification failed";
    case ZO_ERR_GZIP_ID: return "Failed gzip identification values check";
    case ZO_ERR_RESERVED_FLAGS: return "Reserved flag bits set";
    case ZO_ERR_UNSUPPORTED_FLAGS: return "Currently unsupported flags are set";
    case ZO_ERR_INVALID_BUFFER: return "Invalid buffer, unable to uncompress";
    case ZO_ERR_COMPRESS_INTERNAL: return "Unexpected error while compressing";
    case ZO_ERR_END_OF_BUFFER: return "Cannot read further, at end of buffer";
    case ZO_ERR_BYTE_BOUNDARY: return "Must be at a byte boundary";
    case ZO_ERR_BLOCK_HEADER: return "Invalid block header";
    case ZO_ERR_INVALID_SYMBOL: return "Invalid symbol";
    case ZO_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
  }
}

/* ---- cpu_baseline support (bench.py): one compress() or uncompress() per buffer of a batch on
 * `threads` worker threads, each pinned to its own core, buffers handed out first come, first
 * served; the time of the parallel region (tests/bench.nim times single calls; a batch on all cores
 * is what the GPU path is compared with).  dir 0: compress(level, fmt) of srcs;
 * dir 1: uncompress(fmt) of srcs.  Results are kept in outs[] (owned by the caller: zo_free)
 * when outs != NULL, otherwise freed.  Returns the first non-zero status, 0 if none. ---- */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <pthread.h>
#ifdef __GLIBC__
#include <malloc.h>
#endif
#include <sched.h>
#include <time.h>

typedef struct {
  const uint8_t *const *srcs;
  const size_t *lens;
  size_t n;
  int dir, level, fmt, threads, id;
  zo_buf *outs;
  size_t *next;
  int *status;
  pthread_barrier_t *bar;
  size_t scratch_cap; /* the largest result of the batch (an upper bound) */
} zo_mt_job;

static void *zo_mt_worker(void *arg) {
  zo_mt_job *j = (zo_mt_job *)arg;
#ifdef __linux__
  {
    cpu_set_t allowed, one;
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
      int want = j->id % CPU_COUNT(&allowed), seen = 0;
      for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &allowed) && seen++ == want) {
          CPU_ZERO(&one);
          CPU_SET(c, &one);
          (void)pthread_setaffinity_np(pthread_self(), sizeof one, &one);
          break;
        }
    }
  }
#endif
  /* Timed repetitions (results not kept): every thread writes into ONE output buffer of its own, as large as the
   * largest result can get and touched before the clock starts -- no realloc growth, no page fault and no trip
   * through the allocator for the result inside the timed region.  (The codec's own tables and token arrays stay
   * what they are in the reference: allocated a call, deflate.nim:243-252, snappy.nim:24-31.) */
  zo_buf scratch = {0, 0, 0};
  if (!j->outs && j->scratch_cap) {
    scratch.data = (uint8_t *)malloc(j->scratch_cap);
    if (scratch.data) {
      memset(scratch.data, 0, j->scratch_cap);
      scratch.cap = j->scratch_cap;
    }
  }
  pthread_barrier_wait(j->bar); /* all pinned: the clock starts */
  for (;;) {
    size_t i = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
    if (i >= j->n) break;
    zo_buf out = {0, 0, 0};
    zo_buf *o = j->outs ? &out : &scratch;
    o->len = 0;
    int st = j->dir == 0 ? zo_compress(j->srcs[i], j->lens[i], j->level, j->fmt, 0, o)
                         : zo_uncompress(j->srcs[i], j->lens[i], j->fmt, o);
    if (st != ZO_OK) __atomic_store_n(j->status, st, __ATOMIC_RELAXED);
    if (j->outs) j->outs[i] = out;
  }
  pthread_barrier_wait(j->bar); /* the clock stops */
  free(scratch.data);
  return NULL;
}

int zo_batch_mt(const uint8_t *const *srcs, const size_t *lens, size_t n, int dir, int level, int fmt,
                int threads, zo_buf *outs, double *seconds) {
  if (threads < 1) threads = 1;
#ifdef __GLIBC__
  {
    /* The codec grows its result with realloc(); glibc serves blocks of a MiB by mmap()/munmap(), every one of
     * them a trip through the process's address-space lock and a page fault a page -- with a few hundred threads
     * that lock, not the codec, is what gets timed.  Keep such blocks in the (per-thread) arenas instead: after
     * the warm-up repetition a thread's buffers are recycled memory. */
    static int tuned = 0;
    if (!tuned) {
      tuned = 1;
      mallopt(M_MMAP_THRESHOLD, 1 << 30);
      mallopt(M_TRIM_THRESHOLD, 1 << 30);
      mallopt(M_TOP_PAD, 64 << 20);
    }
  }
#endif
  pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof *tid);
  zo_mt_job *jobs = (zo_mt_job *)calloc((size_t)threads, sizeof *jobs);
  pthread_barrier_t bar;
  size_t next = 0;
  int status = ZO_OK;
  if (!tid || !jobs) return ZO_ERR_NOMEM;
  pthread_barrier_init(&bar, NULL, (unsigned)threads + 1u);
  /* the largest result: compress: the stored form and the container; uncompress: ISIZE of a gzip member
   * (gzip.nim:64-66; other formats: the codec's own growth) */
  size_t scratch_cap = 0;
  for (size_t i = 0; i < n; i++) {
    size_t c = 0;
    if (dir == 0) c = lens[i] + 5 * (lens[i] / MAX_UNCOMPRESSED_BLOCK_SIZE + 1) + 64;
    else if (fmt == ZO_DF_GZIP && lens[i] >= 18)
      c = (size_t)srcs[i][lens[i] - 4] | (size_t)srcs[i][lens[i] - 3] << 8 | (size_t)srcs[i][lens[i] - 2] << 16 |
          (size_t)srcs[i][lens[i] - 1] << 24;
    if (c > scratch_cap) scratch_cap = c;
  }
  if (scratch_cap) scratch_cap += 64;
  for (int t = 0; t < threads; t++) {
    zo_mt_job j = {srcs, lens, n, dir, level, fmt, threads, t, outs, &next, &status, &bar, scratch_cap};
    jobs[t] = j;
    pthread_create(&tid[t], NULL, zo_mt_worker, &jobs[t]);
  }
  struct timespec t0, t1;
  pthread_barrier_wait(&bar);
  clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_barrier_wait(&bar);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
  pthread_barrier_destroy(&bar);
  free(tid);
  free(jobs);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return status;
}
