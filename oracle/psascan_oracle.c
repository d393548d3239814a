/*
 * psascan_oracle.c -- CPU restatement of pSAscan's streaming-gap + merge path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path;
 * nothing in the product (psascan_amd/, construct_sa) links, imports or calls
 * it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the shared object built from it.
 *
 * Every function cites the reference file:line (relative to /root/reference/)
 * whose *behaviour* it restates; data layouts are deliberately the simplest
 * possible (plain arrays, one serial chain) so that it is obviously correct.
 *
 * Parity pinning: the reference ships no golden vectors.
 * This oracle is pinned (tests/test_oracle.py) by
 *   (1) brute-force definitions from the full-text ISA,
 *   (2) the sha256 of the reference's own .sa5 outputs on six seeded inputs
 *       recorded in SURVEY.md 8c, and
 *   (3) oracle/_ref: the reference's own hot-path headers compiled where they
 *       lie (oracle/Makefile), stage by stage.
 *
 * Index conventions:
 *   - a "gt" bit array attached to a position range (lo, hi] stores the bit of
 *     text position j at index u = hi - j  (LSB-first inside bytes).  This is
 *     the reference's reversed indexing (bit n-j) shifted by n-hi.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
typedef uint64_t u64;
typedef uint8_t u8;

static inline int bit_get(const u8 *bv, i64 i) { return (bv[i >> 3] >> (i & 7)) & 1; }
static inline void bit_set(u8 *bv, i64 i) { bv[i >> 3] |= (u8)(1u << (i & 7)); }

/* ------------------------------------------------------------------------- */
/* Suffix array of a whole text by prefix doubling (own code; the sorter is   */
/* outside the hot path -- any correct sorter is result-identical).           */
/* Order: lexicographic, shorter-is-smaller (em_compute_initial_ranks.hpp:69) */
/* ------------------------------------------------------------------------- */
static const i64 *g_rank;
static i64 g_k, g_n;
static int cmp_pd(const void *a, const void *b) {
  i64 x = *(const i64 *)a, y = *(const i64 *)b;
  if (g_rank[x] != g_rank[y]) return g_rank[x] < g_rank[y] ? -1 : 1;
  i64 rx = x + g_k < g_n ? g_rank[x + g_k] : -1;
  i64 ry = y + g_k < g_n ? g_rank[y + g_k] : -1;
  return rx < ry ? -1 : (rx > ry ? 1 : 0);
}

int orc_suffix_array(const u8 *t, i64 n, i64 *sa) {
  if (n <= 0) return 0;
  i64 *rk = (i64 *)malloc(sizeof(i64) * n), *tmp = (i64 *)malloc(sizeof(i64) * n);
  if (!rk || !tmp) return -1;
  for (i64 i = 0; i < n; ++i) { sa[i] = i; rk[i] = t[i]; }
  for (i64 k = 1;; k <<= 1) {
    g_rank = rk; g_k = k; g_n = n;
    /* first round sorts by (t[i], t[i+1..]) too: k=1 compares rank and rank+1 */
    qsort(sa, (size_t)n, sizeof(i64), cmp_pd);
    tmp[sa[0]] = 0;
    for (i64 i = 1; i < n; ++i)
      tmp[sa[i]] = tmp[sa[i - 1]] + (cmp_pd(&sa[i - 1], &sa[i]) != 0);
    memcpy(rk, tmp, sizeof(i64) * n);
    if (rk[sa[n - 1]] == n - 1) break;
  }
  free(rk); free(tmp);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Partial SA / BWT / i0 / gt_begin of a (half-)block [beg,end), derived from */
/* the full SA by definition (inmem_bwt_from_sa.hpp:51-54 */
/* for the dummy 0 at i0).  gt_begin bit u (u = end - j, j in (beg,end]) =    */
/* [text[j..n) > text[beg..n)]   (partial_sufsort.hpp:231-233, 357-358).      */
/* ------------------------------------------------------------------------- */
void orc_partial_sa(const u8 *text, i64 n, const i64 *sa, const i64 *isa, i64 beg, i64 end,
                    i64 *psa, u8 *bwt, i64 *i0_out, u8 *gt_begin) {
  i64 m = end - beg, k = 0;
  for (i64 r = 0; r < n; ++r) {
    i64 p = sa[r];
    if (p < beg || p >= end) continue;
    psa[k] = p - beg;
    if (p == beg) { bwt[k] = 0; *i0_out = k; } else bwt[k] = text[p - 1];
    ++k;
  }
  if (gt_begin) {
    memset(gt_begin, 0, (size_t)((m + 7) / 8));
    for (i64 j = beg + 1; j <= end; ++j) {
      i64 rj = j < n ? isa[j] : -1; /* empty suffix is the smallest */
      if (rj > isa[beg]) bit_set(gt_begin, end - j);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* rank(i,c) = #{k < i : bwt[k] == c}; rank(i<=0)=0, rank(i>=m)=count[c].     */
/* Semantics of rank4n<>::rank, rank.hpp:566-568 (+ m_count, rank.hpp:112).   */
/* Layout here: counters every 256 symbols + byte scan.                       */
/* ------------------------------------------------------------------------- */
typedef struct {
  i64 m, nblk;
  const u8 *bwt;
  i64 *occ;      /* [nblk+1][256] */
  i64 count[256];
} orc_rank_t;

orc_rank_t *orc_rank_build(const u8 *bwt, i64 m) {
  orc_rank_t *r = (orc_rank_t *)calloc(1, sizeof(orc_rank_t));
  r->m = m; r->bwt = bwt; r->nblk = (m + 255) / 256;
  r->occ = (i64 *)malloc(sizeof(i64) * 256 * (size_t)(r->nblk + 1));
  i64 run[256]; memset(run, 0, sizeof run);
  for (i64 b = 0; b <= r->nblk; ++b) {
    memcpy(r->occ + 256 * b, run, sizeof run);
    i64 lim = (b + 1) * 256 < m ? (b + 1) * 256 : m;
    for (i64 k = b * 256; k < lim; ++k) run[bwt[k]]++;
  }
  memcpy(r->count, run, sizeof run);
  return r;
}
void orc_rank_free(orc_rank_t *r) { if (r) { free(r->occ); free(r); } }
const i64 *orc_rank_counts(const orc_rank_t *r) { return r->count; }

i64 orc_rank(const orc_rank_t *r, i64 i, int c) {
  if (i <= 0) return 0;
  if (i >= r->m) return r->count[c];
  i64 b = i >> 8, res = r->occ[256 * b + c];
  for (i64 k = b << 8; k < i; ++k) res += (r->bwt[k] == c);
  return res;
}

/* ------------------------------------------------------------------------- */
/* One streaming pass = compute_gap<T> (compute_gap.hpp:61-157) with a single */
/* chain: C array (compute_gap.hpp:77-85), recurrence (stream.hpp:147-158),   */
/* increments (update.hpp:86-96; here exact 64-bit counters, i.e. the VALUE   */
/* m_count[j] + 256*#excess(j) of gap_array.hpp:116-124).                     */
/*   tail range = [tb, te);  init_rank = rank of text[te..n) among the block  */
/*   suffixes;  gt_in bit u  = gt of position te-u w.r.t. the block END;      */
/*   gt_out bit u = [text[te-u..n) > text[block_beg..n)], u in [0, te-tb).    */
/* ------------------------------------------------------------------------- */
i64 orc_stream_pass(const orc_rank_t *r, i64 i0, int last, const u8 *text, i64 tb, i64 te,
                    const u8 *gt_in, i64 init_rank, u64 *gap, u8 *gt_out) {
  i64 C[256], s = 0;
  for (int c = 0; c < 256; ++c) {
    i64 t = r->count[c] + (c == last) - (c == 0);
    C[c] = s; s += t;
  }
  i64 i = init_rank;
  if (gt_out) memset(gt_out, 0, (size_t)((te - tb + 7) / 8));
  for (i64 j = te; j > tb; --j) {
    int c = text[j - 1];
    if (gt_out && i > i0) bit_set(gt_out, te - j);
    int g = gt_in ? bit_get(gt_in, te - j) : 0;
    int delta = (i > i0 && c == 0);
    i = C[c] + orc_rank(r, i, c) - delta;
    if (c == last && g) ++i;
    gap[i]++;
  }
  return i;
}

/* ------------------------------------------------------------------------- */
/* buffered_gap_array::convert_to_bitvector (gap_array.hpp:273-364):          */
/* for j = 0..m: gap[j] ones, then (j < m) a zero.  1 = suffix from the tail. */
/* Returns the number of bits written (m + sum gap).                          */
/* ------------------------------------------------------------------------- */
i64 orc_gap_to_bitvector(const u64 *gap, i64 m, u8 *bv) {
  i64 p = 0;
  for (i64 j = 0; j <= m; ++j) {
    for (u64 k = 0; k < gap[j]; ++k) { bit_set(bv, p); ++p; }
    if (j < m) ++p;
  }
  return p;
}

/* ------------------------------------------------------------------------- */
/* merge_bwt (bwt_merge.hpp:66-140): interleave, patch, return block_i0.      */
/* ------------------------------------------------------------------------- */
i64 orc_merge_bwt(const u8 *lbwt, const u8 *rbwt, i64 ml, i64 mr, i64 left_i0, i64 right_i0,
                  int left_last, const u8 *bv, u8 *out) {
  i64 l = 0, r = 0, block_i0 = -1;
  for (i64 k = 0; k < ml + mr; ++k) {
    if (bit_get(bv, k)) {
      out[k] = (r == right_i0) ? (u8)left_last : rbwt[r]; /* bwt_merge.hpp:128 */
      ++r;
    } else {
      if (l == left_i0) block_i0 = k;                      /* bwt_merge.hpp:133 */
      out[k] = lbwt[l++];
    }
  }
  return block_i0;
}

/* ------------------------------------------------------------------------- */
/* compute_right_gap (compute_right_gap.hpp:55-122,161-305): segment sums of  */
/* block_gap over bv' = bv + sentinel 1 at index block; mr+1 values.          */
/* ------------------------------------------------------------------------- */
void orc_right_gap(const u64 *block_gap, const u8 *bv, i64 ml, i64 mr, u64 *out) {
  i64 block = ml + mr, r = 0; u64 sum = 0;
  for (i64 k = 0; k <= block; ++k) {
    sum += block_gap[k];
    int b = k == block ? 1 : bit_get(bv, k);
    if (b) { out[r++] = sum; sum = 0; }
  }
}
/* compute_left_gap (compute_left_gap.hpp:55-123,162-306): sentinel 0, and    */
/* +1 for every 1-bit inside the segment; ml+1 values.                        */
void orc_left_gap(const u64 *block_gap, const u8 *bv, i64 ml, i64 mr, u64 *out) {
  i64 block = ml + mr, l = 0; u64 sum = 0;
  for (i64 k = 0; k <= block; ++k) {
    sum += block_gap[k];
    int b = k == block ? 0 : bit_get(bv, k);
    if (b) sum += 1; else { out[l++] = sum; sum = 0; }
  }
}

/* ------------------------------------------------------------------------- */
/* vbyte (utils/parallel_utils.hpp:47-136 encoder;                            */
/* io/async_vbyte_stream_reader.hpp:49-186 decoder): LSB-first 7-bit groups,  */
/* 0x80 on all but the last byte.                                             */
/* ------------------------------------------------------------------------- */
i64 orc_vbyte_encode(const u64 *v, i64 cnt, u8 *out) {
  i64 p = 0;
  for (i64 k = 0; k < cnt; ++k) {
    u64 x = v[k];
    while (x > 127) { out[p++] = (u8)((x & 0x7f) | 0x80); x >>= 7; }
    out[p++] = (u8)x;
  }
  return p;
}
i64 orc_vbyte_decode(const u8 *in, i64 nbytes, u64 *v) {
  i64 k = 0, p = 0;
  while (p < nbytes) {
    u64 x = 0; int sh = 0;
    while (in[p] & 0x80) { x |= ((u64)(in[p++] & 0x7f)) << sh; sh += 7; }
    x |= ((u64)in[p++]) << sh;
    v[k++] = x;
  }
  return k;
}

/* ------------------------------------------------------------------------- */
/* merge<T> (merge.hpp:55-180): repeatedly take the LEFTMOST half-block whose */
/* gap head is 0, emit psa+beg, reload its head, decrement all heads to its   */
/* left (merge.hpp:123-158).  Last half-block has head == 0 (merge.hpp:86).   */
/* Output: n uint40 little-endian entries (types/uint40.hpp:42-104).          */
/*   psa[h]: i64[size_h] relative to beg[h]; gap[h]: u64[size_h+1] (h<H-1).   */
/* ------------------------------------------------------------------------- */
void orc_merge(int H, const i64 *beg, const i64 *size, const i64 *const *psa,
               const u64 *const *gap, u8 *out_sa5) {
  i64 n = 0;
  i64 *ptr = (i64 *)calloc((size_t)H, sizeof(i64));
  u64 *head = (u64 *)calloc((size_t)H, sizeof(u64));
  for (int h = 0; h < H; ++h) { n += size[h]; head[h] = (h + 1 < H) ? gap[h][0] : 0; }
  for (i64 i = 0; i < n; ++i) {
    int j = 0;
    while (head[j] != 0) { head[j]--; ++j; }
    u64 v = (u64)(psa[j][ptr[j]] + beg[j]);
    ptr[j]++;
    if (j != H - 1) head[j] = gap[j][ptr[j]];
    for (int b = 0; b < 5; ++b) out_sa5[5 * i + b] = (u8)(v >> (8 * b));
  }
  free(ptr); free(head);
}

/* ------------------------------------------------------------------------- */
/* Chain start ranks (K8): number of suffixes of the block [bb,be) that are    */
/* smaller than text[p..n), found by binary search over the block's partial SA */
/* with a string comparison that may only read block text and that decides a   */
/* comparison reaching `cmp_end` (>= be) by the gt bit of the position the     */
/* pattern has reached: em_compute_initial_ranks.hpp:54-76 (lcp_compare: the   */
/* block suffix runs into block_end, gt w.r.t. block_end decides) and          */
/* :321-363 (lcp_compare_2: text up to tail_begin is read, then gt w.r.t.      */
/* tail_begin).  gt_cmp_end bit u <-> position n - u is NOT used here: the     */
/* array is indexed like every gt array of this file, u = hi - j with hi = n.  */
/* A pattern that ends before the comparison is decided is the smaller one     */
/* (:69-71 "pat_beg + pat_length >= text_length -> -1").                       */
/* ------------------------------------------------------------------------- */
i64 orc_initial_rank(const u8 *text, i64 n, i64 bb, i64 be, const i64 *psa, i64 cmp_end, const u8 *gt_cmp_end, i64 p) {
  if (p >= n) return 0;                               /* :180-183: the empty suffix is the smallest */
  i64 lo = 0, hi = be - bb;                           /* answer in [lo, hi] */
  while (lo < hi) {
    i64 mid = (lo + hi) / 2, s = bb + psa[mid], k = 0;
    int pat_greater;                                  /* text[p..) > text[s..) ? */
    for (;;) {
      if (s + k >= cmp_end) {                         /* the block suffix reached cmp_end: gt of position p + (cmp_end - s) w.r.t. cmp_end */
        i64 q = p + (cmp_end - s);
        pat_greater = q < n ? bit_get(gt_cmp_end, n - q) : 0;
        break;
      }
      if (p + k >= n) { pat_greater = 0; break; }     /* pattern exhausted: it is a proper prefix, hence smaller */
      if (text[s + k] != text[p + k]) { pat_greater = text[p + k] > text[s + k]; break; }
      ++k;
    }
    if (pat_greater) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* ------------------------------------------------------------------------- */
/* Whole run, restating partial_sufsort (partial_sufsort.hpp:558-584) and the */
/* six steps of process_block (partial_sufsort.hpp:67-551) with the sorter    */
/* replaced by "filter the full SA" (orc_partial_sa).  The result of the gap  */
/* streaming + merge path must reproduce the full SA -- tests assert that.    */
/*   ram_use only matters for the last block's left-half size                 */
/*   (partial_sufsort.hpp:86-88).                                             */
/* ------------------------------------------------------------------------- */
typedef struct { i64 beg, size; i64 *psa; u64 *gap; } orc_hb_t;

int orc_psascan(const u8 *text, i64 n, i64 max_block_size, i64 ram_use, u8 *out_sa5) {
  if (n <= 0) return 0;
  i64 *sa = (i64 *)malloc(sizeof(i64) * n), *isa = (i64 *)malloc(sizeof(i64) * n);
  if (orc_suffix_array(text, n, sa)) return -1;
  for (i64 k = 0; k < n; ++k) isa[sa[k]] = k;

  i64 n_blocks = (n + max_block_size - 1) / max_block_size;
  orc_hb_t *hb = (orc_hb_t *)calloc((size_t)(2 * n_blocks), sizeof(orc_hb_t));
  int H = 0;
  /* gt of every position j in (block_beg, n] w.r.t. the current block begin;  */
  /* index u = n - j  (the reference's reversed indexing).                      */
  u8 *tail_gt = (u8 *)calloc((size_t)(n / 8 + 2), 1), *new_gt = (u8 *)calloc((size_t)(n / 8 + 2), 1);

  for (i64 bid = n_blocks - 1; bid >= 0; --bid) {
    i64 b = max_block_size * bid, e = b + max_block_size < n ? b + max_block_size : n;
    i64 bs = e - b;
    int last_block = (e == n);
    i64 ls = last_block ? (bs < (ram_use / 10 > 1 ? ram_use / 10 : 1) ? bs : (ram_use / 10 > 1 ? ram_use / 10 : 1))
                        : (bs / 2 > 1 ? bs / 2 : 1);
    i64 rs = bs - ls, mid = b + ls;
    memset(new_gt, 0, (size_t)(n / 8 + 2));

    /* left half */
    i64 *lpsa = (i64 *)malloc(sizeof(i64) * ls); u8 *lbwt = (u8 *)malloc((size_t)ls);
    u8 *lgt = (u8 *)malloc((size_t)(ls / 8 + 2)); i64 li0 = 0;
    orc_partial_sa(text, n, sa, isa, b, mid, lpsa, lbwt, &li0, lgt);
    for (i64 u = 0; u < ls; ++u) if (bit_get(lgt, u)) bit_set(new_gt, (n - mid) + u);
    free(lgt);
    if (rs == 0) {
      hb[H].beg = b; hb[H].size = ls; hb[H].psa = lpsa; hb[H].gap = NULL; ++H;
      free(lbwt);
      u8 *t = tail_gt; tail_gt = new_gt; new_gt = t;
      continue;
    }
    /* right half */
    i64 *rpsa = (i64 *)malloc(sizeof(i64) * rs); u8 *rbwt = (u8 *)malloc((size_t)rs);
    u8 *rgt = (u8 *)malloc((size_t)(rs / 8 + 2)); i64 ri0 = 0;
    orc_partial_sa(text, n, sa, isa, mid, e, rpsa, rbwt, &ri0, rgt);

    /* step 3: pass A -- stream the right half through rank(left BWT)           */
    /* (partial_sufsort.hpp:403-414).  init rank = #left suffixes < text[e..).   */
    i64 initA = 0;
    for (i64 s = b; s < mid; ++s) initA += (e < n ? isa[s] < isa[e] : 0);
    orc_rank_t *lr = orc_rank_build(lbwt, ls);
    u64 *gapA = (u64 *)calloc((size_t)(ls + 1), sizeof(u64));
    u8 *gtA = (u8 *)malloc((size_t)(rs / 8 + 2));
    orc_stream_pass(lr, li0, text[mid - 1], text, mid, e, rgt, initA, gapA, gtA);
    orc_rank_free(lr);
    for (i64 u = 0; u < rs; ++u) if (bit_get(gtA, u)) bit_set(new_gt, (n - e) + u);
    free(gtA); free(rgt);

    if (last_block) { /* partial_sufsort.hpp:418-429 */
      hb[H].beg = b; hb[H].size = ls; hb[H].psa = lpsa; hb[H].gap = gapA; ++H;
      hb[H].beg = mid; hb[H].size = rs; hb[H].psa = rpsa; hb[H].gap = NULL; ++H;
      free(lbwt); free(rbwt);
      u8 *t = tail_gt; tail_gt = new_gt; new_gt = t;
      continue;
    }
    /* step 4: bitvector + BWT merge (partial_sufsort.hpp:441-471) */
    u8 *bv = (u8 *)calloc((size_t)(bs / 8 + 2), 1);
    orc_gap_to_bitvector(gapA, ls, bv);
    free(gapA);
    u8 *bbwt = (u8 *)malloc((size_t)bs);
    i64 bi0 = orc_merge_bwt(lbwt, rbwt, ls, rs, li0, ri0, text[mid - 1], bv, bbwt);
    free(lbwt); free(rbwt);
    /* step 5: pass B -- stream the tail through rank(block BWT) (:500-514) */
    orc_rank_t *br = orc_rank_build(bbwt, bs);
    u64 *gapB = (u64 *)calloc((size_t)(bs + 1), sizeof(u64));
    u8 *gtB = (u8 *)malloc((size_t)((n - e) / 8 + 2));
    orc_stream_pass(br, bi0, text[e - 1], text, e, n, tail_gt, 0, gapB, gtB);
    orc_rank_free(br); free(bbwt);
    for (i64 u = 0; u < n - e; ++u) if (bit_get(gtB, u)) bit_set(new_gt, u);
    free(gtB);
    /* step 6: split (partial_sufsort.hpp:536-542) */
    u64 *rg = (u64 *)malloc(sizeof(u64) * (size_t)(rs + 1)), *lg = (u64 *)malloc(sizeof(u64) * (size_t)(ls + 1));
    orc_right_gap(gapB, bv, ls, rs, rg);
    orc_left_gap(gapB, bv, ls, rs, lg);
    free(gapB); free(bv);
    hb[H].beg = b; hb[H].size = ls; hb[H].psa = lpsa; hb[H].gap = lg; ++H;
    hb[H].beg = mid; hb[H].size = rs; hb[H].psa = rpsa; hb[H].gap = rg; ++H;
    u8 *t = tail_gt; tail_gt = new_gt; new_gt = t;
  }
  /* merge.hpp:59 -- sort half-blocks by beg (insertion sort, H is tiny) */
  for (int a = 1; a < H; ++a) { orc_hb_t x = hb[a]; int c = a - 1; while (c >= 0 && hb[c].beg > x.beg) { hb[c + 1] = hb[c]; --c; } hb[c + 1] = x; }
  i64 *begs = (i64 *)malloc(sizeof(i64) * H), *sizes = (i64 *)malloc(sizeof(i64) * H);
  const i64 **psas = (const i64 **)malloc(sizeof(void *) * H); const u64 **gaps = (const u64 **)malloc(sizeof(void *) * H);
  for (int h = 0; h < H; ++h) { begs[h] = hb[h].beg; sizes[h] = hb[h].size; psas[h] = hb[h].psa; gaps[h] = hb[h].gap; }
  /* gap files are vbyte on disk in the reference; round-trip them here */
  for (int h = 0; h + 1 < H; ++h) {
    u8 *buf = (u8 *)malloc((size_t)(10 * (sizes[h] + 1)));
    i64 nb = orc_vbyte_encode(gaps[h], sizes[h] + 1, buf);
    orc_vbyte_decode(buf, nb, hb[h].gap);
    free(buf);
  }
  orc_merge(H, begs, sizes, psas, gaps, out_sa5);
  for (int h = 0; h < H; ++h) { free(hb[h].psa); free(hb[h].gap); }
  free(hb); free(begs); free(sizes); free(psas); free(gaps); free(tail_gt); free(new_gt); free(sa); free(isa);
  return 0;
}
