// leaf_tree.hip -- the merging half of the reference's in-memory pSAscan (inmem_psascan_src/inmem_psascan.hpp:64-304:
// max_threads sub-blocks are suffix-sorted, then merged with the streaming machinery itself) in BATCHES.
//
// A range (a half-block of the text) is cut into many small LEAVES that the host cores suffix-sort; the device merges
// them pairwise, level by level.  One level = one launch sequence for ALL pairs (block = left node, tail = right
// node) of that level instead of one sequence per pair:
//   rank structure over the level's BWT array (every node's BWT at its text position: ONE psg_rank_build),
//   start ranks of every chain of every pass by string search (em_compute_initial_ranks.hpp:222-319, one thread each),
//   ONE stream kernel launch (stream_batch_kernel: compute_gap.hpp:61-157 for every pair), ONE rank-log histogram into
//   a shared gap array (slot m of a pass = slot 0 of the next: its unary coding is then the concatenation of the
//   pairs' merge bitvectors, gap_array.hpp:273-364), ONE two-way merge of partial SAs and BWTs (merge.hpp:123-158 +
//   bwt_merge.hpp:66-140), ONE concatenation of gt bits (stream.hpp:150 for the tail, the block's own for the rest).
// 4 GiB of text in 64 KiB leaves are 65 535 pair merges: 17 levels of ~10 launches instead of 65 535 x ~12.
#include "dev_common.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

using namespace psg;

namespace {

struct TreeText {
  const u8 *text;     // addressed by absolute text position
  i64 n;
  i64 text_end;       // text[.. text_end) is readable (a text that stays in host memory is seen through a window)
  int *fail;          // [0]: a comparison would have read behind text_end; [1]: a comparison used up its budget
};

#define LT_BUDGET ((i64)1 << 20)   // symbols one search / one leaf comparison may compare before it gives up

// [text[s..n) < text[p..n)] for s < p; the first k symbols are known to be equal; k returns the common prefix length
__device__ __forceinline__ bool suffix_less_thread(const TreeText &X, i64 s, i64 p, i64 &k, i64 &budget) {
  for (;;) {
    const i64 rem = X.n - (p + k);                       // symbols left in the pattern
    if (rem <= 0) return false;                          // the pattern is a proper prefix of the suffix: it is the smaller one
    if (rem >= 8 && p + k + 8 <= X.text_end) {
      u64 a, b;
      __builtin_memcpy(&a, X.text + s + k, 8);
      __builtin_memcpy(&b, X.text + p + k, 8);
      if (a != b) {
        const int byte = (__ffsll((long long)(a ^ b)) - 1) >> 3;
        k += byte;
        return ((a >> (8 * byte)) & 255u) < ((b >> (8 * byte)) & 255u);
      }
      k += 8;
      if ((budget -= 8) < 0) { X.fail[1] = 1; return false; }
    } else {
      if (p + k >= X.text_end) { X.fail[0] = 1; return false; }
      const u8 a = X.text[s + k], b = X.text[p + k];
      if (a != b) return a < b;
      ++k;
      if (--budget < 0) { X.fail[1] = 1; return false; }
    }
  }
}

// ---- level 0: the leaves as the host sorter delivers them (positions relative to the leaf, 2 or 4 bytes each) ->
// partial SA relative to the range, BWT (dummy 0 at i0, inmem_bwt_from_sa.hpp:51-54), i0, gt bits (bit u <-> position
// leaf_end - u: the suffix is ranked after the leaf's first suffix; bit 0 by comparison).  One workgroup per leaf.
#define LT_GT_LDS_WORDS 8192        // leaves of up to 2^18 positions collect their gt bits in LDS
template <class PSA_T>
__global__ __launch_bounds__(256) void leaf_prepare_kernel(TreeText X, i64 range_beg, const i64 *leaf_beg, const i64 *leaf_gt_word, const PSA_T *in, u32 *psa, u8 *bwt,
                                                            i64 *i0_out, u32 *gt) {
  __shared__ u32 bits[LT_GT_LDS_WORDS];
  __shared__ i64 i0s;
  const i64 lb = leaf_beg[blockIdx.x], le = leaf_beg[blockIdx.x + 1], size = le - lb;   // relative to the range
  const PSA_T *src = in + lb;
  u32 *g = gt + leaf_gt_word[blockIdx.x];
  const i64 nwords = (size + 31) >> 5;
  const bool in_lds = nwords <= LT_GT_LDS_WORDS;
  if (in_lds) for (i64 w = threadIdx.x; w < nwords; w += 256) bits[w] = 0;
  for (i64 k = threadIdx.x; k < size; k += 256) {
    const i64 v = (i64)src[k];
    psa[lb + k] = (u32)(lb + v);
    bwt[lb + k] = v ? X.text[range_beg + lb + v - 1] : (u8)0;
    if (v == 0) i0s = k;
  }
  __syncthreads();
  const i64 i0 = i0s;
  for (i64 k = i0 + 1 + threadIdx.x; k < size; k += 256) {
    const i64 u = size - (i64)src[k];                     // (src[k] != 0 here)
    if (in_lds) atomicOr(&bits[u >> 5], 1u << (u & 31)); else atomicOr(&g[u >> 5], 1u << (u & 31));
  }
  if (threadIdx.x == 0) {
    i0_out[blockIdx.x] = i0;
    // bit 0 <-> position leaf_end: [text[end..) > text[beg..)] by reading on
    const i64 b = range_beg + lb, e = range_beg + le;
    bool gtb = false;
    if (e < X.n) { i64 k = 0, budget = LT_BUDGET; gtb = suffix_less_thread(X, b, e, k, budget); }
    if (gtb) { if (in_lds) atomicOr(&bits[0], 1u); else atomicOr(&g[0], 1u); }
  }
  __syncthreads();
  if (in_lds) for (i64 w = threadIdx.x; w < nwords; w += 256) g[w] = bits[w];
}

// ---- start rank of every chain of every pass: number of block suffixes smaller than the suffix at the chain's start
// (em_compute_initial_ranks.hpp:222-319; comparisons read on in the text, so no gt bits are involved).  One thread per
// chain, binary search over the block's partial SA with the common prefixes of both bounds kept (Manber-Myers).
__global__ __launch_bounds__(PSG_WG) void batch_search_kernel(TreeText X, i64 range_beg, const BatchGeom *geom, const u32 *wg_pass, const u32 *wg_local, const u32 *psa, i64 L,
                                                               i64 *init) {
  const BatchGeom G = geom[wg_pass[blockIdx.x]];
  const i64 k = (i64)wg_local[blockIdx.x] * PSG_WG + threadIdx.x, K = (G.T + L - 1) / L;
  if (k >= K) return;
  const i64 pos = range_beg + G.lbeg + G.m + G.T - k * L;      // chain k starts at the tail's end - k * L
  i64 lo = 0, hi = G.m;
  if (pos >= X.n) hi = 0;                                      // the empty suffix is the smallest
  const u32 *sa = psa + G.lbeg;
  i64 llcp = 0, rlcp = 0, budget = LT_BUDGET;
  while (lo < hi) {
    const i64 md = lo + ((hi - lo) >> 1);
    const i64 s = range_beg + (i64)gload(sa + md);
    i64 l = llcp < rlcp ? llcp : rlcp;
    if (suffix_less_thread(X, s, pos, l, budget)) { lo = md + 1; llcp = l; } else { hi = md; rlcp = l; }
  }
  init[G.kbase + k] = lo;
}

// the rank a chain ends with is the start rank of the next chain of its pass (same check as after an ordinary pass)
__global__ __launch_bounds__(PSG_WG) void batch_handover_kernel(const BatchGeom *geom, const u32 *wg_pass, const u32 *wg_local, i64 L, const i64 *init, const i64 *fin, int *err) {
  const BatchGeom G = geom[wg_pass[blockIdx.x]];
  const i64 k = (i64)wg_local[blockIdx.x] * PSG_WG + threadIdx.x, K = (G.T + L - 1) / L;
  if (k == 0 || k >= K) return;
  if (fin[G.kbase + k - 1] != init[G.kbase + k]) *err = 1;
}

// ---- gt bits of the parents: the tail's positions got theirs from the stream kernel (stream.hpp:150), the block's
// positions keep their own: dst bits [gt_out_word * 32 + T, + m) = src bits [gt_l_word * 32, + m).  Workgroup w copies
// words wg_local[w] * 256 .. of pass wg_pass[w].
__global__ __launch_bounds__(PSG_WG) void batch_gt_concat_kernel(const BatchGeom *geom, const u32 *wg_pass, const u32 *wg_local, const u32 *gt_cur, i64 cur_words, u32 *gt_new) {
  const BatchGeom G = geom[wg_pass[blockIdx.x]];
  const i64 dst0 = G.gt_out_word * 32 + G.T, dst1 = dst0 + G.m, src0 = G.gt_l_word * 32;
  const i64 w = (dst0 >> 5) + (i64)wg_local[blockIdx.x] * PSG_WG + threadIdx.x;
  if (w * 32 >= dst1) return;
  const i64 lo = std::max<i64>(w * 32, dst0), hi = std::min<i64>(w * 32 + 32, dst1);
  const u32 val = get_bits(gt_cur, src0 + (lo - dst0), (int)(hi - lo), cur_words) << (lo - w * 32);
  if (hi - lo == 32) gt_new[w] = val; else if (val) atomicOr(&gt_new[w], val);
}

struct Node { i64 beg, size, gt_word; };

static double wall_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

}  // namespace

extern "C" int psg_merge_leaves(const psg_search_ctx *sc, int64_t range_beg, int64_t range_size, const int64_t *h_leaf_beg, int64_t n_leaves, const void *d_leaf_psa,
                                int psa_bytes, uint32_t *d_psa_out, uint8_t *d_bwt_out, int64_t *i0, uint32_t *d_gt_begin_out, psg_leaf_merge_stats *stats) {
  PSG_REQUIRE(sc && sc->d_text && h_leaf_beg && d_leaf_psa && d_psa_out && d_bwt_out && i0 && d_gt_begin_out && n_leaves >= 1, "psg_merge_leaves");
  PSG_REQUIRE(range_beg >= 0 && range_size >= 1 && range_beg + range_size <= sc->n && range_size < 0xFFFFFFF0ll, "psg_merge_leaves: a range of 1 .. 2^32 - 17 positions inside the text");
  PSG_REQUIRE(psa_bytes == 2 || psa_bytes == 4, "psg_merge_leaves: leaf positions are 2 or 4 bytes wide");
  PSG_REQUIRE(sc->cmp_end == sc->n, "psg_merge_leaves: leaves are ordered as suffixes of the whole text (comparison end = n)");
  PSG_REQUIRE(h_leaf_beg[0] == range_beg && h_leaf_beg[n_leaves] == range_beg + range_size, "psg_merge_leaves: the leaves tile the range");
  for (i64 l = 0; l < n_leaves; ++l)
    PSG_REQUIRE(h_leaf_beg[l + 1] > h_leaf_beg[l] && (psa_bytes == 4 || h_leaf_beg[l + 1] - h_leaf_beg[l] <= 65536), "psg_merge_leaves: empty leaf / leaf too large for 16-bit positions");
  const bool windowed = sc->text_end > 0;
  PSG_REQUIRE(!windowed || (sc->text_begin <= range_beg && range_beg + range_size <= sc->text_end && sc->text_end <= sc->n), "psg_merge_leaves: the range lies outside the text window");
  const double w0 = wall_ms();
  psg_leaf_merge_stats st = {};
  const i64 R = range_size;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int rc;
  // ---- buffers of a level: position-indexed partial SA / BWT, gt arrays of the nodes (each on a 128-bit boundary), i0 per node
  const i64 gt_words_cap = (R + 31) / 32 + 4 * (n_leaves + 2) + 8;
  DevBuf psa[2], bwt[2], gtb[2], i0d[2], fail, err;
  for (int s = 0; s < 2; ++s)
    if ((rc = psa[s].alloc(4 * R + 64)) || (rc = bwt[s].alloc(R + 64)) || (rc = gtb[s].alloc(4 * gt_words_cap)) || (rc = i0d[s].alloc(8 * (n_leaves + 1)))) return rc;
  if ((rc = fail.alloc(16)) || (rc = err.alloc(16))) return rc;
  PSG_HIP(hipMemsetAsync(fail.p, 0, 16, stream()));
  PSG_HIP(hipMemsetAsync(err.p, 0, 16, stream()));
  TreeText X{sc->d_text, sc->n, windowed ? sc->text_end : sc->n, fail.as<int>()};
  const u8 *text_range = sc->d_text + range_beg;
  std::vector<Node> nodes((size_t)n_leaves);
  {
    // level 0
    std::vector<i64> hl((size_t)(2 * n_leaves + 2));
    i64 cursor = 0;
    for (i64 l = 0; l <= n_leaves; ++l) hl[(size_t)l] = h_leaf_beg[l] - range_beg;
    for (i64 l = 0; l < n_leaves; ++l) {
      const i64 sz = h_leaf_beg[l + 1] - h_leaf_beg[l];
      nodes[(size_t)l] = Node{h_leaf_beg[l] - range_beg, sz, cursor};
      hl[(size_t)(n_leaves + 1 + l)] = cursor;
      cursor += (sz + 127) / 128 * 4;
    }
    DevBuf lb;
    if ((rc = lb.alloc((2 * n_leaves + 2) * 8))) return rc;
    if ((rc = psg::copy_h2d(lb.p, hl.data(), hl.size() * 8))) return rc;
    PSG_HIP(hipMemsetAsync(gtb[0].p, 0, (size_t)(4 * gt_words_cap), stream()));
    EventTimer tm; tm.start();
    if (psa_bytes == 2)
      hipLaunchKernelGGL(leaf_prepare_kernel<u16>, dim3((unsigned)n_leaves), dim3(256), 0, stream(), X, (i64)range_beg, lb.as<i64>(), lb.as<i64>() + n_leaves + 1, (const u16 *)d_leaf_psa,
                         psa[0].as<u32>(), bwt[0].as<u8>(), i0d[0].as<i64>(), gtb[0].as<u32>());
    else
      hipLaunchKernelGGL(leaf_prepare_kernel<u32>, dim3((unsigned)n_leaves), dim3(256), 0, stream(), X, (i64)range_beg, lb.as<i64>(), lb.as<i64>() + n_leaves + 1, (const u32 *)d_leaf_psa,
                         psa[0].as<u32>(), bwt[0].as<u8>(), i0d[0].as<i64>(), gtb[0].as<u32>());
    PSG_HIP(hipGetLastError());
    tm.stop();
    PSG_HIP(psg::sync_stream());
    st.prepare_ms = tm.ms();
  }
  int cur = 0;
  const i64 tile = merge_pairs_tile();
  while (nodes.size() > 1) {
    const i64 M = (i64)nodes.size(), P = M / 2;
    const bool carry = (M & 1) != 0;
    const i64 paired_end = nodes[(size_t)(2 * P - 1)].beg + nodes[(size_t)(2 * P - 1)].size;   // the pairs tile [0, paired_end)
    i64 T_total = 0, T_max = 0, m_total = 0;
    for (i64 p = 0; p < P; ++p) { const i64 T = nodes[(size_t)(2 * p + 1)].size; T_total += T; T_max = std::max(T_max, T); m_total += nodes[(size_t)(2 * p)].size; }
    PSG_REQUIRE(m_total < 0xFFFFFFFEll, "psg_merge_leaves: range too large for a 32-bit rank log");
    // ---- rank structure over the level's BWT array
    EventTimer t_rank; t_rank.start();
    // only the left child of a pair is ranked: positions [beg, beg + size] of the even nodes.  The build segments that lie
    // wholly inside right children (half of the array from the second level on) are skipped by the fill kernel.
    DevBuf seg_mask;
    {
      const i64 nseg = cdiv(paired_end, RANK_BUILD_SEG), nw = cdiv(nseg, 32);
      u32 *hm = (u32 *)pinned_buf(15, (size_t)nw * 4);
      if (!hm) { set_error("psg_merge_leaves: pinned host allocation failed"); return PSG_ENOMEM; }
      memset(hm, 0, (size_t)nw * 4);
      for (i64 p = 0; p < P; ++p) {
        const Node &A = nodes[(size_t)(2 * p)];
        for (i64 sgm = A.beg / RANK_BUILD_SEG, last = std::min(nseg - 1, (A.beg + A.size) / RANK_BUILD_SEG); sgm <= last; ++sgm) hm[sgm >> 5] |= 1u << (sgm & 31);
      }
      if ((rc = seg_mask.alloc(nw * 4))) return rc;
      PSG_HIP(hipMemcpyAsync(seg_mask.p, hm, (size_t)nw * 4, hipMemcpyHostToDevice, stream()));
    }
    psg_rank_t *rk = nullptr;
    rank_build_seg_mask = getenv("PSG_LEAF_FULL_RANK") ? nullptr : seg_mask.as<u32>();
    rc = psg_rank_build(bwt[cur].as<u8>(), paired_end, 0, &rk);
    rank_build_seg_mask = nullptr;
    if (rc) return rc;
    struct RankGuard { psg_rank_t *r; ~RankGuard() { psg_rank_free(r); } } rank_guard{rk};
    t_rank.stop();
    // ---- gap array (shared slots), counter width
    DevBuf gap;
    if ((rc = gap.alloc(4 * PSG_GAP_WORDS(m_total)))) return rc;
    int gbits = 32;
    if ((rc = gap_prepare(gap.as<u32>(), m_total, true, &gbits))) return rc;
    const GapExcess gex = gap_excess(gap.as<u32>(), m_total, gbits);
    int mode = T_total >= ((i64)1 << 22) ? 2 : (gbits == 32 ? 0 : 1);
    if (const char *e = getenv("PSG_GAP_MODE")) { if (!strcmp(e, "log")) mode = 2; else if (!strcmp(e, "atomic")) mode = gbits == 32 ? 0 : 1; else if (!strcmp(e, "ovf")) mode = 1; }
    // ---- chain plan: enough chains to fill the chip, and every pass fills its workgroups
    const i64 Ktarget = (i64)stream_batch_blocks_per_cu(rk, mode) * cus * PSG_WG;
    i64 L = cdiv(cdiv(T_total, Ktarget), 32) * 32;
    const i64 Lwg = cdiv(cdiv(T_max, PSG_WG), 32) * 32;
    L = std::max<i64>(32, std::min(L, Lwg));
    if (L >= 128) L = L / 128 * 128;          // whole 16-byte groups of gt words per chain
    if (const char *e = getenv("PSG_BATCH_CHAIN_LEN")) { const i64 v = atoll(e); if (v >= 32) L = v / 32 * 32; }
    // ---- geometry of the passes, workgroup maps
    std::vector<BatchGeom> geom((size_t)P);
    std::vector<u32> wg_pass, wg_local, cp_pass, cp_local, tile_pass((size_t)cdiv(paired_end, tile));
    std::vector<Node> next;
    next.reserve((size_t)(P + 1));
    i64 kbase = 0, gap_base = 0, ones_before = 0, cursor = 0;
    for (i64 p = 0; p < P; ++p) {
      const Node &A = nodes[(size_t)(2 * p)], &Bn = nodes[(size_t)(2 * p + 1)];
      BatchGeom &G = geom[(size_t)p];
      G.lbeg = A.beg; G.m = A.size; G.T = Bn.size; G.kbase = kbase; G.gap_base = gap_base; G.ones_before = ones_before;
      G.gt_in_word = Bn.gt_word; G.gt_l_word = A.gt_word; G.gt_out_word = cursor; G.node = 2 * p;
      next.push_back(Node{A.beg, A.size + Bn.size, cursor});
      cursor += (A.size + Bn.size + 127) / 128 * 4;
      const i64 K = cdiv(G.T, L);
      for (i64 w = 0; w < cdiv(K, PSG_WG); ++w) { wg_pass.push_back((u32)p); wg_local.push_back((u32)w); }
      const i64 cw = ((G.gt_out_word * 32 + G.T + G.m + 31) >> 5) - ((G.gt_out_word * 32 + G.T) >> 5);   // dst words the block's bits touch
      for (i64 w = 0; w < cdiv(cw, PSG_WG); ++w) { cp_pass.push_back((u32)p); cp_local.push_back((u32)w); }
      for (i64 t = cdiv(A.beg, tile); t * tile < A.beg + A.size + Bn.size; ++t) tile_pass[(size_t)t] = (u32)p;   // the pass that holds slot t * tile
      kbase += K; gap_base += G.m; ones_before += G.T;
    }
    if (carry) { const Node &Cn = nodes[(size_t)(M - 1)]; next.push_back(Node{Cn.beg, Cn.size, cursor}); cursor += (Cn.size + 127) / 128 * 4; }
    const i64 Ktotal = kbase, nwg = (i64)wg_pass.size(), ncp = (i64)cp_pass.size(), ntile = (i64)tile_pass.size();
    // one upload: [geom][wg_pass][wg_local][cp_pass][cp_local][tile_pass]
    const size_t geom_b = (size_t)P * sizeof(BatchGeom), maps_b = (size_t)(2 * nwg + 2 * ncp + ntile) * 4;
    char *hp = (char *)pinned_buf(14, geom_b + maps_b + 64);
    if (!hp) { set_error("psg_merge_leaves: pinned host allocation failed"); return PSG_ENOMEM; }
    memcpy(hp, geom.data(), geom_b);
    u32 *hm = (u32 *)(hp + geom_b);
    memcpy(hm, wg_pass.data(), (size_t)nwg * 4); memcpy(hm + nwg, wg_local.data(), (size_t)nwg * 4);
    memcpy(hm + 2 * nwg, cp_pass.data(), (size_t)ncp * 4); memcpy(hm + 2 * nwg + ncp, cp_local.data(), (size_t)ncp * 4);
    memcpy(hm + 2 * nwg + 2 * ncp, tile_pass.data(), (size_t)ntile * 4);
    DevBuf plan_d, init_d, fin_d, log_d, bv;
    if ((rc = plan_d.alloc((i64)(geom_b + maps_b))) || (rc = init_d.alloc(Ktotal * 8)) || (rc = fin_d.alloc(Ktotal * 8)) || (rc = bv.alloc(4 * ((paired_end + 31) / 32 + 2)))) return rc;
    if (mode == 2 && (rc = log_d.alloc(Ktotal * L * 4))) return rc;
    PSG_HIP(hipMemcpyAsync(plan_d.p, hp, geom_b + maps_b, hipMemcpyHostToDevice, stream()));
    const BatchGeom *d_geom = plan_d.as<BatchGeom>();
    const u32 *d_maps = (const u32 *)(plan_d.as<char>() + geom_b);
    const u32 *d_wg_pass = d_maps, *d_wg_local = d_maps + nwg, *d_cp_pass = d_maps + 2 * nwg, *d_cp_local = d_maps + 2 * nwg + ncp, *d_tile_pass = d_maps + 2 * nwg + 2 * ncp;
    BatchTables *tabs = stream_batch_tables_create(rk, P);
    if (!tabs) { set_error("psg_merge_leaves: allocation of the pass tables failed"); return PSG_ENOMEM; }
    struct TabGuard { BatchTables *t; ~TabGuard() { stream_batch_tables_free(t); } } tab_guard{tabs};
    const int nx = cur ^ 1;
    PSG_HIP(hipMemsetAsync(gtb[nx].p, 0, (size_t)(4 * (cursor + 8)), stream()));
    if (mode < 2) PSG_HIP(hipMemsetAsync(gap.p, 0, (size_t)(4 * (m_total + 1)), stream()));
    // ---- start ranks, the passes, the hand-over check
    EventTimer t_search, t_stream, t_hist, t_merge;
    t_search.start();
    hipLaunchKernelGGL(batch_search_kernel, dim3((unsigned)nwg), dim3(PSG_WG), 0, stream(), X, (i64)range_beg, d_geom, d_wg_pass, d_wg_local, psa[cur].as<u32>(), L, init_d.as<i64>());
    PSG_HIP(hipGetLastError());
    t_search.stop();
    if ((rc = stream_batch_setup(rk, tabs, d_geom, P, i0d[cur].as<i64>(), text_range, gtb[cur].as<u32>(), gtb[nx].as<u32>(), L, init_d.as<i64>(), fin_d.as<i64>(), log_d.as<u32>(),
                                 Ktotal, gap.as<u32>(), gex))) return rc;
    t_stream.start();
    if ((rc = stream_batch_launch(rk, tabs, d_wg_pass, d_wg_local, nwg, mode))) return rc;
    t_stream.stop();
    hipLaunchKernelGGL(batch_handover_kernel, dim3((unsigned)nwg), dim3(PSG_WG), 0, stream(), d_geom, d_wg_pass, d_wg_local, L, init_d.as<i64>(), fin_d.as<i64>(), err.as<int>());
    PSG_HIP(hipGetLastError());
    // ---- gap array -> the pairs' merge bitvectors, back to back
    double hist_ms = 0;
    if (mode == 2) {
      if ((rc = gap_hist_from_log(log_d.as<u32>(), Ktotal * L, m_total, gap.as<u32>(), &hist_ms, true, gex))) return rc;
      log_d.alloc(16);
    }
    {
      int h[8];
      PSG_HIP(hipMemcpyAsync(pinned_buf(3, 64), fail.p, 8, hipMemcpyDeviceToHost, stream()));
      PSG_HIP(hipMemcpyAsync((char *)pinned_buf(3, 64) + 8, err.p, 4, hipMemcpyDeviceToHost, stream()));
      PSG_HIP(psg::sync_stream());
      memcpy(h, pinned_buf(3, 64), 12);
      if (h[0]) { set_error("psg_merge_leaves: a comparison ran past the end of the text window (repeats longer than the window's look-ahead)"); return PSG_EWINDOW; }
      if (h[1]) { set_error("psg_merge_leaves: a start-rank search exceeded its comparison budget (long repeats across leaves)"); return PSG_EUNRESOLVED; }
      if (h[2]) { set_error("psg_merge_leaves: chain hand-over check failed"); return PSG_ECHECK; }
    }
    i64 nb = 0;
    EventTimer t_bv; t_bv.start();
    if ((rc = psg_gap_to_bitvector(gap.as<u32>(), m_total, bv.as<u32>(), paired_end, &nb))) return rc;
    t_bv.stop();
    if (nb != paired_end) { set_error("psg_merge_leaves: gap sum mismatch (" + std::to_string(nb) + " bits, expected " + std::to_string(paired_end) + ")"); return PSG_ECHECK; }
    gap.alloc(16);
    // ---- merge partial SAs + BWTs, concatenate gt bits
    t_merge.start();
    if ((rc = merge_pairs_launch(bv.as<u32>(), paired_end, d_geom, P, d_tile_pass, psa[cur].as<u32>(), bwt[cur].as<u8>(), text_range, psa[nx].as<u32>(), bwt[nx].as<u8>(), i0d[nx].as<i64>()))) return rc;
    hipLaunchKernelGGL(batch_gt_concat_kernel, dim3((unsigned)ncp), dim3(PSG_WG), 0, stream(), d_geom, d_cp_pass, d_cp_local, gtb[cur].as<u32>(), gt_words_cap, gtb[nx].as<u32>());
    PSG_HIP(hipGetLastError());
    if (carry) {
      const Node &Cn = nodes[(size_t)(M - 1)], &Dn = next.back();
      PSG_HIP(hipMemcpyAsync(psa[nx].as<u32>() + Cn.beg, psa[cur].as<u32>() + Cn.beg, (size_t)Cn.size * 4, hipMemcpyDeviceToDevice, stream()));
      PSG_HIP(hipMemcpyAsync(bwt[nx].as<u8>() + Cn.beg, bwt[cur].as<u8>() + Cn.beg, (size_t)Cn.size, hipMemcpyDeviceToDevice, stream()));
      PSG_HIP(hipMemcpyAsync(gtb[nx].as<u32>() + Dn.gt_word, gtb[cur].as<u32>() + Cn.gt_word, (size_t)((Cn.size + 31) / 32) * 4, hipMemcpyDeviceToDevice, stream()));
      PSG_HIP(hipMemcpyAsync(i0d[nx].as<i64>() + P, i0d[cur].as<i64>() + (M - 1), 8, hipMemcpyDeviceToDevice, stream()));
    }
    t_merge.stop();
    PSG_HIP(psg::sync_stream());
    st.levels += 1; st.passes += P; st.suffixes += T_total;
    st.rank_ms += t_rank.ms(); st.search_ms += t_search.ms(); st.stream_ms += t_stream.ms(); st.hist_ms += hist_ms; st.bitvector_ms += t_bv.ms(); st.merge_ms += t_merge.ms();
    nodes.swap(next);
    cur = nx;
  }
  // ---- the root
  PSG_HIP(hipMemcpyAsync(d_psa_out, psa[cur].p, (size_t)R * 4, hipMemcpyDeviceToDevice, stream()));
  PSG_HIP(hipMemcpyAsync(d_bwt_out, bwt[cur].p, (size_t)R, hipMemcpyDeviceToDevice, stream()));
  PSG_HIP(hipMemcpyAsync(d_gt_begin_out, gtb[cur].as<u32>() + nodes[0].gt_word, (size_t)((R + 31) / 32) * 4, hipMemcpyDeviceToDevice, stream()));
  i64 h_i0 = -1;
  if ((rc = psg::copy_d2h(&h_i0, i0d[cur].p, 8))) return rc;
  {
    int h[4];
    if ((rc = psg::copy_d2h(h, fail.p, 8))) return rc;
    if (h[0]) { set_error("psg_merge_leaves: a comparison ran past the end of the text window (repeats longer than the window's look-ahead)"); return PSG_EWINDOW; }
    if (h[1]) { set_error("psg_merge_leaves: a comparison exceeded its budget (long repeats)"); return PSG_EUNRESOLVED; }
  }
  if (h_i0 < 0 || h_i0 >= R) { set_error("psg_merge_leaves: i0 of the merged range out of bounds"); return PSG_ECHECK; }
  *i0 = h_i0;
  st.total_ms = wall_ms() - w0;
  note_kernel_ms(st.stream_ms);
  if (stats) *stats = st;
  return 0;
}
