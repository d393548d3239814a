// ref_harness.cpp -- thin extern "C" wrappers around the REFERENCE's own hot-path
// headers, compiled from where they lie under /root/reference (see Makefile).
//
// TEST INFRASTRUCTURE ONLY (same rule as psascan_oracle.c).  Nothing from the
// reference is copied into this repository: this file is our own glue and the
// built library lands in oracle/_ref/ (git-ignored).  The full construct_sa of
// the reference is NOT buildable here (libdivsufsort/libsais are absent and we
// do not write stand-ins); the headers on the streaming-gap + merge path
// (rank.hpp, compute_gap.hpp, gap_array.hpp, bwt_merge.hpp,
// compute_{left,right}_gap.hpp, merge.hpp, em_compute_initial_ranks.hpp with
// approx_rank.hpp / sparse_isa.hpp) do not need the sorter and compile as they are.
//
// The reference works on files; each wrapper stages its inputs into `workdir`
// and reads the reference's output files back.
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <unistd.h>

#include "utils/utils.hpp"
#include "types/uint40.hpp"
#include "io/multifile.hpp"
#include "io/distributed_file.hpp"
#include "io/async_vbyte_stream_reader.hpp"
#include "rank.hpp"
#include "gap_array.hpp"
#include "bitvector.hpp"
#include "half_block_info.hpp"
#include "bwt_merge.hpp"
#include "compute_gap.hpp"
#include "compute_right_gap.hpp"
#include "compute_left_gap.hpp"
#include "merge.hpp"
#include "em_compute_initial_ranks.hpp"

using namespace psascan_private;

static double g_last_seconds = 0;   // wall time spent inside the reference call proper
static double g_rank_seconds = 0;
static double now_s() { return (double)utils::wclock(); }

static void silence(bool on) {
  static int saved = -1;
  if (on) { fflush(stderr); saved = dup(2); FILE *f = fopen("/dev/null", "w"); dup2(fileno(f), 2); fclose(f); }
  else if (saved >= 0) { fflush(stderr); dup2(saved, 2); close(saved); saved = -1; }
}

static buffered_gap_array *gap_from_values(const uint64_t *v, long len, const std::string &store) {
  buffered_gap_array *g = new buffered_gap_array(len, store);
  for (long j = 0; j < len; ++j) {
    g->m_count[j] = (unsigned char)(v[j] & 255);
    for (uint64_t k = 0; k < (v[j] >> 8); ++k) g->add_excess(j);   // each entry = +256 (gap_array.hpp:116-124)
  }
  return g;
}
static void gap_to_values(buffered_gap_array *g, uint64_t *out) {
  g->start_sequential_access();
  for (long j = 0; j < g->m_length; ++j) out[j] = (uint64_t)g->get_next();
  g->stop_sequential_access();
}

static long g_threads = 2;
static int g_uint40 = 0;   // 1: instantiate compute_gap / merge with T = uint40 (what psascan.hpp:117-125 picks for n >= 2^31)

template <typename T>
static int merge_T(int H, const long *beg, const long *size, const int *const *psa, const uint64_t *const *gap,
                   long ram_use, const char *workdir, uint8_t *out_sa5) {
  std::string wd(workdir), out_fn = wd + "/ref_merge_out.sa5";
  std::vector<half_block_info<T> > hbs;
  long n = 0;
  silence(true);
  for (int h = 0; h < H; ++h) {
    half_block_info<T> hb;
    hb.beg = beg[h]; hb.end = beg[h] + size[h]; n += size[h];
    std::vector<T> vals((size_t)size[h]);
    for (long k = 0; k < size[h]; ++k) vals[(size_t)k] = T((long)psa[h][k]);
    hb.psa = new distributed_file<T>(out_fn, std::max(4L, ram_use / 20L), vals.data(), vals.data() + size[h]);
    if (h + 1 < H) {
      hb.gap_filename = wd + "/ref_gap." + utils::random_string_hash();
      buffered_gap_array *g = gap_from_values(gap[h], size[h] + 1, wd + "/ref_excess4");
      g->save_to_file(hb.gap_filename);
      g->erase_disk_excess();
      delete g;
    }
    hbs.push_back(hb);
  }
  double t0 = now_s();
  merge<T>(out_fn, ram_use, hbs);
  g_last_seconds = now_s() - t0;
  silence(false);
  unsigned char *buf = NULL; long len = 0;
  utils::read_objects_from_file(buf, len, out_fn);
  memcpy(out_sa5, buf, std::min(len, 5 * n)); free(buf);
  utils::file_delete(out_fn);
  return len == 5 * n ? 0 : 1;
}

extern "C" {

double ref_last_seconds(void) { return g_last_seconds; }
double ref_last_rank_build_seconds(void) { return g_rank_seconds; }
void ref_set_threads(long t) { g_threads = t > 0 ? t : 1; }
void ref_set_uint40(int on) { g_uint40 = on ? 1 : 0; }

// rank4n<>::rank (rank.hpp:566-708) and m_count (rank.hpp:112)
int ref_rank(const uint8_t *bwt, long m, const long *qi, const uint8_t *qc, long nq, long *out, long *counts256) {
  rank4n<> *r = new rank4n<>(bwt, (unsigned long)m, 2);
  for (long k = 0; k < nq; ++k) out[k] = r->rank(qi[k], qc[k]);
  if (counts256) for (int c = 0; c < 256; ++c) counts256[c] = r->m_count[c];
  delete r;
  return 0;
}

// compute_gap<int> (compute_gap.hpp:61-157) on in-memory inputs.
//   init_ranks[t] = rank at the END of stream chunk t (stream.hpp:66,108);
//   gt_in / gt_out: bit u <-> position te-u.
int ref_compute_gap(const uint8_t *bwt, long m, long i0, int last, const uint8_t *text, long n,
                    long tb, long te, const uint8_t *gt_in, const long *init_ranks, long n_threads,
                    const char *workdir, uint64_t *gap_out, uint8_t *gt_out) {
  std::string wd(workdir);
  std::string text_fn = wd + "/ref_text.bin", base = wd + "/ref_out";
  utils::write_objects_to_file(text, n, text_fn);
  multifile *gt_in_mf = new multifile();
  {
    std::string fn = wd + "/ref_gtin.bin";
    long nbytes = (te - tb + 7) / 8;
    std::vector<uint8_t> z(nbytes + 1, 0);
    if (gt_in) memcpy(z.data(), gt_in, nbytes);
    utils::write_objects_to_file(z.data(), nbytes, fn);
    gt_in_mf->add_file(n - te, n - tb, fn);
  }
  multifile *gt_out_mf = new multifile();
  double tr0 = now_s();
  rank4n<> *r = new rank4n<>(bwt, (unsigned long)m, (unsigned)g_threads);
  g_rank_seconds = now_s() - tr0;
  buffered_gap_array *gap = new buffered_gap_array(m + 1, base + ".excess");
  std::vector<long> ir(init_ranks, init_ranks + n_threads);
  silence(true);
  double t0 = now_s();
  if (g_uint40) compute_gap<uint40>(r, gap, tb, te, n, n_threads, i0, 1L << 21, (unsigned char)last, ir, text_fn, base, gt_in_mf, gt_out_mf);
  else compute_gap<int>(r, gap, tb, te, n, n_threads, i0, 1L << 21, (unsigned char)last, ir, text_fn, base, gt_in_mf, gt_out_mf);
  g_last_seconds = now_s() - t0;
  silence(false);
  gap_to_values(gap, gap_out);
  memset(gt_out, 0, (te - tb + 7) / 8);
  for (size_t f = 0; f < gt_out_mf->files_info.size(); ++f) {
    const single_file_info &fi = gt_out_mf->files_info[f];
    unsigned char *buf = NULL; long len = 0;
    utils::read_objects_from_file(buf, len, fi.m_filename);
    for (long idx = fi.m_beg; idx < fi.m_end; ++idx) {
      long k = idx - fi.m_beg;
      if (buf[k >> 3] & (1 << (k & 7))) { long u = idx - (n - te); gt_out[u >> 3] |= (uint8_t)(1 << (u & 7)); }
    }
    free(buf);
  }
  gap->erase_disk_excess();
  delete gap; delete r; delete gt_in_mf; delete gt_out_mf;
  utils::file_delete(text_fn);
  return 0;
}

// buffered_gap_array::convert_to_bitvector (gap_array.hpp:273-364)
int ref_gap_to_bitvector(const uint64_t *gap, long m, const char *workdir, uint8_t *bv_out, long nbytes) {
  buffered_gap_array *g = gap_from_values(gap, m + 1, std::string(workdir) + "/ref_excess1");
  silence(true);
  double t0 = now_s();
  bitvector *bv = g->convert_to_bitvector(g_threads);
  g_last_seconds = now_s() - t0;
  silence(false);
  std::string fn = std::string(workdir) + "/ref_bv.bin";
  bv->save(fn);
  unsigned char *buf = NULL; long len = 0;
  utils::read_objects_from_file(buf, len, fn);
  memcpy(bv_out, buf, std::min(len, nbytes));
  free(buf); utils::file_delete(fn);
  g->erase_disk_excess();
  delete bv; delete g;
  return 0;
}

// merge_bwt (bwt_merge.hpp:66-140)
long ref_merge_bwt(const uint8_t *lbwt, const uint8_t *rbwt, long ml, long mr, long li0, long ri0, int left_last,
                   const uint8_t *bv_bytes, uint8_t *out) {
  bitvector bv(ml + mr + 1);
  for (long k = 0; k < ml + mr; ++k) if (bv_bytes[k >> 3] & (1 << (k & 7))) bv.set(k);
  return merge_bwt(lbwt, rbwt, ml, mr, li0, ri0, (unsigned char)left_last, out, &bv, 2);
}

// gap_array_2n (gap_array.hpp:386-529) + compute_right_gap / compute_left_gap.
// Returns values decoded by the reference's own vbyte reader, and the raw vbyte bytes.
static int split_common(int right, const uint64_t *block_gap, const uint8_t *bv_bytes, long ml, long mr,
                        const char *workdir, uint64_t *out_vals, uint8_t *out_vbyte, long *out_nbytes) {
  long block = ml + mr;
  std::string wd(workdir);
  buffered_gap_array *g = gap_from_values(block_gap, block + 1, wd + "/ref_excess2");
  g->flush_excess_to_disk();
  gap_array_2n *g2 = new gap_array_2n(g, 2);
  delete g;
  silence(true);
  g2->apply_excess_from_disk(1L << 20, 2);
  bitvector bv(block + 1);
  for (long k = 0; k < block; ++k) if (bv_bytes[k >> 3] & (1 << (k & 7))) bv.set(k);
  std::string fn = wd + (right ? "/ref_rgap.vb" : "/ref_lgap.vb");
  if (right) compute_right_gap(ml, mr, g2, &bv, fn, 2, 1L << 20);
  else compute_left_gap(ml, mr, g2, &bv, fn, 2, 1L << 20);
  silence(false);
  g2->erase_disk_excess();
  delete g2;
  long cnt = (right ? mr : ml) + 1;
  {
    async_vbyte_stream_reader<long> rd(fn, 1L << 16);
    for (long k = 0; k < cnt; ++k) out_vals[k] = (uint64_t)rd.read();
  }
  if (out_vbyte) {
    unsigned char *buf = NULL; long len = 0;
    utils::read_objects_from_file(buf, len, fn);
    memcpy(out_vbyte, buf, len); *out_nbytes = len; free(buf);
  }
  utils::file_delete(fn);
  return 0;
}
int ref_right_gap(const uint64_t *bg, const uint8_t *bv, long ml, long mr, const char *wd, uint64_t *vals, uint8_t *vb, long *nb) {
  return split_common(1, bg, bv, ml, mr, wd, vals, vb, nb);
}
int ref_left_gap(const uint64_t *bg, const uint8_t *bv, long ml, long mr, const char *wd, uint64_t *vals, uint8_t *vb, long *nb) {
  return split_common(0, bg, bv, ml, mr, wd, vals, vb, nb);
}

// buffered_gap_array::save_to_file (gap_array.hpp:156-182) -> vbyte bytes
long ref_gap_save_vbyte(const uint64_t *gap, long len, const char *workdir, uint8_t *out) {
  std::string wd(workdir), fn = wd + "/ref_gap.vb";
  buffered_gap_array *g = gap_from_values(gap, len, wd + "/ref_excess3");
  silence(true);
  g->save_to_file(fn);
  silence(false);
  g->erase_disk_excess();
  delete g;
  unsigned char *buf = NULL; long nb = 0;
  utils::read_objects_from_file(buf, nb, fn);
  memcpy(out, buf, nb); free(buf); utils::file_delete(fn);
  return nb;
}

// merge<T> (merge.hpp:55-180), T = int or uint40. psa[h] int32 relative to beg[h]; gap[h] u64[size+1] for h < H-1.
int ref_merge(int H, const long *beg, const long *size, const int *const *psa, const uint64_t *const *gap,
              long ram_use, const char *workdir, uint8_t *out_sa5) {
  return g_uint40 ? merge_T<uint40>(H, beg, size, psa, gap, ram_use, workdir, out_sa5)
                  : merge_T<int>(H, beg, size, psa, gap, ram_use, workdir, out_sa5);
}

// em_compute_initial_ranks, first overload (em_compute_initial_ranks.hpp:222-319; call sites partial_sufsort.hpp:189,385):
// ranks of the stream-chunk starts block_end + t * ceil(tail/threads) among the block's suffixes; a comparison that
// reaches block_end is decided by gt bits w.r.t. block_end (gt bit u <-> position tail_end - u, u in [0, tail_end - block_end)),
// ranges that stay open after max_lcp symbols by the sparse ISA (approx_rank.hpp, sparse_isa.hpp).
long ref_initial_ranks(const uint8_t *text, long n, long block_beg, long block_end, const int *psa, const uint8_t *bwt, long i0,
                       long tail_end, const uint8_t *gt, long rank_after_tail, long max_threads, const char *workdir, long *out) {
  std::string wd(workdir), text_fn = wd + "/ref_text_ir.bin", gfn = wd + "/ref_gt_ir.bin";
  utils::write_objects_to_file(text, n, text_fn);
  long nbytes = (tail_end - block_end + 7) / 8;
  std::vector<uint8_t> z(nbytes + 1, 0);
  if (gt) memcpy(z.data(), gt, nbytes);
  utils::write_objects_to_file(z.data(), nbytes, gfn);
  multifile *mf = new multifile();
  mf->add_file(n - tail_end, n - block_end, gfn);
  std::vector<long> res;
  silence(true);
  if (g_uint40) {
    std::vector<uint40> p40((size_t)(block_end - block_beg));
    for (long k = 0; k < block_end - block_beg; ++k) p40[(size_t)k] = uint40((long)psa[k]);
    em_compute_initial_ranks<uint40>(text + block_beg, p40.data(), bwt, i0, block_beg, block_end, n, text_fn, mf, res, max_threads, tail_end, rank_after_tail);
  } else {
    em_compute_initial_ranks<int>(text + block_beg, psa, bwt, i0, block_beg, block_end, n, text_fn, mf, res, max_threads, tail_end, rank_after_tail);
  }
  silence(false);
  for (size_t t = 0; t < res.size(); ++t) out[t] = res[t];
  delete mf;   // (the multifile deletes its files)
  utils::file_delete(text_fn);
  return (long)res.size();
}

// second overload (em_compute_initial_ranks.hpp:513-561; call site partial_sufsort.hpp:311): the block is followed by a
// "mid block" [block_end, tail_begin) whose text is read, comparisons reaching tail_begin are decided by gt bits w.r.t.
// tail_begin (gt bit u <-> position n - u).
long ref_initial_ranks2(const uint8_t *text, long n, long block_beg, long block_end, const int *psa, long tail_begin,
                        const uint8_t *gt, long max_threads, const char *workdir, long *out) {
  std::string wd(workdir), text_fn = wd + "/ref_text_ir2.bin", gfn = wd + "/ref_gt_ir2.bin";
  utils::write_objects_to_file(text, n, text_fn);
  long nbytes = (n - tail_begin + 7) / 8;
  std::vector<uint8_t> z(nbytes + 1, 0);
  if (gt) memcpy(z.data(), gt, nbytes);
  utils::write_objects_to_file(z.data(), nbytes, gfn);
  multifile *mf = new multifile();
  mf->add_file(0, n - tail_begin, gfn);
  std::vector<long> res;
  silence(true);
  if (g_uint40) {
    std::vector<uint40> p40((size_t)(block_end - block_beg));
    for (long k = 0; k < block_end - block_beg; ++k) p40[(size_t)k] = uint40((long)psa[k]);
    em_compute_initial_ranks<uint40>(text + block_beg, p40.data(), block_beg, block_end, n, text_fn, mf, res, max_threads, tail_begin);
  } else {
    em_compute_initial_ranks<int>(text + block_beg, psa, block_beg, block_end, n, text_fn, mf, res, max_threads, tail_begin);
  }
  silence(false);
  for (size_t t = 0; t < res.size(); ++t) out[t] = res[t];
  delete mf;
  utils::file_delete(text_fn);
  return (long)res.size();
}

}  // extern "C"
