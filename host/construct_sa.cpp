// construct_sa -- drop-in for the reference's command line (src/main.cpp:48-246) with the
// streaming-gap + merge path running on an MI355X through the C ABI (include/psascan_amd.h).
//
//   construct_sa [-g GAPFILE] [-m MEM] [-o OUTFILE] [-v] [-h] FILE        (same flags, same defaults)
//   extension:   --block-size N   force max_block_size (bypasses the -m formula; for tests)
//                --chains N       cap the number of backward-search chains per pass
//                --check[=N]      verify the output on the device while it is merged (permutation sum + N sampled
//                                 adjacent pairs per slice in suffix order); failure = exit status 1
//                --discard-output produce the .sa5 bytes (they reach host memory) but write no file
//                --leaf-size N / --fanout F / --no-device-merge
//                                 half-blocks are suffix-sorted as leaves of <= N symbols on the host cores (32 KiB:
//                                 a leaf stays in a core's L2 and hands over 16-bit positions) and merged on the
//                                 device, pairwise, every level of the leaf tree in ONE batch of passes
//                                 (psg_merge_leaves) -- the in-memory pSAscan of the reference
//                                 (inmem_psascan.hpp:64-304) with the GPU as the merger.  --fanout F: the first
//                                 version of that merging, F sub-ranges per step and one pass at a time
//                --device-sort    NOT the reference's placement (north_star keeps the half-block suffix sort on host
//                                 cores, and that is the default): sort the half-blocks on the device with the
//                                 bench's prefix-key sorter (psascan_amd_extras.h; texts whose repeats stay below a
//                                 few hundred symbols -- anything else falls back to the host sorter)
//                --checkpoint DIR after every block the run's state goes to DIR (partial SAs as part files, merge
//                                 bitvectors, gt bits, a manifest renamed into place last); the same command started
//                                 again resumes behind the last finished block.  The reference has no such thing
//                                 (SURVEY 8f row 4): a 1 TiB run that dies in block 40 starts over.
//                --text-on-host [--tail-chunk N]
//                                 the text stays in host memory (automatic beyond 55 % of the device memory): every
//                                 pass uploads its tail in chunks of N symbols (default 1 Gi) and streams them one
//                                 after the other with the exact hand-over rank -- stream.hpp:104-106 reads the tail
//                                 from the text file the same way
//                --hbm-limit N    the external-memory schedule with HBM in the place of the reference's RAM budget: the
//                                 library hands out at most N bytes of device memory (psg_set_memory_limit); the text
//                                 stays in host memory (tails uploaded chunk by chunk), so do the gt bits (a chunk's
//                                 words go up and come back with it: stream.hpp:104-106 reads and writes them through
//                                 files), the partial SAs, and every finished merge bitvector (psg_mbv_spill; the
//                                 reference's gap files, gap_array.hpp:156-182) -- the merge streams all of them back
//                                 slice by slice (merge.hpp:72-81,143-145)
//                --spill-psa      keep the partial suffix arrays in part files next to GAPFILE (-g; default: the
//                                 output name) instead of host memory: `GAPFILE.psa.<beg>` is written when a block
//                                 is done and mapped back for the merge (the reference's distributed_file,
//                                 io/distributed_file.hpp:58-67); host memory then holds one block at a time
//
// Memory tiers: the text, the gt bits and every half-block's merge bitvector stay in HBM; the partial suffix
// arrays stay in HOST memory where the sorter wrote them (the reference keeps them in part files,
// io/distributed_file.hpp:58-67) and are streamed through the device slice by slice during the final merge
// (psg_merge_stream); the input file is memory-mapped, never copied.
//
// The block schedule is the reference's process_block (partial_sufsort.hpp:67-551) and pSAscan
// driver (psascan.hpp:53-131): same block / half-block boundaries for the same -m and thread
// count, so the same passes are executed.  The per-half-block suffix sort stays on the host
// (halfblock.hpp); every hot-path step is a psg_* call.  Output: 5*n bytes, 40-bit LE entries.
#include <fcntl.h>
#include <getopt.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../include/psascan_amd.h"
#include "../include/psascan_amd_extras.h"
#include "halfblock.hpp"
#include "leafsort.hpp"

using psa_host::HalfBlock;

static const char *program_name = "construct_sa";
static bool g_verbose = false;

static double wclock() { timeval t; gettimeofday(&t, nullptr); return t.tv_sec + t.tv_usec * 1e-6; }

static void usage(int status) {
  printf("Usage: %s [OPTION]... FILE\n"
         "Construct the suffix array of text stored in FILE.\n"
         "\n"
         "Mandatory arguments to long options are mandatory for short options too.\n"
         "  -h, --help              display this help and exit\n"
         "  -g, --gap=GAPFILE       accepted for compatibility (prefix of the reference's temporary\n"
         "                          files; this implementation keeps intermediates in HBM)\n"
         "  -m, --mem=MEM           memory budget in bytes that determines the block size exactly as\n"
         "                          in the reference. Metric and IEC suffixes are recognized, e.g.,\n"
         "                          -m 10k, -m 1Mi, -m 3G. Default: 3584Mi\n"
         "  -o, --output=OUTFILE    specify output filename. Default: FILE.sa5\n"
         "  -v, --verbose           print detailed information\n"
         "      --block-size=N      force the maximal block size (extension)\n"
         "      --chains=N          cap the number of chains per streaming pass (extension)\n"
         "      --check[=N]         verify the output on the device: permutation sum and N (default\n"
         "                          4096) sampled adjacent pairs per slice in suffix order (extension)\n"
         "      --discard-output    do everything but write the output file (extension)\n"
         "      --spill-psa         partial suffix arrays in part files GAPFILE.psa.* instead of host\n"
         "                          memory; merge bitvectors that live in host memory (--hbm-limit, or\n"
         "                          too many blocks for the device) in GAPFILE.mbv.* as well (extension)\n"
         "      --leaf-size=N       half-blocks larger than N (default 32Ki) are cut into leaves of at most N\n"
         "                          symbols that are suffix-sorted on the host and merged on the device\n"
         "      --fanout=F          merge F sub-ranges per step, one pass at a time (default: pairwise, every\n"
         "                          level of the leaf tree in one batch of passes; leaves of at most 64Ki)\n"
         "      --no-device-merge   sort every half-block in one piece on the host (extension)\n"
         "      --device-sort       suffix-sort the half-blocks on the device (extension; the default keeps\n"
         "                          the sort on the host cores like the reference)\n"
         "      --checkpoint=DIR    keep the state of the run in DIR after every block (partial SAs, merge\n"
         "                          bitvectors, gt bits); a run started again with the same arguments resumes\n"
         "                          behind the last finished block (the reference starts over)\n"
         "      --hbm-limit=N       use at most N bytes of device memory: the text, the gt bits, the partial suffix\n"
         "                          arrays and the finished merge bitvectors then live in host memory and pass\n"
         "                          through the device piece by piece (the external-memory schedule of the\n"
         "                          reference with HBM in the place of RAM; blocks need ~50 bytes per symbol)\n"
         "      --text-on-host      keep the text in host memory and upload the tail of every pass in chunks\n"
         "                          (automatic for texts beyond 55%% of the device memory); --tail-chunk=N\n",
         program_name);
  std::exit(status);
}

// digits + optional k|m|g|t (decimal) or ki|mi|gi|ti (binary), case-insensitive (main.cpp:76-131)
static bool parse_number(const char *str, uint64_t *ret) {
  *ret = 0;
  size_t len = strlen(str), nd = 0;
  while (nd < len && isdigit((unsigned char)str[nd])) { *ret = *ret * 10 + (uint64_t)(str[nd] - '0'); ++nd; }
  if (nd == 0) return false;
  size_t sl = len - nd;
  if (sl == 0) return true;
  if (sl > 2) return false;
  char a = (char)tolower((unsigned char)str[nd]);
  if (sl == 2 && tolower((unsigned char)str[nd + 1]) != 'i') return false;
  int idx = a == 'k' ? 1 : a == 'm' ? 2 : a == 'g' ? 3 : a == 't' ? 4 : 0;
  if (!idx) return false;
  if (sl == 1) { for (int k = 0; k < idx; ++k) *ret *= 1000; } else *ret <<= (10 * idx);
  return true;
}

static bool file_exists(const std::string &f) { FILE *fp = fopen(f.c_str(), "r"); if (fp) fclose(fp); return fp != nullptr; }

#define CK(call)                                                                                      \
  do { int rc_ = (call); if (rc_ != 0) throw std::runtime_error(std::string(#call) + ": " + psg_last_error()); } while (0)

static void pending_downloads_wait();
static double g_alloc_seconds = 0, g_free_seconds = 0;   // time inside the device allocator (verbose summary)
static double wclock_raw() { timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + tv.tv_usec * 1e-6; }
struct Dev {  // owning device buffer
  void *p = nullptr;
  int64_t bytes = 0;
  Dev() {}
  explicit Dev(int64_t b, bool zero = false) { alloc(b, zero); }
  Dev(const Dev &) = delete;
  Dev &operator=(const Dev &) = delete;
  Dev(Dev &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; }
  Dev &operator=(Dev &&o) noexcept { release(); p = o.p; bytes = o.bytes; o.p = nullptr; return *this; }
  ~Dev() { release(); }
  void alloc(int64_t b, bool zero = false) {
    release(); bytes = b < 16 ? 16 : b;
    const double t0 = wclock_raw();
    if (psg_malloc(&p, bytes) == PSG_ENOMEM) { pending_downloads_wait(); CK(psg_malloc(&p, bytes)); }   // partial SAs on their way to the host hold device memory
    g_alloc_seconds += wclock_raw() - t0;
    if (zero) CK(psg_memset(p, 0, bytes));
  }
  void release() { if (p) { const double t0 = wclock_raw(); psg_free(p); p = nullptr; g_free_seconds += wclock_raw() - t0; } }
  template <class T> T *as() const { return (T *)p; }
};

// the partial SA of a half-block that was built on the device, on its way to host memory in the background
// (psg_d2h_begin: worker thread + own stream; the device copy is freed by the worker the moment it is drained)
struct PendingPsa {
  psg_copy_t *c_lo = nullptr, *c_hi = nullptr;
  static std::mutex &mu() { static std::mutex m; return m; }
  static std::vector<PendingPsa *> &all() { static std::vector<PendingPsa *> v; return v; }
  PendingPsa() { std::lock_guard<std::mutex> lk(mu()); all().push_back(this); }
  void wait() {
    const int r1 = psg_copy_wait(c_lo); c_lo = nullptr;
    const int r2 = psg_copy_wait(c_hi); c_hi = nullptr;
    if (r1 || r2) throw std::runtime_error(std::string("download of a partial suffix array: ") + psg_last_error());
  }
  ~PendingPsa() {
    (void)psg_copy_wait(c_lo); (void)psg_copy_wait(c_hi);
    std::lock_guard<std::mutex> lk(mu());
    auto &v = all();
    v.erase(std::remove(v.begin(), v.end(), this), v.end());
  }
  // device memory is short: wait until every download in flight has handed its source back
  static void wait_all() { std::lock_guard<std::mutex> lk(mu()); for (PendingPsa *p : all()) p->wait(); }
};

static std::function<void()> g_evict_resident;   // set by run(): moves the partial SAs that were kept in HBM to host memory
static void pending_downloads_wait() { PendingPsa::wait_all(); if (g_evict_resident) g_evict_resident(); }
// a library call that allocates its own temporaries: once more after the downloads in flight have released theirs
template <class F> static int with_memory_retry(F &&f) {
  int rc = f();
  if (rc == PSG_ENOMEM) { pending_downloads_wait(); (void)psg_trim(); rc = f(); }
  return rc;
}

// a finished half-block: the partial SA stays in host memory (or in a part file, --spill-psa), the merge bitvector in HBM
struct DoneHalfBlock {
  int64_t beg = 0, size = 0;
  psa_host::PsaVec psa_lo;
  psa_host::PsaHiVec psa_hi;
  Dev psa_dev, psa_hi_dev;        // ... or the partial SA stays in HBM (it fits next to everything else: no PCIe round trip)
  Dev mbv;
  int64_t mbv_bits = 0;           // length of the merge bitvector
  psa_host::PsaVec mbv_host;      // the merge bitvector lives in host memory (psg_mbv_spill_begin; not value-initialised: the download is the first touch) ...
  std::vector<uint64_t> mbv_samp; // ... with its rank samples
  void spill_mbv() {
    if (!mbv.p || mbv_bits <= 0) return;
    mbv_host.resize((size_t)psg_mbv_spill_words(mbv_bits));
    mbv_samp.resize((size_t)((mbv_bits + 4095) / 4096 + 1));
    std::unique_ptr<PendingPsa> P(new PendingPsa());        // in the background, like the partial SAs: the schedule goes on
    if (psg_mbv_spill_begin(mbv.as<uint32_t>(), mbv_bits, mbv_host.data(), mbv_samp.data(), &P->c_lo) != 0) throw std::runtime_error(std::string("psg_mbv_spill_begin: ") + psg_last_error());
    mbv.p = nullptr; mbv.bytes = 0;                          // the library frees it when it is drained
    mbv_pend = std::move(P);
  }
  // --spill-psa with the merge bitvectors in host memory: they go to a file as well ([words][rank samples]) and are
  // mapped for the merge -- the reference's gap files (gap_array.hpp:156-182), here in the unary form the merge reads
  std::string mbv_file;
  void *mbv_map = nullptr;
  size_t mbv_map_bytes = 0, mbv_file_words = 0;
  void spill_mbv_file(const std::string &prefix) {
    if (mbv_host.empty() || !mbv_file.empty()) return;
    settle();
    mbv_file = prefix + ".mbv." + std::to_string(beg);
    mbv_file_words = mbv_host.size();
    mbv_map_bytes = 4 * mbv_host.size() + 8 * mbv_samp.size();
    FILE *f = fopen(mbv_file.c_str(), "wb");
    bool ok = f && fwrite(mbv_host.data(), 4, mbv_host.size(), f) == mbv_host.size() && fwrite(mbv_samp.data(), 8, mbv_samp.size(), f) == mbv_samp.size();
    if (f) ok = fclose(f) == 0 && ok;
    if (!ok) throw std::runtime_error("cannot write the merge bitvector file " + mbv_file);
    psa_host::PsaVec().swap(mbv_host);
    std::vector<uint64_t>().swap(mbv_samp);
  }
  const uint32_t *mbv_w() const { return mbv_map ? (const uint32_t *)mbv_map : (mbv_host.empty() ? nullptr : mbv_host.data()); }
  const uint64_t *mbv_s() const { return mbv_map ? (const uint64_t *)((const char *)mbv_map + 4 * mbv_file_words) : (mbv_samp.empty() ? nullptr : mbv_samp.data()); }
  std::unique_ptr<PendingPsa> pend;   // psa_lo / psa_hi are still being written (settle() before the host reads them)
  std::unique_ptr<PendingPsa> mbv_pend;   // mbv_host likewise
  void settle() {
    if (pend) { pend->wait(); pend.reset(); }
    if (mbv_pend) { mbv_pend->wait(); mbv_pend.reset(); psg_mbv_spill_finish(mbv_host.data(), mbv_bits); }
  }
  // device memory is short (or the partial SA has to go to a file): the resident copy moves to host memory
  void to_host() {
    if (!psa_dev.p) return;
    psa_lo.resize((size_t)size);
    if (psg_d2h(psa_lo.data(), psa_dev.p, 4 * size) != 0) throw std::runtime_error(std::string("download of a partial suffix array: ") + psg_last_error());
    psa_dev.release();
    if (psa_hi_dev.p) {
      psa_hi.resize((size_t)size);
      if (psg_d2h(psa_hi.data(), psa_hi_dev.p, size) != 0) throw std::runtime_error(std::string("download of a partial suffix array: ") + psg_last_error());
      psa_hi_dev.release();
    }
  }
  std::string part_file;          // --spill-psa: [size x u32 low words][size x u8 high bytes, if any]
  bool part_has_hi = false;
  bool keep_part = false;         // checkpointed run: the file outlives this process until the run completes
  void *map = nullptr;
  size_t map_bytes = 0;
  DoneHalfBlock() {}
  DoneHalfBlock(const DoneHalfBlock &) = delete;
  DoneHalfBlock &operator=(const DoneHalfBlock &) = delete;
  DoneHalfBlock(DoneHalfBlock &&o) noexcept { take(o); }
  DoneHalfBlock &operator=(DoneHalfBlock &&o) noexcept { if (this != &o) { drop(); take(o); } return *this; }
  ~DoneHalfBlock() { drop(); }
  void drop() {
    if (mbv_map) munmap(mbv_map, mbv_map_bytes);
    mbv_map = nullptr;
    if (!mbv_file.empty()) remove(mbv_file.c_str());
    mbv_file.clear();
    drop_psa();
  }
  void drop_psa() { pend.reset(); mbv_pend.reset(); if (map) munmap(map, map_bytes); map = nullptr; if (!part_file.empty() && !keep_part) remove(part_file.c_str()); part_file.clear(); }
  void take(DoneHalfBlock &o) {
    beg = o.beg; size = o.size; psa_lo = std::move(o.psa_lo); psa_hi = std::move(o.psa_hi); mbv = std::move(o.mbv); pend = std::move(o.pend); mbv_pend = std::move(o.mbv_pend);
    psa_dev = std::move(o.psa_dev); psa_hi_dev = std::move(o.psa_hi_dev);
    mbv_bits = o.mbv_bits; mbv_host = std::move(o.mbv_host); mbv_samp = std::move(o.mbv_samp);
    mbv_file = std::move(o.mbv_file); o.mbv_file.clear(); mbv_map = o.mbv_map; o.mbv_map = nullptr; mbv_map_bytes = o.mbv_map_bytes; mbv_file_words = o.mbv_file_words;
    part_file = std::move(o.part_file); o.part_file.clear(); part_has_hi = o.part_has_hi; map = o.map; o.map = nullptr; map_bytes = o.map_bytes;
    keep_part = o.keep_part;
  }
  void spill(const std::string &prefix) {
    const std::string want = prefix + ".psa." + std::to_string(beg);
    if (!part_file.empty()) {           // already on disk: under this prefix, or under another one (--spill-psa, then --checkpoint)
      if (part_file == want) return;
      if (map) { munmap(map, map_bytes); map = nullptr; }
      if (rename(part_file.c_str(), want.c_str()) != 0) {   // another file system: copy
        FILE *in = fopen(part_file.c_str(), "rb"), *o = fopen(want.c_str(), "wb");
        bool ok = in && o;
        std::vector<char> buf((size_t)1 << 22);
        for (size_t got; ok && (got = fread(buf.data(), 1, buf.size(), in)) > 0;) ok = fwrite(buf.data(), 1, got, o) == got;
        if (in) { ok = !ferror(in) && ok; fclose(in); }
        if (o) ok = fclose(o) == 0 && ok;
        if (!ok) { remove(want.c_str()); throw std::runtime_error("cannot move the part file " + part_file + " to " + want); }
        remove(part_file.c_str());
      }
      part_file = want;
      return;
    }
    settle();
    to_host();
    part_file = prefix + ".psa." + std::to_string(beg);
    FILE *f = fopen(part_file.c_str(), "wb");
    bool ok = f && fwrite(psa_lo.data(), 4, (size_t)size, f) == (size_t)size;
    part_has_hi = !psa_hi.empty();
    if (ok && part_has_hi) ok = fwrite(psa_hi.data(), 1, (size_t)size, f) == (size_t)size;
    if (f) { ok = fflush(f) == 0 && fsync(fileno(f)) == 0 && ok; ok = fclose(f) == 0 && ok; }   // on the disk before a manifest names it
    if (!ok) throw std::runtime_error("cannot write the part file " + part_file);
    psa_host::PsaVec().swap(psa_lo);
    psa_host::PsaHiVec().swap(psa_hi);
  }
  void map_back() {
    if (!mbv_file.empty() && !mbv_map) {
      int fd = open(mbv_file.c_str(), O_RDONLY);
      mbv_map = fd >= 0 ? mmap(nullptr, mbv_map_bytes, PROT_READ, MAP_SHARED, fd, 0) : MAP_FAILED;
      if (fd >= 0) close(fd);
      if (mbv_map == MAP_FAILED) { mbv_map = nullptr; throw std::runtime_error("cannot map the merge bitvector file " + mbv_file); }
    }
    if (part_file.empty()) return;
    int fd = open(part_file.c_str(), O_RDONLY);
    map_bytes = (size_t)size * (part_has_hi ? 5 : 4);
    map = fd >= 0 ? mmap(nullptr, map_bytes, PROT_READ, MAP_SHARED, fd, 0) : MAP_FAILED;
    if (fd >= 0) close(fd);
    if (map == MAP_FAILED) { map = nullptr; throw std::runtime_error("cannot map the part file " + part_file); }
  }
  const uint32_t *lo() const { return map ? (const uint32_t *)map : psa_lo.data(); }
  const uint8_t *hi() const { return map ? (part_has_hi ? (const uint8_t *)map + 4 * (size_t)size : nullptr) : (psa_hi.empty() ? nullptr : psa_hi.data()); }
};

struct RankGuard {     // a rank structure is hundreds of MiB of HBM: freed on every path out of its scope
  psg_rank_t *r = nullptr;
  RankGuard() {}
  RankGuard(const RankGuard &) = delete;
  RankGuard &operator=(const RankGuard &) = delete;
  ~RankGuard() { reset(); }
  void reset() { if (r) psg_rank_free(r); r = nullptr; }
};
struct PlanGuard {
  psg_merge_plan_t *p = nullptr;
  PlanGuard() {}
  PlanGuard(const PlanGuard &) = delete;
  PlanGuard &operator=(const PlanGuard &) = delete;
  ~PlanGuard() { if (p) psg_merge_plan_free(p); }
};

// The .sa5 (5 bytes per symbol) reaches the page cache at ~10.5 GB/s the first time its pages are written and at ~16 GB/s
// when they are there already (tools/writebench.cpp on the GPU box; parallel writers, O_DIRECT and mmap are all slower).
// So while the blocks are processed and nothing can be written yet (the first entry of the suffix array is known only
// after the last block), one helper thread fills the output file with zeros; the merge then overwrites.  It is stopped
// before the first real byte is written.
struct OutputPrefill {
  std::thread th;
  std::atomic<bool> stop{false};
  std::atomic<int64_t> done{0};
  void start(const std::string &fn, int64_t bytes) {
    th = std::thread([this, fn, bytes]() {
      const int fd = open(fn.c_str(), O_WRONLY);
      if (fd < 0) return;
      const size_t piece = (size_t)32 << 20;
      std::vector<char> z(piece, 0);
      int64_t off = 0;
      while (off < bytes && !stop.load(std::memory_order_relaxed)) {
        const ssize_t w = pwrite(fd, z.data(), (size_t)std::min<int64_t>((int64_t)piece, bytes - off), (off_t)off);
        if (w <= 0) break;                                   // (disk full: the real write reports it)
        off += w;
        done.store(off, std::memory_order_relaxed);
      }
      close(fd);
    });
  }
  void finish() { stop = true; if (th.joinable()) th.join(); }
  ~OutputPrefill() { finish(); }
};

struct MappedFile {   // read-only view of the input (the page cache is the host copy of the text)
  const uint8_t *p = nullptr;
  int64_t n = 0;
  explicit MappedFile(const std::string &fn) {
    int fd = open(fn.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + fn);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("cannot stat " + fn); }
    n = (int64_t)st.st_size;
    if (n > 0) {
      void *m = mmap(nullptr, (size_t)n, PROT_READ, MAP_SHARED, fd, 0);
      if (m == MAP_FAILED) { close(fd); throw std::runtime_error("cannot map " + fn); }
      (void)madvise(m, (size_t)n, MADV_WILLNEED);
      p = (const uint8_t *)m;
    }
    close(fd);
  }
  ~MappedFile() { if (p) munmap((void *)p, (size_t)n); }
  const uint8_t *data() const { return p; }
};

static Dev upload(const void *h, int64_t bytes, int64_t pad = 16) {
  Dev d((bytes + pad - 1) / pad * pad + pad, true);
  if (bytes) CK(psg_h2d(d.p, h, bytes));
  return d;
}

static void log_phase(const char *what, double t0, int64_t units = 0) {
  double dt = wclock() - t0;
  if (units) fprintf(stderr, "    %s: %.2fs (%.2f MiB/s)\n", what, dt, units / 1048576.0 / std::max(dt, 1e-9));
  else fprintf(stderr, "    %s: %.2fs\n", what, dt);
}

struct Options {
  int64_t forced_block = 0, max_chains = 0, check_samples = -1, leaf_size = 0;
  int fanout = 0;                 // 0: leaves merged pairwise, a whole tree level per launch sequence (psg_merge_leaves); >= 2: one pass at a time, F sub-ranges per step
  int64_t tail_chunk = (int64_t)1 << 30;
  bool discard = false, spill_psa = false, hierarchical = true, device_sort = false, text_on_host = false;
  std::string gap_prefix;
  int64_t hbm_limit = 0;          // --hbm-limit N: the device hands out at most N bytes; what does not fit lives in host memory
  std::string checkpoint_dir;     // --checkpoint DIR: state on disk after every block; an interrupted run resumes from it
  int64_t stop_after = 0;         // test hook: stop (exit status 3) after this many blocks of this invocation
};

static void run(const std::string &text_fn, const std::string &out_fn, uint64_t ram_use, long max_threads, const Options &opt_in) {
  Options opt = opt_in;   // adjusted below when the text stays in host memory
  const int64_t forced_block = opt.forced_block, max_chains = opt.max_chains;
  // ---- planner: psascan.hpp:57-91 ----
  if (ram_use < 6) throw std::runtime_error("not enough memory to run pSAscan.");
  MappedFile text(text_fn);
  const int64_t n = text.n;
  fprintf(stderr, "Input filename = %s\nOutput filename = %s\nInput length = %ld (%.1fMiB)\n\n", text_fn.c_str(), out_fn.c_str(), (long)n, n / 1048576.0);
  const int64_t g = 1 << 21;
  int64_t ram_for_threads = 2 * max_threads * g;
  if ((long double)ram_use / 5.2L < (long double)(1LL << 31)) ram_for_threads += max_threads * g;
  else ram_for_threads += (int64_t)(((4.L / 5) * max_threads) * g);
  ram_for_threads += max_threads * g + max_threads * (int64_t)(6 << 20);
  int64_t max_block_size;
  if (forced_block > 0) max_block_size = std::max<int64_t>(2, forced_block);
  else {
    int64_t excl = (int64_t)ram_use - ram_for_threads;
    if (excl < 6) {
      long req = (long)((ram_for_threads + (1 << 20) - 1) / (1 << 20));
      fprintf(stderr, "Error: not enough memory to start threads. You need at least %ldMiB\n", req + 1);
      std::exit(EXIT_FAILURE);
    }
    max_block_size = std::max<int64_t>(2, (int64_t)(excl / 5.2L));
  }
  fprintf(stderr, "RAM budget = %lu (%.1fMiB)\nMax block size = %ld (%.1fMiB)\n\n", (unsigned long)ram_use, ram_use / 1048576.0, (long)max_block_size, max_block_size / 1048576.0);
  double start = wclock();
  if (g_verbose) {   // how long the process existed before this clock started (loader, HIP fat binary registration, mapping the input)
    double up = 0; unsigned long long st = 0;
    if (FILE *f = fopen("/proc/uptime", "r")) { if (fscanf(f, "%lf", &up) != 1) up = 0; fclose(f); }
    if (FILE *f = fopen("/proc/self/stat", "r")) {
      char buf[1024]; size_t k = fread(buf, 1, sizeof buf - 1, f); buf[k] = 0; fclose(f);
      const char *p = strrchr(buf, ')');
      if (p) { int field = 2; for (++p; *p && field < 22; ++p) if (*p == ' ') ++field; if (field == 22) sscanf(p, "%llu", &st); }
    }
    if (up > 0 && st > 0) fprintf(stderr, "Process age when the clock starts: %.2fs\n", up - (double)st / (double)sysconf(_SC_CLK_TCK));
  }
  FILE *out = opt.discard ? nullptr : fopen(out_fn.c_str(), "wb");
  if (!opt.discard && !out) throw std::runtime_error("cannot open output " + out_fn);
  if (n == 0) { if (out) fclose(out); return; }
  CK(psg_init(0));
  if (g_verbose) fprintf(stderr, "Device initialised after %.2fs\n", wclock() - start);
  // everything the host does from here on is copying between pinned buffers, the page cache and the device: keep the
  // threads (this one and all it starts) on the socket the device hangs on
  int numa_node = -1;
  CK(psgx_bind_threads_near_device(&numa_node));
  if (g_verbose && numa_node >= 0) fprintf(stderr, "Host threads bound to the CPUs of NUMA node %d (the device's)\n", numa_node);
  OutputPrefill prefill;
  if (out && n >= ((int64_t)64 << 20) && !getenv("PSASCAN_NO_PREFILL")) prefill.start(out_fn, 5 * n);

  char devname[256];
  CK(psg_device_name(devname, sizeof devname));
  fprintf(stderr, "Device = %s\n\n", devname);
  // ---- where the text lives on the device.  Default: all of it in HBM (1 byte per symbol next to ~45 bytes per block
  // symbol of temporaries).  A text that would take more than 55 % of the device (or --text-on-host) stays in host
  // memory: every pass then uploads its tail in chunks and streams chunk after chunk with the exact hand-over rank
  // (the reference reads the tail from the text file in every pass, stream.hpp:104-106).
  if (opt.hbm_limit > 0) {
    CK(psg_set_memory_limit(opt.hbm_limit));
    fprintf(stderr, "Device memory budget = %ld (%.1fMiB): the text, the gt bits, the partial suffix arrays and the finished merge bitvectors live in host memory\n\n",
            (long)opt.hbm_limit, opt.hbm_limit / 1048576.0);
    if (48 * (double)max_block_size > (double)opt.hbm_limit)
      throw std::runtime_error("the blocks are too large for --hbm-limit: a block of " + std::to_string(max_block_size) + " symbols needs ~50 bytes of device memory per symbol (use a smaller -m)");
  }
  int64_t dev_free = 0, dev_total = 0;
  CK(psg_device_memory(&dev_free, &dev_total));
  const bool text_on_host = opt_in.text_on_host || opt.hbm_limit > 0 || (double)n > 0.55 * (double)dev_total;
  const bool host_tier = opt.hbm_limit > 0;      // gt bits and merge bitvectors in host memory as well
  if (text_on_host) {
    opt.device_sort = false;   // the device sorter reads the text up to its end; the leaf merging works through a window per half-block
    fprintf(stderr, "Text stays in host memory: tails are uploaded in chunks of %ld symbols\n\n", (long)opt_in.tail_chunk);
  }
  Dev d_text(16);                                   // allocated and filled further down, while the host threads sort the first leaves
  Dev tail_buf[2] = {Dev(text_on_host ? opt.tail_chunk + 64 : 16), Dev(text_on_host ? opt.tail_chunk + 64 : 16)};
  const int64_t gt_words = (n + 31) / 32 + 2;
  // gt bits of every position behind the current block w.r.t. its begin (bit n - j), the reference's tail_gt_begin_rev
  // multifile (partial_sufsort.hpp:573-579): in HBM, or -- host tier -- in host memory, a chunk's words travelling with it
  struct GtStore {
    Dev d;
    std::vector<uint32_t> h;
    bool on_host = false;
    void init(int64_t words, bool host) { on_host = host; if (host) h.assign((size_t)words, 0u); else d.alloc(4 * words, true); }
    void zero() { if (on_host) std::fill(h.begin(), h.end(), 0u); else CK(psg_memset(d.p, 0, d.bytes)); }
    uint32_t *dev() const { return on_host ? nullptr : d.as<uint32_t>(); }
    // bits [dst_bit, dst_bit + nbits) = bits [0, nbits) of a device bit array
    void put(int64_t dst_bit, const uint32_t *d_src, int64_t nbits) {
      if (nbits <= 0) return;
      if (!on_host) { CK(psg_bitcopy(d.as<uint32_t>(), dst_bit, d_src, 0, nbits)); return; }
      std::vector<uint32_t> src((size_t)((nbits + 31) / 32 + 1), 0u);
      CK(psg_d2h(src.data(), d_src, 4 * ((nbits + 31) / 32)));
      if (nbits & 31) src[(size_t)((nbits - 1) >> 5)] &= (1u << (nbits & 31)) - 1u;
      const int sh = (int)(dst_bit & 31);
      const int64_t w0 = dst_bit >> 5, nw = (nbits + 31) / 32;
      // (the destination bits are zero: the array is cleared for every block and every bit is written once)
      for (int64_t k = 0; k < nw; ++k) {
        const uint64_t v = (uint64_t)src[(size_t)k] << sh;
        h[(size_t)(w0 + k)] |= (uint32_t)v;
        if (sh) h[(size_t)(w0 + k + 1)] |= (uint32_t)(v >> 32);
      }
    }
    void release() { d.release(); std::vector<uint32_t>().swap(h); }
  } gt_cur, gt_new;
  gt_cur.init(gt_words, host_tier); gt_new.init(gt_words, host_tier);
  std::vector<DoneHalfBlock> hbs;
  std::vector<uint32_t> cur_host;
  g_evict_resident = [&hbs]() { for (DoneHalfBlock &h : hbs) h.to_host(); };
  struct EvictReset { ~EvictReset() { g_evict_resident = nullptr; } } evict_reset;


  const int64_t n_blocks = (n + max_block_size - 1) / max_block_size;

  // ---- placement plan, from the block layout alone.  The merge bitvector of a half-block has one bit per suffix from its
  // begin to the end of the text: with H half-blocks that is ~ H/2 * n/8 bytes in all -- quadratic in the number of
  // blocks (204 GiB for 32 GiB of text in the default 646 MiB blocks), where the reference keeps its gap files on disk
  // (gap_array.hpp:156-182).  What the plan says does not fit in HBM next to the text, the gt bits and the passes'
  // temporaries goes to host memory as soon as it is finished: first the partial SAs (the oldest stay), then the
  // merge bitvectors as well.  The final merge takes either kind from either place.
  int64_t mbv_total = 0;
  for (int64_t bid = n_blocks - 1; bid >= 0; --bid) {
    const int64_t b = max_block_size * bid, e = std::min(b + max_block_size, n);
    const int64_t ls = e == n ? std::min<int64_t>(e - b, std::max<int64_t>(1, (int64_t)(ram_use / 10))) : std::max<int64_t>(1, (e - b) / 2);
    mbv_total += (n - b) / 8 + (e < n ? (n - (b + ls)) / 8 : 0) + 64;   // left half: n - b bits; right half (not in the last block): n - mid
  }
  const int64_t place_budget = (int64_t)(0.92 * (double)dev_total);
  const int64_t place_base = (text_on_host ? 0 : n) + (host_tier ? 0 : 8 * gt_words) + 30 * max_block_size + ((int64_t)8 << 30);
  const bool mbv_on_host = host_tier || place_base + mbv_total > place_budget || getenv("PSASCAN_MBV_ON_HOST") != nullptr;
  int64_t psa_resident_budget = std::max<int64_t>(0, place_budget - place_base - (mbv_on_host ? 0 : mbv_total)), psa_resident = 0;
  if (g_verbose || mbv_on_host)
    fprintf(stderr, "Placement: merge bitvectors %.1f GiB in all -> %s; up to %.1f GiB of partial suffix arrays stay in HBM\n\n", mbv_total / 1073741824.0,
            mbv_on_host ? "host memory" : "HBM", psa_resident_budget / 1073741824.0);

  // ---- --checkpoint DIR.  After every block: the new half-blocks' partial SAs as part files, their merge bitvectors,
  // the gt bits the next block starts from, then a manifest renamed into place -- whatever a crash leaves behind, the
  // manifest names only files that were complete before it was written.  Started again with the same text size, block
  // size and budget, the run loads that state and goes on with the next block.
  const bool ckpt = !opt.checkpoint_dir.empty();
  const std::string ck_prefix = opt.checkpoint_dir + "/ckpt";
  int64_t first_bid = n_blocks - 1;
  auto write_file = [&](const std::string &fn, const void *data, size_t bytes) {
    FILE *f = fopen(fn.c_str(), "wb");
    bool ok = f && fwrite(data, 1, bytes, f) == bytes;
    if (f) { ok = fflush(f) == 0 && fsync(fileno(f)) == 0 && ok; ok = fclose(f) == 0 && ok; }
    if (!ok) throw std::runtime_error("cannot write the checkpoint file " + fn);
  };
  auto read_file = [&](const std::string &fn, void *data, size_t bytes) {
    FILE *f = fopen(fn.c_str(), "rb");
    bool ok = f && fread(data, 1, bytes, f) == bytes;
    if (f) fclose(f);
    if (!ok) throw std::runtime_error("cannot read the checkpoint file " + fn);
  };
  // cheap fingerprint of the text (256 sampled pages): an edited input of the same length does not resume
  auto text_fingerprint = [&]() -> uint64_t {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    const int64_t pages = 256, psz = 4096;
    for (int64_t k = 0; k < pages; ++k) {
      const int64_t off = n <= psz ? 0 : (int64_t)((__int128)(n - psz) * k / (pages - 1));
      const int64_t len = std::min<int64_t>(psz, n - off);
      for (int64_t i = 0; i < len; ++i) { h ^= text.p[off + i]; h *= 1099511628211ull; }
    }
    return h;
  };
  size_t ck_saved = 0;                                       // half-blocks already on disk
  auto checkpoint_block = [&](int64_t bid) {                 // after block bid: hbs, gt_cur are its results
    if (!ckpt) return;
    double t1 = wclock();
    std::vector<uint8_t> hostbuf;
    for (; ck_saved < hbs.size(); ++ck_saved) {
      DoneHalfBlock &h = hbs[ck_saved];
      h.keep_part = true;
      h.settle();
      h.spill(ck_prefix);
      if (h.mbv.p) {
        hostbuf.resize((size_t)h.mbv.bytes);
        CK(psg_d2h(hostbuf.data(), h.mbv.p, h.mbv.bytes));
        write_file(ck_prefix + ".mbv." + std::to_string(h.beg), hostbuf.data(), hostbuf.size());
      } else if (!h.mbv_host.empty()) write_file(ck_prefix + ".mbv." + std::to_string(h.beg), h.mbv_host.data(), 4 * h.mbv_host.size());
    }
    if (gt_cur.on_host) write_file(ck_prefix + ".gt." + std::to_string(bid), gt_cur.h.data(), (size_t)(4 * gt_words));
    else {
      hostbuf.resize((size_t)(4 * gt_words));
      CK(psg_d2h(hostbuf.data(), gt_cur.d.p, 4 * gt_words));
      write_file(ck_prefix + ".gt." + std::to_string(bid), hostbuf.data(), hostbuf.size());
    }
    const std::string mf = ck_prefix + ".manifest", tmp = mf + ".tmp";
    FILE *f = fopen(tmp.c_str(), "w");
    if (!f) throw std::runtime_error("cannot write " + tmp);
    fprintf(f, "psascan-mi355x-checkpoint 2\nn %ld block %ld ram %ld fp %lx\nnext_block %ld\nhalfblocks %zu\n", (long)n, (long)max_block_size, (long)ram_use,
            (unsigned long)text_fingerprint(), (long)(bid - 1), hbs.size());
    for (const DoneHalfBlock &h : hbs) fprintf(f, "%ld %ld %d %ld\n", (long)h.beg, (long)h.size, (int)h.part_has_hi, (long)(h.mbv.p ? h.mbv.bytes : (int64_t)(4 * h.mbv_host.size())));
    bool ok = fflush(f) == 0 && fsync(fileno(f)) == 0;
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), mf.c_str()) != 0) throw std::runtime_error("cannot write " + mf);
    { int dfd = open(opt.checkpoint_dir.c_str(), O_RDONLY); if (dfd >= 0) { (void)fsync(dfd); close(dfd); } }   // the rename itself
    remove((ck_prefix + ".gt." + std::to_string(bid + 1)).c_str());
    if (g_verbose) fprintf(stderr, "    checkpoint: %.2fs\n", wclock() - t1);
  };
  auto checkpoint_clear = [&]() {                            // the run is complete: nothing to resume
    if (!ckpt) return;
    remove((ck_prefix + ".manifest").c_str());
    remove((ck_prefix + ".gt.0").c_str());
    for (DoneHalfBlock &h : hbs) {
      remove((ck_prefix + ".mbv." + std::to_string(h.beg)).c_str());
      h.keep_part = false;                                   // its destructor removes the part file
    }
  };
  if (ckpt) {
    mkdir(opt.checkpoint_dir.c_str(), 0777);
    FILE *f = fopen((ck_prefix + ".manifest").c_str(), "r");
    if (f) {
      long mn = -1, mb = -1, mr = -1, next = -2; size_t cnt = 0; int ver = 0; unsigned long fp = 0;
      bool ok = fscanf(f, "psascan-mi355x-checkpoint %d n %ld block %ld ram %ld fp %lx next_block %ld halfblocks %zu", &ver, &mn, &mb, &mr, &fp, &next, &cnt) == 7 && ver == 2;
      if (ok && (mn != (long)n || mb != (long)max_block_size || mr != (long)ram_use || fp != (unsigned long)text_fingerprint())) {
        fclose(f);
        throw std::runtime_error("the checkpoint in " + opt.checkpoint_dir + " belongs to another run (text, block size or memory budget differ)");
      }
      for (size_t k = 0; ok && k < cnt; ++k) {
        long hb = 0, hs = 0, mv = 0; int hh = 0;
        ok = fscanf(f, "%ld %ld %d %ld", &hb, &hs, &hh, &mv) == 4;
        if (!ok) break;
        DoneHalfBlock h;
        h.beg = hb; h.size = hs; h.part_has_hi = hh != 0; h.keep_part = true;
        h.part_file = ck_prefix + ".psa." + std::to_string(hb);
        if (mv > 0) {
          std::vector<uint8_t> buf((size_t)mv);
          read_file(ck_prefix + ".mbv." + std::to_string(hb), buf.data(), buf.size());
          h.mbv.alloc(mv);
          CK(psg_h2d(h.mbv.p, buf.data(), mv));
          h.mbv_bits = n - hb;
          if (mbv_on_host) h.spill_mbv();
        }
        struct stat sb;
        if (stat(h.part_file.c_str(), &sb) != 0 || sb.st_size != (off_t)((size_t)hs * (hh ? 5 : 4))) throw std::runtime_error("checkpoint part file missing or short: " + h.part_file);
        hbs.push_back(std::move(h));
      }
      fclose(f);
      if (!ok) throw std::runtime_error("unreadable checkpoint manifest in " + opt.checkpoint_dir);
      std::vector<uint8_t> buf((size_t)(4 * gt_words));
      read_file(ck_prefix + ".gt." + std::to_string(next + 1), buf.data(), buf.size());
      if (gt_cur.on_host) memcpy(gt_cur.h.data(), buf.data(), (size_t)(4 * gt_words)); else CK(psg_h2d(gt_cur.d.p, buf.data(), 4 * gt_words));
      first_bid = next;
      ck_saved = hbs.size();
      fprintf(stderr, "Resuming from the checkpoint in %s: %ld of %ld blocks done\n\n", opt.checkpoint_dir.c_str(), (long)(n_blocks - 1 - first_bid), (long)n_blocks);
    }
  }
  int64_t blocks_this_run = 0;
  auto end_block = [&](int64_t bid) {
    checkpoint_block(bid);
    if (opt.stop_after > 0 && ++blocks_this_run >= opt.stop_after && bid > 0) {
      fprintf(stderr, "Stopping after %ld blocks (--stop-after)\n", (long)blocks_this_run);
      fflush(stderr);
      _exit(3);
    }
  };

  // One streaming pass.  The chain start ranks normally come out of the warm-up on the device; on text with long
  // repeats some stay open: the pass then reports PSG_EUNRESOLVED, the partial SAs of the block's halves are
  // uploaded (they live in host memory) and the pass is repeated with a search context -- the open starts are
  // found by string search over them (em_compute_initial_ranks.hpp:222-319), still in one kernel launch.
  struct PartRef { int64_t beg, size; const psa_host::PsaVec *lo; const psa_host::PsaHiVec *hi; DoneHalfBlock *owner; };   // (owner->psa_dev: already on the device)
  Dev gt_chunk_in, gt_chunk_out;   // host tier: the gt words of the tail chunk being streamed
  std::vector<uint32_t> gt_chunk_host;
  auto stream_pass = [&](psg_rank_t *rank, int64_t i0, int last_sym, int64_t tail_beg, int64_t T, const uint32_t *d_gt_in, int64_t rank_at_end,
                         uint32_t *d_gap, uint32_t *d_gt_out, int64_t cmp_end, const uint32_t *d_gt_cmp_end, const std::vector<PartRef> &parts,
                         psg_stream_stats *st, GtStore *h_gt_in = nullptr, GtStore *h_gt_out = nullptr) {
    if (text_on_host) {
      // the tail in chunks, right to left; u = distance from the tail end (chunk bounds are multiples of 64, so the gt
      // words of a chunk start on a word boundary); every chunk starts from the rank the previous one ended with
      const int64_t C = std::max<int64_t>(64, opt.tail_chunk / 64 * 64);
      psg_stream_stats acc{};
      int64_t fin = rank_at_end;
      // two chunk buffers: chunk k+1 goes up in the background (psg_h2d_begin) while chunk k is streamed
      struct Up { psg_copy_t *c = nullptr; ~Up() { (void)psg_copy_wait(c); } } up[2];
      auto chunk_of = [&](int64_t u_lo, int64_t &len, int64_t &pos) { const int64_t u_hi = std::min(T, u_lo + C); len = u_hi - u_lo; pos = tail_beg + (T - u_hi); };
      auto begin_upload = [&](int64_t u_lo, int slot) {
        int64_t len, pos;
        chunk_of(u_lo, len, pos);
        if (len > 0) CK(psg_h2d_begin(tail_buf[slot].p, text.data() + pos, len, &up[slot].c));
      };
      begin_upload(0, 0);
      int slot = 0;
      for (int64_t u_lo = 0; u_lo < T || (T == 0 && u_lo == 0); u_lo += C, slot ^= 1) {
        int64_t len, pos;
        chunk_of(u_lo, len, pos);
        { psg_copy_t *c = up[slot].c; up[slot].c = nullptr; CK(psg_copy_wait(c)); }
        if (u_lo + C < T) begin_upload(u_lo + C, slot ^ 1);   // (the other buffer's pass has completed: passes are synchronous)
        psg_stream_args a{};
        a.rank = rank; a.block_i0 = i0; a.block_last_symbol = last_sym; a.d_tail = tail_buf[slot].as<uint8_t>(); a.tail_len = len; a.right_context = 0;
        a.d_gt_in = d_gt_in ? d_gt_in + (u_lo >> 5) : nullptr; a.rank_at_context_end = fin; a.d_gap = d_gap; a.d_gt_out = d_gt_out ? d_gt_out + (u_lo >> 5) : nullptr;
        const int64_t cw = (len + 31) / 32;                    // gt words of this chunk (chunks start on word boundaries)
        if (h_gt_in && h_gt_in->on_host) {                     // host tier: the chunk's gt words go up with it ...
          if (gt_chunk_in.bytes < 4 * (C / 32 + 8)) gt_chunk_in.alloc(4 * (C / 32 + 8));
          CK(psg_h2d(gt_chunk_in.p, h_gt_in->h.data() + (u_lo >> 5), 4 * (cw + 1)));
          a.d_gt_in = gt_chunk_in.as<uint32_t>();
        }
        if (h_gt_out && h_gt_out->on_host) {
          if (gt_chunk_out.bytes < 4 * (C / 32 + 8)) gt_chunk_out.alloc(4 * (C / 32 + 8));
          CK(psg_memset(gt_chunk_out.p, 0, 4 * (cw + 4)));
          a.d_gt_out = gt_chunk_out.as<uint32_t>();
        }
        a.max_chains = max_chains; a.flags = u_lo == 0 ? PSG_GAP_UNINITIALIZED : 0; a.search = nullptr; a.tail_begin_abs = pos;
        psg_stream_stats s1{};
        if (psg_stream_gap_args(&a, &fin, &s1)) throw std::runtime_error(std::string("psg_stream_gap_args (tail chunk): ") + psg_last_error());
        if (h_gt_out && h_gt_out->on_host) {                   // ... and the new ones come back
          gt_chunk_host.resize((size_t)cw);
          CK(psg_d2h(gt_chunk_host.data(), gt_chunk_out.p, 4 * cw));
          if (len & 31) gt_chunk_host[(size_t)cw - 1] &= (1u << (len & 31)) - 1u;
          for (int64_t w = 0; w < cw; ++w) h_gt_out->h[(size_t)((u_lo >> 5) + w)] |= gt_chunk_host[(size_t)w];
        }
        acc.n_chains = std::max(acc.n_chains, s1.n_chains); acc.chain_len = s1.chain_len; acc.warmup_steps = std::max(acc.warmup_steps, s1.warmup_steps);
        acc.unresolved += s1.unresolved; acc.rounds += s1.rounds; acc.kernel_ms += s1.kernel_ms; acc.total_ms += s1.total_ms; acc.hist_ms += s1.hist_ms;
        if (T == 0) break;
      }
      *st = acc;
      return;
    }
    psg_stream_args a{};
    a.rank = rank; a.block_i0 = i0; a.block_last_symbol = last_sym; a.d_tail = d_text.as<uint8_t>() + tail_beg; a.tail_len = T; a.right_context = 0;
    a.d_gt_in = d_gt_in; a.rank_at_context_end = rank_at_end; a.d_gap = d_gap; a.d_gt_out = d_gt_out; a.max_chains = max_chains;
    a.flags = PSG_GAP_UNINITIALIZED | PSG_FAIL_IF_UNRESOLVED; a.search = nullptr; a.tail_begin_abs = tail_beg;
    int rc = with_memory_retry([&] { return psg_stream_gap_args(&a, nullptr, st); });
    if (rc == PSG_EUNRESOLVED) {
      double t1 = wclock();
      psg_search_ctx sc{};
      sc.d_text = d_text.as<uint8_t>(); sc.n = n; sc.cmp_end = cmp_end; sc.d_gt_cmp_end = cmp_end == n ? nullptr : d_gt_cmp_end;
      sc.nparts = (int)parts.size();
      std::vector<Dev> up;
      for (size_t k = 0; k < parts.size(); ++k) {
        sc.part[k].beg = parts[k].beg; sc.part[k].size = parts[k].size;
        if (parts[k].owner->psa_dev.p) {                      // kept in HBM
          sc.part[k].d_psa_lo = parts[k].owner->psa_dev.as<uint32_t>(); sc.part[k].d_psa_hi = parts[k].owner->psa_hi_dev.as<uint8_t>();
          continue;
        }
        parts[k].owner->settle();
        up.push_back(upload(parts[k].lo->data(), 4 * parts[k].size));
        sc.part[k].d_psa_lo = up.back().as<uint32_t>(); sc.part[k].d_psa_hi = nullptr;
        if (!parts[k].hi->empty()) { up.push_back(upload(parts[k].hi->data(), parts[k].size)); sc.part[k].d_psa_hi = up.back().as<uint8_t>(); }
      }
      a.flags = PSG_GAP_UNINITIALIZED; a.search = &sc;
      rc = psg_stream_gap_args(&a, nullptr, st);
      if (g_verbose) fprintf(stderr, "      long repeats: %ld chain starts found by string search (partial SAs uploaded, %.2fs)\n", (long)st->unresolved, wclock() - t1);
    }
    if (rc) throw std::runtime_error(std::string("psg_stream_gap_args: ") + psg_last_error());
  };

  // ---- look-ahead sorter: LEAVES on all host cores, right to left (the order the schedule needs them).
  // A half-block larger than the leaf size is cut into leaves that ARE suffix-sorted on the host; their partial
  // SAs are merged on the device with the hot path itself -- the reference's in-memory pSAscan does the same with
  // max_threads sub-blocks per block (inmem_psascan.hpp:64-304).  Small leaves stay inside the host caches (a 1 GiB
  // half-block sorts at ~7 MB/s per core, 1-2 MiB leaves at 18-30), and the merging is what the GPU is fast at.  1 MiB
  // balances the two sides on a 16-thread box: 4 GiB of English-like text in 13.2 s with 2 MiB leaves (device busy
  // 6.4 s), 11.95 s with 1 MiB (9.1 s), 15.3 s with 512 KiB (12.5 s: the device is the bound).
  // Comparisons that run past a leaf's end are decided by reading on in the text, which is in host memory, instead
  // of by gt bits of what lies to the right (initial_partial_sufsort.hpp:61-80) -- so a leaf depends on nothing.  A
  // leaf whose comparisons run longer than LOOKAHEAD_CAP symbols (periodic text) fails; its half-block is then sorted
  // in one piece in the sequential schedule, with gt bits from the streaming passes, as in the reference.
  const int64_t LOOKAHEAD_CAP = 1 << 16;
  const bool lookahead = max_threads > 1 && !getenv("PSASCAN_NO_LOOKAHEAD") && !opt.device_sort;
  // Default: leaves of 64 KiB merged in batches.  (The first version -- 1 MiB leaves, four sub-ranges per step, one pass
  // at a time, still there as --fanout F -- spent 9 of its 12 s per 4 GiB on 4095 device passes of ~2 ms fixed cost, and
  // smaller leaves, which the host sorts faster, made it slower: 8190 passes.)
  const bool batched_ok = opt.fanout < 2 && !getenv("PSASCAN_TEST_WIDE_NODES");
  // 32 KiB: the sorter's two key arrays (16 bytes per symbol) stay inside a core's L2 -- 147 MB/s per core on an EPYC 9575F
  // against 109 with 64 KiB leaves (tools/leafbench2.cpp), for one more level on the device
  const int64_t leaf_size = opt.leaf_size > 0 ? opt.leaf_size : (batched_ok ? (int64_t)1 << 15 : (int64_t)1 << 20);
  const bool batched = batched_ok && leaf_size <= 65536;
  struct Task { int64_t beg, end, half_ord; };
  std::vector<Task> tasks;                                   // in the order the schedule consumes them
  std::vector<std::vector<int64_t>> half_tasks((size_t)(2 * n_blocks));   // half id -> its task ids, LEFT to RIGHT
  std::vector<int64_t> half_ord((size_t)(2 * n_blocks), -1);              // half id -> its place in the schedule
  int64_t n_halves_sched = 0, max_half = 1;
  auto half_range = [&](int64_t id, int64_t &hb_beg, int64_t &hb_end) {
    const int64_t bid = id / 2, b = max_block_size * bid, e = std::min(b + max_block_size, n), bs = e - b;
    const bool last = e == n;
    const int64_t ls = last ? std::min<int64_t>(bs, std::max<int64_t>(1, (int64_t)(ram_use / 10))) : std::max<int64_t>(1, bs / 2);
    if (id & 1) { hb_beg = b + ls; hb_end = e; } else { hb_beg = b; hb_end = b + ls; }
  };
  for (int64_t bid = first_bid; bid >= 0; --bid)
    for (int side = 1; side >= 0; --side) {                  // right half first
      int64_t hb, he;
      half_range(2 * bid + side, hb, he);
      if (he <= hb) continue;
      const int64_t nleaves = lookahead && opt.hierarchical ? std::max<int64_t>(1, (he - hb + leaf_size - 1) / leaf_size) : 1;
      std::vector<int64_t> ids;
      half_ord[(size_t)(2 * bid + side)] = n_halves_sched;
      max_half = std::max(max_half, he - hb);
      for (int64_t k = nleaves - 1; k >= 0; --k) {           // rightmost leaf first
        const int64_t lb = hb + (__int128)(he - hb) * k / nleaves, le = hb + (__int128)(he - hb) * (k + 1) / nleaves;
        ids.push_back((int64_t)tasks.size());
        tasks.push_back(Task{lb, le, n_halves_sched});
      }
      ++n_halves_sched;
      std::reverse(ids.begin(), ids.end());
      half_tasks[(size_t)(2 * bid + side)] = ids;
    }
  struct Pre { std::unique_ptr<HalfBlock> hb; bool done = false, failed = false; };
  std::vector<Pre> pre(tasks.size());
  std::mutex pre_mu;
  std::condition_variable pre_cv;
  std::atomic<int64_t> next_task{0};
  // how far the sorters may run ahead of the schedule: a finished leaf holds ~5 bytes per symbol until it is
  // consumed and a running sort ~25 more, so the window is bounded by a quarter of the physical memory
  int64_t consumed = 0;            // tasks the schedule has taken; guarded by pre_mu
  bool stop_workers = false;       // set when the schedule ends (normally or by an exception); guarded by pre_mu
  int64_t window = 2;
  {
    int64_t task_bytes = 1;
    for (const Task &t : tasks) task_bytes = std::max(task_bytes, t.end - t.beg);
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    const int64_t phys = pages > 0 && psz > 0 ? (int64_t)pages * psz : ((int64_t)16 << 30);
    window = std::max<int64_t>(2 * max_threads, std::min<int64_t>((int64_t)1 << 16, (phys / 4) / (30 * task_bytes)));
  }
  // Batched merging: a leaf hands over 16-bit positions only, written straight into the pinned staging buffer of its
  // half-block (one DMA per half-block later); RING half-blocks may be in the making at a time.
  const int RING = 3;
  struct PinnedRing {
    void *p[3] = {nullptr, nullptr, nullptr};
    ~PinnedRing() { for (void *q : p) if (q) psg_host_free(q); }
  } ring;
  int64_t ring_ready = 0, halves_consumed = 0;   // guarded by pre_mu
  const bool use_batched = lookahead && opt.hierarchical && batched && max_half < 0xFFFFFFF0ll;
  if (use_batched) { CK(psg_host_alloc(&ring.p[0], 2 * max_half + 64)); ring_ready = 1; }
  std::vector<std::thread> workers;
  if (lookahead) {
    const long nthreads = std::min<long>(max_threads, (long)tasks.size());
    for (long t = 0; t < nthreads; ++t)
      workers.emplace_back([&]() {
        psa_host::LeafScratch scratch;
        for (;;) {
          const int64_t k = next_task.fetch_add(1);
          if (k >= (int64_t)tasks.size()) return;
          const int64_t ord = tasks[(size_t)k].half_ord;
          {
            std::unique_lock<std::mutex> lk(pre_mu);
            if (use_batched) pre_cv.wait(lk, [&] { return stop_workers || ord < halves_consumed + ring_ready; });
            else pre_cv.wait(lk, [&] { return stop_workers || k < consumed + window; });
            if (stop_workers) return;
          }
          const int64_t hb_beg = tasks[(size_t)k].beg, hb_end = tasks[(size_t)k].end;
          if (use_batched) {
            // the half-block this leaf belongs to: the tasks of a half are contiguous, its first position is the begin of
            // its leftmost leaf = the LAST task of the run with this half_ord
            int64_t kk = k;
            while (kk + 1 < (int64_t)tasks.size() && tasks[(size_t)(kk + 1)].half_ord == ord) ++kk;
            const int64_t half_beg = tasks[(size_t)kk].beg;
            uint16_t *dst = (uint16_t *)ring.p[ord % RING] + (hb_beg - half_beg);
            bool ok = false;
            try { ok = psa_host::sort_leaf16(text.data(), n, hb_beg, hb_end, dst, LOOKAHEAD_CAP, scratch); } catch (...) { ok = false; }
            if (ok && memchr(text.data() + hb_beg, 255, (size_t)(hb_end - hb_beg))) ok = false;   // reported by the sequential path
            std::lock_guard<std::mutex> lk(pre_mu);
            pre[(size_t)k].failed = !ok; pre[(size_t)k].done = true;
            pre_cv.notify_all();
            continue;
          }
          std::unique_ptr<HalfBlock> h;
          bool failed = false;
          try {
            h.reset(new HalfBlock());
            if (!psa_host::sort_halfblock_radix(text.data(), n, hb_beg, hb_end, *h, LOOKAHEAD_CAP)) {   // text with repeats: SA-IS
              h.reset(new HalfBlock());
              psa_host::sort_halfblock(text.data(), n, hb_beg, hb_end, psa_host::gt_tail_direct(text.data(), n, hb_end, LOOKAHEAD_CAP), *h, LOOKAHEAD_CAP);
            }
          } catch (const psa_host::GtCapExceeded &) { h.reset(); failed = true; }
            catch (...) { h.reset(); failed = true; }       // e.g. byte 255: reported by the sequential path
          std::lock_guard<std::mutex> lk(pre_mu);
          pre[(size_t)k].hb = std::move(h); pre[(size_t)k].failed = failed; pre[(size_t)k].done = true;
          pre_cv.notify_all();
        }
      });
  }
  struct Joiner {
    std::vector<std::thread> &w; std::mutex &mu; std::condition_variable &cv; bool &stop;
    ~Joiner() {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv.notify_all();
      for (auto &t : w) if (t.joinable()) t.join();
    }
  } joiner{workers, pre_mu, pre_cv, stop_workers};
  if (!text_on_host && n) {
    const double tu = wclock();
    d_text.alloc((n + 15) / 16 * 16 + 16);
    CK(psg_memset(d_text.as<uint8_t>() + n / 16 * 16, 0, (n + 15) / 16 * 16 + 16 - n / 16 * 16));   // the padding behind the text
    CK(psg_h2d(d_text.p, text.data(), n));                     // (the leaf sorters are running by now)
    if (g_verbose) fprintf(stderr, "Text on the device after %.2fs (upload %.2fs)\n\n", wclock() - start, wclock() - tu);
  }
  // the other staging buffers (page-locking 0.65 GiB takes ~0.1 s), while the first leaves are being sorted.  (On this
  // thread: a helper thread calling hipHostMalloc while this one copies through the library's staging buffers crashed
  // inside the HIP runtime.)
  if (use_batched)
    for (int r = 1; r < RING && r < n_halves_sched; ++r) {
      void *q = nullptr;
      CK(psg_host_alloc(&q, 2 * max_half + 64));
      std::lock_guard<std::mutex> lk(pre_mu);
      ring.p[r] = q; ring_ready = r + 1;
      pre_cv.notify_all();
    }
  // the pre-sorted leaves of half `id`, left to right; empty when there are none (look-ahead off) or one gave up
  auto take_leaves = [&](int64_t id) -> std::vector<std::unique_ptr<HalfBlock>> {
    std::vector<std::unique_ptr<HalfBlock>> out;
    if (!lookahead) return out;
    const std::vector<int64_t> &ids = half_tasks[(size_t)id];
    std::unique_lock<std::mutex> lk(pre_mu);
    pre_cv.wait(lk, [&] { for (int64_t k : ids) if (!pre[(size_t)k].done) return false; return true; });
    consumed += (int64_t)ids.size();                       // the tasks are taken in order, one half at a time
    pre_cv.notify_all();
    bool ok = true;
    for (int64_t k : ids) ok = ok && pre[(size_t)k].hb != nullptr;
    for (int64_t k : ids) { if (ok) out.push_back(std::move(pre[(size_t)k].hb)); else pre[(size_t)k].hb.reset(); }
    return out;
  };

  // batched merging: wait for the leaves of half `id` (16-bit partial SAs in the half's staging buffer), upload them in
  // one DMA and release the buffer to the sorters.  false: a leaf gave up (periodic text, byte 255).
  auto take_staged = [&](int64_t id, Dev &d_leaf, std::vector<int64_t> &bounds) -> bool {
    const std::vector<int64_t> &ids = half_tasks[(size_t)id];
    const int64_t ord = half_ord[(size_t)id];
    bool ok = true;
    {
      std::unique_lock<std::mutex> lk(pre_mu);
      pre_cv.wait(lk, [&] { for (int64_t k : ids) if (!pre[(size_t)k].done) return false; return true; });
      for (int64_t k : ids) ok = ok && !pre[(size_t)k].failed;
    }
    bounds.clear();
    for (int64_t k : ids) bounds.push_back(tasks[(size_t)k].beg);
    bounds.push_back(tasks[(size_t)ids.back()].end);
    const int64_t size = bounds.back() - bounds.front();
    if (ok) {
      d_leaf.alloc(2 * size + 64);
      CK(psg_h2d(d_leaf.p, ring.p[ord % RING], 2 * size));
    }
    std::lock_guard<std::mutex> lk(pre_mu);
    halves_consumed = ord + 1;                                // the staging buffer is free for half ord + RING
    consumed += (int64_t)ids.size();
    pre_cv.notify_all();
    return ok;
  };

  // ---- in-HBM merge of sorted sub-ranges into the partial SA of their union (inmem_psascan.hpp:233-304): the block
  // schedule in small -- sub-range i streams the sub-ranges to its right through its rank structure, the gap array
  // becomes its merge bitvector, one merge yields the union's partial SA; BWT, i0 and gt_begin follow from it.
  // (a node of 2^32 positions or more -- the 8 GiB half-blocks of 16 GiB blocks -- carries bits 32..39 in psa_hi)
  struct DevNode { int64_t beg = 0, size = 0, i0 = 0; Dev psa, psa_hi, bwt, gt; };
  const int64_t wide_from = getenv("PSASCAN_TEST_WIDE_NODES") ? 1 : ((int64_t)1 << 32);   // test hook: every merged node gets a high plane
  int64_t inner_passes = 0, inner_suffixes = 0, inner_levels = 0;
  double bt_prepare = 0, bt_rank = 0, bt_search = 0, bt_stream = 0, bt_hist = 0, bt_bv = 0, bt_merge = 0, bt_total = 0;   // batched merging, ms
  double tm_upload = 0, tm_search = 0, tm_rank = 0, tm_stream = 0, tm_bv = 0, tm_merge = 0, tm_finish = 0;   // where the merging spends its time
  psg_search_ctx sc_text{};                                  // comparisons by reading on in the text (cmp_end = n)
  sc_text.d_text = d_text.as<uint8_t>(); sc_text.n = n; sc_text.cmp_end = n; sc_text.d_gt_cmp_end = nullptr; sc_text.nparts = 0;
  // Text in host memory: the merging of a half-block's leaves sees the text through a WINDOW on the device -- the
  // half-block and the look-ahead behind it -- addressed like the whole text (pointer to position 0 = window - begin).
  // A comparison that would leave the window fails the call (PSG_EWINDOW: repeats longer than the look-ahead that
  // span leaves); that half-block is then sorted in one piece on the host.
  struct WindowExceeded {};
  Dev text_window;
  auto set_window = [&](int64_t hb, int64_t he) {
    if (!text_on_host) return;
    const int64_t wend = std::min<int64_t>(n, he + LOOKAHEAD_CAP + 4096);
    text_window.alloc(wend - hb + 64);
    CK(psg_h2d(text_window.p, text.data() + hb, wend - hb));
    sc_text.d_text = text_window.as<uint8_t>() - hb;
    sc_text.text_begin = hb; sc_text.text_end = wend;
  };
  auto ckw = [&](int rc, const char *what) {
    if (rc == PSG_EWINDOW || (rc == PSG_EUNRESOLVED && !strcmp(what, "psg_merge_leaves"))) throw WindowExceeded();
    if (rc) throw std::runtime_error(std::string(what) + ": " + psg_last_error());
  };
  auto upload_leaf = [&](HalfBlock &h) {
    const double tu = wclock();
    struct Acc { double &a; double t0; ~Acc() { a += wclock() - t0; } } acc{tm_upload, tu};
    DevNode d;
    d.beg = h.beg; d.size = h.size; d.i0 = h.i0;
    d.psa = upload(h.psa_lo.data(), 4 * h.size);
    d.bwt = upload(h.bwt.data(), h.size);
    d.gt = upload(h.gt_begin.data(), 4 * (int64_t)h.gt_begin.size());
    return d;
  };
  auto merge_nodes = [&](std::vector<DevNode> &ch) -> DevNode {
    const int f = (int)ch.size();
    const int64_t b = ch[0].beg, e = ch[(size_t)f - 1].beg + ch[(size_t)f - 1].size, R = e - b;
    const int64_t gtw = (R + 31) / 32 + 4;
    Dev gt_c(4 * gtw, true), gt_n(4 * gtw, true);            // bit u <-> position e - u, w.r.t. the begin of the sub-range being processed
    std::vector<Dev> mbv((size_t)f);
    CK(psg_bitcopy(gt_c.as<uint32_t>(), 0, ch[(size_t)f - 1].gt.as<uint32_t>(), 0, ch[(size_t)f - 1].size));
    for (int i = f - 2; i >= 0; --i) {
      DevNode &c = ch[(size_t)i];
      const int64_t x1 = c.beg + c.size, T = e - x1;
      psg_search_ctx sc = sc_text;
      sc.nparts = 1; sc.part[0].beg = c.beg; sc.part[0].size = c.size; sc.part[0].d_psa_lo = c.psa.as<uint32_t>(); sc.part[0].d_psa_hi = c.psa_hi.as<uint8_t>();
      int64_t r_end = 0;
      double tq = wclock();
      if (e < n) ckw(psg_initial_ranks(&sc, &e, 1, &r_end), "psg_initial_ranks");
      tm_search += wclock() - tq; tq = wclock();
      RankGuard rkg;
      psg_rank_t *&rk = rkg.r;
      CK(with_memory_retry([&] { return psg_rank_build(c.bwt.as<uint8_t>(), c.size, 0, &rk); }));
      tm_rank += wclock() - tq; tq = wclock();
      Dev gap(4 * psg_gap_words(c.size), false);
      CK(psg_memset(gt_n.p, 0, gt_n.bytes));
      psg_stream_args a{};
      a.rank = rk; a.block_i0 = c.i0; a.block_last_symbol = text.p[(size_t)x1 - 1]; a.d_tail = sc_text.d_text + x1; a.tail_len = T; a.right_context = 0;
      a.d_gt_in = gt_c.as<uint32_t>(); a.rank_at_context_end = r_end; a.d_gap = gap.as<uint32_t>(); a.d_gt_out = gt_n.as<uint32_t>(); a.max_chains = max_chains;
      a.flags = PSG_GAP_UNINITIALIZED | PSG_SEARCH_ALL_STARTS; a.search = &sc; a.tail_begin_abs = x1;   // leaves sorted with a bounded look-ahead: no long repeats here
      psg_stream_stats st;
      ckw(with_memory_retry([&] { return psg_stream_gap_args(&a, nullptr, &st); }), "psg_stream_gap_args (sub-range)");
      rkg.reset();
      tm_stream += wclock() - tq; tq = wclock();
      ++inner_passes; inner_suffixes += T;
      mbv[(size_t)i].alloc(4 * ((c.size + T + 31) / 32 + 2), true);
      int64_t nb = 0;
      CK(psg_gap_to_bitvector(gap.as<uint32_t>(), c.size, mbv[(size_t)i].as<uint32_t>(), c.size + T, &nb));
      if (nb != c.size + T) throw std::runtime_error("gap sum mismatch in a sub-range pass");
      CK(psg_bitcopy(gt_n.as<uint32_t>(), T, c.gt.as<uint32_t>(), 0, c.size));   // positions (c.beg, x1]
      std::swap(gt_c, gt_n);
      tm_bv += wclock() - tq;
    }
    double tq = wclock();
    std::vector<psg_hb_desc> desc((size_t)f);
    for (int i = 0; i < f; ++i)
      desc[(size_t)i] = psg_hb_desc{ch[(size_t)i].beg - b, ch[(size_t)i].size, ch[(size_t)i].psa.as<uint32_t>(), ch[(size_t)i].psa_hi.as<uint8_t>(), i + 1 < f ? mbv[(size_t)i].as<uint32_t>() : nullptr};
    PlanGuard pg;
    psg_merge_plan_t *&plan = pg.p;
    CK(psg_merge_plan_create(desc.data(), f, &plan));
    DevNode out;
    out.beg = b; out.size = R;
    out.psa.alloc(4 * R + 16);
    const bool wide = R >= wide_from;
    if (wide) out.psa_hi.alloc(R + 16);
    int mrc = wide ? psg_merge_run_planes(plan, 0, R, out.psa.as<uint32_t>(), out.psa_hi.as<uint8_t>()) : psg_merge_run_u32(plan, 0, R, out.psa.as<uint32_t>());
    psg_merge_plan_free(plan); plan = nullptr;
    if (mrc) throw std::runtime_error(std::string("merge of sub-ranges: ") + psg_last_error());
    ch.clear();                                               // the children's arrays are dead
    tm_merge += wclock() - tq; tq = wclock();
    out.bwt.alloc(R + 16);
    out.gt.alloc(4 * gtw, true);
    ckw(psg_halfblock_from_psa40(&sc_text, b, R, out.psa.as<uint32_t>(), out.psa_hi.as<uint8_t>(), out.bwt.as<uint8_t>(), &out.i0, out.gt.as<uint32_t>()), "psg_halfblock_from_psa40");
    tm_finish += wclock() - tq;
    return out;
  };
  const int fanout = std::max(2, opt.fanout);
  std::function<DevNode(std::vector<std::unique_ptr<HalfBlock>> &, size_t, size_t)> build_tree =
      [&](std::vector<std::unique_ptr<HalfBlock>> &lv, size_t lo, size_t hi) -> DevNode {
    if (hi - lo == 1) { DevNode d = upload_leaf(*lv[lo]); lv[lo].reset(); return d; }
    std::vector<DevNode> ch;
    const size_t cnt = hi - lo, groups = std::min<size_t>((size_t)fanout, cnt);
    for (size_t g = 0; g < groups; ++g) ch.push_back(build_tree(lv, lo + cnt * g / groups, lo + cnt * (g + 1) / groups));
    return merge_nodes(ch);
  };

  // one half-block, ready for the block schedule: BWT and gt_begin in HBM, the partial SA in host memory
  struct Half { int64_t beg = 0, size = 0, i0 = 0; Dev bwt, gt; psa_host::PsaVec psa_lo; psa_host::PsaHiVec psa_hi; Dev psa_dev, psa_hi_dev; std::vector<uint32_t> gt_host; std::unique_ptr<PendingPsa> pend;
                int64_t query_rank = 0; bool have_query_rank = false; };
  auto gt_host_of = [&](Half &h) -> const std::vector<uint32_t> & {   // gt_begin on the host (sequential sorter of the half to the left)
    if (h.gt_host.empty()) { h.gt_host.resize((size_t)((h.size + 31) / 32 + 1)); CK(psg_d2h(h.gt_host.data(), h.gt.p, 4 * (int64_t)h.gt_host.size())); }
    return h.gt_host;
  };
  // query_pos >= 0: also the rank of text[query_pos..) among the half-block's suffixes, if the half-block is built on the device
  auto make_half = [&](int64_t id, int64_t hb, int64_t he, const psa_host::GtTail &gt_tail, const char *what, int64_t query_pos) {
    Half H;
    H.beg = hb; H.size = he - hb;
    const double t0 = wclock();
    auto take_root = [&](DevNode &root) {                      // the merged node becomes the half-block: partial SA to the host
      H.i0 = root.i0;
      H.bwt = std::move(root.bwt); H.gt = std::move(root.gt);
      if (query_pos >= 0) {                                   // while the partial SA is on the device: K8 (em_compute_initial_ranks.hpp:222-319)
        psg_search_ctx sc = sc_text;
        sc.nparts = 1; sc.part[0].beg = hb; sc.part[0].size = H.size; sc.part[0].d_psa_lo = root.psa.as<uint32_t>(); sc.part[0].d_psa_hi = root.psa_hi.as<uint8_t>();
        CK(psg_initial_ranks(&sc, &query_pos, 1, &H.query_rank));
        H.have_query_rank = true;
      }
      // A partial SA that fits in HBM next to what the rest of the run needs (~50 bytes per block symbol at the peak of
      // pass B) stays there: the final merge then reads it where it lies instead of 4-5 bytes per symbol crossing PCIe
      // twice.  If device memory runs short after all, the resident ones are moved to host memory (g_evict_resident).
      {
        int64_t in_use = 0, peak = 0, reserved = 0;
        psg_mem_stats(&in_use, &peak, &reserved);
        const int64_t reserve = 30 * max_block_size + ((int64_t)6 << 30);
        const int64_t mine = root.psa.bytes + root.psa_hi.bytes;
        const bool fits = in_use + reserve < place_budget && psa_resident + mine <= psa_resident_budget;   // now, and by the plan
        if (fits && !text_on_host && !opt.spill_psa && !ckpt && !getenv("PSASCAN_PSA_ON_HOST")) {
          H.psa_dev = std::move(root.psa); H.psa_hi_dev = std::move(root.psa_hi);
          psa_resident += mine;
          return;
        }
      }
      H.psa_lo.resize((size_t)H.size);                        // (default-initialised: the download is the first touch)
      std::unique_ptr<PendingPsa> P(new PendingPsa());
      CK(psg_d2h_begin(H.psa_lo.data(), root.psa.p, 4 * H.size, 1, &P->c_lo));
      root.psa.p = nullptr;                                   // the worker frees it when it is drained
      if (root.psa_hi.p) {
        H.psa_hi.resize((size_t)H.size);
        CK(psg_d2h_begin(H.psa_hi.data(), root.psa_hi.p, H.size, 1, &P->c_hi));
        root.psa_hi.p = nullptr;
      }
      H.pend = std::move(P);                                  // in the background: the schedule goes on with the next sort / pass
    };
    if (opt.device_sort) {
      // same input restriction as the host sorter and the reference (initial_partial_sufsort.hpp:141-146)
      if (memchr(text.data() + hb, 255, (size_t)H.size)) throw std::runtime_error("the input contains byte 255");
      // the device sorter holds 32-bit positions: a half-block of 2^32 symbols or more (or --leaf-size with --device-sort:
      // pieces of that size) is sorted in pieces that are merged like host-sorted leaves
      const int64_t piece_max = opt.leaf_size > 0 ? opt.leaf_size : ((int64_t)1 << 31);
      const int64_t np = H.size < ((int64_t)1 << 32) && opt.leaf_size <= 0 ? 1 : (H.size + piece_max - 1) / piece_max;
      std::vector<DevNode> pieces;
      bool ok = true;
      for (int64_t k = 0; k < np && ok; ++k) {
        DevNode d;
        d.beg = hb + H.size * k / np; d.size = hb + H.size * (k + 1) / np - d.beg;
        d.psa.alloc(4 * d.size + 16); d.bwt.alloc(d.size + 16); d.gt.alloc(4 * ((d.size + 31) / 32 + 2), true);
        int64_t ties = 0;
        const int src = with_memory_retry([&] { return psgx_sort_halfblock(d_text.as<uint8_t>(), n, d.beg, d.beg + d.size, d.psa.as<uint32_t>(), d.bwt.as<uint8_t>(), &d.i0, d.gt.as<uint32_t>(), &ties); });
        if (src == PSG_ENOMEM) throw std::runtime_error(std::string("device sufsort: ") + psg_last_error());   // not a reason to sort gigabytes on one host thread
        ok = src == 0;
        if (ok) pieces.push_back(std::move(d));
      }
      if (ok) {
        if (np == 1) take_root(pieces[0]);
        else {
          const double t1 = wclock();
          const int64_t p0 = inner_passes;
          DevNode root = merge_nodes(pieces);
          const double t2 = wclock();
          take_root(root);
          if (g_verbose) fprintf(stderr, "      hand-over of the partial SA: %.2fs\n", wclock() - t2);
          if (g_verbose) fprintf(stderr, "      %ld pieces sorted on the device, merged in %.2fs (%ld passes)\n", (long)np, wclock() - t1, (long)(inner_passes - p0));
        }
        log_phase((std::string("device sufsort (") + what + " half)").c_str(), t0, H.size);
        return H;
      }
      fprintf(stderr, "    device sufsort (%s half) gave up (%s): host sorter\n", what, psg_last_error());
    }
    if (use_batched) {
      Dev d_leaf;
      std::vector<int64_t> bounds;
      const bool staged = take_staged(id, d_leaf, bounds);
      const double t_wait = wclock() - t0;
      bool merged = false;
      if (staged) {
        try {
          set_window(hb, he);
          DevNode root;
          root.beg = hb; root.size = H.size;
          root.psa.alloc(4 * H.size + 16); root.bwt.alloc(H.size + 16); root.gt.alloc(4 * ((H.size + 31) / 32 + 4), true);
          psg_leaf_merge_stats ls{};
          ckw(with_memory_retry([&] { return psg_merge_leaves(&sc_text, hb, H.size, bounds.data(), (int64_t)bounds.size() - 1, d_leaf.p, 2, root.psa.as<uint32_t>(),
                                                              root.bwt.as<uint8_t>(), &root.i0, root.gt.as<uint32_t>(), &ls); }), "psg_merge_leaves");
          d_leaf.release();
          inner_passes += ls.passes; inner_suffixes += ls.suffixes; inner_levels += ls.levels;
          bt_prepare += ls.prepare_ms; bt_rank += ls.rank_ms; bt_search += ls.search_ms; bt_stream += ls.stream_ms; bt_hist += ls.hist_ms; bt_bv += ls.bitvector_ms;
          bt_merge += ls.merge_ms; bt_total += ls.total_ms;
          take_root(root);
          merged = true;
          fprintf(stderr, "    sufsort (%s half): %zu leaves sorted ahead on the host (waited %.2fs), merged on the device in %.2fs (%ld levels, %ld passes, %.1f Mi suffixes streamed)\n",
                  what, bounds.size() - 1, t_wait, wclock() - t0 - t_wait, (long)ls.levels, (long)ls.passes, ls.suffixes / 1048576.0);
        } catch (const WindowExceeded &) {
          fprintf(stderr, "    sufsort (%s half): a comparison between leaves ran past the text window on the device: host sorter\n", what);
        }
        text_window.release();
      }
      if (!merged) {                                           // periodic text, long repeats across leaves: one sort with gt bits, as in the reference
        const double t1 = wclock();
        HalfBlock h;
        psa_host::sort_halfblock(text.data(), n, hb, he, gt_tail, h);
        H.i0 = h.i0;
        H.bwt = upload(h.bwt.data(), h.size);
        H.gt = upload(h.gt_begin.data(), 4 * (int64_t)h.gt_begin.size());
        H.gt_host.swap(h.gt_begin);
        H.psa_lo.swap(h.psa_lo); H.psa_hi.swap(h.psa_hi);
        fprintf(stderr, "    host sufsort (%s half): %.2fs (%.2f MiB/s)\n", what, wclock() - t1, H.size / 1048576.0 / std::max(wclock() - t1, 1e-9));
      }
      return H;
    }
    std::vector<std::unique_ptr<HalfBlock>> leaves = take_leaves(id);
    const double t_wait = wclock() - t0;
    auto from_host = [&](HalfBlock &h) {
      H.i0 = h.i0;
      H.bwt = upload(h.bwt.data(), h.size);
      H.gt = upload(h.gt_begin.data(), 4 * (int64_t)h.gt_begin.size());
      H.gt_host.swap(h.gt_begin);
      H.psa_lo.swap(h.psa_lo); H.psa_hi.swap(h.psa_hi);
    };
    bool merged = false;
    if (leaves.size() > 1) {
      const size_t nl = leaves.size();
      const int64_t p0 = inner_passes, s0 = inner_suffixes;
      try {
        set_window(hb, he);
        DevNode root = build_tree(leaves, 0, nl);
        take_root(root);
        merged = true;
        fprintf(stderr, "    sufsort (%s half): %zu leaves sorted ahead on the host (waited %.2fs), merged on the device in %.2fs (%ld passes, %.1f Mi suffixes streamed)\n",
                what, nl, t_wait, wclock() - t0 - t_wait, (long)(inner_passes - p0), (inner_suffixes - s0) / 1048576.0);
      } catch (const WindowExceeded &) {
        fprintf(stderr, "    sufsort (%s half): a comparison between leaves ran past the text window on the device: host sorter\n", what);
        leaves.clear();
      }
      text_window.release();
    }
    if (merged) {
    } else if (leaves.size() == 1) {
      from_host(*leaves[0]);
      fprintf(stderr, "    host sufsort (%s half, sorted ahead; waited): %.2fs\n", what, t_wait);
    } else {                                                  // sequential schedule: one sort with gt bits
      HalfBlock h;
      psa_host::sort_halfblock(text.data(), n, hb, he, gt_tail, h);
      from_host(h);
      fprintf(stderr, "    host sufsort (%s half): %.2fs (%.2f MiB/s)\n", what, wclock() - t0, H.size / 1048576.0 / std::max(wclock() - t0, 1e-9));
    }
    return H;
  };
  auto keep_half = [&](Half &h) {   // the partial SA stays on the host; everything else of the half-block is dropped
    DoneHalfBlock d;
    d.beg = h.beg; d.size = h.size;
    d.psa_lo.swap(h.psa_lo); d.psa_hi.swap(h.psa_hi);
    d.psa_dev = std::move(h.psa_dev); d.psa_hi_dev = std::move(h.psa_hi_dev);
    d.pend = std::move(h.pend);
    return d;
  };

  for (int64_t bid = first_bid; bid >= 0; --bid) {      // partial_sufsort.hpp:568
    const int64_t b = max_block_size * bid, e = std::min(b + max_block_size, n), bs = e - b;
    const bool last_block = e == n;
    if (!last_block && bs <= 1) throw std::runtime_error("any block other than the last has to be of length at least two.");
    const int64_t ls = last_block ? std::min<int64_t>(bs, std::max<int64_t>(1, (int64_t)(ram_use / 10))) : std::max<int64_t>(1, bs / 2);
    const int64_t rs = bs - ls, mid = b + ls;
    fprintf(stderr, "Process block %ld/%ld [%ld..%ld):\n", (long)(n_blocks - bid), (long)n_blocks, (long)b, (long)e);
    if (g_verbose) fprintf(stderr, "    [%.2fs since start]\n", wclock() - start);
    gt_new.zero();
    bool have_cur_host = false;
    auto gt_tail_e = [&](int64_t v) {     // [text[e+v..) > text[e..)] from the previous block's passes (fetched when a sequential sort asks)
      int64_t idx = n - (e + v);
      if (gt_cur.on_host) return (bool)((gt_cur.h[(size_t)(idx >> 5)] >> (idx & 31)) & 1u);
      if (!have_cur_host) { cur_host.resize((size_t)gt_words); CK(psg_d2h(cur_host.data(), gt_cur.d.p, 4 * gt_words)); have_cur_host = true; }
      return (bool)((cur_host[(size_t)(idx >> 5)] >> (idx & 31)) & 1u);
    };
    Half R, L;
    if (rs > 0) R = make_half(2 * bid + 1, mid, e, gt_tail_e, "right", -1);
    auto gt_tail_mid = [&](int64_t v) {  // position mid+v in (mid, e]: right half's gt_begin, u = e - j
      if (rs == 0) return gt_tail_e(v);
      int64_t u = e - (mid + v);
      return (bool)((gt_host_of(R)[(size_t)(u >> 5)] >> (u & 31)) & 1u);
    };
    L = make_half(2 * bid, b, mid, gt_tail_mid, "left", rs > 0 && !text_on_host ? e : -1);   // (text on the host: e lies outside the half-block's window)
    double t0 = wclock();
    if (rs == 0) {
      DoneHalfBlock hbL = keep_half(L);
      if (opt.spill_psa) hbL.spill(opt.gap_prefix);
      gt_new.put(n - mid, L.gt.as<uint32_t>(), ls);
      hbs.push_back(std::move(hbL));
      std::swap(gt_cur, gt_new);
      end_block(bid);
      continue;
    }
    Dev &d_lbwt = L.bwt, &d_rbwt = R.bwt, &d_rgt = R.gt, &d_lgt = L.gt;
    // ---- pass A (partial_sufsort.hpp:403-414)
    t0 = wclock();
    RankGuard rankLg, rankBg;
    psg_rank_t *&rankL = rankLg.r;
    CK(with_memory_retry([&] { return psg_rank_build(d_lbwt.as<uint8_t>(), ls, 0, &rankL); }));
    log_phase("Construct rank (left half, device)", t0, ls);
    Dev gapA(4 * psg_gap_words(ls), false), gtA(4 * ((rs + 31) / 32 + 2), true);   // fresh gap array: PSG_GAP_UNINITIALIZED
    int64_t initA = 0;                                        // rank of text[e..) among the left half's suffixes
    if (L.have_query_rank) initA = L.query_rank;              // found on the device before the partial SA left it
    else {                                                  // host search over the partial SA
      if (L.pend) L.pend->wait();
      HalfBlock Lview;
      Lview.beg = b; Lview.size = ls; Lview.psa_lo.swap(L.psa_lo); Lview.psa_hi.swap(L.psa_hi);
      initA = psa_host::rank_by_search(text.data(), n, Lview, e);
      L.psa_lo.swap(Lview.psa_lo); L.psa_hi.swap(Lview.psa_hi);
    }
    const int64_t L_i0 = L.i0, R_i0 = R.i0;
    DoneHalfBlock hbL = keep_half(L), hbR = keep_half(R);
    psg_stream_stats st;
    t0 = wclock();
    stream_pass(rankL, L_i0, text.p[(size_t)mid - 1], mid, rs, d_rgt.as<uint32_t>(), initA, gapA.as<uint32_t>(), gtA.as<uint32_t>(), e, gt_cur.dev(),
                {PartRef{hbL.beg, hbL.size, &hbL.psa_lo, &hbL.psa_hi, &hbL}}, &st);
    log_phase("Stream (right half through left half, device)", t0, rs);
    if (g_verbose) fprintf(stderr, "      chains=%ld len=%ld warmup=%ld unresolved=%ld rounds=%ld kernel=%.2fms histogram=%.2fms call=%.2fms\n", (long)st.n_chains, (long)st.chain_len, (long)st.warmup_steps, (long)st.unresolved, (long)st.rounds, st.kernel_ms, st.hist_ms, st.total_ms);
    rankLg.reset();
    Dev bvA(4 * ((bs + 31) / 32 + 2), true);
    int64_t nb = 0;
    CK(psg_gap_to_bitvector(gapA.as<uint32_t>(), ls, bvA.as<uint32_t>(), bs, &nb));
    if (nb != bs) throw std::runtime_error("gap sum mismatch after pass A");
    gapA.release();
    if (last_block) {  // :418-429 -- the left half's gap array is its merge bitvector
      hbL.mbv = std::move(bvA);
      hbL.mbv_bits = bs;
      if (mbv_on_host) hbL.spill_mbv();
      gt_new.put(n - e, gtA.as<uint32_t>(), rs);
      gt_new.put(n - mid, d_lgt.as<uint32_t>(), ls);
      if (opt.spill_psa) {
      hbL.spill(opt.gap_prefix); hbR.spill(opt.gap_prefix);
      if (mbv_on_host && !ckpt) { hbL.spill_mbv_file(opt.gap_prefix); hbR.spill_mbv_file(opt.gap_prefix); }   // (a checkpoint keeps its own copy of them)
    }
      hbs.push_back(std::move(hbL)); hbs.push_back(std::move(hbR));
      std::swap(gt_cur, gt_new);
      end_block(bid);
      continue;
    }
    // ---- BWT merge (:468-471)
    t0 = wclock();
    Dev d_bbwt(bs + 16);
    int64_t block_i0 = -1;
    CK(psg_merge_bwt(d_lbwt.as<uint8_t>(), d_rbwt.as<uint8_t>(), ls, rs, L_i0, R_i0, text.p[(size_t)mid - 1], bvA.as<uint32_t>(), d_bbwt.as<uint8_t>(), &block_i0));
    d_lbwt.release(); d_rbwt.release();
    log_phase("Merge BWTs of half-blocks (device)", t0, bs);
    // ---- pass B (:500-514)
    t0 = wclock();
    psg_rank_t *&rankB = rankBg.r;
    CK(with_memory_retry([&] { return psg_rank_build(d_bbwt.as<uint8_t>(), bs, 0, &rankB); }));
    d_bbwt.release();
    log_phase("Construct rank (block, device)", t0, bs);
    const int64_t T = n - e;
    Dev gapB(4 * psg_gap_words(bs), false);
    t0 = wclock();
    stream_pass(rankB, block_i0, text.p[(size_t)e - 1], e, T, gt_cur.dev(), 0, gapB.as<uint32_t>(), gt_new.dev(), e, gt_cur.dev(),
                {PartRef{hbL.beg, hbL.size, &hbL.psa_lo, &hbL.psa_hi, &hbL}, PartRef{hbR.beg, hbR.size, &hbR.psa_lo, &hbR.psa_hi, &hbR}}, &st, &gt_cur, &gt_new);
    log_phase("Stream (tail through block, device)", t0, T);
    if (g_verbose) fprintf(stderr, "      chains=%ld len=%ld warmup=%ld unresolved=%ld rounds=%ld kernel=%.2fms histogram=%.2fms call=%.2fms\n", (long)st.n_chains, (long)st.chain_len, (long)st.warmup_steps, (long)st.unresolved, (long)st.rounds, st.kernel_ms, st.hist_ms, st.total_ms);
    rankBg.reset();
    gt_new.put(n - e, gtA.as<uint32_t>(), rs);
    gt_new.put(n - mid, d_lgt.as<uint32_t>(), ls);
    // ---- split (:536-542)
    t0 = wclock();
    hbL.mbv.alloc(4 * ((bs + T + 31) / 32 + 2));
    hbR.mbv.alloc(4 * ((rs + T + 31) / 32 + 2));
    CK(psg_split_gap(gapB.as<uint32_t>(), bvA.as<uint32_t>(), ls, rs, T, hbL.mbv.as<uint32_t>(), hbR.mbv.as<uint32_t>()));
    log_phase("Compute gaps of half-blocks (device)", t0, bs);
    hbL.mbv_bits = bs + T; hbR.mbv_bits = rs + T;
    if (mbv_on_host) { const double ts = wclock(); hbL.spill_mbv(); hbR.spill_mbv(); if (g_verbose) fprintf(stderr, "    merge bitvectors to host memory: %.2fs\n", wclock() - ts); }
    if (opt.spill_psa) {
      hbL.spill(opt.gap_prefix); hbR.spill(opt.gap_prefix);
      if (mbv_on_host && !ckpt) { hbL.spill_mbv_file(opt.gap_prefix); hbR.spill_mbv_file(opt.gap_prefix); }   // (a checkpoint keeps its own copy of them)
    }
    hbs.push_back(std::move(hbL)); hbs.push_back(std::move(hbR));
    std::swap(gt_cur, gt_new);
    end_block(bid);
  }
  if (inner_passes) {
    fprintf(stderr, "\nIn-HBM merging of host-sorted leaves: %ld passes, %.2f Gi suffixes streamed\n", (long)inner_passes, inner_suffixes / 1073741824.0);
    if (g_verbose) {
      double drv = 0; int64_t segs = 0, segb = 0;
      (void)psgx_arena_stats(&drv, &segs, &segb);
      fprintf(stderr, "    device allocator: %.2fs in psg_malloc, %.2fs in psg_free (host side); %ld arena segments (%.1f GiB) fetched from the driver in %.2fs\n", g_alloc_seconds, g_free_seconds,
              (long)segs, segb / 1073741824.0, drv);
    }
    if (g_verbose && inner_levels) fprintf(stderr, "    batched merging: %ld levels, %.2fs in the calls; kernels (ms): leaves -> nodes %.0f, rank builds %.0f, start-rank search %.0f, stream %.0f, "
                                                   "rank-log histogram %.0f, gap->bitvector %.0f, merge + gt %.0f\n", (long)inner_levels, bt_total / 1e3, bt_prepare, bt_rank, bt_search, bt_stream, bt_hist, bt_bv, bt_merge);
    if (g_verbose) fprintf(stderr, "    seconds: leaf upload %.2f, start-rank search %.2f, rank build %.2f, stream pass %.2f, gap->bitvector %.2f, merge %.2f, BWT/gt from PSA %.2f\n",
                           tm_upload, tm_search, tm_rank, tm_stream, tm_bv, tm_merge, tm_finish);
  }
  gt_cur.release(); gt_new.release();
  if (opt.check_samples < 0) d_text.release();     // the device-side check compares suffixes of the text

  // ---- merge (psascan.hpp:117-125, merge.hpp:55-180): PSAs streamed from host memory, .sa5 slices to the file
  fprintf(stderr, "\nMerge partial suffix arrays:\n");
  prefill.finish();                                          // no zero may land behind a real byte
  if (g_verbose) fprintf(stderr, "    [%.2fs since start; %.1f of %.1f GiB of the output file were in the page cache ahead of time]\n", wclock() - start,
                         prefill.done.load() / 1073741824.0, 5.0 * n / 1073741824.0);
  double t0 = wclock();
  std::sort(hbs.begin(), hbs.end(), [](const DoneHalfBlock &a, const DoneHalfBlock &b) { return a.beg < b.beg; });
  std::vector<psg_hb_host_desc> desc(hbs.size());
  {
    int64_t files = 0, bytes = 0;
    for (const DoneHalfBlock &h : hbs) if (!h.mbv_file.empty()) { ++files; bytes += (int64_t)h.mbv_map_bytes; }
    if (files && g_verbose) fprintf(stderr, "    %ld merge bitvector files (%.1f MiB) mapped for the merge\n", (long)files, bytes / 1048576.0);
  }
  for (size_t h = 0; h < hbs.size(); ++h) {
    hbs[h].settle();
    hbs[h].map_back();
    desc[h] = psg_hb_host_desc{hbs[h].beg, hbs[h].size, hbs[h].lo(), hbs[h].hi(), h + 1 < hbs.size() ? hbs[h].mbv.as<uint32_t>() : nullptr,
                               hbs[h].psa_dev.as<uint32_t>(), hbs[h].psa_hi_dev.as<uint8_t>(),
                               h + 1 < hbs.size() ? hbs[h].mbv_w() : nullptr,
                               h + 1 < hbs.size() ? hbs[h].mbv_s() : nullptr};
  }
  // --check with the text in host memory: the same property check on the host (sum of all entries, sampled adjacent
  // pairs compared in the memory-mapped text), slice by slice as the output arrives
  struct HostCheck {
    const uint8_t *text; int64_t n, samples; bool on;
    uint64_t sum = 0, rng = 88172645463325252ull; int64_t bad = 0, undecided = 0;
    static int64_t entry(const uint8_t *p) { return (int64_t)p[0] | (int64_t)p[1] << 8 | (int64_t)p[2] << 16 | (int64_t)p[3] << 24 | (int64_t)p[4] << 32; }
    void slice(const uint8_t *sa5, int64_t cnt) {
      const int nt = 8;
      uint64_t part[nt] = {0};
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { uint64_t a = 0; for (int64_t k = cnt * t / nt; k < cnt * (t + 1) / nt; ++k) a += (uint64_t)entry(sa5 + 5 * k); part[t] = a; });
      for (auto &x : th) x.join();
      for (int t = 0; t < nt; ++t) sum += part[t];
      for (int64_t q = 0; q < samples && cnt > 1; ++q) {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        const int64_t k = (int64_t)(rng % (uint64_t)(cnt - 1)), a = entry(sa5 + 5 * k), b = entry(sa5 + 5 * (k + 1));
        if (a < 0 || a >= n || b < 0 || b >= n || a == b) { ++bad; continue; }
        int64_t l = 0;
        const int64_t cap = (int64_t)1 << 22;
        while (l < cap && a + l < n && b + l < n && text[a + l] == text[b + l]) ++l;
        if (l >= cap) { ++undecided; continue; }
        const bool less = a + l >= n ? true : (b + l >= n ? false : text[a + l] < text[b + l]);
        if (!less) ++bad;
      }
    }
  } hcheck{text.data(), n, opt.check_samples, opt.check_samples >= 0 && text_on_host};
  struct SinkCtx { FILE *out; bool ok; int64_t entries; HostCheck *hc; } sctx{out, true, 0, &hcheck};
  psg_sink_fn sink = [](void *c, const uint8_t *h_sa5, int64_t, int64_t cnt) -> int {
    SinkCtx *x = (SinkCtx *)c;
    x->entries += cnt;
    if (x->hc->on) x->hc->slice(h_sa5, cnt);
    if (x->out && fwrite(h_sa5, 1, (size_t)(5 * cnt), x->out) != (size_t)(5 * cnt)) { x->ok = false; return 1; }
    return 0;
  };
  psg_merge_check chk{d_text.as<uint8_t>(), n, opt.check_samples, 12345, 0, 0, 0};
  psg_merge_stream_stats ms;
  const int64_t slice = 64LL << 20;  // output entries per slice
  int mrc = psg_merge_stream(desc.data(), (int)desc.size(), std::min(slice, n), opt.check_samples >= 0 && !hcheck.on ? &chk : nullptr, sink, &sctx, &ms);
  if (hcheck.on) { chk.sum = hcheck.sum; chk.bad_pairs = hcheck.bad; }
  if (!sctx.ok) throw std::runtime_error("write failed on " + out_fn);
  if (mrc) throw std::runtime_error(std::string("psg_merge_stream: ") + psg_last_error());
  if (sctx.entries != n) throw std::runtime_error("merge produced a wrong number of entries");
  if (out) {   // the last buffer of the .sa5 reaches the disk here: a failure keeps the checkpoint and exits non-zero
    const bool flushed = fflush(out) == 0;          // (no fsync: 5 bytes per symbol forced to the disk would dominate the run; the reference does not either)
    const bool closed = fclose(out) == 0;
    out = nullptr;
    if (!flushed || !closed) throw std::runtime_error("write failed on " + out_fn);
  }
  log_phase("merge + write", t0, 5 * n);
  if (g_verbose) fprintf(stderr, "      slices=%ld merge kernels=%.1fms staging=%.1fms sink=%.1fms h2d=%.1fMiB d2h=%.1fMiB\n", (long)ms.slices, ms.kernel_ms, ms.stage_ms, ms.sink_ms, ms.h2d_bytes / 1048576.0, ms.d2h_bytes / 1048576.0);
  if (opt.check_samples >= 0) {
    const unsigned __int128 nn = (unsigned __int128)n * (unsigned __int128)(n - 1) / 2;
    const bool sum_ok = chk.sum == (uint64_t)nn;
    const std::string und = !hcheck.on && chk.undecided_pairs ? ", " + std::to_string((long)chk.undecided_pairs) + " pairs undecided within 16 Mi symbols" : "";
    fprintf(stderr, "    check: permutation sum %s, %ld of %ld sampled adjacent pairs out of order%s%s\n", sum_ok ? "ok" : "WRONG", (long)chk.bad_pairs, (long)(ms.slices * opt.check_samples),
            und.c_str(), hcheck.on ? (hcheck.undecided ? " (on the host; some pairs undecided within 4 Mi symbols)" : " (on the host)") : "");
    if (!sum_ok || chk.bad_pairs) throw std::runtime_error("output check failed");
  }
  checkpoint_clear();
  {
    int64_t in_use = 0, peak = 0, reserved = 0;
    psg_mem_stats(&in_use, &peak, &reserved);
    fprintf(stderr, "    device memory: peak in use %.2f GiB, reserved %.2f GiB\n", peak / 1073741824.0, reserved / 1073741824.0);
  }
  double total = wclock() - start;
  fprintf(stderr, "\n\nComputation finished. Summary:\n  elapsed time: %.2fs (%.4fs/MiB)\n  speed: %.2fMiB/s\n", total, total / (n / 1048576.0), (n / 1048576.0) / total);
  // Everything is written and closed.  Leave now: unwinding would hand 4-5 bytes per symbol of partial SAs back to the
  // C library page by page and then run the HIP runtime's teardown of the device arena -- ten seconds after a 32 GiB run.
  for (DoneHalfBlock &h : hbs) {
    if (!h.part_file.empty() && !h.keep_part) remove(h.part_file.c_str());
    if (!h.mbv_file.empty()) remove(h.mbv_file.c_str());
  }
  fflush(stdout); fflush(stderr);
  if (getenv("PSASCAN_NORMAL_EXIT")) return;      // under a profiler: its tool library writes its files at a regular exit
  _exit(EXIT_SUCCESS);
}

int main(int argc, char **argv) {
  program_name = argv[0];
  static struct option long_options[] = {{"help", no_argument, NULL, 'h'}, {"gap", required_argument, NULL, 'g'}, {"mem", required_argument, NULL, 'm'},
                                         {"output", required_argument, NULL, 'o'}, {"verbose", no_argument, NULL, 'v'},
                                         {"block-size", required_argument, NULL, 1000}, {"chains", required_argument, NULL, 1001},
                                         {"check", optional_argument, NULL, 1002}, {"discard-output", no_argument, NULL, 1003},
                                         {"spill-psa", no_argument, NULL, 1004}, {"leaf-size", required_argument, NULL, 1005},
                                         {"fanout", required_argument, NULL, 1006}, {"no-device-merge", no_argument, NULL, 1007},
                                         {"device-sort", no_argument, NULL, 1008}, {"text-on-host", no_argument, NULL, 1009},
                                         {"tail-chunk", required_argument, NULL, 1010}, {"checkpoint", required_argument, NULL, 1011},
                                         {"stop-after", required_argument, NULL, 1012}, {"hbm-limit", required_argument, NULL, 1013}, {NULL, 0, NULL, 0}};
  uint64_t ram_use = (uint64_t)3584 << 20;
  std::string output_filename, gap_filename;
  Options opt;
  int c;
  while ((c = getopt_long(argc, argv, "g:hm:o:v", long_options, NULL)) != -1) {
    switch (c) {
      case 'g': gap_filename = optarg; break;
      case 'h': usage(EXIT_FAILURE); break;             // the reference exits with failure here too (main.cpp:159-161)
      case 'm':
        if (!parse_number(optarg, &ram_use)) { fprintf(stderr, "Error: parsing RAM limit (%s) failed\n\n", optarg); usage(EXIT_FAILURE); }
        if (ram_use == 0) { fprintf(stderr, "Error: invalid RAM limit (%lu)\n\n", (unsigned long)ram_use); usage(EXIT_FAILURE); }
        break;
      case 'o': output_filename = optarg; break;
      case 'v': g_verbose = true; break;
      case 1000: { uint64_t v; if (!parse_number(optarg, &v) || v == 0) { fprintf(stderr, "Error: bad --block-size\n\n"); usage(EXIT_FAILURE); } opt.forced_block = (int64_t)v; break; }
      case 1001: opt.max_chains = atoll(optarg); break;
      case 1002: opt.check_samples = optarg ? atoll(optarg) : 4096; if (opt.check_samples < 0) opt.check_samples = 0; break;
      case 1003: opt.discard = true; break;
      case 1004: opt.spill_psa = true; break;
      case 1005: { uint64_t v; if (!parse_number(optarg, &v) || v == 0) { fprintf(stderr, "Error: bad --leaf-size\n\n"); usage(EXIT_FAILURE); } opt.leaf_size = (int64_t)v; break; }
      case 1006: opt.fanout = atoi(optarg); break;
      case 1007: opt.hierarchical = false; break;
      case 1008: opt.device_sort = true; break;
      case 1009: opt.text_on_host = true; break;
      case 1011: opt.checkpoint_dir = optarg; break;
      case 1012: opt.stop_after = atoll(optarg); break;
      case 1013: { uint64_t v; if (!parse_number(optarg, &v) || v < ((uint64_t)64 << 20)) { fprintf(stderr, "Error: bad --hbm-limit\n\n"); usage(EXIT_FAILURE); } opt.hbm_limit = (int64_t)v; break; }
      case 1010: { uint64_t v; if (!parse_number(optarg, &v) || v < 64) { fprintf(stderr, "Error: bad --tail-chunk\n\n"); usage(EXIT_FAILURE); } opt.tail_chunk = (int64_t)v; break; }
      default: usage(EXIT_FAILURE); break;
    }
  }
  if (optind >= argc) { fprintf(stderr, "Error: FILE not provided\n\n"); usage(EXIT_FAILURE); }
  std::string text_filename = argv[optind++];
  if (optind < argc) fprintf(stderr, "Warning: multiple input files provided. Only the first will be processed.\n");
  if (output_filename.empty()) output_filename = text_filename + ".sa5";
  if (gap_filename.empty()) gap_filename = output_filename;
  opt.gap_prefix = gap_filename;
  if (!file_exists(text_filename)) { fprintf(stderr, "Error: input file (%s) does not exist\n\n", text_filename.c_str()); usage(EXIT_FAILURE); }
  if (!opt.discard && file_exists(output_filename)) {   // main.cpp:216-238
    char *line = NULL; size_t buflen = 0; ssize_t len = 0;
    do {
      printf("Output file (%s) exists. Overwrite? [y/n]: ", output_filename.c_str());
      if ((len = getline(&line, &buflen, stdin)) == -1) { printf("\nError: failed to read answer\n\n"); fflush(stdout); usage(EXIT_FAILURE); }
    } while (len != 2 || (line[0] != 'y' && line[0] != 'n'));
    if (line[0] == 'n') { free(line); std::exit(EXIT_FAILURE); }
    free(line);
  }
  long max_threads = 0;
  if (const char *e = getenv("OMP_NUM_THREADS")) max_threads = atol(e);   // the reference's only thread knob (main.cpp:241)
  if (max_threads <= 0) max_threads = (long)std::max(1u, std::thread::hardware_concurrency());
  try {
    run(text_filename, output_filename, ram_use, max_threads, opt);
  } catch (const std::exception &ex) {
    fprintf(stderr, "Error: %s\n", ex.what());
    if (!opt.discard && file_exists(output_filename)) remove(output_filename.c_str());
    return EXIT_FAILURE;
  }
  // everything is written, closed and cleaned up: leave without the HIP runtime's static teardown, which hipFree's
  // the device arena segment by segment (~30 ms per GiB: ten seconds after a 32 GiB run)
  fflush(stdout); fflush(stderr);
  if (getenv("PSASCAN_NORMAL_EXIT")) return EXIT_SUCCESS;
  _exit(EXIT_SUCCESS);
}
