// legacy_ops.cpp -- Calculate3Dpoint / CudaComputeHref / g2o::CudaComputeH on top
// of the C-ABI (include/nid/nid_c.h, include/nid/nid_multi.h).  See include/nid/legacy_ops.h.
//
// The three operators keep the reference's signatures and its ownership rules (the caller owns every buffer
// and hands ALL of them over on every call, computeH.cu:373-502), but not its per-call cost (>= 10 mallocs,
// 52-367 MB of memsets, 11 MB of uploads): the frame-pair state stays resident on the device(s) between calls.
// What is resident is identified by CONTENT, not by pointer: a caller that frees and re-mallocs its buffers per
// pair (NID_pose_estimation.cpp:229-251, 385-392) usually gets the same addresses back.
//   * CudaComputeHref is called once per pair: it always uploads the reference (im0, points3d) afresh;
//   * CudaComputeH keeps, per big caller buffer (im0, points3d, im1, bs_ref), a key: address, length, a quick
//     fingerprint of 64 samples and the 64-bit hashes of its kSlices (128) contiguous slices; per small array
//     (bs_counter, Href: 1-2 KB) address, length and ONE hash, recomputed on every call.
//   * EVERY read of a caller buffer happens between the entry and the return of an operator (round 6; round 5's default
//     let worker threads read them between calls -- a use-after-free for the reference's main(), which frees its
//     buffers right after the last CudaComputeH, NID_pose_estimation.cpp:388-395).  Two modes
//     (nid_legacy_set_verify_mode):
//     ROTATING (default): a call checks address, length and the quick fingerprint of the four big buffers (about a
//     microsecond), hands a few of their 128 slices -- 1/32 of every buffer per call on average -- to the pool's worker
//     threads, runs the evaluation on the device, and JOINS the workers before it returns.  A slice that no longer hashes
//     to what its key holds means the buffer was rewritten in place: the call says so on stderr, counts it
//     (nid_legacy_stale_detections), uploads the buffer's current content and evaluates again before it returns -- an
//     undeclared change in place is followed within 43 calls (the reference's LM makes 40-60 per pair) and the caller
//     is told.  NID_LEGACY_VERIFY_EVERY_CALL=1 (mode EVERY_CALL) checks every slice on every call: the reference's
//     guarantee exactly (it re-uploads everything on every call, computeH.cu:420-429).
//     TRUSTED (nid_legacy_set_trust_buffers(1) or NID_LEGACY_TRUST_BUFFERS=1): the cheap check only; the full hash is
//     recomputed -- and decides whether the buffer is uploaded -- when the cheap part changed, on every kRehashEvery-th
//     call of a pair, and after nid_legacy_invalidate().  No thread is woken by this mode's calls once the pair is set up.
//     In both modes a new frame pair (new buffers, or CudaComputeHref) is noticed at once, and
//     nid_legacy_invalidate(parts) makes the next call recompute the named parts' hashes in full.
//     NID_LEGACY_ALWAYS_UPLOAD=1 uploads everything on every call.
#include "nid/legacy_ops.h"

#include <atomic>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "nid/nid_c.h"
#include "nid/nid_multi.h"

namespace {

constexpr int kSlices = NID_LEGACY_SLICES;   // a big buffer's content key: one hash per slice; the default mode checks one slice per call

struct LegacyState {
  nid_multi *m = nullptr;
  int rows = 0, cols = 0, cell = 0, bins = 0;
  double intr[4] = {0, 0, 0, 0};
  // keys of what is resident on the device(s)
  struct Key {
    const void *addr = nullptr;
    size_t n = 0;             // elements
    uint64_t quick = 0, full = 0;
    uint64_t slice[kSlices] = {};  // big buffers: the slices' hashes (`full` is their combination)
    bool valid = false;
    bool full_known = true;   // false: the hashes were never taken -- a full check then counts as "changed"
  };
  Key k_im0, k_points, k_im1, k_bs_ref, k_counter, k_href;
  bool have_ref = false, have_target = false, have_href = false;
  unsigned long calls = 0;      // CudaComputeH calls on this frame pair (CudaComputeHref starts a new count)
  unsigned long rot = 0;        // the rotating verification's cursor: the next slice to check
  unsigned force_full = 0;      // nid_legacy_invalidate: parts whose full hash the next call recomputes
  unsigned fresh = 0;           // parts hashed in full or uploaded by the CURRENT call: nothing to verify behind them
};
constexpr unsigned long kRehashEvery = 128;

LegacyState g_state;
std::vector<int32_t> g_devices = {0};
int g_rank = 0, g_world = 1;
bool g_have_id = false;
uint8_t g_id[NID_RCCL_ID_BYTES];
int g_jac_bound = NID_JACBOUND_CPU;
int g_math_mode = NID_MATH_FAST;
int g_reduce_rccl = 0;
// Blocking single-pose calls are latency bound: with up to 256 cells per shard (one workgroup per CU) 512-thread
// workgroups for the Jacobian launch, 256 beyond; cost-only launches shaped per launch (profiles/r02_launch_cost_*.txt).
int g_jac_threads = -1, g_cost_threads = 0;  // nid_legacy_set_launch_shape; -1 = by shard size
int g_resident = -1;                         // nid_legacy_set_resident; -1 = the NID_LEGACY_RESIDENT environment variable
bool resident_wanted() {
  if (g_resident >= 0) return g_resident != 0;
  const char *e = getenv("NID_LEGACY_RESIDENT");
  return e && e[0] == '1';
}

int jac_threads_for(int cells, int shards) {
  if (g_jac_threads >= 0) return g_jac_threads;
  return (cells + shards - 1) / shards <= 256 ? 512 : 256;
}
nid_comm *g_comm = nullptr;  // lives across nid_legacy_reset(): one communicator per process, however many pairs / levels
long g_uploads = 0;

// NID_LEGACY_TRACE=1: microseconds of every step of the per-pair setup on stderr (tools/pair_setup.py collects them
// into profiles/r06_pair_setup.txt)
struct StepTrace {
  bool on;
  const char *what;
  std::chrono::steady_clock::time_point t;
  explicit StepTrace(const char *w) : what(w) {
    static const bool env = getenv("NID_LEGACY_TRACE") != nullptr;
    on = env;
    if (on) t = std::chrono::steady_clock::now();
  }
  void step(const char *name) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[nid trace] %s: %s %.1f us\n", what, name, std::chrono::duration<double, std::micro>(n - t).count());
    t = n;
  }
};

int g_verify_mode = -1;    // nid_legacy_set_verify_mode; -1 = the environment (NID_LEGACY_TRUST_BUFFERS=1, NID_LEGACY_VERIFY_EVERY_CALL=1) or the default
int g_verify_slices = -1;  // nid_legacy_set_verify_slices; -1 = by mode (ROTATING 1, EVERY_CALL kSlices) or NID_LEGACY_VERIFY_SLICES
int verify_mode() {
  if (g_verify_mode >= 0) return g_verify_mode;
  static const int env = [] {
    const char *t = getenv("NID_LEGACY_TRUST_BUFFERS"), *e = getenv("NID_LEGACY_VERIFY_EVERY_CALL");
    if (t && t[0] == '1') return (int)NID_LEGACY_VERIFY_TRUSTED;
    if (e && e[0] == '1') return (int)NID_LEGACY_VERIFY_EVERY_CALL;
    return (int)NID_LEGACY_VERIFY_ROTATING;
  }();
  return env;
}
bool trust_buffers() { return verify_mode() == NID_LEGACY_VERIFY_TRUSTED; }
// slices of every big buffer a CudaComputeH call verifies (0: trusted buffers)
int slices_per_call() {
  const int mode = verify_mode();
  if (mode == NID_LEGACY_VERIFY_TRUSTED) return 0;
  if (mode == NID_LEGACY_VERIFY_EVERY_CALL) return kSlices;
  if (g_verify_slices > 0) return std::min(g_verify_slices, kSlices);
  static const int env = [] { const char *e = getenv("NID_LEGACY_VERIFY_SLICES"); return e ? std::max(1, std::min(kSlices, atoi(e))) : NID_LEGACY_SLICES_PER_CALL; }();
  return env;
}
long g_stale_detections = 0;

bool always_upload() {
  static const bool v = getenv("NID_LEGACY_ALWAYS_UPLOAD") != nullptr;
  return v;
}

// 64 samples spread over the array + its length, FNV-1a over their bytes (never 0)
template <typename T>
uint64_t fingerprint(const T *a, size_t n) {
  if (!a) return 1;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void *p, size_t bytes) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 1099511628211ull; }
  };
  mix(&n, sizeof(n));
  const size_t samples = n < 64 ? n : 64;
  for (size_t k = 0; k < samples; k++) {
    const size_t i = samples > 1 ? (size_t)((unsigned __int128)k * (n - 1) / (samples - 1)) : 0;
    mix(&a[i], sizeof(T));
  }
  return h ? h : 2;
}

// ---- content hashes -------------------------------------------------------------------------------------------
// A frame pair's big buffers (points3d 7.4 MB, bs_value 9.8 MB at 640x480) are hashed once per pair in full and one
// slice per call after that; one core does ~20 GB/s.  The slices go to a small pool of worker threads (created at the
// first use, parked on a condition variable in between) beside the caller.  The pool's size: NID_LEGACY_HASH_THREADS
// workers (default 3; 0: the caller hashes alone, no thread is created).  A job lives INSIDE one operator call: start()
// takes the pool, finish() -- always before the operator returns -- gives it back; no worker touches a caller buffer
// outside start() .. finish().
class HashPool {
 public:
  static HashPool &get() { static HashPool *p = new HashPool;  return *p; }  // (never destroyed: its parked workers end with the process)
  // start(n, job): job(0..n-1) is handed to the workers -- each takes the next part that nobody has taken -- and the
  // call returns; finish(): the caller takes what is left and waits for the rest.  One job at a time, started and
  // finished by the same thread; the job must stay valid until finish().
  // Parts are claimed with one atomic increment on the JOB's own counter (a worker that comes late holds the old job
  // object and finds nothing left in it: it can never touch the next job's parts); the mutex is for parking only.
  void start(int nparts, std::function<void(int)> fn) {
    call_.lock();
    auto j = std::make_shared<Job>();
    j->fn = std::move(fn);
    j->nparts = nparts;
    j->remaining.store(nparts, std::memory_order_relaxed);
    mine_ = j;
    std::atomic_store(&current_, j);
    generation_.fetch_add(1);  // (sequentially consistent with parked_: either this thread sees a parking worker, or the worker sees the new generation)
    if (parked_.load() > 0 && has_workers()) {  // (a fork()ed child has no workers: finish() does all parts)
      std::lock_guard<std::mutex> g(m_);
      wake_.notify_all();
    }
  }
  void finish() {
    Job &j = *mine_;
    help(j);
    while (j.remaining.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();  // (parts in other threads' hands: microseconds)
    std::atomic_store(&current_, std::shared_ptr<Job>());
    mine_.reset();
    call_.unlock();
  }
  void run(int nparts, std::function<void(int)> job) { start(nparts, std::move(job)); finish(); }
  bool has_workers() const { return !workers_.empty() && getpid() == owner_; }

 private:
  struct Job {
    std::function<void(int)> fn;
    int nparts = 0;
    std::atomic<int> next{0}, remaining{0};
  };
  HashPool() : owner_(getpid()) {
    int n = 3;
    if (const char *e = getenv("NID_LEGACY_HASH_THREADS")) n = std::max(0, std::min(15, atoi(e)));
    if (const char *e = getenv("NID_LEGACY_HASH_SPIN_US")) spin_us_ = std::max(0, std::min(100000, atoi(e)));
    for (int w = 0; w < n; w++) workers_.emplace_back([this] { loop(); });
    // fork(): threads do not survive it, locks do -- a child forked while a worker held m_ would wait for it for ever.
    // The prepare handler takes call_ (free whenever no operator is inside start() .. finish(): a job never outlives
    // the call that started it, so a fork() between two calls finds it free; a fork() from another thread waits for
    // the running call's finish()) and then m_, so the child inherits both free, with no job in flight (and finds
    // itself without workers by its pid).
    pthread_atfork([] { HashPool &p = get(); p.call_.lock(); p.m_.lock(); },
                   [] { HashPool &p = get(); p.m_.unlock(); p.call_.unlock(); },
                   [] { HashPool &p = get(); p.m_.unlock(); p.call_.unlock(); });
  }
  static void help(Job &j) {  // take parts until none is left
    for (;;) {
      const int part = j.next.fetch_add(1, std::memory_order_acq_rel);
      if (part >= j.nparts) return;
      j.fn(part);
      j.remaining.fetch_sub(1, std::memory_order_release);
    }
  }
  // A worker that has just served a job keeps POLLING for the next one for spin_us_ microseconds before it parks on the
  // condition variable: the operators are called back to back by the LM (one call per 30-60 us), and a parked thread
  // takes 10-20 us to come back -- as long as the whole evaluation it is supposed to hide behind
  // (profiles/r06_legacy_call_cost.txt).  NID_LEGACY_HASH_SPIN_US (default 150; 0: park at once).
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      bool got = false;
      for (unsigned spins = 0; spin_us_ > 0; spins++) {
        if (generation_.load(std::memory_order_acquire) != seen) { got = true; break; }
        __builtin_ia32_pause();
        if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
      }
      if (!got) {
        std::unique_lock<std::mutex> g(m_);
        parked_.fetch_add(1);
        wake_.wait(g, [&] { return generation_.load() != seen; });
        parked_.fetch_sub(1);
      }
      seen = generation_.load(std::memory_order_acquire);
      if (std::shared_ptr<Job> j = std::atomic_load(&current_)) help(*j);
    }
  }
  std::mutex call_, m_;
  std::condition_variable wake_;
  std::shared_ptr<Job> current_, mine_;
  std::vector<std::thread> workers_;
  std::atomic<unsigned long> generation_{0};
  std::atomic<int> parked_{0};
  int spin_us_ = 150;
  const pid_t owner_;
};

// one contiguous run of bytes, 64 bits: eight interleaved multiply-xor lanes over the 8-byte words (eight independent
// dependency chains keep the multiplier busy: memory-bound, ~0.05 ms/MB on one core)
uint64_t hash_run(const unsigned char *b, size_t bytes, uint64_t salt) {
  const size_t words = bytes / 8;
  constexpr int L = 8;
  uint64_t h[L] = {0x9E3779B97F4A7C15ull ^ salt, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull,
                   0x85EBCA77C2B2AE63ull, 0xD6E8FEB86659FD93ull, 0xA0761D6478BD642Full, 0xE7037ED1A0B428DBull};
  size_t i = 0;
  for (; i + L <= words; i += L) {
    uint64_t w[L];
    std::memcpy(w, b + 8 * i, 8 * L);
    for (int k = 0; k < L; k++) { h[k] = (h[k] ^ w[k]) * 0x9FB21C651E98DF25ull; h[k] ^= h[k] >> 29; }
  }
  for (; i < words; i++) { uint64_t w; std::memcpy(&w, b + 8 * i, 8); h[i % L] = (h[i % L] ^ w) * 0x9FB21C651E98DF25ull; h[i % L] ^= h[i % L] >> 29; }
  for (size_t t = 8 * words; t < bytes; t++) h[0] = (h[0] ^ b[t]) * 0x100000001B3ull;
  uint64_t r = h[0];
  for (int k = 1; k < L; k++) r = (r ^ h[k]) * 0xFF51AFD7ED558CCDull + k;
  r ^= r >> 32;
  return r;
}

// slice s of a buffer of `bytes` bytes: [lo, hi), split at multiples of 64 bytes (whole bs_value rows); the last slice
// takes the remainder (a buffer below 1 KB is its last slice alone)
inline void slice_range(size_t bytes, int s, size_t *lo, size_t *hi) {
  const size_t part = (bytes / kSlices) & ~(size_t)63;
  *lo = part * (size_t)s;
  *hi = s == kSlices - 1 ? bytes : part * (size_t)(s + 1);
}
inline uint64_t combine_slices(const uint64_t *h) {
  uint64_t r = h[0];
  for (int p = 1; p < kSlices; p++) r = (r ^ h[p]) * 0xFF51AFD7ED558CCDull + p;
  r ^= r >> 32;
  return r ? r : 2;
}

// Slices [first, first + count) (mod kSlices) of up to four buffers as ONE pooled job: begin() hands the parts to the
// workers and returns, end() takes what is left, waits, and leaves every requested slice's hash in out[buffer][slice].
// Both are called inside one operator call.
class SliceHasher {
 public:
  struct Req { const void *data = nullptr; size_t bytes = 0; };
  void begin(const Req *reqs, int n, int first, int count) {
    parts_.clear();
    for (int r = 0; r < n && r < 4; r++) {
      const unsigned char *b = static_cast<const unsigned char *>(reqs[r].data);
      if (!b) continue;
      for (int c = 0; c < count && c < kSlices; c++) {
        const int s = (first + c) % kSlices;
        size_t lo, hi;
        slice_range(reqs[r].bytes, s, &lo, &hi);
        parts_.push_back({b + lo, hi - lo, (uint64_t)reqs[r].bytes + (uint64_t)s, &out[r][s]});
      }
    }
    running_ = true;
    HashPool::get().start((int)parts_.size(), [this](int k) {
      const Part &pt = parts_[(size_t)k];
      *pt.out = hash_run(pt.b, pt.bytes, pt.salt);
    });
  }
  bool running() const { return running_; }
  void end() {
    if (!running_) return;
    HashPool::get().finish();
    running_ = false;
  }
  uint64_t out[4][kSlices] = {};

 private:
  struct Part { const unsigned char *b; size_t bytes; uint64_t salt; uint64_t *out; };
  std::vector<Part> parts_;
  bool running_ = false;
};

// the whole content of one big buffer into its key (all slices, on the pool); begin / end so that the caller can do
// something else in between (upload_reference hashes the points while they cross PCIe)
struct FullHash {
  SliceHasher sh;
  void begin(const void *data, size_t bytes) { SliceHasher::Req r; r.data = data; r.bytes = bytes; sh.begin(&r, 1, 0, kSlices); }
  void end(LegacyState::Key *k) {
    sh.end();
    std::memcpy(k->slice, sh.out[0], sizeof(k->slice));
    k->full = combine_slices(k->slice);
    k->full_known = true;
  }
};
template <typename T>
void full_hash_into(LegacyState::Key *k, const T *a, size_t n) {
  if (!a) { std::memset(k->slice, 0, sizeof(k->slice)); k->full = 1; k->full_known = true; return; }
  FullHash f;
  f.begin(a, n * sizeof(T));
  f.end(k);
}
// the small per-cell arrays: one run, on the caller
template <typename T>
uint64_t small_hash(const T *a, size_t n) {
  if (!a) return 1;
  const uint64_t r = hash_run(reinterpret_cast<const unsigned char *>(a), n * sizeof(T), n * sizeof(T));
  return r ? r : 2;
}

// Does the caller's big buffer still hold what is resident?  Updates the key; `force`: recompute the hashes even if
// address, length and the quick fingerprint are unchanged.  *hashed: the key's hashes were taken by this call.
template <typename T>
bool same_content(LegacyState::Key &k, const T *a, size_t n, bool force, bool *hashed) {
  const uint64_t q = fingerprint(a, n);
  const bool cheap_same = k.valid && k.addr == (const void *)a && k.n == n && k.quick == q;
  if (cheap_same && !force) return true;
  LegacyState::Key now;
  full_hash_into(&now, a, n);
  if (hashed) *hashed = true;
  // a key whose hashes were never taken (trusted mode: CudaComputeHref's bs_value) is completed by its first full
  // check; in the default mode a key without them counts as "changed"
  bool same = k.valid && k.n == n && k.full_known && k.full == now.full;
  if (cheap_same && !k.full_known && trust_buffers()) same = true;
  now.addr = a; now.n = n; now.quick = q; now.valid = true;
  k = now;
  return same;
}
template <typename T>
bool same_small(LegacyState::Key &k, const T *a, size_t n) {
  const uint64_t f = small_hash(a, n);
  const bool same = k.valid && k.addr == (const void *)a && k.n == n && k.full == f;
  k.addr = a; k.n = n; k.quick = 0; k.full = f; k.valid = true; k.full_known = true;
  return same;
}
template <typename T>
void remember_small(LegacyState::Key &k, const T *a, size_t n) { (void)same_small(k, a, n); }
template <typename T>
void remember(LegacyState::Key &k, const T *a, size_t n) {
  full_hash_into(&k, a, n);
  k.addr = a; k.n = n; k.quick = fingerprint(a, n); k.valid = true;
}

bool to_u8(const double *im, size_t n, std::vector<uint8_t> *out);
// An f64 image carrying u8 values (NID_pose_estimation.cpp:245-251), keyed like every big buffer on the caller's f64
// bytes.  `have`: something is resident to compare with.  On return *same says whether the resident image can stay; u8
// holds the converted image (the conversion checks every value: an integer in [0, 255] or the call fails) when it cannot.
int check_image(LegacyState::Key &k, const double *im, size_t n, bool force, bool have, std::vector<uint8_t> *u8, bool *same,
                bool *hashed) {
  u8->clear();
  *same = same_content(k, im, n, force || !have, hashed) && have;
  if (*same) return NID_OK;
  if (!to_u8(im, n, u8)) { k.valid = false; return NID_ERR_UNSUPPORTED; }
  return NID_OK;
}

void report(const char *where, int rc, nid_multi *m) {
  // the reference prints CUDA errors and carries on (computeH.cu:454-473)
  std::fprintf(stderr, "[nid legacy] %s failed: %s (%d) %s\n", where, nid_status_string(rc), rc,
               m ? nid_multi_last_error(m) : "");
}

// Contexts of OTHER geometries than the current one, most recently used last (round 6): a coarse-to-fine schedule walks
// through three or four geometries per frame pair (host/nid_pyramid.cpp), and creating a context is ~6 ms of allocations
// (18 of the pyramid's 33 ms per pair went there).  At most kParkedMax of them are kept; a parked context holds no
// resident kernel and its caller-buffer keys are dropped.
constexpr size_t kParkedMax = 4;
std::vector<LegacyState> g_parked;

void apply_options(nid_multi *m, int cell);

nid_multi *get_multi(int rows, int cols, int cell, int bins, int deg, const double *intr) {
  LegacyState &S = g_state;
  auto matches = [&](const LegacyState &T) {
    return T.m && T.rows == rows && T.cols == cols && T.cell == cell && T.bins == bins &&
           T.intr[0] == intr[0] && T.intr[1] == intr[1] && T.intr[2] == intr[2] && T.intr[3] == intr[3];
  };
  if (matches(S)) return S.m;
  if (S.m) {  // park the current context
    (void)nid_multi_resident_pause(S.m);
    LegacyState keep;
    keep.m = S.m; keep.rows = S.rows; keep.cols = S.cols; keep.cell = S.cell; keep.bins = S.bins;
    std::memcpy(keep.intr, S.intr, sizeof(keep.intr));
    g_parked.push_back(keep);
    if (g_parked.size() > kParkedMax) { nid_multi_destroy(g_parked.front().m); g_parked.erase(g_parked.begin()); }
  }
  S = LegacyState();
  for (size_t k = 0; k < g_parked.size(); k++)
    if (matches(g_parked[k])) {
      S = g_parked[k];  // (no keys, nothing "resident": the next call hands its pair over afresh)
      g_parked.erase(g_parked.begin() + (long)k);
      apply_options(S.m, cell);
      return S.m;
    }
  nid_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.rows = rows; cfg.cols = cols; cfg.cell_num = cell; cfg.bin_num = bins; cfg.bs_degree = deg;
  cfg.fx = intr[0]; cfg.fy = intr[1]; cfg.cx = intr[2]; cfg.cy = intr[3];
  nid_multi *m = nullptr;
  int rc = g_world > 1 ? nid_multi_create_rank(&cfg, g_devices[0], g_rank, g_world, &m)
                       : nid_multi_create(&cfg, g_devices.data(), (int32_t)g_devices.size(), &m);
  if (rc != NID_OK) { report("nid_multi_create", rc, nullptr); return nullptr; }
  if (g_world > 1 || g_reduce_rccl) {
    if (!g_comm) {
      if (g_world > 1 && !g_have_id) { report("nid_legacy_set_rank: no RCCL id", NID_ERR_STATE, m); nid_multi_destroy(m); return nullptr; }
      rc = g_world > 1 ? nid_comm_create_rank(g_id, g_rank, g_world, g_devices[0], &g_comm)
                       : nid_comm_create_local(g_devices.data(), (int32_t)g_devices.size(), &g_comm);
      if (rc != NID_OK) { report("RCCL communicator", rc, m); nid_multi_destroy(m); return nullptr; }
    }
    rc = nid_multi_attach_comm(m, g_comm);
    if (rc != NID_OK) { report("nid_multi_attach_comm", rc, m); nid_multi_destroy(m); return nullptr; }
  }
  apply_options(m, cell);
  S.m = m; S.rows = rows; S.cols = cols; S.cell = cell; S.bins = bins;
  std::memcpy(S.intr, intr, sizeof(S.intr));
  return m;
}

void apply_options(nid_multi *m, int cell) {
  // the operator signatures carry a 4x4 matrix: computeH.cu:152-154 semantics for the transform
  nid_multi_set_options(m, g_jac_bound, NID_XFORM_MATRIX);
  nid_multi_set_math_mode(m, g_math_mode);
  nid_multi_set_href_nan_markers(m, 1);  // CudaComputeHref's bs_value with the CUDA operator's NaN rows, written on the device
  // the operators are blocking, one pose (or one LM rejection chain) at a time: latency matters, not the
  // pipelined throughput the 128-thread default is tuned for (nid_set_launch_shape, tools/latency_sweep.py)
  nid_multi_set_launch_shape(m, jac_threads_for(cell * cell, g_world > 1 ? g_world : (int)g_devices.size()), g_cost_threads);
  (void)nid_multi_set_resident(m, resident_wanted() ? 1 : 0);  // (unsupported platform: the launched form)
}

bool to_u8(const double *im, size_t n, std::vector<uint8_t> *out) {
  out->resize(n);
  return nid_set_reference_image_f64(im, (int64_t)n, out->data()) == NID_OK;
}

// (`im`: im0 converted already, with its key up to date -- or empty: converted and keyed here)
int upload_reference(LegacyState &S, const double *im0, const double *points3d, std::vector<uint8_t> *im, bool points_keyed) {
  const size_t N = (size_t)S.rows * S.cols;
  std::vector<uint8_t> local;
  if (!im || im->empty()) {
    bool dummy;
    int rc = check_image(S.k_im0, im0, N, true, false, &local, &dummy, nullptr);
    if (rc != NID_OK) return rc;
    im = &local;
  }
  // the points' content key is taken WHILE they cross PCIe (7.4 MB at 640x480: 0.2 ms on the pool's threads, hidden)
  FullHash fh;
  if (!points_keyed) fh.begin(points3d, 3 * N * sizeof(double));
  int rc = nid_multi_set_reference_points(S.m, points3d, im->data());
  if (!points_keyed) {
    LegacyState::Key k;
    fh.end(&k);  // (on every path: the pool is held until then)
    k.addr = points3d; k.n = 3 * N; k.quick = fingerprint(points3d, 3 * N); k.valid = true;
    if (rc == NID_OK) S.k_points = k;
  }
  if (rc != NID_OK) return rc;
  S.have_ref = true; S.have_href = false;
  S.fresh |= NID_LEGACY_REFERENCE;
  g_uploads++;
  return NID_OK;
}

int ensure_reference(LegacyState &S, const double *im0, const double *points3d, bool force) {
  const size_t N = (size_t)S.rows * S.cols;
  std::vector<uint8_t> im;
  bool points_keyed = false;
  if (S.have_ref && !always_upload()) {
    bool a = false, hashed_a = false, hashed_b = false;
    int rc = check_image(S.k_im0, im0, N, force, true, &im, &a, &hashed_a);
    if (rc != NID_OK) return rc;
    const bool b = same_content(S.k_points, points3d, 3 * N, force, &hashed_b);
    points_keyed = true;
    if (hashed_a && hashed_b) S.fresh |= NID_LEGACY_REFERENCE;
    if (a && b) return NID_OK;
  }
  return upload_reference(S, im0, points3d, &im, points_keyed);
}

}  // namespace

void Calculate3Dpoint(double *depth, double *pose_c2w, double *points_3d, double *camera_intrincis, int rows,
                      int cols) {
  StepTrace tr("Calculate3Dpoint");
  int rc = nid_backproject(depth, pose_c2w, camera_intrincis[0], camera_intrincis[1], camera_intrincis[2],
                           camera_intrincis[3], rows, cols, g_devices[0], points_3d);
  if (rc != NID_OK) report("Calculate3Dpoint", rc, nullptr);
  tr.step("upload depth, kernel, points back");
}

void CudaComputeHref(double *im0, double *points3d, double *pose, double *camera_intrincis, int bin_num,
                     int bs_degree, int cell_num, int rows, int cols, double *bs_value, int *bs_index,
                     int *bs_counter, double *Href) {
  StepTrace tr("CudaComputeHref");
  nid_multi *m = get_multi(rows, cols, cell_num, bin_num, bs_degree, camera_intrincis);
  if (!m) return;
  tr.step("context (created or reused)");
  LegacyState &S = g_state;
  int rc = upload_reference(S, im0, points3d, nullptr, false);  // once per frame pair: always fresh
  if (rc != NID_OK) { report("CudaComputeHref(reference upload)", rc, m); return; }
  tr.step("reference: im0 -> u8, content keys, upload, tile kernel");
  S.have_target = false;                        // a new pair: the next CudaComputeH re-checks its target
  const size_t N = (size_t)rows * cols;
  const int ncell = cell_num * cell_num;
  static_assert(sizeof(int) == sizeof(int32_t), "bs_index is handed through as int32_t");
  std::vector<int32_t> cnt(ncell);
  std::vector<double> href(ncell);
  // (every pixel of a cell is written by the shard that owns the cell; what belongs to no cell of this process -- trailing
  // rows / columns of a size the cell count does not divide, other ranks' cells -- must read as zero)
  const bool every_pixel_ours = g_world == 1 && rows % cell_num == 0 && cols % cell_num == 0;
  if (bs_value && !every_pixel_ours) std::fill(bs_value, bs_value + 4 * N, (double)NAN);  // (no sample there: the NaN rows of the convention below)
  if (bs_index && !every_pixel_ours) std::memset(bs_index, 0, N * sizeof(int));
  tr.step("output buffers cleared");
  rc = nid_multi_compute_href_matrix(m, pose, cnt.data(), href.data(), bs_value, reinterpret_cast<int32_t *>(bs_index));
  if (rc != NID_OK) { report("CudaComputeHref", rc, m); return; }
  tr.step("k_href + bs_value / bs_index back to the caller");
  for (int c = 0; c < ncell; c++) {
    bs_counter[c] = cnt[c];
    // CudaComputeHref.cu:205-220: NaN when inactive, otherwise subtract onto the caller's value
    Href[c] = std::isnan(href[c]) ? NAN : Href[c] + href[c];
  }
  // the device already holds these weights (CPU-edge convention: 0 instead of NaN)
  S.have_href = true;
  S.calls = 0;  // (the rotation / the trusted mode's periodic full check count the calls of THIS pair)
  if (bs_value && !trust_buffers()) {
    // default mode: the CudaComputeH calls compare the caller's bs_ref, slice by slice, with these hashes (9.8 MB at
    // 640x480, on the pool).  (The legacy NaN rows, CudaComputeHref.cu:82-87, 126-130, were written on the device:
    // nid_set_href_nan_markers.)
    remember(S.k_bs_ref, bs_value, 4 * N);
  } else if (bs_value) {
    // trusted buffers: address, length and the sampled fingerprint now and NO full hash (0.25 ms even on the pool); the
    // first full check that reaches the key -- the kRehashEvery-th call of the pair, nid_legacy_invalidate -- takes it.
    S.k_bs_ref = LegacyState::Key();
    S.k_bs_ref.addr = bs_value; S.k_bs_ref.n = 4 * N; S.k_bs_ref.full = 0; S.k_bs_ref.full_known = false;
    S.k_bs_ref.quick = fingerprint(bs_value, 4 * N); S.k_bs_ref.valid = true;
  } else {
    remember(S.k_bs_ref, bs_value, 0);
  }
  tr.step("content key of bs_value (default: slice hashes; trusted: fingerprint)");
  remember_small(S.k_counter, bs_counter, (size_t)ncell);
  remember_small(S.k_href, Href, (size_t)ncell);
  tr.step("content keys of the per-cell outputs");
}

namespace {
struct CallArgs {
  double *im0, *im1, *points3d;
  int *bs_counter;
  double *bs_ref;
  int *bs_index_ref;
  double *camera_intrincis;
  int bin_num, bs_degree, cell_num, rows, cols;
  double *Href;
};
nid_multi *ensure_state(const CallArgs &a);
void verify_begin(const CallArgs &a, bool long_call);
unsigned verify_end(const CallArgs &a);
nid_multi *follow_change(const CallArgs &a, unsigned stale);
}

namespace g2o {

void CudaComputeH(bool calculate_der, double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref,
                  int *bs_index_ref, double *pose, double *camera_intrincis, int bin_num, int bs_degree,
                  int cell_num, int rows, int cols, double *Href, double *pro_target, double *pro_joint,
                  double *Htarget, double *Hjoint, double *der) {
  (void)pro_target; (void)pro_joint;  // accepted, never read or written (computeH.cu:373-502)
  const CallArgs a = {im0, im1, points3d, bs_counter, bs_ref, bs_index_ref, camera_intrincis, bin_num, bs_degree, cell_num, rows, cols, Href};
  nid_multi *m = ensure_state(a);
  if (!m) return;
  const int ncell = cell_num * cell_num;
  std::vector<double> ht(ncell), hj(ncell);
  // this call's slices of the caller's buffers are hashed by the pool's workers WHILE the device evaluates ...
  verify_begin(a, calculate_der);
  int rc = nid_multi_evaluate_matrix(m, pose, calculate_der ? 1 : 0, ht.data(), hj.data(), nullptr, calculate_der ? der : nullptr);
  // ... and the workers are joined here, on every path: no read of a caller buffer outlives the call
  const unsigned stale = verify_end(a);
  if (rc != NID_OK) { report("CudaComputeH", rc, m); return; }
  if (stale) {
    // a buffer rewritten in place: what was just evaluated may be its old content -- upload what differs, evaluate again
    m = follow_change(a, stale);
    if (!m) return;
    rc = nid_multi_evaluate_matrix(m, pose, calculate_der ? 1 : 0, ht.data(), hj.data(), nullptr, calculate_der ? der : nullptr);
    if (rc != NID_OK) { report("CudaComputeH", rc, m); return; }
  }
  for (int c = 0; c < ncell; c++) {
    // CalculateHKernel: NaN for bs_counter < 300, else `-=` onto the caller's (zeroed) value
    Htarget[c] = std::isnan(ht[c]) ? NAN : Htarget[c] + ht[c];
    Hjoint[c] = std::isnan(hj[c]) ? NAN : Hjoint[c] + hj[c];
  }
}

}  // namespace g2o

namespace {

// Everything CudaComputeH does before its kernels: the frame-pair state on the device(s), keyed on content.
nid_multi *ensure_state(const CallArgs &a) {
  nid_multi *m = get_multi(a.rows, a.cols, a.cell_num, a.bin_num, a.bs_degree, a.camera_intrincis);
  if (!m) return nullptr;
  LegacyState &S = g_state;
  const size_t N = (size_t)a.rows * a.cols;
  const int ncell = a.cell_num * a.cell_num;
  const bool first_of_pair = !S.have_target;
  StepTrace tr("CudaComputeH state check");
  tr.on = tr.on && first_of_pair;  // (every later call of a pair takes about a microsecond: not traced)
  // Full hashes: whenever a key's cheap part changed (same_content), after an invalidate, and with trusted buffers on
  // every kRehashEvery-th call of the pair.  The default mode's per-call slices: verify_begin / verify_end.
  S.calls++;
  S.fresh = 0;
  const bool periodic = (trust_buffers() || always_upload()) && S.calls % kRehashEvery == 0;
  const unsigned force = S.force_full;
  S.force_full = 0;
  int rc = ensure_reference(S, a.im0, a.points3d, periodic || (force & NID_LEGACY_REFERENCE));
  if (rc != NID_OK) { report("CudaComputeH(reference upload)", rc, m); return nullptr; }
  {
    std::vector<uint8_t> im;
    bool same = false, hashed = false;
    const bool have = S.have_target && !always_upload();
    rc = check_image(S.k_im1, a.im1, N, periodic || (force & NID_LEGACY_TARGET), have, &im, &same, &hashed);
    if (rc != NID_OK) { report("CudaComputeH(im1 is not u8-valued)", rc, m); return nullptr; }
    if (hashed) S.fresh |= NID_LEGACY_TARGET;
    if (!same) {
      rc = nid_multi_set_target_u8(m, im.data());
      if (rc != NID_OK) { report("CudaComputeH(target upload)", rc, m); S.k_im1.valid = false; return nullptr; }
      S.have_target = true;
      g_uploads++;
    }
  }
  tr.step("reference keys checked, target -> u8 + upload + margins");
  // Href is only READ by the reference with calculate_der (computeH.cu:428-429); the kernels also need it to know
  // which cells are active.  A NULL Href gives zeros (cost-only outputs do not depend on it) and the first call
  // that brings one replaces them: Href is part of the key.  The per-cell arrays are small: full hashes every call.
  const bool fh = periodic || (force & NID_LEGACY_HREF_STATE);
  bool same = S.have_href && !always_upload();
  bool keyed = false;  // bs_ref's slice hashes were taken by this call
  if (same) {
    const bool x = same_content(S.k_bs_ref, a.bs_ref, 4 * N, fh, &keyed), y = same_small(S.k_counter, a.bs_counter, (size_t)ncell);
    const bool z = !a.Href || same_small(S.k_href, a.Href, (size_t)ncell);
    if (keyed) S.fresh |= NID_LEGACY_HREF_STATE;
    same = x && y && z;
  }
  if (!same) {
    std::vector<double> href(ncell, 0.0);
    if (a.Href) for (int c = 0; c < ncell; c++) href[c] = a.Href[c];
    rc = nid_multi_set_href_state(m, a.bs_counter, href.data(), a.bs_ref, a.bs_index_ref);
    if (rc != NID_OK) { report("CudaComputeH(href state upload)", rc, m); return nullptr; }
    if (!keyed) remember(S.k_bs_ref, a.bs_ref, 4 * N);
    remember_small(S.k_counter, a.bs_counter, (size_t)ncell);
    if (a.Href) remember_small(S.k_href, a.Href, (size_t)ncell); else S.k_href = LegacyState::Key();
    S.have_href = true;
    S.fresh |= NID_LEGACY_HREF_STATE;
    g_uploads++;
  }
  tr.step("href state keys checked");
  return m;
}

// The default mode's verification of ONE call: slices [calls * k, calls * k + k) mod kSlices of im0, points3d, im1 and
// bs_ref -- those whose keys this very call has not just taken -- go to the pool's workers (verify_begin, right before the
// evaluation is launched); verify_end joins them and returns the parts (NID_LEGACY_*) whose slice no longer hashes to the
// key's value.  Both inside the operator: the caller's buffers are not read once it has returned.
SliceHasher g_verify;
struct VerifyPlan { const LegacyState::Key *key[4]; const void *addr[4]; size_t n[4]; bool on[4]; int first, count; } g_plan;

void verify_begin(const CallArgs &a, bool long_call) {
  LegacyState &S = g_state;
  int k = always_upload() ? 0 : slices_per_call();
  if (k <= 0) return;
  // a call hashes what hides behind its own evaluation: an evaluation with the Jacobian phase takes twice a cost-only one
  // (20 against 10 us of device time at 640x480; the pool reads ~45 MB/ms), so it takes 1.5 k slices and a cost-only call
  // 0.75 k -- the LM's pattern of one in four covers 15/16 k per call on average (profiles/r06_legacy_call_cost.txt)
  if (k < kSlices && k >= 4) k = long_call ? k + k / 2 : k - k / 4;
  const size_t N = (size_t)a.rows * a.cols;
  static const unsigned part[4] = {NID_LEGACY_REFERENCE, NID_LEGACY_REFERENCE, NID_LEGACY_TARGET, NID_LEGACY_HREF_STATE};
  const LegacyState::Key *key[4] = {&S.k_im0, &S.k_points, &S.k_im1, &S.k_bs_ref};
  const void *addr[4] = {a.im0, a.points3d, a.im1, a.bs_ref};
  const size_t cnt[4] = {N, 3 * N, N, 4 * N};
  SliceHasher::Req req[4];
  bool any = false;
  for (int b = 0; b < 4; b++) {
    g_plan.key[b] = key[b]; g_plan.addr[b] = addr[b]; g_plan.n[b] = cnt[b];
    g_plan.on[b] = !(S.fresh & part[b]) && key[b]->valid && key[b]->full_known && key[b]->addr == addr[b] && key[b]->n == cnt[b] && addr[b];
    req[b].data = g_plan.on[b] ? addr[b] : nullptr;
    req[b].bytes = cnt[b] * sizeof(double);
    any = any || g_plan.on[b];
  }
  if (!any) return;
  g_plan.count = k;
  g_plan.first = (int)(S.rot % kSlices);
  S.rot += (unsigned long)k;
  g_verify.begin(req, 4, g_plan.first, g_plan.count);
}

unsigned verify_end(const CallArgs &a) {
  (void)a;
  if (!g_verify.running()) return 0;
  g_verify.end();
  static const unsigned part[4] = {NID_LEGACY_REFERENCE, NID_LEGACY_REFERENCE, NID_LEGACY_TARGET, NID_LEGACY_HREF_STATE};
  static const char *what[4] = {"im0", "points3d", "im1", "bs_ref"};
  unsigned stale = 0;
  for (int b = 0; b < 4; b++) {
    if (!g_plan.on[b]) continue;
    for (int c = 0; c < g_plan.count; c++) {
      const int s = (g_plan.first + c) % kSlices;
      if (g_verify.out[b][s] == g_plan.key[b]->slice[s]) continue;
      stale |= part[b];
      if (g_plan.count < kSlices) {  // (every slice on every call: nothing stale was ever evaluated -- followed silently, like the reference)
        std::fprintf(stderr, "[nid legacy] the caller's %s buffer was rewritten IN PLACE between two CudaComputeH calls (slice %d of %d no "
                             "longer matches what is on the device): earlier calls may have evaluated its old content; uploading the new "
                             "content and evaluating again.  nid_legacy_invalidate() announces such a change, "
                             "nid_legacy_set_verify_mode(NID_LEGACY_VERIFY_EVERY_CALL) checks every slice on every call.\n", what[b], s, kSlices);
        g_stale_detections++;
      }
      break;
    }
  }
  return stale;
}

nid_multi *follow_change(const CallArgs &a, unsigned stale) {
  g_state.force_full |= stale;
  return ensure_state(a);
}

}  // namespace

extern "C" {

void nid_legacy_set_jacobian_bound(int mode) {
  g_jac_bound = mode ? NID_JACBOUND_CUDA : NID_JACBOUND_CPU;
  if (g_state.m) nid_multi_set_options(g_state.m, g_jac_bound, NID_XFORM_MATRIX);
}

void nid_legacy_set_math_mode(int mode) {
  g_math_mode = mode ? NID_MATH_STRICT : NID_MATH_FAST;
  if (g_state.m) nid_multi_set_math_mode(g_state.m, g_math_mode);
}

void nid_legacy_set_resident(int on) {
  g_resident = on ? 1 : 0;
  if (g_state.m) (void)nid_multi_set_resident(g_state.m, g_resident);
}

void nid_legacy_set_launch_shape(int jac_threads, int cost_threads) {
  g_jac_threads = jac_threads; g_cost_threads = cost_threads;
  if (g_state.m)
    nid_multi_set_launch_shape(g_state.m, jac_threads_for(g_state.cell * g_state.cell, g_world > 1 ? g_world : (int)g_devices.size()),
                               cost_threads);
}

void nid_legacy_set_device(int device) {
  const int32_t d = device;
  nid_legacy_set_devices(&d, 1, 0);
}

static void drop_comm() {
  if (g_comm) { nid_comm_destroy(g_comm); g_comm = nullptr; }
}

void nid_legacy_set_devices(const int32_t *devices, int n, int reduce_rccl) {
  if (!devices || n < 1 || n > NID_MAX_SHARDS) return;
  nid_legacy_reset();
  drop_comm();
  g_devices.assign(devices, devices + n);
  g_rank = 0; g_world = 1; g_have_id = false;
  g_reduce_rccl = reduce_rccl ? 1 : 0;
}

void nid_legacy_set_rank(int device, int rank, int world, const uint8_t *rccl_id128) {
  nid_legacy_reset();
  drop_comm();
  g_devices.assign(1, device);
  g_rank = rank; g_world = world < 1 ? 1 : world;
  g_have_id = rccl_id128 != nullptr;
  if (rccl_id128) std::memcpy(g_id, rccl_id128, sizeof(g_id));
  g_reduce_rccl = g_world > 1 ? 1 : 0;
}

void nid_legacy_invalidate(unsigned parts) { g_state.force_full |= parts; }

void nid_legacy_set_verify_mode(int mode) {
  g_verify_mode = (mode == NID_LEGACY_VERIFY_EVERY_CALL || mode == NID_LEGACY_VERIFY_TRUSTED) ? mode : (int)NID_LEGACY_VERIFY_ROTATING;
}
void nid_legacy_set_verify_slices(int per_call) { g_verify_slices = per_call < 1 ? -1 : std::min(per_call, kSlices); }
void nid_legacy_set_trust_buffers(int on) { nid_legacy_set_verify_mode(on ? NID_LEGACY_VERIFY_TRUSTED : NID_LEGACY_VERIFY_ROTATING); }
long nid_legacy_stale_detections(void) { return g_stale_detections; }

void nid_legacy_reset(void) {
  if (g_state.m) nid_multi_destroy(g_state.m);
  for (LegacyState &T : g_parked) nid_multi_destroy(T.m);
  g_parked.clear();
  g_state = LegacyState();
  (void)nid_backproject_release();  // Calculate3Dpoint's scratch
}

void nid_legacy_quiesce(void) {
  if (g_state.m) (void)nid_multi_resident_pause(g_state.m);
}

nid_multi *nid_legacy_prepare(double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref,
                              int *bs_index_ref, double *camera_intrincis, int bin_num, int bs_degree, int cell_num,
                              int rows, int cols, double *Href) {
  const CallArgs a = {im0, im1, points3d, bs_counter, bs_ref, bs_index_ref, camera_intrincis, bin_num, bs_degree, cell_num, rows, cols, Href};
  nid_multi *m = ensure_state(a);
  if (!m) return nullptr;
  // (no evaluation to overlap with: this call's slices are hashed on the spot)
  verify_begin(a, false);
  const unsigned stale = verify_end(a);
  return stale ? follow_change(a, stale) : m;
}

// The frame pair in the driver's own formats on the context the operators use (same cache, same devices / ranks /
// communicator), through nid_multi_set_pair_u16: for hosts that need none of the operators' per-pixel arrays (the fused LM
// flows of g2o_min).  The caller-buffer keys are dropped: a legacy call that follows re-uploads whatever it is handed.
nid_multi *nid_legacy_set_pair_u16(const uint16_t *depth_u16, const uint8_t *im0, const uint8_t *im1, const double *T_wc0_colmajor16,
                                   const double *pose0_colmajor16, const double *camera_intrincis, int bin_num, int bs_degree,
                                   int cell_num, int rows, int cols, int32_t *bs_counter, double *Href) {
  nid_multi *m = get_multi(rows, cols, cell_num, bin_num, bs_degree, camera_intrincis);
  if (!m) return nullptr;
  LegacyState &S = g_state;
  S.k_im0 = S.k_points = S.k_im1 = S.k_bs_ref = S.k_counter = S.k_href = LegacyState::Key();
  S.have_ref = S.have_target = S.have_href = false;
  S.calls = 0;
  int rc = nid_multi_set_pair_u16(m, depth_u16, camera_intrincis[4], im0, im1, T_wc0_colmajor16, nullptr, pose0_colmajor16, bs_counter, Href);
  if (rc != NID_OK) { report("nid_legacy_set_pair_u16", rc, m); return nullptr; }
  g_uploads++;
  return m;
}

nid_multi *nid_legacy_multi(void) { return g_state.m; }

nid_ctx *nid_legacy_context(void) { return g_state.m ? nid_multi_shard(g_state.m, 0) : nullptr; }

long nid_legacy_upload_count(void) { return g_uploads; }

}  // extern "C"
