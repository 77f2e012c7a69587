// legacy_ops.cpp -- Calculate3Dpoint / CudaComputeHref / g2o::CudaComputeH on top
// of the C-ABI (include/nid/nid_c.h, include/nid/nid_multi.h).  See include/nid/legacy_ops.h.
//
// The three operators keep the reference's signatures and its ownership rules (the caller owns every buffer
// and hands ALL of them over on every call, computeH.cu:373-502), but not its per-call cost (>= 10 mallocs,
// 52-367 MB of memsets, 11 MB of uploads): the frame-pair state stays resident on the device(s) between calls.
// What is resident is identified by CONTENT, not by pointer: a caller that frees and re-mallocs its buffers per
// pair (NID_pose_estimation.cpp:229-251, 385-392) usually gets the same addresses back.
//   * CudaComputeHref is called once per pair: it always uploads the reference (im0, points3d) afresh;
//   * CudaComputeH keeps, per caller buffer (im0, points3d, im1, bs_ref, bs_counter, Href), a key: address, length, a
//     quick fingerprint of 64 samples and a FULL 64-bit hash of the content.
//     Three modes (nid_legacy_set_verify_mode; round 5):
//     BACKGROUND (default): a call checks address, length and the quick fingerprint (about a microsecond for all six
//     buffers) and evaluates; beside it a small pool of worker threads takes the full hashes of the four big buffers --
//     one verification at a time, started by a call whenever none is running (22 MB at 640x480: ~0.4 ms on three
//     workers) -- and the first call after a verification that found a buffer changed IN PLACE says so loudly on stderr,
//     counts it (nid_legacy_stale_detections), and uploads the buffer's current content: a caller that rewrites a buffer
//     in place without saying so is followed within a verification's time, at most a few calls evaluated on the old
//     content, and is told.  (Verifying INSIDE every call costs 0.36 ms per call -- ten times the evaluation,
//     profiles/r05_pair_setup.txt -- which is why it is not the default.)
//     EVERY_CALL: every call recomputes the full hash of every buffer before it evaluates -- the caller's bytes are read on
//     every call, like the reference, which re-uploads them on every call (computeH.cu:420-429) -- and uploads what
//     differs: a change in place is followed on the NEXT call, with nothing asked of the caller.
//     TRUSTED (nid_legacy_set_trust_buffers(1) or NID_LEGACY_TRUST_BUFFERS=1; round 4's default): the cheap check only;
//     the full hash is recomputed -- and decides whether the buffer is uploaded -- when the cheap part changed, on every
//     kRehashEvery-th call of a pair, and after nid_legacy_invalidate().  No thread is started by this mode's calls
//     once the pair is set up.
//     In every mode a new frame pair (new buffers, or CudaComputeHref) is noticed at once, and nid_legacy_invalidate(parts)
//     makes the next call recompute the named parts' full hashes.
//     The two per-cell arrays (1-2 KB) are fully hashed on every call in both modes; the images' keys carry the hash
//     of their u8 conversion (what is uploaded) and, in the default mode, of the caller's f64 bytes (what is verified).
//     NID_LEGACY_ALWAYS_UPLOAD=1 uploads everything on every call.
#include "nid/legacy_ops.h"

#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "nid/nid_c.h"
#include "nid/nid_multi.h"

namespace {

struct LegacyState {
  nid_multi *m = nullptr;
  int rows = 0, cols = 0, cell = 0, bins = 0;
  double intr[4] = {0, 0, 0, 0};
  // keys of what is resident on the device(s)
  struct Key {
    const void *addr = nullptr;
    size_t n = 0;
    uint64_t quick = 0, full = 0;
    bool valid = false;
    bool full_known = true;  // false: `full` was never taken -- a full check then counts as "changed"
    uint64_t raw = 0;        // images: the hash of the caller's f64 bytes (`full` is that of the u8 conversion)
    bool raw_known = false;
  };
  Key k_im0, k_points, k_im1, k_bs_ref, k_counter, k_href;
  bool have_ref = false, have_target = false, have_href = false;
  unsigned long calls = 0;      // CudaComputeH calls on this frame pair (CudaComputeHref starts a new count)
  unsigned long epoch = 0;      // bumped by every upload: a background verification that straddles one is discarded
  unsigned force_full = 0;      // nid_legacy_invalidate: parts whose full hash the next call recomputes
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
// into profiles/r04_pair_setup.txt)
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

int g_verify_mode = -1;  // nid_legacy_set_verify_mode; -1 = the environment (NID_LEGACY_TRUST_BUFFERS=1, NID_LEGACY_VERIFY_EVERY_CALL=1) or the default
int verify_mode() {
  if (g_verify_mode >= 0) return g_verify_mode;
  static const int env = [] {
    const char *t = getenv("NID_LEGACY_TRUST_BUFFERS"), *e = getenv("NID_LEGACY_VERIFY_EVERY_CALL");
    if (t && t[0] == '1') return (int)NID_LEGACY_VERIFY_TRUSTED;
    if (e && e[0] == '1') return (int)NID_LEGACY_VERIFY_EVERY_CALL;
    return (int)NID_LEGACY_VERIFY_BACKGROUND;
  }();
  return env;
}
bool trust_buffers() { return verify_mode() == NID_LEGACY_VERIFY_TRUSTED; }
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
// A frame pair's big buffers (points3d 7.4 MB, bs_value 9.8 MB at 640x480) are hashed once per pair -- and, in the
// default mode, on every CudaComputeH call; one core does ~20 GB/s, i.e. 0.4-0.5 ms each.  Buffers of 1 MB and more are
// therefore hashed in kHashParts contiguous parts by a small pool of worker threads (created at the first use, parked
// on a condition variable in between), and the key is the combination of the parts' hashes in order.  The pool's size:
// NID_LEGACY_HASH_THREADS workers beside the caller (default 3; 0: the caller hashes alone, no thread is created).
constexpr int kHashParts = 4;
constexpr size_t kHashParallelBytes = 1u << 20;

class HashPool {
 public:
  static HashPool &get() { static HashPool *p = new HashPool;  return *p; }  // (never destroyed: its parked workers end with the process)
  // start(n, job): job(0..n-1) is handed to the workers -- each takes the next part that nobody has taken -- and the
  // call returns; finish(): the caller takes what is left and waits for the rest.  One job at a time (start() holds a
  // lock until finish()); the job must stay valid until finish().
  void start(int nparts, std::function<void(int)> job) {
    call_.lock();
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = std::move(job);
      next_ = 0;
      nparts_ = nparts;
      remaining_ = nparts;
      generation_++;
    }
    if (!workers_.empty() && getpid() == owner_) wake_.notify_all();  // (a fork()ed child has no workers: finish() does all parts)
  }
  void finish() {
    help();
    {
      std::unique_lock<std::mutex> g(m_);
      done_.wait(g, [&] { return remaining_ == 0; });
      job_ = nullptr;
    }
    call_.unlock();
  }
  void run(int nparts, std::function<void(int)> job) { start(nparts, std::move(job)); finish(); }
  // a started job whose parts the WORKERS have all done (nobody called finish() yet)?  false without workers: finish() does the work then
  bool done_by_workers() {
    std::lock_guard<std::mutex> g(m_);
    return remaining_ == 0;
  }
  bool has_workers() const { return !workers_.empty() && getpid() == owner_; }

 private:
  HashPool() : owner_(getpid()) {
    int n = 3;
    if (const char *e = getenv("NID_LEGACY_HASH_THREADS")) n = std::max(0, std::min(15, atoi(e)));
    for (int w = 0; w < n; w++) workers_.emplace_back([this] { loop(); });
    // fork(): threads do not survive it, locks do -- a child forked while a worker held m_ would wait for it for ever.
    // The handlers take both locks around the fork, so the child inherits them free (and finds itself without workers
    // by its pid).
    pthread_atfork([] { HashPool &p = get(); p.call_.lock(); p.m_.lock(); },
                   [] { HashPool &p = get(); p.m_.unlock(); p.call_.unlock(); },
                   [] { HashPool &p = get(); p.m_.unlock(); p.call_.unlock(); });
  }
  void help() {  // take parts until none is left
    for (;;) {
      std::function<void(int)> *job;
      int part;
      {
        std::lock_guard<std::mutex> g(m_);
        if (next_ >= nparts_ || !job_) return;
        part = next_++;
        job = &job_;
      }
      (*job)(part);
      bool last;
      { std::lock_guard<std::mutex> g(m_); last = --remaining_ == 0; }
      if (last) done_.notify_all();
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        wake_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
      }
      help();
    }
  }
  std::mutex call_, m_;
  std::condition_variable wake_, done_;
  std::function<void(int)> job_;
  std::vector<std::thread> workers_;
  unsigned long generation_ = 0;
  int next_ = 0, nparts_ = 0, remaining_ = 0;
  const pid_t owner_;
};

// one contiguous run of bytes, 64 bits: eight interleaved multiply-xor lanes over the 8-byte words (eight independent
// dependency chains keep the multiplier busy: memory-bound, ~0.05 ms/MB on one core).  MARK (unused since the NaN rows of
// bs_value are written on the device): rows of four doubles, an all-zero row becomes four NaNs first and is hashed as such.
template <bool MARK>
uint64_t hash_run(unsigned char *b, size_t bytes, uint64_t salt) {
  const size_t words = bytes / 8;
  constexpr int L = 8;
  uint64_t h[L] = {0x9E3779B97F4A7C15ull ^ salt, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull,
                   0x85EBCA77C2B2AE63ull, 0xD6E8FEB86659FD93ull, 0xA0761D6478BD642Full, 0xE7037ED1A0B428DBull};
  auto mark = [](double *w) {
    if (w[0] == 0.0 && w[1] == 0.0 && w[2] == 0.0 && w[3] == 0.0) w[0] = w[1] = w[2] = w[3] = NAN;
  };
  size_t i = 0;
  for (; i + L <= words; i += L) {
    if (MARK) { mark(reinterpret_cast<double *>(b) + i); mark(reinterpret_cast<double *>(b) + i + 4); }
    uint64_t w[L];
    std::memcpy(w, b + 8 * i, 8 * L);
    for (int k = 0; k < L; k++) { h[k] = (h[k] ^ w[k]) * 0x9FB21C651E98DF25ull; h[k] ^= h[k] >> 29; }
  }
  if (MARK && i < words) mark(reinterpret_cast<double *>(b) + i);  // (a run of whole rows: at most one row is left)
  for (; i < words; i++) { uint64_t w; std::memcpy(&w, b + 8 * i, 8); h[i % L] = (h[i % L] ^ w) * 0x9FB21C651E98DF25ull; h[i % L] ^= h[i % L] >> 29; }
  for (size_t t = 8 * words; t < bytes; t++) h[0] = (h[0] ^ b[t]) * 0x100000001B3ull;
  uint64_t r = h[0];
  for (int k = 1; k < L; k++) r = (r ^ h[k]) * 0xFF51AFD7ED558CCDull + k;
  r ^= r >> 32;
  return r;
}

// the whole content: one run, or kHashParts runs (split at multiples of 64 bytes -- whole bs_value rows) combined in order.
// PendingHash: the parts are on the pool's workers; get() takes what is left, waits and combines (the caller may have
// done something else in between: upload_reference hashes the points while they cross PCIe)
struct PendingHash {
  uint64_t h[kHashParts] = {};
  uint64_t direct = 0;
  bool pooled = false;
  uint64_t get() {
    uint64_t r = direct;
    if (pooled) {
      HashPool::get().finish();
      pooled = false;
      r = h[0];
      for (int p = 1; p < kHashParts; p++) r = (r ^ h[p]) * 0xFF51AFD7ED558CCDull + p;
      r ^= r >> 32;
    }
    return r ? r : 2;
  }
};
void background_drain();
template <bool MARK>
void hash_bytes_begin(unsigned char *b, size_t bytes, PendingHash *ph) {
  background_drain();
  if (bytes < kHashParallelBytes) { ph->direct = hash_run<MARK>(b, bytes, bytes); ph->pooled = false; return; }
  const size_t part = (bytes / kHashParts) & ~(size_t)63;
  ph->pooled = true;
  HashPool::get().start(kHashParts, [b, bytes, part, ph](int p) {
    const size_t lo = part * p, hi = p == kHashParts - 1 ? bytes : part * (p + 1);
    ph->h[p] = hash_run<MARK>(b + lo, hi - lo, bytes + p);
  });
}
template <bool MARK>
uint64_t hash_bytes(unsigned char *b, size_t bytes) {
  PendingHash ph;
  hash_bytes_begin<MARK>(b, bytes, &ph);
  return ph.get();
}

template <typename T>
uint64_t full_hash(const T *a, size_t n) {
  if (!a) return 1;
  return hash_bytes<false>(reinterpret_cast<unsigned char *>(const_cast<T *>(a)), n * sizeof(T));
}

// Several buffers in ONE pooled job (the per-call verification of the default mode: four buffers, sixteen parts): each
// buffer's value is exactly what full_hash gives for it alone (same partition, same combination).
struct HashReq {
  const void *data = nullptr;
  size_t bytes = 0;
  uint64_t result = 0;
};
class ManyHash {
 public:
  // begin(): the parts go to the pool's workers and the call returns; ready(): the workers have done them all;
  // end(): take what is left, wait, combine -- results in req[].result
  void begin(const HashReq *reqs, int n) {
    parts_.clear();
    n_ = n;
    for (int r = 0; r < n; r++) req[r] = reqs[r];
    h_.assign((size_t)n * kHashParts, 0);
    for (int r = 0; r < n; r++) {
      unsigned char *b = static_cast<unsigned char *>(const_cast<void *>(req[r].data));
      const size_t bytes = req[r].bytes;
      if (!b) continue;
      if (bytes < kHashParallelBytes) { parts_.push_back({r, -1, b, bytes, bytes}); continue; }
      const size_t part = (bytes / kHashParts) & ~(size_t)63;
      for (int p = 0; p < kHashParts; p++) {
        const size_t lo = part * p, hi = p == kHashParts - 1 ? bytes : part * (p + 1);
        parts_.push_back({r, p, b + lo, hi - lo, bytes + p});
      }
    }
    HashPool::get().start((int)parts_.size(), [this](int k) {
      const Part &pt = parts_[(size_t)k];
      h_[(size_t)pt.req * kHashParts + (pt.idx < 0 ? 0 : pt.idx)] = hash_run<false>(pt.b, pt.bytes, pt.salt);
    });
    running_ = true;
  }
  bool running() const { return running_; }
  bool ready() { return running_ && HashPool::get().done_by_workers(); }
  void end() {
    if (!running_) return;
    HashPool::get().finish();
    running_ = false;
    for (int r = 0; r < n_; r++) {
      if (!req[r].data) { req[r].result = 1; continue; }
      uint64_t v = h_[(size_t)r * kHashParts];
      if (req[r].bytes >= kHashParallelBytes) {
        for (int p = 1; p < kHashParts; p++) v = (v ^ h_[(size_t)r * kHashParts + p]) * 0xFF51AFD7ED558CCDull + p;
        v ^= v >> 32;
      }
      req[r].result = v ? v : 2;
    }
  }
  HashReq req[4];

 private:
  struct Part { int req, idx; unsigned char *b; size_t bytes; uint64_t salt; };
  std::vector<Part> parts_;
  std::vector<uint64_t> h_;
  int n_ = 0;
  bool running_ = false;
};

// The BACKGROUND verification (the default mode): the full hashes of the four big buffers, taken by the pool's workers
// while the caller carries on; looked at by the next CudaComputeH call that finds them ready.  What they are compared
// with is what the keys held when the verification STARTED; an upload in between (`epoch`) makes the result moot.
struct Background {
  ManyHash mh;
  const void *addr[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t n[4] = {0, 0, 0, 0};
  uint64_t expect[4] = {0, 0, 0, 0};
  bool have[4] = {false, false, false, false};
  unsigned long epoch = 0;
  bool finished = false;  // its results wait to be looked at
  std::chrono::steady_clock::time_point last_end{};
} g_bg;
// between two verifications: what the workers take of the caller's cores and memory bandwidth (back to back they slowed a
// loop of CudaComputeH calls by 60 %: profiles/r05_pair_setup.txt); a change in place is reported within this + ~0.4 ms
constexpr std::chrono::microseconds kBackgroundPause(2000);

// every SYNCHRONOUS use of the pool first lets a background verification end (it holds the pool until then)
void background_drain() {
  if (g_bg.mh.running()) {
    g_bg.mh.end();
    g_bg.finished = true;
    g_bg.last_end = std::chrono::steady_clock::now();
  }
}

void hash_many(HashReq *reqs, int n) {
  background_drain();
  ManyHash m;
  m.begin(reqs, n);
  m.end();
  for (int r = 0; r < n; r++) reqs[r].result = m.req[r].result;
}

// Does the caller's buffer still hold what is resident?  Updates the key; `force`: recompute the full hash even if
// address, length and the quick fingerprint are unchanged (`pre`: that hash, if the caller has taken it already).
template <typename T>
bool same_content(LegacyState::Key &k, const T *a, size_t n, bool force, const uint64_t *pre = nullptr) {
  const uint64_t q = fingerprint(a, n);
  const bool cheap_same = k.valid && k.addr == (const void *)a && k.n == n && k.quick == q;
  if (cheap_same && !force) return true;
  const uint64_t f = pre ? *pre : full_hash(a, n);
  // a key whose full hash was never taken (trusted mode: CudaComputeHref's bs_value) is completed by its first full
  // check; in the default mode a key without one counts as "changed"
  bool same = k.valid && k.n == n && k.full_known && k.full == f;
  if (cheap_same && !k.full_known && trust_buffers()) same = true;
  k.addr = a; k.n = n; k.quick = q; k.full = f; k.valid = true; k.full_known = true;
  return same;
}
template <typename T>
void remember(LegacyState::Key &k, const T *a, size_t n) {
  k.addr = a; k.n = n; k.quick = fingerprint(a, n); k.full = full_hash(a, n); k.valid = true; k.full_known = true;
}

bool to_u8(const double *im, size_t n, std::vector<uint8_t> *out);
// An f64 image carrying u8 values (NID_pose_estimation.cpp:245-251).  Its key's `full` hash is that of the CONVERTED image
// -- the conversion checks every value (an integer in [0, 255] or the call fails), so the u8 image determines the
// buffer's content, and the conversion is needed for the upload anyway; in the default mode the key also carries the
// hash of the caller's f64 bytes (`raw`), which is what a per-call verification compares (no conversion while nothing
// changes; `pre_raw`: that hash, if the caller has taken it already).  `have`: something is resident to compare with.
// On return *same says whether the resident image can stay; u8 holds the converted image whenever the conversion ran.
int check_image(LegacyState::Key &k, const double *im, size_t n, bool force, bool have, std::vector<uint8_t> *u8, bool *same,
                const uint64_t *pre_raw = nullptr) {
  const uint64_t q = fingerprint(im, n);
  u8->clear();
  const bool cheap_same = have && k.valid && k.addr == (const void *)im && k.n == n && k.quick == q;
  if (cheap_same && !force) { *same = true; return NID_OK; }
  const bool keep_raw = !trust_buffers();
  uint64_t raw = 0;
  if (keep_raw) {
    raw = pre_raw ? *pre_raw : full_hash(im, n);
    if (cheap_same && k.raw_known && k.raw == raw) { *same = true; return NID_OK; }
  }
  if (!to_u8(im, n, u8)) return NID_ERR_UNSUPPORTED;
  const uint64_t f = full_hash(u8->data(), n);
  *same = have && k.valid && k.n == n && k.full == f;
  k.addr = im; k.n = n; k.quick = q; k.full = f; k.valid = true; k.full_known = true;
  k.raw = raw; k.raw_known = keep_raw;
  return NID_OK;
}

void report(const char *where, int rc, nid_multi *m) {
  // the reference prints CUDA errors and carries on (computeH.cu:454-473)
  std::fprintf(stderr, "[nid legacy] %s failed: %s (%d) %s\n", where, nid_status_string(rc), rc,
               m ? nid_multi_last_error(m) : "");
}

nid_multi *get_multi(int rows, int cols, int cell, int bins, int deg, const double *intr) {
  LegacyState &S = g_state;
  const bool same = S.m && S.rows == rows && S.cols == cols && S.cell == cell && S.bins == bins &&
                    S.intr[0] == intr[0] && S.intr[1] == intr[1] && S.intr[2] == intr[2] && S.intr[3] == intr[3];
  if (same) return S.m;
  if (S.m) nid_multi_destroy(S.m);
  S = LegacyState();
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
  // the operator signatures carry a 4x4 matrix: computeH.cu:152-154 semantics for the transform
  nid_multi_set_options(m, g_jac_bound, NID_XFORM_MATRIX);
  nid_multi_set_math_mode(m, g_math_mode);
  nid_multi_set_href_nan_markers(m, 1);  // CudaComputeHref's bs_value with the CUDA operator's NaN rows, written on the device
  // the operators are blocking, one pose (or one LM rejection chain) at a time: latency matters, not the
  // pipelined throughput the 128-thread default is tuned for (nid_set_launch_shape, tools/latency_sweep.py)
  nid_multi_set_launch_shape(m, jac_threads_for(cell * cell, g_world > 1 ? g_world : (int)g_devices.size()), g_cost_threads);
  if (resident_wanted()) (void)nid_multi_set_resident(m, 1);  // (unsupported platform: the launched form)
  S.m = m; S.rows = rows; S.cols = cols; S.cell = cell; S.bins = bins;
  std::memcpy(S.intr, intr, sizeof(S.intr));
  return m;
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
    int rc = check_image(S.k_im0, im0, N, true, false, &local, &dummy);
    if (rc != NID_OK) return rc;
    im = &local;
  }
  // the points' content key is taken WHILE they cross PCIe (7.4 MB at 640x480: 0.2 ms on the pool's threads, hidden)
  PendingHash ph;
  if (!points_keyed) hash_bytes_begin<false>(reinterpret_cast<unsigned char *>(const_cast<double *>(points3d)), 3 * N * sizeof(double), &ph);
  int rc = nid_multi_set_reference_points(S.m, points3d, im->data());
  if (!points_keyed) {
    const uint64_t f = ph.get();  // (on every path: the pool is held until then)
    if (rc == NID_OK) { S.k_points.addr = points3d; S.k_points.n = 3 * N; S.k_points.quick = fingerprint(points3d, 3 * N); S.k_points.full = f; S.k_points.valid = true; S.k_points.full_known = true; }
  }
  if (rc != NID_OK) return rc;
  S.have_ref = true; S.have_href = false;
  S.epoch++;
  g_uploads++;
  return NID_OK;
}

int ensure_reference(LegacyState &S, const double *im0, const double *points3d, bool force, const uint64_t *pre_im0 = nullptr,
                     const uint64_t *pre_points = nullptr) {
  const size_t N = (size_t)S.rows * S.cols;
  std::vector<uint8_t> im;
  bool points_keyed = false;
  if (S.have_ref && !always_upload()) {
    bool a = false;
    int rc = check_image(S.k_im0, im0, N, force, true, &im, &a, pre_im0);
    if (rc != NID_OK) return rc;
    const bool b = same_content(S.k_points, points3d, 3 * N, force, pre_points);
    points_keyed = true;
    if (a && b) return NID_OK;
  }
  return upload_reference(S, im0, points3d, &im, points_keyed);
}

}  // namespace

void Calculate3Dpoint(double *depth, double *pose_c2w, double *points_3d, double *camera_intrincis, int rows,
                      int cols) {
  StepTrace tr("Calculate3Dpoint");
  background_drain();       // (points_3d may be the buffer a verification is reading)
  g_bg.finished = false;
  int rc = nid_backproject(depth, pose_c2w, camera_intrincis[0], camera_intrincis[1], camera_intrincis[2],
                           camera_intrincis[3], rows, cols, g_devices[0], points_3d);
  if (rc != NID_OK) report("Calculate3Dpoint", rc, nullptr);
  tr.step("upload depth, kernel, points back");
}

void CudaComputeHref(double *im0, double *points3d, double *pose, double *camera_intrincis, int bin_num,
                     int bs_degree, int cell_num, int rows, int cols, double *bs_value, int *bs_index,
                     int *bs_counter, double *Href) {
  StepTrace tr("CudaComputeHref");
  background_drain();       // (a verification of the previous pair's buffers: over before this pair's are written)
  g_bg.finished = false;
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
  S.calls = 0;  // (the trusted mode's periodic full check counts the calls of THIS pair)
  if (bs_value && !trust_buffers()) {
    // default mode: every CudaComputeH call compares the caller's bs_ref with this hash (9.8 MB at 640x480, on the pool).
    // (The legacy NaN rows, CudaComputeHref.cu:82-87, 126-130, were written on the device: nid_set_href_nan_markers.)
    remember(S.k_bs_ref, bs_value, 4 * N);
  } else if (bs_value) {
    // trusted buffers: address, length and the sampled fingerprint now and NO full hash (0.25 ms even on the pool); the
    // first full check that reaches the key -- the kRehashEvery-th call of the pair, nid_legacy_invalidate -- takes it.
    S.k_bs_ref.addr = bs_value; S.k_bs_ref.n = 4 * N; S.k_bs_ref.full = 0; S.k_bs_ref.full_known = false;
    S.k_bs_ref.quick = fingerprint(bs_value, 4 * N); S.k_bs_ref.valid = true;
  } else {
    remember(S.k_bs_ref, bs_value, 0);
  }
  tr.step("content key of bs_value (default: full hash; trusted: fingerprint)");
  remember(S.k_counter, bs_counter, (size_t)ncell);
  remember(S.k_href, Href, (size_t)ncell);
  tr.step("content keys of the per-cell outputs");
}

namespace {
nid_multi *ensure_state(double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref, int *bs_index_ref,
                        double *camera_intrincis, int bin_num, int bs_degree, int cell_num, int rows, int cols,
                        double *Href);
}

namespace g2o {

void CudaComputeH(bool calculate_der, double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref,
                  int *bs_index_ref, double *pose, double *camera_intrincis, int bin_num, int bs_degree,
                  int cell_num, int rows, int cols, double *Href, double *pro_target, double *pro_joint,
                  double *Htarget, double *Hjoint, double *der) {
  (void)pro_target; (void)pro_joint;  // accepted, never read or written (computeH.cu:373-502)
  nid_multi *m = ensure_state(im0, im1, points3d, bs_counter, bs_ref, bs_index_ref, camera_intrincis, bin_num, bs_degree,
                              cell_num, rows, cols, Href);
  if (!m) return;
  const int ncell = cell_num * cell_num;
  std::vector<double> ht(ncell), hj(ncell);
  int rc = nid_multi_evaluate_matrix(m, pose, calculate_der ? 1 : 0, ht.data(), hj.data(), nullptr, calculate_der ? der : nullptr);
  if (rc != NID_OK) { report("CudaComputeH", rc, m); return; }
  for (int c = 0; c < ncell; c++) {
    // CalculateHKernel: NaN for bs_counter < 300, else `-=` onto the caller's (zeroed) value
    Htarget[c] = std::isnan(ht[c]) ? NAN : Htarget[c] + ht[c];
    Hjoint[c] = std::isnan(hj[c]) ? NAN : Hjoint[c] + hj[c];
  }
}

}  // namespace g2o

namespace {

// Everything CudaComputeH does before its kernels: the frame-pair state on the device(s), keyed on content.
nid_multi *ensure_state(double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref, int *bs_index_ref,
                        double *camera_intrincis, int bin_num, int bs_degree, int cell_num, int rows, int cols,
                        double *Href) {
  nid_multi *m = get_multi(rows, cols, cell_num, bin_num, bs_degree, camera_intrincis);
  if (!m) return nullptr;
  LegacyState &S = g_state;
  const size_t N = (size_t)rows * cols;
  const int ncell = cell_num * cell_num;
  const bool first_of_pair = !S.have_target;
  StepTrace tr("CudaComputeH state check");
  tr.on = tr.on && first_of_pair;  // (every later call of a pair takes about a microsecond: not traced)
  // The full hashes.  Default: on every call, all four big buffers in one pooled job (the caller's bytes are read on every
  // call, like the reference's uploads).  Trusted buffers: whenever a key's cheap part changed (same_content), on every
  // kRehashEvery-th call of the pair, after an invalidate.
  S.calls++;
  const int mode = always_upload() ? (int)NID_LEGACY_VERIFY_TRUSTED : verify_mode();
  const bool verify = mode == NID_LEGACY_VERIFY_EVERY_CALL;
  const bool periodic = verify || (mode == NID_LEGACY_VERIFY_TRUSTED && S.calls % kRehashEvery == 0);
  unsigned force = S.force_full;
  S.force_full = 0;
  // BACKGROUND: a verification the workers have finished is looked at now.  A buffer whose content no longer hashes to
  // what its key held when the verification started -- same address, same length, no upload in between -- was rewritten
  // in place: said loudly, counted, and followed (its part takes the full check below, i.e. it is uploaded).
  if (mode == NID_LEGACY_VERIFY_BACKGROUND) {
    if (g_bg.mh.running() && (g_bg.mh.ready() || !HashPool::get().has_workers())) background_drain();
    if (g_bg.finished) {
      g_bg.finished = false;
      if (g_bg.epoch == S.epoch) {
        const void *now_addr[4] = {im0, points3d, im1, bs_ref};
        const size_t now_n[4] = {N, 3 * N, N, 4 * N};
        static const char *what[4] = {"im0", "points3d", "im1", "bs_ref"};
        static const unsigned part[4] = {NID_LEGACY_REFERENCE, NID_LEGACY_REFERENCE, NID_LEGACY_TARGET, NID_LEGACY_HREF_STATE};
        for (int k = 0; k < 4; k++)
          if (g_bg.have[k] && g_bg.addr[k] == now_addr[k] && g_bg.n[k] == now_n[k] && g_bg.mh.req[k].result != g_bg.expect[k] && !(force & part[k])) {
            std::fprintf(stderr, "[nid legacy] the caller's %s buffer was rewritten IN PLACE between two CudaComputeH calls (found by the background "
                                 "verification): calls since then may have evaluated its old content; uploading the new content now.  "
                                 "nid_legacy_invalidate() announces such a change, nid_legacy_set_verify_mode(NID_LEGACY_VERIFY_EVERY_CALL) checks every call.\n", what[k]);
            force |= part[k];
            g_stale_detections++;
          }
      }
    }
  } else {
    background_drain();
    g_bg.finished = false;
  }
  uint64_t pre[4] = {0, 0, 0, 0};
  const bool have_pre = verify && S.have_ref && S.have_target && S.have_href;
  if (have_pre) {
    HashReq req[4];
    req[0].data = im0; req[0].bytes = N * sizeof(double);
    req[1].data = points3d; req[1].bytes = 3 * N * sizeof(double);
    req[2].data = im1; req[2].bytes = N * sizeof(double);
    req[3].data = bs_ref; req[3].bytes = 4 * N * sizeof(double);
    hash_many(req, 4);
    for (int k = 0; k < 4; k++) pre[k] = req[k].result;
    tr.on = tr.on || (StepTrace("").on && S.calls == 2);  // (traced once per pair: the second call is the first that only verifies)
    tr.step("per-call verification: full hashes of im0, points3d, im1, bs_ref");
  }
  int rc = ensure_reference(S, im0, points3d, periodic || (force & NID_LEGACY_REFERENCE), have_pre ? &pre[0] : nullptr, have_pre ? &pre[1] : nullptr);
  if (rc != NID_OK) { report("CudaComputeH(reference upload)", rc, m); return nullptr; }
  {
    std::vector<uint8_t> im;
    bool same = false;
    const bool have = S.have_target && !always_upload();
    rc = check_image(S.k_im1, im1, N, periodic || (force & NID_LEGACY_TARGET), have, &im, &same, have_pre ? &pre[2] : nullptr);
    if (rc != NID_OK) { report("CudaComputeH(im1 is not u8-valued)", rc, m); return nullptr; }
    if (!same) {
      rc = nid_multi_set_target_u8(m, im.data());
      if (rc != NID_OK) { report("CudaComputeH(target upload)", rc, m); return nullptr; }
      S.have_target = true;
      S.epoch++;
      g_uploads++;
    }
  }
  tr.step("reference keys checked, target -> u8 + upload + margins");
  // Href is only READ by the reference with calculate_der (computeH.cu:428-429); the kernels also need it to know
  // which cells are active.  A NULL Href gives zeros (cost-only outputs do not depend on it) and the first call
  // that brings one replaces them: Href is part of the key.  The per-cell arrays are small: full hashes every call.
  const bool fh = periodic || (force & NID_LEGACY_HREF_STATE);
  bool same = S.have_href && !always_upload();
  if (same) {
    const bool a = same_content(S.k_bs_ref, bs_ref, 4 * N, fh, have_pre ? &pre[3] : nullptr), b = same_content(S.k_counter, bs_counter, (size_t)ncell, true);
    const bool c = !Href || same_content(S.k_href, Href, (size_t)ncell, true);
    same = a && b && c;
  }
  if (!same) {
    std::vector<double> href(ncell, 0.0);
    if (Href) for (int c = 0; c < ncell; c++) href[c] = Href[c];
    rc = nid_multi_set_href_state(m, bs_counter, href.data(), bs_ref, bs_index_ref);
    if (rc != NID_OK) { report("CudaComputeH(href state upload)", rc, m); return nullptr; }
    remember(S.k_bs_ref, bs_ref, 4 * N); remember(S.k_counter, bs_counter, (size_t)ncell);
    if (Href) remember(S.k_href, Href, (size_t)ncell); else S.k_href = LegacyState::Key();
    S.have_href = true;
    S.epoch++;
    g_uploads++;
  }
  tr.step("href state keys checked");
  // BACKGROUND: start the next verification (one at a time, with a pause between two) against what the keys hold NOW
  // (not by a pair's first call: everything was hashed or uploaded by this very call, and a caller that continues on the
  // nid_multi_* interface -- nid_legacy_prepare: g2o_min's fused flows -- never comes back to look at the result)
  if (mode == NID_LEGACY_VERIFY_BACKGROUND && !first_of_pair && !g_bg.mh.running() && !g_bg.finished && HashPool::get().has_workers() &&
      std::chrono::steady_clock::now() - g_bg.last_end > kBackgroundPause) {
    HashReq req[4];
    const void *addr[4] = {im0, points3d, im1, bs_ref};
    const size_t cnt[4] = {N, 3 * N, N, 4 * N};
    const LegacyState::Key *key[4] = {&S.k_im0, &S.k_points, &S.k_im1, &S.k_bs_ref};
    for (int k = 0; k < 4; k++) {
      const bool image = k == 0 || k == 2;
      g_bg.have[k] = key[k]->valid && key[k]->addr == addr[k] && key[k]->n == cnt[k] && (image ? key[k]->raw_known : key[k]->full_known);
      g_bg.addr[k] = addr[k]; g_bg.n[k] = cnt[k];
      g_bg.expect[k] = image ? key[k]->raw : key[k]->full;
      req[k].data = g_bg.have[k] ? addr[k] : nullptr;
      req[k].bytes = cnt[k] * sizeof(double);
    }
    g_bg.epoch = S.epoch;
    g_bg.mh.begin(req, 4);
  }
  return m;
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
  background_drain();
  g_bg.finished = false;
  g_verify_mode = (mode == NID_LEGACY_VERIFY_EVERY_CALL || mode == NID_LEGACY_VERIFY_TRUSTED) ? mode : (int)NID_LEGACY_VERIFY_BACKGROUND;
}
void nid_legacy_set_trust_buffers(int on) { nid_legacy_set_verify_mode(on ? NID_LEGACY_VERIFY_TRUSTED : NID_LEGACY_VERIFY_BACKGROUND); }
long nid_legacy_stale_detections(void) { return g_stale_detections; }

void nid_legacy_reset(void) {
  background_drain();  // (it reads the caller's buffers: "before the caller frees its buffers")
  g_bg.finished = false;
  if (g_state.m) nid_multi_destroy(g_state.m);
  g_state = LegacyState();
  (void)nid_backproject_release();  // Calculate3Dpoint's scratch
}

void nid_legacy_quiesce(void) {
  background_drain();  // (no worker reads the caller's buffers once this returns)
  if (g_state.m) (void)nid_multi_resident_pause(g_state.m);
}

nid_multi *nid_legacy_prepare(double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref,
                              int *bs_index_ref, double *camera_intrincis, int bin_num, int bs_degree, int cell_num,
                              int rows, int cols, double *Href) {
  return ensure_state(im0, im1, points3d, bs_counter, bs_ref, bs_index_ref, camera_intrincis, bin_num, bs_degree, cell_num,
                      rows, cols, Href);
}

nid_multi *nid_legacy_multi(void) { return g_state.m; }

nid_ctx *nid_legacy_context(void) { return g_state.m ? nid_multi_shard(g_state.m, 0) : nullptr; }

long nid_legacy_upload_count(void) { return g_uploads; }

}  // extern "C"
