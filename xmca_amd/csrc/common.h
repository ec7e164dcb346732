// Common host/device helpers for the xmca_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/xmca_hip.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace xmca {

// status codes: the XMCA_* macros of include/xmca_hip.h

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define XMCA_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      throw ::xmca::Error(XMCA_ERR_HIP, std::string(#expr) + ": " +              \
                                                    hipGetErrorString(_e) + " (" +       \
                                                    __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    }                                                                                    \
  } while (0)

#define XMCA_CHECK(cond, code, msg)                        \
  do {                                                     \
    if (!(cond)) throw ::xmca::Error((code), (msg));       \
  } while (0)

// Device memory pool of one handle (one stream).  A solve takes and returns dozens of temporaries; hipMalloc maps pages
// (~0.1-1 ms for the sizes here) and hipFree waits for the WHOLE device, which also serialises handles that work side by
// side on different streams (the surrogate lanes of rule_n).  Blocks given back are kept and handed out again by size.
// Re-use is safe without events because a pool serves exactly one stream: whatever still reads a returned block was
// enqueued on that stream before whatever writes it next.  The calling thread names its pool with a PoolScope (API entry
// points, lane threads); a DevBuf remembers the pool it came from and returns there from any thread.
// Blocks above `keep_limit` in total (XMCA_POOL_LIMIT_GB, default 16) are freed straight away; XMCA_POOL=0 switches
// pooling off; xmca_trim_pool returns what is held.  Every pool of the process is registered: when hipMalloc fails, the
// blocks parked in ALL pools (sibling lanes, the parent handle, other handles) are given back before giving up - behind a
// hipDeviceSynchronize, so a block another stream released earlier is safe to free here.  The lane pools of
// rule_n / bootstrap are emptied when their call ends if they hold more than 4 GB (run_lanes, xmca_hip.cpp).
struct DevPool {
  struct Registry {
    std::mutex mu;
    std::vector<DevPool*> pools;
  };
  static Registry& registry() {
    static Registry* r = new Registry;     // never destroyed: pools of static handles may outlive any static of this file
    return *r;
  }
  int device = 0;                          // the device current when the pool was made: where its blocks live
  DevPool() {
    (void)hipGetDevice(&device);
    std::lock_guard<std::mutex> g(registry().mu);
    registry().pools.push_back(this);
  }
  static void trim_all() {
    // Blocks parked in OTHER pools may still be read by work queued on their streams: wait for the device before any of
    // them is freed (advisor, round 3: this used to rest on hipFree's implicit synchronisation alone).  The registry stays
    // locked for the loop - a pool must not be destroyed under it - but no longer across a device-wide wait per block.
    // Every pool's OWN device is waited for (advisor, round 4: only the caller's current device was).
    int orig = 0;
    (void)hipGetDevice(&orig);
    std::lock_guard<std::mutex> g(registry().mu);
    std::vector<int> devs;
    for (DevPool* p : registry().pools) {
      bool seen = false;
      for (int d : devs) seen = seen || d == p->device;
      if (!seen) devs.push_back(p->device);
    }
    if (devs.empty()) devs.push_back(orig);
    int cur = orig;
    for (int d : devs) {
      if (d != cur) { (void)hipSetDevice(d); cur = d; }
      (void)hipDeviceSynchronize();
    }
    if (cur != orig) (void)hipSetDevice(orig);
    for (DevPool* p : registry().pools) p->trim();
  }
  std::mutex mu;
  std::multimap<size_t, void*> blocks;     // capacity in bytes -> free block
  size_t held = 0;
  size_t keep_limit = [] {
    const char* e = std::getenv("XMCA_POOL_LIMIT_GB");
    const double gb = e ? std::atof(e) : 16.0;
    return (size_t)((gb > 0.0 ? gb : 0.0) * (double)((size_t)1 << 30));
  }();
  static bool enabled() {
    static const bool on = [] { const char* e = std::getenv("XMCA_POOL"); return !(e && e[0] == '0'); }();
    return on;
  }
  void* take(size_t bytes, size_t* cap) {
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = blocks.lower_bound(bytes);
      if (it != blocks.end() && it->first <= bytes + bytes / 4 + ((size_t)1 << 20)) {
        void* p = it->second;
        *cap = it->first;
        held -= it->first;
        blocks.erase(it);
        return p;
      }
    }
    void* p = nullptr;
    const size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {               // out of memory: give the kept blocks of every pool back and try once more
      (void)hipGetLastError();
      trim_all();
      e = hipMalloc(&p, want);
    }
    if (e != hipSuccess)
      throw Error(XMCA_ERR_HIP, std::string("hipMalloc of ") + std::to_string(want) + " bytes: " + hipGetErrorString(e));
    *cap = want;
    return p;
  }
  void give(void* p, size_t cap) {
    {
      std::lock_guard<std::mutex> g(mu);
      if (held + cap <= keep_limit) {
        blocks.emplace(cap, p);
        held += cap;
        return;
      }
    }
    (void)hipFree(p);
  }
  void trim() {
    std::lock_guard<std::mutex> g(mu);
    for (auto& b : blocks) (void)hipFree(b.second);
    blocks.clear();
    held = 0;
  }
  ~DevPool() {
    {
      std::lock_guard<std::mutex> g(registry().mu);
      auto& v = registry().pools;
      for (size_t i = 0; i < v.size(); ++i)
        if (v[i] == this) { v.erase(v.begin() + (std::ptrdiff_t)i); break; }
    }
    trim();
  }
  DevPool(const DevPool&) = delete;
  DevPool& operator=(const DevPool&) = delete;
};

inline DevPool*& current_pool() {
  thread_local DevPool* p = nullptr;
  return p;
}
struct PoolScope {
  DevPool* prev;
  explicit PoolScope(DevPool* p) : prev(current_pool()) { current_pool() = DevPool::enabled() ? p : nullptr; }
  ~PoolScope() { current_pool() = prev; }
  PoolScope(const PoolScope&) = delete;
  PoolScope& operator=(const PoolScope&) = delete;
};

// Owning device buffer.  Grows on demand, never shrinks; from the calling thread's pool when it has one, else hipMalloc.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;            // elements
  size_t cap_bytes = 0;      // size of the underlying block
  DevPool* owner = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), cap_bytes(o.cap_bytes), owner(o.owner) { o.p = nullptr; o.cap = 0; o.cap_bytes = 0; o.owner = nullptr; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p; cap = o.cap; cap_bytes = o.cap_bytes; owner = o.owner;
      o.p = nullptr; o.cap = 0; o.cap_bytes = 0; o.owner = nullptr;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) {
      if (owner) owner->give(p, cap_bytes);
      else (void)hipFree(p);
    }
    p = nullptr;
    cap = 0;
    cap_bytes = 0;
    owner = nullptr;
  }
  T* ensure(size_t n) {
    if (n > cap) {
      release();
      DevPool* pool = current_pool();
      if (pool) {
        p = static_cast<T*>(pool->take(n * sizeof(T), &cap_bytes));
        owner = pool;
        n = cap_bytes / sizeof(T);
      } else {
        XMCA_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
        cap_bytes = n * sizeof(T);
      }
      cap = n;
    }
    return p;
  }
  T* get() const { return p; }
};

// One gate per device for EVERY persistent kernel of the library - kernels whose workgroups wait for each other inside the
// launch (trd_resident_kernel: every CU, exclusively; varimax_persistent_kernel: one CU per workgroup of its grid).  Two such
// grids in flight on one device - the surrogate lanes of rule_n / bootstrapping run on their own streams and threads - can
// each hold a part of the CUs and wait for the rest: bounded spins end it, but only after ~0.2 s and a rerun.  A launch
// claims the CUs its grid needs and waits on the HOST for its turn; the claim is given back when the stream has finished the
// kernel.  Ordinary kernels are not counted: they always finish, so a persistent grid only waits for them.
struct PersistGate {
  std::mutex mu;
  std::condition_variable cv;
  int n_cus = 256;
  int used = 0;
  void acquire(int cus) {
    std::unique_lock<std::mutex> lk(mu);
    if (cus > n_cus) cus = n_cus;
    cv.wait(lk, [&] { return used + cus <= n_cus; });
    used += cus;
  }
  void release(int cus) {
    { std::lock_guard<std::mutex> lk(mu); used -= cus > n_cus ? n_cus : cus; }
    cv.notify_all();
  }
  struct Claim {                       // RAII: acquire on construction (cus > 0), release on destruction / release()
    PersistGate* g; int cus;
    Claim(PersistGate& gate, int c) : g(&gate), cus(c) { if (cus > 0) g->acquire(cus); }
    void release() { if (cus > 0) { g->release(cus); cus = 0; } }
    ~Claim() { release(); }
    Claim(const Claim&) = delete;
    Claim& operator=(const Claim&) = delete;
  };
};
// persistent launches of this process that ran out of their bounded spins (a workgroup never became resident) and were
// repeated on the launch-per-step path: 0 unless another process holds CUs (xmca_persistent_giveups)
inline std::atomic<long long>& persist_giveups() {
  static std::atomic<long long> n{0};
  return n;
}
inline PersistGate& persist_gate() {   // of the calling thread's current device
  static std::mutex mu;
  static std::map<int, PersistGate*>* gates = new std::map<int, PersistGate*>;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  auto it = gates->find(dev);
  if (it != gates->end()) return *it->second;
  PersistGate* g = new PersistGate;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) g->n_cus = n;
  (*gates)[dev] = g;
  return *g;
}

// true on a host thread that runs one of SEVERAL surrogate lanes (rule_n / bootstrapping: xmca_hip.cpp run_lanes).  Lanes keep
// to one stream each: with a second stream per lane (the eigensolver's compact-WY factors, tridiag_vec.h) the persistent
// reductions of four lanes ran out of their bounded spins now and then (measured: give-ups in 2 of 6 processes of the gate
// test against 0 of 6 - eight streams of one process on one device, plus the exclusive grids, is more than its hardware queues
// take without time-slicing).
inline bool& in_surrogate_lanes() {
  thread_local bool v = false;
  return v;
}

// XMCA_TRACE=jacobi,solve,rot (any subset, or "all"): progress lines of the eigensolver sweeps / the solver's route and
// consistency decisions / the rotation loop on stderr.  One switch instead of three.
static inline bool xmca_trace(const char* what) {
  const char* e = std::getenv("XMCA_TRACE");
  if (!e || !e[0]) return false;
  const std::string s(e);
  return s.find("all") != std::string::npos || s.find(what) != std::string::npos;
}

static inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace xmca
