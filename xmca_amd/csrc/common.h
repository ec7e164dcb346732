// Common host/device helpers for the xmca_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/xmca_hip.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
// blocks parked in ALL pools (sibling lanes, the parent handle, other handles) are given back before giving up - hipFree
// synchronises the device, so a block another stream released earlier is safe to free here.  The lane pools of
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
  DevPool() {
    std::lock_guard<std::mutex> g(registry().mu);
    registry().pools.push_back(this);
  }
  static void trim_all() {
    std::lock_guard<std::mutex> g(registry().mu);
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

static inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace xmca
