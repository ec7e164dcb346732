// Common host/device helpers for the xmca_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/xmca_hip.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

// Owning device buffer (hipMalloc).  Grows on demand, never shrinks.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  T* ensure(size_t n) {
    if (n > cap) {
      release();
      XMCA_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
      cap = n;
    }
    return p;
  }
  T* get() const { return p; }
};

static inline int ceil_div(int64_t a, int64_t b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace xmca
