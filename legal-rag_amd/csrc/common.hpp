// Shared host-side plumbing for libamdretrieval (error strings, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>

#include "../../include/amdretrieval.h"

namespace amdr {

std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define AMDR_HIP(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return ::amdr::fail(_e == hipErrorOutOfMemory ? AMDR_ENOMEM : AMDR_EHIP, "%s failed: %s (%s:%d)", \
                          #expr, hipGetErrorString(_e), __FILE__, __LINE__);                    \
  } while (0)

#define AMDR_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return ::amdr::fail(AMDR_EINVAL, __VA_ARGS__); \
  } while (0)

// Growable device buffer owned by a handle (never shrinks).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return AMDR_OK;
    if (p) {
      AMDR_HIP(hipFree(p));
      p = nullptr;
      cap = 0;
    }
    AMDR_HIP(hipMalloc(&p, bytes));
    cap = bytes;
    return AMDR_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// top-k staging capacity: power of two, >= k + 64 (one wave of appends always fits)
inline int topk_cap(int k) { return next_pow2(k + 64) < 128 ? 128 : next_pow2(k + 64); }

int check_device(int device);

}  // namespace amdr
