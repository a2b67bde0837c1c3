// Shared host-side helpers for libnncf_hip.so (error state, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/nncf_hip.h"

namespace nncf {

// Thread-local message behind nncf_last_error().
void set_error(const char* fmt, ...);
int fail(int status, const char* fmt, ...);

#define NNCF_HIP_CHECK(expr)                                                                    \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return ::nncf::fail(NNCF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                          __FILE__, __LINE__);                                                  \
  } while (0)

#define NNCF_REQUIRE(cond, ...)                                      \
  do {                                                               \
    if (!(cond)) return ::nncf::fail(NNCF_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// Affine map x -> a*x + c over Z/2^64 (one or many LCG steps composed).
struct Affine {
  uint64_t a, c;
};
__host__ __device__ inline Affine compose(Affine f, Affine g) {  // g after f
  return Affine{g.a * f.a, g.a * f.c + g.c};
}
__host__ __device__ inline uint64_t apply(Affine f, uint64_t x) { return f.a * x + f.c; }

constexpr uint64_t kLcgA = 25214903917ULL;  // nodesampler.cpp:24
constexpr uint64_t kLcgC = 11ULL;
constexpr int64_t kTableSize = 100000000LL;  // nodesampler.cpp:14

// f^n for the LCG step f (O(log n)).
inline Affine lcg_power(uint64_t n) {
  Affine result{1, 0}, base{kLcgA, kLcgC};
  while (n) {
    if (n & 1) result = compose(result, base);
    base = compose(base, base);
    n >>= 1;
  }
  return result;
}

// Library-wide measurement hook (nncf_profile_begin / _end): while a kernel name is selected,
// every launch site of that kernel brackets the launch with HIP events on the launch stream.
hipEvent_t* gprof_start(const char* kernel, hipStream_t stream);  // NULL when not selected / full
void gprof_stop(hipEvent_t* stop, hipStream_t stream);
struct ProfScope {
  hipEvent_t* stop;
  hipStream_t stream;
  ProfScope(const char* kernel, hipStream_t s) : stop(gprof_start(kernel, s)), stream(s) {}
  ~ProfScope() { gprof_stop(stop, stream); }
};

// Makes `device` current for the scope and restores the caller's device on every exit path.
struct DeviceGuard {
  int prev = -1;
  hipError_t status;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    status = hipSetDevice(device);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: a flag per process would leave a second device of the
// same process without the opt-in.  One of these per kernel instantiation (function-static at the launch site).
constexpr int kMaxDevices = 64;
struct KernelDeviceState {
  bool attr_set[kMaxDevices] = {};
  int capacity[kMaxDevices];
  KernelDeviceState() { for (int i = 0; i < kMaxDevices; ++i) capacity[i] = -1; }
};
inline int ensure_dynamic_lds(const void* kern, size_t lds, KernelDeviceState& st) {
  int dev = 0;
  NNCF_HIP_CHECK(hipGetDevice(&dev));
  NNCF_REQUIRE(dev >= 0 && dev < kMaxDevices, "device index %d beyond %d", dev, kMaxDevices);
  if (!st.attr_set[dev]) {
    NNCF_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    st.attr_set[dev] = true;
  }
  return NNCF_OK;
}

}  // namespace nncf
