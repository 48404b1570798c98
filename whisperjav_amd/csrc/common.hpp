// common.hpp -- shared host/device helpers for libwjhip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/wjhip.h"

namespace wj {

// ---- error plumbing (thread-local message, no exceptions across the C ABI) ----------------
void set_error(const char* fmt, ...);
const char* get_error();

#define WJ_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::wj::set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__,                \
                      hipGetErrorString(_e));                                             \
      return WJ_E_HIP;                                                                    \
    }                                                                                     \
  } while (0)

#define WJ_REQUIRE(cond, ...)                                                             \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      ::wj::set_error(__VA_ARGS__);                                                       \
      return WJ_E_INVALID;                                                                \
    }                                                                                     \
  } while (0)

#define WJ_LAUNCH_CHECK()                                                                 \
  do {                                                                                    \
    hipError_t _e = hipGetLastError();                                                    \
    if (_e != hipSuccess) {                                                               \
      ::wj::set_error("kernel launch failed at %s:%d: %s", __FILE__, __LINE__,            \
                      hipGetErrorString(_e));                                             \
      return WJ_E_HIP;                                                                    \
    }                                                                                     \
  } while (0)

}  // namespace wj

struct wj_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int cu_count = 0;
  // grow-only scratch owned by the context (log-mel intermediates, test entry points)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int ensure_scratch(size_t bytes);
  hipStream_t pick(void* s) const { return s ? reinterpret_cast<hipStream_t>(s) : stream; }
  // event-pair profiler (wj_profile_start/stop): per-tag kernel time measured with HIP events on the
  // stream the kernels are launched on
  struct wj_profiler* prof = nullptr;
};

namespace wj {

// ---- bf16 as raw uint16 (RNE conversion; finite inputs only on this path) -----------------
typedef uint16_t bf16_t;

__host__ __device__ inline float bf2f(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = static_cast<uint32_t>(v) << 16;
  return c.f;
}
__host__ __device__ inline bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  u += 0x7FFFu + ((u >> 16) & 1u);
  return static_cast<bf16_t>(u >> 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int dtype = WJ_F32;
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int dtype = WJ_BF16;
  __device__ static inline float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static inline void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// store 4 consecutive elements (address must be aligned to 4 elements)
__device__ inline void st4(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ inline void st4(bf16_t* p, const float v[4]) {
  uint2 u;
  u.x = static_cast<uint32_t>(f2bf(v[0])) | (static_cast<uint32_t>(f2bf(v[1])) << 16);
  u.y = static_cast<uint32_t>(f2bf(v[2])) | (static_cast<uint32_t>(f2bf(v[3])) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}
// load 8 consecutive elements as floats (address aligned to 8 elements)
__device__ inline void ld8(const float* p, float v[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ inline void ld8(const bf16_t* p, float v[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}

__device__ inline float gelu_exact(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// bf16 compute type: erfc by Abramowitz-Stegun 7.1.26 (|abs err| < 5e-7 on the GELU, far inside the bf16
// rounding of the result) -- 1 rcp + 1 exp + 7 FMA instead of libm erff's ~40-instruction branchy path, which
// cost the fc1 epilogue as much as a third of its main loop.  The float32 compute type keeps gelu_exact.
__device__ inline float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float erfc = p * t * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
  return 0.5f * x * (x >= 0.f ? 2.0f - erfc : erfc);
}
template <typename T> __device__ inline float gelu_for(float x) {
  if constexpr (sizeof(T) == 2) return gelu_fast(x);
  else return gelu_exact(x);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace wj
