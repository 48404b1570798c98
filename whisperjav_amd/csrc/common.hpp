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
  int64_t prof_units = 1;   // work units of the launches being profiled (windows of the batch), set by the callers
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

// ---- fp16 as _Float16 (a distinct 2-byte type, so templates can tell it from bf16_t) -------
typedef _Float16 f16_t;

__host__ __device__ inline float h2f(f16_t v) { return static_cast<float>(v); }
__host__ __device__ inline f16_t f2h(float f) { return static_cast<f16_t>(f); }   // RNE (v_cvt_f16_f32)

inline bool is16(int dtype) { return dtype == WJ_BF16 || dtype == WJ_F16; }
inline size_t dtype_size(int dtype) { return dtype == WJ_F32 ? 4 : 2; }

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

template <> struct Elem<f16_t> {
  static constexpr int dtype = WJ_F16;
  __device__ static inline float ld(const f16_t* p) { return h2f(*p); }
  __device__ static inline void st(f16_t* p, float v) { *p = f2h(v); }
};

// value of x after a round trip through the 16-bit storage type T (what a producing epilogue would have stored)
template <typename T> __device__ inline float round_T(float x);
template <> __device__ inline float round_T<bf16_t>(float x) { return bf2f(f2bf(x)); }
template <> __device__ inline float round_T<f16_t>(float x) { return h2f(f2h(x)); }
template <> __device__ inline float round_T<float>(float x) { return x; }

// two floats -> one dword holding (lo, hi) in storage type T
template <typename T> __device__ inline uint32_t pack2(float lo, float hi);
template <> __device__ inline uint32_t pack2<bf16_t>(float lo, float hi) {
  return static_cast<uint32_t>(f2bf(lo)) | (static_cast<uint32_t>(f2bf(hi)) << 16);
}
template <> __device__ inline uint32_t pack2<f16_t>(float lo, float hi) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  h2_t v;
  v[0] = f2h(lo);
  v[1] = f2h(hi);
  return __builtin_bit_cast(uint32_t, v);
}
template <typename T> __device__ inline void unpack2(uint32_t w, float& lo, float& hi);
template <> __device__ inline void unpack2<bf16_t>(uint32_t w, float& lo, float& hi) {
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xFFFF0000u);
}
template <> __device__ inline void unpack2<f16_t>(uint32_t w, float& lo, float& hi) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t v = __builtin_bit_cast(h2_t, w);
  lo = h2f(v[0]);
  hi = h2f(v[1]);
}

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
__device__ inline void st4(f16_t* p, const float v[4]) {
  uint2 u;
  u.x = pack2<f16_t>(v[0], v[1]);
  u.y = pack2<f16_t>(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}
// split store: hi = T(v) at p, lo = T(v - hi) at p + lo_off (4 consecutive elements each)
template <typename T> __device__ inline void st4_split(T* p, int64_t lo_off, const float v[4]) {
  float r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = v[i] - round_T<T>(v[i]);
  st4(p, v);
  st4(p + lo_off, r);
}
template <typename T> __device__ inline void st_split(T* p, int64_t lo_off, float v) {
  Elem<T>::st(p, v);
  Elem<T>::st(p + lo_off, v - round_T<T>(v));
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

__device__ inline void ld8(const f16_t* p, float v[8]) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) unpack2<f16_t>(w[i], v[2 * i], v[2 * i + 1]);
}

// ---- 16x16x32 MFMA for both 16-bit operand types (fp32 accumulate) -----------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> { typedef bf16x8_t type; };
template <> struct Vec8<f16_t> { typedef f16x8_t type; };

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16(f16x8_t a, f16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ typename Vec8<T>::type as_vec8(uint4 v) {
  return __builtin_bit_cast(typename Vec8<T>::type, v);
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

// hipFuncSetAttribute applies to the CURRENT device: a "done once" flag per kernel has to remember which devices it was
// done on, or a process that opens contexts on two devices launches on the second one without the attribute.
struct AttrOnce {
  uint64_t mask = 0, cur = 0;
  bool need() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    cur = 1ull << (dev & 63);
    return !(mask & cur);
  }
  void done() { mask |= cur; }
};

}  // namespace wj
