// norm.hip -- LayerNorm, layout/convert helpers, decoder embedding, decode-state bookkeeping.
// All HBM-bound elementwise work: one wavefront per row, 64-lane coalesced strides.
#include "kernels.hpp"

namespace wj {

// One wave per row; two-pass (mean, then centred variance) in registers like torch's CPU kernel.
// Every load (x, gamma, beta) is issued up front from a clamped, always-valid column index and the
// VALUE is masked: predicated loads compile to a branch + s_waitcnt per element, which serialises
// ~20 L2 round trips per row (measured 10 us per launch for 16 rows before this change).
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, T* __restrict__ out, int M,
                                                        int D, int split) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * D;
  float v[MAXV], g[MAXV], be[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = min(lane + i * 64, D - 1);
    v[i] = xr[c];
    g[i] = w[c];
    be[i] = b[c];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) s += (lane + i * 64 < D) ? v[i] : 0.f;
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const float d = (lane + i * 64 < D) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
  T* o = out + (int64_t)row * D * (split ? 2 : 1);     // split: [hi(D) | lo(D)] per row (GemmArgs::split)
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < D) {
      const float y = (v[i] - mean) * rstd * g[i] + be[i];
      if constexpr (sizeof(T) == 2) {
        if (split) { st_split<T>(o + c, D, y); continue; }
      }
      Elem<T>::st(o + c, y);
    }
  }
}


// Vectorised form for D % 256 == 0 (the encoder widths: 1280 = 5 x 256): one wave per row, lane owns columns
// 256 i + 4 lane .. + 3, so x / gamma / beta are 16-byte loads and the output an 8-byte store (the scalar kernel above moves
// 4 + 2 bytes per lane and instruction: 1.7 TB/s on the 144 000 x 1280 encoder rows).  blk != 0: the output goes out in the
// blocked GEMM-operand layout [ceil(M / 256)][D / 32][256][32] (GemmArgs::blk) -- a lane's 4 columns never straddle a
// 32-column block.  Same two-pass statistics; the summation order differs from the scalar kernel's (different partial sums).
template <typename T, int NV4, bool RESID>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(float* __restrict__ x, const float* __restrict__ partial, int ksplit,
                                                            const float* __restrict__ bias, const float* __restrict__ w,
                                                            const float* __restrict__ b, T* __restrict__ out, int M, int blk, int split) {
  constexpr int D = NV4 * 256;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float4* xr = reinterpret_cast<float4*>(x + (int64_t)row * D);
  float4 v[NV4], g[NV4], be[NV4];
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    v[i] = xr[i * 64 + lane];
    g[i] = reinterpret_cast<const float4*>(w)[i * 64 + lane];
    be[i] = reinterpret_cast<const float4*>(b)[i * 64 + lane];
  }
  if constexpr (RESID) {      // x += bias + sum_s partial[s] in slice order (deterministic), written back: the split-K consumer
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const float4 bb = reinterpret_cast<const float4*>(bias)[i * 64 + lane];
      v[i].x += bb.x; v[i].y += bb.y; v[i].z += bb.z; v[i].w += bb.w;
    }
    for (int sidx = 0; sidx < ksplit; ++sidx) {
      const float4* pr = reinterpret_cast<const float4*>(partial + ((int64_t)sidx * M + row) * D);
#pragma unroll
      for (int i = 0; i < NV4; ++i) {
        const float4 p = pr[i * 64 + lane];
        v[i].x += p.x; v[i].y += p.y; v[i].z += p.z; v[i].w += p.w;
      }
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) xr[i * 64 + lane] = v[i];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const float d0 = v[i].x - mean, d1 = v[i].y - mean, d2 = v[i].z - mean, d3 = v[i].w - mean;
    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    const int c = i * 256 + lane * 4;
    float y[4] = {(v[i].x - mean) * rstd * g[i].x + be[i].x, (v[i].y - mean) * rstd * g[i].y + be[i].y,
                  (v[i].z - mean) * rstd * g[i].z + be[i].z, (v[i].w - mean) * rstd * g[i].w + be[i].w};
    if (split) {             // [hi(D) | lo(D)] rows (GemmArgs::split)
      st4_split<T>(out + (int64_t)row * 2 * D + c, D, y);
      continue;
    }
    const int64_t off = blk ? (((int64_t)(row >> 8) * (D >> 5) + (c >> 5)) << 13) + ((row & 255) << 5) + (c & 31)
                            : (int64_t)row * D + c;
    st4(out + off, y);
  }
}

template <typename T, bool RESID>
static bool launch_ln_vec(float* x, const float* partial, int ksplit, const float* bias, const float* w, const float* b, T* out, int M,
                          int D, int blk, int split, hipStream_t s) {
  dim3 grid(ceil_div(M, 4));
#define WJ_LNV(NV) hipLaunchKernelGGL((layernorm_vec_kernel<T, NV, RESID>), grid, dim3(256), 0, s, x, partial, ksplit, bias, w, b, out, M, blk, split)
  switch (D) {
    case 256: WJ_LNV(1); return true;
    case 512: WJ_LNV(2); return true;
    case 768: WJ_LNV(3); return true;
    case 1024: WJ_LNV(4); return true;
    case 1280: WJ_LNV(5); return true;
    default: return false;
  }
#undef WJ_LNV
}

int g_ln_vec = 1;   // wj_tune("ln_vec"): 0 = always the scalar kernel (A/B)

int launch_layernorm(int dtype, const float* x, const float* w, const float* b, void* out, int M, int D,
                     hipStream_t s, int split, int blk) {
  if (D > 64 * 20 || D <= 0) { set_error("layernorm: D=%d unsupported (max 1280)", D); return WJ_E_INVALID; }
  if (M <= 0) return WJ_OK;
  if (blk && (split || !is16(dtype) || (D % 256))) { set_error("layernorm: a blocked output needs a 16-bit type, D %% 256 == 0, no split"); return WJ_E_INVALID; }
  if ((blk || g_ln_vec) && is16(dtype) && (D % 256) == 0) {
    float* xm = const_cast<float*>(x);      // not written without RESID
    const bool ok = dtype == WJ_F16 ? launch_ln_vec<f16_t, false>(xm, nullptr, 0, nullptr, w, b, (f16_t*)out, M, D, blk, split, s)
                                    : launch_ln_vec<bf16_t, false>(xm, nullptr, 0, nullptr, w, b, (bf16_t*)out, M, D, blk, split, s);
    if (ok) { WJ_LAUNCH_CHECK(); return WJ_OK; }
    if (blk) { set_error("layernorm: no blocked-output kernel for D=%d", D); return WJ_E_INVALID; }
  }
  dim3 grid(ceil_div(M, 4));
#define WJ_LN(NV)                                                                                          \
  do {                                                                                                     \
    if (dtype == WJ_F32)                                                                                   \
      hipLaunchKernelGGL((layernorm_kernel<float, NV>), grid, dim3(256), 0, s, x, w, b, (float*)out, M, D, 0); \
    else if (dtype == WJ_F16)                                                                              \
      hipLaunchKernelGGL((layernorm_kernel<f16_t, NV>), grid, dim3(256), 0, s, x, w, b, (f16_t*)out, M, D, split); \
    else                                                                                                   \
      hipLaunchKernelGGL((layernorm_kernel<bf16_t, NV>), grid, dim3(256), 0, s, x, w, b, (bf16_t*)out, M, D, split); \
  } while (0)
  if (D <= 64 * 6) WJ_LN(6);
  else if (D <= 64 * 12) WJ_LN(12);
  else WJ_LN(20);
#undef WJ_LN
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_resid_kernel(float* __restrict__ x, const float* __restrict__ partial,
                                                              int ksplit, const float* __restrict__ bias,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              T* __restrict__ out, int M, int D, int split) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float* xr = x + (int64_t)row * D;
  float v[MAXV], g[MAXV], be[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = min(lane + i * 64, D - 1);
    v[i] = xr[c] + bias[c];
    g[i] = w[c];
    be[i] = b[c];
  }
  for (int s = 0; s < ksplit; ++s) {
    const float* pr = partial + ((int64_t)s * M + row) * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) v[i] += pr[min(lane + i * 64, D - 1)];
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < D) xr[c] = v[i];
    sum += c < D ? v[i] : 0.f;
  }
  const float mean = wave_sum(sum) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const float d = (lane + i * 64 < D) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
  T* o = out + (int64_t)row * D * (split ? 2 : 1);     // split: [hi(D) | lo(D)] per row (GemmArgs::split)
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < D) {
      const float y = (v[i] - mean) * rstd * g[i] + be[i];
      if constexpr (sizeof(T) == 2) {
        if (split) { st_split<T>(o + c, D, y); continue; }
      }
      Elem<T>::st(o + c, y);
    }
  }
}

int launch_layernorm_resid(int dtype, float* x, const float* partial, int ksplit, const float* bias, const float* w,
                           const float* b, void* out, int M, int D, hipStream_t s, int split) {
  if (D > 64 * 20 || D <= 0) { set_error("layernorm: D=%d unsupported (max 1280)", D); return WJ_E_INVALID; }
  if (M <= 0) return WJ_OK;
  if (g_ln_vec && is16(dtype) && (D % 256) == 0) {      // 16-byte loads of x / slabs / gamma / beta, 8-byte stores (round 4)
    const bool ok = dtype == WJ_F16 ? launch_ln_vec<f16_t, true>(x, partial, ksplit, bias, w, b, (f16_t*)out, M, D, 0, split, s)
                                    : launch_ln_vec<bf16_t, true>(x, partial, ksplit, bias, w, b, (bf16_t*)out, M, D, 0, split, s);
    if (ok) { WJ_LAUNCH_CHECK(); return WJ_OK; }
  }
  dim3 grid(ceil_div(M, 4));
#define WJ_LNR(NV)                                                                                              \
  do {                                                                                                          \
    if (dtype == WJ_F32)                                                                                        \
      hipLaunchKernelGGL((layernorm_resid_kernel<float, NV>), grid, dim3(256), 0, s, x, partial, ksplit, bias, w, b, \
                         (float*)out, M, D, 0);                                                                 \
    else if (dtype == WJ_F16)                                                                                   \
      hipLaunchKernelGGL((layernorm_resid_kernel<f16_t, NV>), grid, dim3(256), 0, s, x, partial, ksplit, bias, w, b, \
                         (f16_t*)out, M, D, split);                                                             \
    else                                                                                                        \
      hipLaunchKernelGGL((layernorm_resid_kernel<bf16_t, NV>), grid, dim3(256), 0, s, x, partial, ksplit, bias, w, b, \
                         (bf16_t*)out, M, D, split);                                                            \
  } while (0)
  if (D <= 64 * 6) WJ_LNR(6);
  else if (D <= 64 * 12) WJ_LNR(12);
  else WJ_LNR(20);
#undef WJ_LNR
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

// mel f32 [B][C][F] -> rows T [B][F+2][C] (pad rows untouched = zero); 32x32 LDS transpose tiles
template <typename T>
__global__ __launch_bounds__(256) void mel_to_rows_kernel(const float* __restrict__ mel, T* __restrict__ out, int C,
                                                          int F) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int f0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, f = f0 + tx;
    tile[r][tx] = (c < C && f < F) ? mel[((int64_t)b * C + c) * F + f] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int f = f0 + r, c = c0 + tx;
    if (c < C && f < F) Elem<T>::st(out + ((int64_t)b * (F + 2) + f + 1) * C + c, tile[tx][r]);
  }
}

int launch_mel_to_rows(int dtype, const float* mel, void* out, int B, int n_mels, int frames, hipStream_t s) {
  dim3 grid(ceil_div(frames, 32), ceil_div(n_mels, 32), B);
  if (dtype == WJ_F32)
    hipLaunchKernelGGL(mel_to_rows_kernel<float>, grid, dim3(256), 0, s, mel, (float*)out, n_mels, frames);
  else if (dtype == WJ_F16)
    hipLaunchKernelGGL(mel_to_rows_kernel<f16_t>, grid, dim3(256), 0, s, mel, (f16_t*)out, n_mels, frames);
  else
    hipLaunchKernelGGL(mel_to_rows_kernel<bf16_t>, grid, dim3(256), 0, s, mel, (bf16_t*)out, n_mels, frames);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const T* __restrict__ emb, const float* __restrict__ pos_emb,
                                                    const int32_t* __restrict__ tokens, int64_t tok_stride,
                                                    const int* __restrict__ pos_ptr, float* __restrict__ x, int D) {
  const int r = blockIdx.x;
  const int pos = *pos_ptr;
  const int tok = tokens[(int64_t)r * tok_stride + pos];
  const T* e = emb + (int64_t)tok * D;
  const float* p = pos_emb + (int64_t)pos * D;
  for (int c = threadIdx.x; c < D; c += 256) x[(int64_t)r * D + c] = Elem<T>::ld(e + c) + p[c];
}

// full-sequence pass: row r = window * Tp + t embeds token t of the window's history at position t
template <typename T>
__global__ __launch_bounds__(256) void embed_seq_kernel(const T* __restrict__ emb, const float* __restrict__ pos_emb,
                                                        const int32_t* __restrict__ tokens, int64_t tok_stride, int Tp,
                                                        float* __restrict__ x, int D) {
  const int r = blockIdx.x, w = r / Tp, t = r % Tp;
  const int tok = tokens[(int64_t)w * tok_stride + t];
  const T* e = emb + (int64_t)tok * D;
  const float* p = pos_emb + (int64_t)t * D;
  for (int c = threadIdx.x; c < D; c += 256) x[(int64_t)r * D + c] = Elem<T>::ld(e + c) + p[c];
}

int launch_embed_seq(int dtype, const void* tok_emb, const float* pos_emb, const int32_t* tokens, int64_t tok_stride,
                     int Tp, float* x, int B, int D, hipStream_t s) {
  if (dtype == WJ_F32)
    hipLaunchKernelGGL(embed_seq_kernel<float>, dim3(B * Tp), dim3(256), 0, s, (const float*)tok_emb, pos_emb, tokens,
                       tok_stride, Tp, x, D);
  else if (dtype == WJ_F16)
    hipLaunchKernelGGL(embed_seq_kernel<f16_t>, dim3(B * Tp), dim3(256), 0, s, (const f16_t*)tok_emb, pos_emb, tokens,
                       tok_stride, Tp, x, D);
  else
    hipLaunchKernelGGL(embed_seq_kernel<bf16_t>, dim3(B * Tp), dim3(256), 0, s, (const bf16_t*)tok_emb, pos_emb, tokens,
                       tok_stride, Tp, x, D);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

int launch_embed(int dtype, const void* tok_emb, const float* pos_emb, const int32_t* tokens, int64_t tok_stride,
                 const int* pos_ptr, float* x, int R, int D, hipStream_t s) {
  if (dtype == WJ_F32)
    hipLaunchKernelGGL(embed_kernel<float>, dim3(R), dim3(256), 0, s, (const float*)tok_emb, pos_emb, tokens,
                       tok_stride, pos_ptr, x, D);
  else if (dtype == WJ_F16)
    hipLaunchKernelGGL(embed_kernel<f16_t>, dim3(R), dim3(256), 0, s, (const f16_t*)tok_emb, pos_emb, tokens,
                       tok_stride, pos_ptr, x, D);
  else
    hipLaunchKernelGGL(embed_kernel<bf16_t>, dim3(R), dim3(256), 0, s, (const bf16_t*)tok_emb, pos_emb, tokens,
                       tok_stride, pos_ptr, x, D);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void f32_to_T_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    Elem<T>::st(out + i, in[i]);
}

int launch_f32_to_T(int dtype, const float* in, void* out, int64_t n, hipStream_t s) {
  if (n <= 0) return WJ_OK;
  const int blocks = (int)min((int64_t)4096, ceil_div64(n, 256));
  if (dtype == WJ_F32)
    hipLaunchKernelGGL(f32_to_T_kernel<float>, dim3(blocks), dim3(256), 0, s, in, (float*)out, n);
  else if (dtype == WJ_F16)
    hipLaunchKernelGGL(f32_to_T_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, in, (f16_t*)out, n);
  else
    hipLaunchKernelGGL(f32_to_T_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, in, (bf16_t*)out, n);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

__global__ void advance_pos_kernel(int* pos) { if (threadIdx.x == 0) *pos += 1; }

int launch_advance_pos(int* pos_ptr, hipStream_t s) {
  hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(64), 0, s, pos_ptr);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

// Beam search re-binding: logical row r continues the history of old logical row parent[r].
// row_map[r][j] = physical KV-cache row that holds position j of r's history.
__global__ __launch_bounds__(256) void rebind_rows_kernel(const int32_t* __restrict__ old_map,
                                                          int32_t* __restrict__ new_map,
                                                          const int32_t* __restrict__ parent,
                                                          const int* __restrict__ pos_ptr, int stride) {
  const int r = blockIdx.x;
  const int pos = *pos_ptr;
  const int p = parent[r];
  for (int j = threadIdx.x; j < stride; j += 256) {
    int32_t v = r;
    if (j < pos) v = old_map[(int64_t)p * stride + j];
    new_map[(int64_t)r * stride + j] = v;
  }
}

// Compaction of a beam-search batch (finished windows leave it): new logical row r takes over old logical row src[r].
// Physical KV-cache rows never move -- the row map keeps pointing at them; from the current position on row r writes its
// own physical row r, which no live history references at those positions (a cell (row, position) is written once).
__global__ __launch_bounds__(256) void compact_rows_kernel(const int32_t* __restrict__ src_rows, const int32_t* __restrict__ old_map,
                                                           int32_t* __restrict__ new_map, int map_stride,
                                                           const int* __restrict__ pos_ptr, const int32_t* __restrict__ old_hist,
                                                           int32_t* __restrict__ new_hist, int64_t tok_stride,
                                                           const float* __restrict__ old_score, float* __restrict__ new_score) {
  const int r = blockIdx.x, src = src_rows[r];
  const int pos = *pos_ptr;
  for (int j = threadIdx.x; j < map_stride; j += 256)
    new_map[(int64_t)r * map_stride + j] = j < pos ? old_map[(int64_t)src * map_stride + j] : r;
  for (int64_t j = threadIdx.x; j < tok_stride; j += 256) new_hist[(int64_t)r * tok_stride + j] = old_hist[(int64_t)src * tok_stride + j];
  if (threadIdx.x == 0) new_score[r] = old_score[src];
}

int launch_compact_rows(const int32_t* src_rows, int R_new, const int32_t* old_map, int32_t* new_map, int map_stride,
                        const int* pos_ptr, const int32_t* old_hist, int32_t* new_hist, int64_t tok_stride,
                        const float* old_score, float* new_score, hipStream_t s) {
  hipLaunchKernelGGL(compact_rows_kernel, dim3(R_new), dim3(256), 0, s, src_rows, old_map, new_map, map_stride, pos_ptr, old_hist,
                     new_hist, tok_stride, old_score, new_score);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

int launch_rebind_rows(const int32_t* old_map, int32_t* new_map, const int32_t* parent, const int* pos_ptr, int R,
                       int stride, hipStream_t s) {
  hipLaunchKernelGGL(rebind_rows_kernel, dim3(R), dim3(256), 0, s, old_map, new_map, parent, pos_ptr, stride);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

}  // namespace wj
