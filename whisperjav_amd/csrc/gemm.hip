// gemm.hip -- C[M,N] = A[M,K] . W[N,K]^T with fused epilogues, gfx950.
//
// Three kernels:
//   * gemm_bf16_tile  : 128x128x64 tile, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, register-prefetched
//                       global->LDS staging with an XOR-swizzled LDS image, double-buffered LDS.
//                       Used for the encoder (M = batch*1500) and the cross-K/V projection.
//   * gemm_bf16_skinny: decode-step GEMM (M = hypotheses <= a few hundred).  HBM-bound on W: one
//                       workgroup owns 16 output columns, its 8 waves split K, W fragments go
//                       straight from HBM to VGPRs (streamed once), partials meet in LDS.
//   * gemm_f32        : exact-fp32 VALU kernel (fmaf chains, k ascending) for the float32 compute
//                       type that carries the 1e-3 log-prob parity bar.
//
// Both operands are K-contiguous ([M][K] activations, [N][K] = PyTorch Linear weights), so every
// MFMA fragment is one 16-byte load and no transposition is ever needed.  Accumulators use the
// "NM" orientation (mfma(W_frag, A_frag)): a lane ends up with 4 consecutive output columns of
// one row, i.e. one 8-byte (bf16) / 16-byte (fp32) store.  EPI_VT flips the operand order so a
// lane holds 4 consecutive rows instead (V is written transposed per head for the attention
// kernel's P.V operand).
#include <stdlib.h>

#include "kernels.hpp"

namespace wj {

// --------------------------------------------------------------------------------------------
// epilogues
// --------------------------------------------------------------------------------------------
template <int EPI, typename T>
__device__ __forceinline__ void epi_nm(const GemmArgs& g, int z, int m, int n, float v[4]) {
  // v[j] belongs to (row m, column n + j); n % 4 == 0; caller guarantees m < M and n < N.
  if constexpr (EPI == EPI_PARTIAL_F32) {   // z = K-slice index
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + ((int64_t)z * g.M + m) * g.ldc + n) =
        make_float4(v[0], v[1], v[2], v[3]);
    return;
  } else if constexpr (EPI == EPI_F32) {
    float* o = reinterpret_cast<float*>(g.out) + (int64_t)z * g.c_batch + (int64_t)m * g.ldc + n;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < g.N) o[j] = v[j] + (g.bias ? g.bias[n + j] : 0.0f);
    return;
  } else {
    if (g.bias) {
      const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if constexpr (EPI == EPI_T || EPI == EPI_GELU_T) {
      if constexpr (EPI == EPI_GELU_T) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gelu_for<T>(v[j]);
      }
      T* o = reinterpret_cast<T*>(g.out) + (int64_t)z * g.c_batch + (int64_t)m * g.ldc + n;
      if constexpr (sizeof(T) == 2) {
        if (g.split_out) { st4_split<T>(o, g.N, v); return; }
      }
      st4(o, v);
    } else if constexpr (EPI == EPI_RESID_F32) {
      float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (int64_t)z * g.c_batch +
                                            (int64_t)m * g.ldc + n);
      float4 x = *o;
      x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
      *o = x;
    } else if constexpr (EPI == EPI_GELU_POS_F32) {
      const float4 p = *reinterpret_cast<const float4*>(g.pos + (int64_t)m * g.N + n);
      float4 x;
      x.x = gelu_for<T>(v[0]) + p.x; x.y = gelu_for<T>(v[1]) + p.y;
      x.z = gelu_for<T>(v[2]) + p.z; x.w = gelu_for<T>(v[3]) + p.w;
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (int64_t)z * g.c_batch +
                                 (int64_t)m * g.ldc + n) = x;
    } else if constexpr (EPI == EPI_QK_HEADS || EPI == EPI_CKV) {
      const int which = n >= g.D;
      const int nn = n - which * g.D;
      const int h = nn >> 6, dd = nn & 63;
      T* base = reinterpret_cast<T*>(which ? g.out2 : g.out);
      st4(base + (((int64_t)z * g.H + h) * g.Tpad + m) * 64 + dd, v);
    } else if constexpr (EPI == EPI_QKV_DEC) {
      const int which = n / g.D;
      const int nn = n - which * g.D;
      if (which == 0) {
        st4(reinterpret_cast<T*>(g.out) + (int64_t)m * g.D + nn, v);
      } else {
        const int h = nn >> 6, dd = nn & 63;
        const int pos = g.seq_tp ? m % g.seq_tp : *g.pos_ptr;
        const int64_t crow = g.seq_tp ? m / g.seq_tp : m;
        T* base = reinterpret_cast<T*>(which == 1 ? g.out2 : g.out3);
        st4(base + ((crow * g.H + h) * g.cache_len + pos) * 64 + dd, v);
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void epi_vt(const GemmArgs& g, int z, int m, int n, float v[4]) {
  // v[i] belongs to (row m + i, column n); m % 4 == 0; m < M (M % 4 == 0), n < N.
  const float b = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += b;
  const int h = n >> 6, dd = n & 63;
  st4(reinterpret_cast<T*>(g.out) + (((int64_t)z * g.H + h) * 64 + dd) * g.Tpad + m, v);
}

// --------------------------------------------------------------------------------------------
// bf16 MFMA, 128x128x64 tile
// --------------------------------------------------------------------------------------------
constexpr int TBM = 128, TBN = 128, TBK = 64;

// Predicated 16-byte load: the address is always valid (callers clamp it) and the VALUE is
// selected.  Writing `p ? *ptr : zero` instead lets the compiler select between the global pointer
// and a stack slot, which turns every load into a flat_load plus a scratch store.
__device__ __forceinline__ uint4 ldg16_pred(const void* p, bool pred) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  v.x = pred ? v.x : 0u; v.y = pred ? v.y : 0u; v.z = pred ? v.z : 0u; v.w = pred ? v.w : 0u;
  return v;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * TBK + ((chunk ^ (row & 7)) << 3); }

template <typename T, int EPI, bool GLDS>
__global__ __launch_bounds__(256) void gemm_h_tile_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 2 * TBM * TBK];  // [buf][A|W][128][64] = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z;
  // XCD-aware tile order: workgroup b is dispatched to XCD b % 8 (each XCD has its own 4 MiB L2).
  // Remap so every XCD walks a CONTIGUOUS run of tiles in m-major order: the n-tiles that share one
  // A row-panel then hit the same L2 instead of fetching the panel once per XCD (PMC before the remap:
  // ~5x the algorithmic bytes on the fc2 GEMM).  Bijective for any tile count.
  const int nx = gridDim.x, ntiles = gridDim.x * gridDim.y;
  const int lin = blockIdx.y * nx + blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = lin & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  // inside the run, walk groups of GM row-panels column by column: the ~64 tiles an XCD has in
  // flight then form an 8 x 8 patch and re-use both the A and the W panels ~8x from L2
  constexpr int GM = 8;
  const int per_group = GM * nx, group = tile / per_group, first_m = group * GM;
  const int gsz = min(GM, (int)gridDim.y - first_m), in_group = tile - group * per_group;
  const int m0 = (first_m + in_group % gsz) * TBM, n0 = (in_group / gsz) * TBN;
  // split-K (EPI_PARTIAL_F32): blockIdx.z selects a K slice instead of a batch entry.  Split activations
  // (g.split, LDS-DMA path only): A rows are [hi | lo], the k loop runs over 2 K and W wraps at K.
  const int ktot = g.split ? 2 * g.K : g.K;
  const int kslice = (EPI == EPI_PARTIAL_F32) ? ktot / g.ksplit : ktot;
  const int64_t koff = (EPI == EPI_PARTIAL_F32) ? (int64_t)z * kslice : 0;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A) + ((EPI == EPI_PARTIAL_F32) ? koff : (int64_t)z * g.a_batch);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W) + (g.split ? 0 : koff);
  const int wwrap = g.split ? g.K : 0x7fffffff;      // W column of logical k (split): (koff + k) mod K
  const int wbase = g.split ? (int)koff : 0;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const int nk = (kslice + TBK - 1) / TBK;
  // staging registers are individually named (an indexed array ends up in scratch memory)
#define WJ_GLOAD1(i, k0)                                                                      \
  {                                                                                           \
    const int idx = tid + (i) * 256;                                                          \
    const int row = idx >> 3, ch = idx & 7;                                                   \
    const int k = (k0) + ch * 8;                                                              \
    const bool kin = k < kslice;                                                              \
    const int kc = kin ? k : 0;                                                               \
    ra##i = ldg16_pred(A + (int64_t)min(m0 + row, g.M - 1) * g.lda + kc, kin && m0 + row < g.M); \
    rb##i = ldg16_pred(W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + kc, kin && n0 + row < g.N); \
  }
#define WJ_GLOAD(k0) WJ_GLOAD1(0, k0) WJ_GLOAD1(1, k0) WJ_GLOAD1(2, k0) WJ_GLOAD1(3, k0)
#define WJ_SSTORE1(i, buf)                                                                    \
  {                                                                                           \
    const int idx = tid + (i) * 256;                                                          \
    const int row = idx >> 3, ch = idx & 7;                                                   \
    const int off = swz(row, ch);                                                             \
    *reinterpret_cast<uint4*>(&lds[((buf) * 2 + 0) * TBM * TBK + off]) = ra##i;               \
    *reinterpret_cast<uint4*>(&lds[((buf) * 2 + 1) * TBM * TBK + off]) = rb##i;               \
  }
#define WJ_SSTORE(buf) WJ_SSTORE1(0, buf) WJ_SSTORE1(1, buf) WJ_SSTORE1(2, buf) WJ_SSTORE1(3, buf)

  // GLDS: LDS-DMA staging (global_load_lds_dwordx4): each wave instruction deposits 64 x 16 B =
  // 8 tile rows, lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address (lane ->
  // physical chunk p = lane & 7 of row r0 + lane / 8 fetches logical chunk p ^ (row & 7)).  Requires
  // K % 64 == 0 (no zero fill on this path); out-of-range rows read a clamped (valid) row and are
  // dropped by the epilogue.
#define WJ_GLDS_STAGE(buf, k0)                                                                     \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
    const int r0 = (wave * 4 + q) * 8;                                                             \
    const int row = r0 + (lane >> 3);                                                              \
    const int c = (lane & 7) ^ (row & 7);                                                          \
    int wk = wbase + (k0);                                                                         \
    wk = wk >= wwrap ? wk - wwrap : wk;                                                            \
    const bf16_t* ga = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + (k0) + c * 8;                 \
    const bf16_t* gw = W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + wk + c * 8;                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,            \
        (__attribute__((address_space(3))) void*)(&lds[((buf) * 2 + 0) * TBM * TBK + r0 * TBK]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,            \
        (__attribute__((address_space(3))) void*)(&lds[((buf) * 2 + 1) * TBM * TBK + r0 * TBK]), 16, 0, 0); \
  }

  if constexpr (GLDS) {
    WJ_GLDS_STAGE(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    WJ_GLOAD(0)
    WJ_SSTORE(0)
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      if constexpr (GLDS) { WJ_GLDS_STAGE(cur ^ 1, (kt + 1) * TBK) } else { WJ_GLOAD((kt + 1) * TBK) }
    }
    const bf16_t* la = &lds[(cur * 2 + 0) * TBM * TBK];
    const bf16_t* lb = &lds[(cur * 2 + 1) * TBM * TBK];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Vec8<T>::type af[4], wf[4];
      const int ch = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&la[swz(row, ch)]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + (lane & 15);
        wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&lb[swz(row, ch)]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (EPI == EPI_VT)
            acc[i][j] = mfma16(af[i], wf[j], acc[i][j]);
          else
            acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        }
    }
    if constexpr (GLDS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of tile kt+1 has landed
    } else {
      if (kt + 1 < nk) { WJ_SSTORE(cur ^ 1) }
    }
    __syncthreads();
  }
#undef WJ_GLDS_STAGE
#undef WJ_GLOAD
#undef WJ_SSTORE
#undef WJ_GLOAD1
#undef WJ_SSTORE1

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if constexpr (EPI == EPI_VT) {
        const int m = m0 + wm * 64 + i * 16 + (lane >> 4) * 4;
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        if (m < g.M && n < g.N) epi_vt<T>(g, z, m, n, v);
      } else {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        if (m < g.M && n < g.N) epi_nm<EPI, T>(g, z, m, n, v);
      }
    }
}

// --------------------------------------------------------------------------------------------
// bf16 MFMA, 256x256x64 tile, 8 waves (2 x 4; a wave owns 128 x 64 of C in 128 accumulator registers).
// Same LDS image and LDS-DMA staging as the 128-tile kernel, but each operand byte read from LDS feeds
// 1.5x more MFMAs ((8 + 4) fragment reads for 32 MFMAs per k-half instead of (4 + 4) for 16) and the two
// waves that share a SIMD overlap one wave's ds_reads with the other's MFMAs.  2 x 64 KiB of LDS.
// Used for the big encoder GEMMs (N % 256 == 0, K % 64 == 0).
// --------------------------------------------------------------------------------------------
constexpr int BBM = 256, BBN = 256;

template <typename T, int EPI, int SCHED>
__global__ __launch_bounds__(512) void gemm_h_big_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_big[];   // [buf][A|W][256][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int z = blockIdx.z;
  const int nx = gridDim.x, ntiles = gridDim.x * gridDim.y;
  const int lin = blockIdx.y * nx + blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = lin & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  constexpr int GM = 8;
  const int per_group = GM * nx, group = tile / per_group, first_m = group * GM;
  const int gsz = min(GM, (int)gridDim.y - first_m), in_group = tile - group * per_group;
  const int m0 = (first_m + in_group % gsz) * BBM, n0 = (in_group / gsz) * BBN;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)z * g.a_batch;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / TBK;
  constexpr int STAGE = 2 * BBM * TBK;   // elements per buffer (A then W)
#define WJ_BIG_STAGE(buf, k0)                                                                      \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
    const int r0 = (wave * 4 + q) * 8;                                                             \
    const int row = r0 + (lane >> 3);                                                              \
    const int c = (lane & 7) ^ (row & 7);                                                          \
    const bf16_t* ga = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + (k0) + c * 8;                 \
    const bf16_t* gw = W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + (k0) + c * 8;                 \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,            \
        (__attribute__((address_space(3))) void*)(&lds_big[(buf) * STAGE + r0 * TBK]), 16, 0, 0);  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,            \
        (__attribute__((address_space(3))) void*)(&lds_big[(buf) * STAGE + BBM * TBK + r0 * TBK]), 16, 0, 0); \
  }

  WJ_BIG_STAGE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) { WJ_BIG_STAGE(cur ^ 1, (kt + 1) * TBK) }
    const bf16_t* la = &lds_big[cur * STAGE];
    const bf16_t* lb = la + BBM * TBK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Vec8<T>::type af[8], wf[4];
      const int ch = ks * 4 + (lane >> 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + (lane & 15);
        wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&lb[swz(row, ch)]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&la[swz(row, ch)]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (EPI == EPI_VT)
            acc[i][j] = mfma16(af[i], wf[j], acc[i][j]);
          else
            acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
        }
    }
    if constexpr (SCHED == 1) {
      // Issue-order hint for the whole k-step (both k-halves are one basic block): the first half's 12 fragment reads,
      // then the second half's reads trickle in between the first half's MFMAs (1 ds_read per 2 MFMAs) so the LDS
      // latency of half 2 hides under the matrix pipe, the 8 LDS-DMA requests of the next stage are spread over the
      // second half's MFMAs.  Scheduling only: the data flow is unchanged.
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);      // DS read
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read (LDS-DMA)
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of tile kt+1 has landed
    __syncthreads();
  }
#undef WJ_BIG_STAGE

  if constexpr (EPI == EPI_RESID_F32) {
    // x += acc + bias: all 16 residual loads of a half-tile are issued before the first add/store (written as
    // load-add-store per fragment this epilogue is a chain of 32 dependent memory round trips)
    const int nb = n0 + wn * 64 + (lane >> 4) * 4;
    float4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bias4[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + nb + j * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* xo = reinterpret_cast<float*>(g.out) + (int64_t)z * g.c_batch;
#pragma unroll
    for (int ih = 0; ih < 8; ih += 4) {
      float4 r[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + wm * 128 + (ih + i) * 16 + (lane & 15), g.M - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[i][j] = *reinterpret_cast<const float4*>(xo + (int64_t)m * g.ldc + nb + j * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + (ih + i) * 16 + (lane & 15);
        if (m < g.M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4_t a4 = acc[ih + i][j];
            float4 x = r[i][j];
            x.x += a4[0] + bias4[j].x; x.y += a4[1] + bias4[j].y; x.z += a4[2] + bias4[j].z; x.w += a4[3] + bias4[j].w;
            *reinterpret_cast<float4*>(xo + (int64_t)m * g.ldc + nb + j * 16) = x;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if constexpr (EPI == EPI_VT) {
        const int m = m0 + wm * 128 + i * 16 + (lane >> 4) * 4;
        const int n = n0 + wn * 64 + j * 16 + (lane & 15);
        if (m < g.M && n < g.N) epi_vt<T>(g, z, m, n, v);
      } else {
        const int m = m0 + wm * 128 + i * 16 + (lane & 15);
        const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        if (m < g.M && n < g.N) epi_nm<EPI, T>(g, z, m, n, v);
      }
    }
}

// --------------------------------------------------------------------------------------------
// bf16 MFMA, 128x128x64 tile with an NS-stage LDS-DMA pipeline (decode GEMMs, M = a few hundred rows).
// A decode GEMM gives a workgroup only 5-20 k-steps of 32 MFMAs per wave: with one stage of prefetch every
// k-step costs a full global->LDS round trip (~1 us), so the kernel is latency- not MFMA-bound.  Here NS-1
// stages are in flight: stage kt+NS-1 is requested right after the barrier that retires stage kt-1, and the
// wait before computing stage kt lets the younger stages stay outstanding (s_waitcnt vmcnt(8 * younger)).
// One barrier per k-step.  NS * 32 KiB of LDS.
// --------------------------------------------------------------------------------------------
template <typename T, int EPI, int NS>
__global__ __launch_bounds__(256) void gemm_h_tile_ms_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_ms[];   // [stage][A|W][128][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z;
  const int nx = gridDim.x, ntiles = gridDim.x * gridDim.y;
  const int lin = blockIdx.y * nx + blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = lin & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  constexpr int GM = 8;
  const int per_group = GM * nx, group = tile / per_group, first_m = group * GM;
  const int gsz = min(GM, (int)gridDim.y - first_m), in_group = tile - group * per_group;
  const int m0 = (first_m + in_group % gsz) * TBM, n0 = (in_group / gsz) * TBN;
  const int ktot = g.split ? 2 * g.K : g.K;
  const int kslice = (EPI == EPI_PARTIAL_F32) ? ktot / g.ksplit : ktot;
  const int64_t koff = (EPI == EPI_PARTIAL_F32) ? (int64_t)z * kslice : 0;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A) + ((EPI == EPI_PARTIAL_F32) ? koff : (int64_t)z * g.a_batch);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W) + (g.split ? 0 : koff);
  const int wwrap = g.split ? g.K : 0x7fffffff;
  const int wbase = g.split ? (int)koff : 0;

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = kslice / TBK;
  constexpr int STAGE = 2 * TBM * TBK;
#define WJ_MS_STAGE(buf, k0)                                                                       \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
    const int r0 = (wave * 4 + q) * 8;                                                             \
    const int row = r0 + (lane >> 3);                                                              \
    const int c = (lane & 7) ^ (row & 7);                                                          \
    int wk = wbase + (k0);                                                                         \
    wk = wk >= wwrap ? wk - wwrap : wk;                                                            \
    const bf16_t* ga = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + (k0) + c * 8;                 \
    const bf16_t* gw = W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + wk + c * 8;                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,            \
        (__attribute__((address_space(3))) void*)(&lds_ms[(buf) * STAGE + r0 * TBK]), 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,            \
        (__attribute__((address_space(3))) void*)(&lds_ms[(buf) * STAGE + TBM * TBK + r0 * TBK]), 16, 0, 0); \
  }

#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nk) { WJ_MS_STAGE(st, st * TBK) }

  for (int kt = 0; kt < nk; ++kt) {
    const int younger = min(NS - 2, nk - 1 - kt);     // stages requested after stage kt and still allowed in flight
    if (younger >= 4) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (younger == 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // stage kt visible to all; buffer of stage kt-1 is free
    if (kt + NS - 1 < nk) { WJ_MS_STAGE((kt + NS - 1) % NS, (kt + NS - 1) * TBK) }
    const bf16_t* la = &lds_ms[(kt % NS) * STAGE];
    const bf16_t* lb = la + TBM * TBK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename Vec8<T>::type af[4], wf[4];
      const int ch = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&la[swz(wm * 64 + i * 16 + (lane & 15), ch)]);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&lb[swz(wn * 64 + j * 16 + (lane & 15), ch)]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
    }
  }
#undef WJ_MS_STAGE

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const int m = m0 + wm * 64 + i * 16 + (lane & 15);
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (m < g.M && n < g.N) epi_nm<EPI, T>(g, z, m, n, v);
    }
}

template <typename T, int EPI, int NS>
static int launch_ms_inst(const GemmArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)NS * 2 * TBM * TBK * sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_tile_ms_kernel<T, EPI, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%d KiB LDS): %s", (int)(smem >> 10), hipGetErrorString(e)); return WJ_E_HIP; }
    attr_set = true;
  }
  dim3 grid(ceil_div(a.N, TBN), ceil_div(a.M, TBM), EPI == EPI_PARTIAL_F32 ? a.ksplit : a.nbatch);
  hipLaunchKernelGGL((gemm_h_tile_ms_kernel<T, EPI, NS>), grid, dim3(256), smem, s, a);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T, int EPI>
static int launch_ms(const GemmArgs& a, hipStream_t s, int ns) {
  if constexpr (EPI == EPI_T || EPI == EPI_GELU_T || EPI == EPI_F32 || EPI == EPI_RESID_F32 || EPI == EPI_QKV_DEC ||
                EPI == EPI_PARTIAL_F32) {
    const int ks = EPI == EPI_PARTIAL_F32 ? a.ksplit : 1;
    if ((a.split ? 2 * a.K : a.K) % (TBK * ks) || (a.split && a.K % TBK)) {
      set_error("gemm: the multi-stage tile kernel needs K %% (64 * ksplit) == 0");
      return WJ_E_INVALID;
    }
    switch (ns) {
      case 3: return launch_ms_inst<T, EPI, 3>(a, s);
      case 5: return launch_ms_inst<T, EPI, 5>(a, s);
      default: return launch_ms_inst<T, EPI, 4>(a, s);
    }
  } else {
    set_error("gemm: the multi-stage tile kernel does not carry epilogue %d", (int)EPI);
    return WJ_E_INVALID;
  }
}

int g_gemm_big = 1;   // wj_tune("gemm_big"): 0 disables the 256-tile kernel, 2 selects its issue-order-hinted build

template <typename T, int EPI>
static int launch_big(const GemmArgs& a, hipStream_t s) {
  if constexpr (EPI == EPI_PARTIAL_F32) {
    set_error("gemm: the 256-tile kernel has no split-K mode");
    return WJ_E_INVALID;
  } else {
    constexpr size_t smem = 2 * 2 * BBM * TBK * sizeof(bf16_t);   // 128 KiB
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_kernel<T, EPI, 0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_kernel<T, EPI, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(128 KiB LDS): %s", hipGetErrorString(e)); return WJ_E_HIP; }
      attr_set = true;
    }
    dim3 grid(ceil_div(a.N, BBN), ceil_div(a.M, BBM), a.nbatch);
    if (g_gemm_big == 2) hipLaunchKernelGGL((gemm_h_big_kernel<T, EPI, 1>), grid, dim3(512), smem, s, a);
    else hipLaunchKernelGGL((gemm_h_big_kernel<T, EPI, 0>), grid, dim3(512), smem, s, a);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  }
}

// --------------------------------------------------------------------------------------------
// bf16 MFMA, skinny (decode) kernel: 16 output columns per workgroup, 8 waves split K
// --------------------------------------------------------------------------------------------
template <typename T, int EPI, int MT, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_h_skinny_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float red[8 * MT * 16 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt0 = blockIdx.x * 16;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);
  // split-K (EPI_PARTIAL_F32): blockIdx.y selects one of g.ksplit K slices; the 8 waves split that slice
  const int kz = (EPI == EPI_PARTIAL_F32) ? blockIdx.y : 0;
  const int ksteps_all = (g.K + 31) / 32;
  const int ksteps = (EPI == EPI_PARTIAL_F32) ? ksteps_all / g.ksplit : ksteps_all;
  const int per = (ksteps + 7) / 8;
  const int ks_begin = kz * ksteps + wave * per;
  const int ks_end = min(kz * ksteps + ksteps, ks_begin + per);
  const int wrow = nt0 + (lane & 15);
  const int kq = (lane >> 4) * 8;

  for (int mc = 0; mc < g.M; mc += MT * 16) {
    f32x4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // UNR k-steps of W and A fragments are requested before the first MFMA of the group so that
    // each wave keeps several 1 KiB loads in flight (the kernel is latency / HBM bound, not MFMA bound)
    constexpr int UNR = SPLIT ? (MT <= 2 ? 2 : 1) : (MT <= 2 ? 4 : (MT == 4 ? 3 : 2));
    for (int ks0 = ks_begin; ks0 < ks_end; ks0 += UNR) {
      uint4 wv[UNR], av[UNR][MT], al[SPLIT ? UNR : 1][SPLIT ? MT : 1];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = (ks0 + u) * 32 + kq;
        const bool kin = (ks0 + u) < ks_end && k < g.K;
        const int kc = kin ? k : 0;
        wv[u] = ldg16_pred(W + (int64_t)min(wrow, g.N - 1) * g.ldw + kc, kin && wrow < g.N);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int arow = mc + t * 16 + (lane & 15);
          av[u][t] = ldg16_pred(A + (int64_t)min(arow, g.M - 1) * g.lda + kc, kin && arow < g.M);
          if constexpr (SPLIT)   // rounding residuals of the same activations: the second half of the row
            al[u][t] = ldg16_pred(A + (int64_t)min(arow, g.M - 1) * g.lda + g.K + kc, kin && arow < g.M);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const typename Vec8<T>::type wf = as_vec8<T>(wv[u]);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          acc[t] = mfma16(wf, as_vec8<T>(av[u][t]), acc[t]);
          if constexpr (SPLIT) acc[t] = mfma16(wf, as_vec8<T>(al[u][t]), acc[t]);
        }
      }
    }
    // partials -> LDS: red[wave][m_local][n_local]
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      float* p = &red[((wave * MT * 16) + t * 16 + (lane & 15)) * 16 + (lane >> 4) * 4];
      *reinterpret_cast<float4*>(p) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
    __syncthreads();
    for (int o = tid; o < MT * 64; o += 512) {
      const int mm = o >> 2, q = o & 3;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const float4 p = *reinterpret_cast<const float4*>(&red[((w * MT * 16) + mm) * 16 + q * 4]);
        v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
      }
      const int m = mc + mm, n = nt0 + q * 4;
      if (m < g.M && n < g.N) epi_nm<EPI, T>(g, kz, m, n, v);
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------------
// bf16 MFMA, "rows" (decode) kernel: 16 output columns per workgroup, ONE WAVE PER 16-ROW BLOCK.
// No LDS, no barrier, no cross-wave reduction: a wave owns a 16x16 output tile, both MFMA operands are
// 16-byte loads straight from L2/HBM, and UNR k-steps (2 A + 2 W loads each) are requested before the
// first MFMA -- with K (or the K slice) <= 640 the whole operand stream of a wave is ONE round trip.
// The waves of a workgroup share the W fragments through the CU's vector L1; every workgroup re-reads A
// (M x K bf16, L2 resident).  Latency, not bandwidth, bounds a decode-step GEMM (M <= ~200 rows against
// 3-13 MB of weights): what matters is the number of dependent memory round trips per wave and the
// number of launches, which is what this kernel minimises.  blockIdx.y = K slice (EPI_PARTIAL_F32),
// blockIdx.z = group of 8 row blocks.
// --------------------------------------------------------------------------------------------
template <typename T, int EPI, int UNR, bool SPLIT>
__global__ __launch_bounds__(512) void gemm_h_rows_kernel(const GemmArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int m0 = (blockIdx.z * 8 + wave) * 16;
  if (m0 >= g.M) return;
  const int kz = (EPI == EPI_PARTIAL_F32) ? blockIdx.y : 0;
  const int kslice = (EPI == EPI_PARTIAL_F32) ? g.K / g.ksplit : g.K;
  const int kb = kz * kslice, ke = kb + kslice;
  const bf16_t* __restrict__ Ap = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)min(m0 + li, g.M - 1) * g.lda + lg * 8;
  const bf16_t* __restrict__ Wp = reinterpret_cast<const bf16_t*>(g.W) + (int64_t)min(n0 + li, g.N - 1) * g.ldw + lg * 8;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = kb; k0 < ke; k0 += 64 * UNR) {
    uint4 av[UNR][2], wv[UNR][2], al[SPLIT ? UNR : 1][2];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = min(k0 + 64 * u, ke - 64);     // clamped (never predicated); surplus steps are skipped below
      wv[u][0] = *reinterpret_cast<const uint4*>(Wp + k);
      wv[u][1] = *reinterpret_cast<const uint4*>(Wp + k + 32);
      av[u][0] = *reinterpret_cast<const uint4*>(Ap + k);
      av[u][1] = *reinterpret_cast<const uint4*>(Ap + k + 32);
      if constexpr (SPLIT) {                       // rounding residuals: second half of the activation row
        al[u][0] = *reinterpret_cast<const uint4*>(Ap + g.K + k);
        al[u][1] = *reinterpret_cast<const uint4*>(Ap + g.K + k + 32);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (k0 + 64 * u < ke) {
        acc = mfma16(as_vec8<T>(wv[u][0]), as_vec8<T>(av[u][0]), acc);
        acc = mfma16(as_vec8<T>(wv[u][1]), as_vec8<T>(av[u][1]), acc);
        if constexpr (SPLIT) {
          acc = mfma16(as_vec8<T>(wv[u][0]), as_vec8<T>(al[u][0]), acc);
          acc = mfma16(as_vec8<T>(wv[u][1]), as_vec8<T>(al[u][1]), acc);
        }
      }
    }
  }
  const int m = m0 + li, n = n0 + lg * 4;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (m < g.M && n < g.N) epi_nm<EPI, T>(g, kz, m, n, v);
}

template <typename T, int EPI>
static int launch_rows(const GemmArgs& a, hipStream_t s, int unr) {
  if constexpr (EPI == EPI_T || EPI == EPI_GELU_T || EPI == EPI_F32 || EPI == EPI_RESID_F32 || EPI == EPI_QKV_DEC ||
                EPI == EPI_PARTIAL_F32) {
    const int ks = EPI == EPI_PARTIAL_F32 ? a.ksplit : 1;
    if (a.nbatch != 1 || (a.K % (64 * ks))) {
      set_error("gemm: the rows kernel needs nbatch == 1 and K %% (64 * ksplit) == 0 (K=%d ksplit=%d)", a.K, ks);
      return WJ_E_INVALID;
    }
    const int steps = a.K / ks / 64;
    if (unr <= 0) unr = steps % 10 == 0 ? 10 : (steps % 5 == 0 ? 5 : (steps >= 8 ? 8 : 4));
    if (a.split) unr = unr == 10 ? 5 : (unr == 8 ? 4 : unr);      // six 16-byte loads per k-step instead of four
    const dim3 grid(ceil_div(a.N, 16), ks, ceil_div(a.M, 128));
    const dim3 block(64 * min(8, ceil_div(a.M, 16)));
#define WJ_ROWS(U)                                                                                        \
  do {                                                                                                    \
    if (a.split) hipLaunchKernelGGL((gemm_h_rows_kernel<T, EPI, U, true>), grid, block, 0, s, a);         \
    else hipLaunchKernelGGL((gemm_h_rows_kernel<T, EPI, U, false>), grid, block, 0, s, a);                \
  } while (0)
    switch (unr) {
      case 4: WJ_ROWS(4); break;
      case 5: WJ_ROWS(5); break;
      case 8: WJ_ROWS(8); break;
      case 10: WJ_ROWS(10); break;
      default: set_error("gemm: rows kernel unroll %d not instantiated (4, 5, 8, 10)", unr); return WJ_E_INVALID;
    }
#undef WJ_ROWS
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  } else {
    set_error("gemm: the rows kernel does not carry epilogue %d", (int)EPI);
    return WJ_E_INVALID;
  }
}

// --------------------------------------------------------------------------------------------
// fp32 VALU kernel (parity compute type)
// --------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[16][68];
  __shared__ __attribute__((aligned(16))) float Bs[16][68];
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;
  const int z = blockIdx.z;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* __restrict__ A = reinterpret_cast<const float*>(g.A) + (int64_t)z * g.a_batch;
  const float* __restrict__ W = reinterpret_cast<const float*>(g.W);
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < g.K; k0 += 16) {
    const bool kin = k0 + lk < g.K;
    const int kc = kin ? k0 + lk : 0;
    const uint4 au = ldg16_pred(A + (int64_t)min(m0 + lrow, g.M - 1) * g.lda + kc, kin && m0 + lrow < g.M);
    const uint4 bu = ldg16_pred(W + (int64_t)min(n0 + lrow, g.N - 1) * g.ldw + kc, kin && n0 + lrow < g.N);
    const float4 a = make_float4(__uint_as_float(au.x), __uint_as_float(au.y), __uint_as_float(au.z), __uint_as_float(au.w));
    const float4 b = make_float4(__uint_as_float(bu.x), __uint_as_float(bu.y), __uint_as_float(bu.z), __uint_as_float(bu.w));
    As[lk + 0][lrow] = a.x; As[lk + 1][lrow] = a.y; As[lk + 2][lrow] = a.z; As[lk + 3][lrow] = a.w;
    Bs[lk + 0][lrow] = b.x; Bs[lk + 1][lrow] = b.y; Bs[lk + 2][lrow] = b.z; Bs[lk + 3][lrow] = b.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 a4 = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b4 = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
      const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  if constexpr (EPI == EPI_VT) {
    const int m = m0 + ty * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      float v[4] = {acc[0][j], acc[1][j], acc[2][j], acc[3][j]};
      if (m < g.M && n < g.N) epi_vt<float>(g, z, m, n, v);
    }
  } else {
    const int n = n0 + tx * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + ty * 4 + i;
      float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
      if (m < g.M && n < g.N) epi_nm<EPI, float>(g, z, m, n, v);
    }
  }
}

// --------------------------------------------------------------------------------------------
// split-K consumer: out = EPI(sum_s slab[s]) for epilogues that have no natural consumer kernel
// (QKV + cache append, cross-q, fc1 + GELU).  One thread per 4 consecutive columns of a row.
// --------------------------------------------------------------------------------------------
template <int EPI, typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g, const float* __restrict__ slab, int ks) {
  const int quads = g.N >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)g.M * quads) return;
  const int m = (int)(idx / quads), n = (int)(idx % quads) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < ks; ++s) {
    const float4 p = *reinterpret_cast<const float4*>(slab + ((int64_t)s * g.M + m) * g.N + n);
    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
  }
  epi_nm<EPI, T>(g, 0, m, n, v);
}

template <int EPI>
static int launch_reduce_epi(int dtype, const GemmArgs& a, const float* slab, int ks, hipStream_t s) {
  const int64_t total = (int64_t)a.M * (a.N >> 2);
  dim3 grid((unsigned)ceil_div64(total, 256));
  if (dtype == WJ_F32) hipLaunchKernelGGL((splitk_reduce_kernel<EPI, float>), grid, dim3(256), 0, s, a, slab, ks);
  else if (dtype == WJ_F16) hipLaunchKernelGGL((splitk_reduce_kernel<EPI, f16_t>), grid, dim3(256), 0, s, a, slab, ks);
  else hipLaunchKernelGGL((splitk_reduce_kernel<EPI, bf16_t>), grid, dim3(256), 0, s, a, slab, ks);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

int launch_splitk_reduce(int dtype, Epi epi, const GemmArgs& a, const float* slab, int ks, hipStream_t s) {
  if (a.N % 4) { set_error("splitk_reduce: N must be a multiple of 4"); return WJ_E_INVALID; }
  switch (epi) {
    case EPI_T: return launch_reduce_epi<EPI_T>(dtype, a, slab, ks, s);
    case EPI_GELU_T: return launch_reduce_epi<EPI_GELU_T>(dtype, a, slab, ks, s);
    case EPI_QKV_DEC: return launch_reduce_epi<EPI_QKV_DEC>(dtype, a, slab, ks, s);
    default: set_error("splitk_reduce: unsupported epilogue %d", (int)epi); return WJ_E_INVALID;
  }
}

// --------------------------------------------------------------------------------------------
// dispatch
// --------------------------------------------------------------------------------------------
template <typename T, int EPI>
static int launch_epi16(const GemmArgs& a, hipStream_t s, int variant) {
  if (variant == 5 || (variant >= 50 && variant < 70)) return launch_rows<T, EPI>(a, s, variant == 5 ? 0 : variant - 50);
  if (variant == 7 || (variant >= 73 && variant <= 75)) return launch_ms<T, EPI>(a, s, variant == 7 ? 4 : variant - 70);
  const bool skinny_ok = (EPI != EPI_VT) && a.nbatch == 1;
  bool skinny = skinny_ok && a.M <= 512;
  if (variant == 1 || variant == 3 || variant == 4) skinny = false;
  if (variant == 2) {
    if (!skinny_ok) { set_error("skinny GEMM does not support this epilogue/batching"); return WJ_E_INVALID; }
    skinny = true;
  }
  if (skinny) {
    if constexpr (EPI != EPI_VT) {
      dim3 grid(ceil_div(a.N, 16), EPI == EPI_PARTIAL_F32 ? a.ksplit : 1);
#define WJ_SKINNY(MT)                                                                                     \
  do {                                                                                                    \
    if (a.split) hipLaunchKernelGGL((gemm_h_skinny_kernel<T, EPI, MT, true>), grid, dim3(512), 0, s, a);  \
    else hipLaunchKernelGGL((gemm_h_skinny_kernel<T, EPI, MT, false>), grid, dim3(512), 0, s, a);         \
  } while (0)
      if (a.M <= 16) WJ_SKINNY(1);
      else if (a.M <= 32) WJ_SKINNY(2);
      else if (a.M <= 64) WJ_SKINNY(4);
      else WJ_SKINNY(8);
#undef WJ_SKINNY
      WJ_LAUNCH_CHECK();
    }
    return WJ_OK;
  }
  // big encoder GEMMs: 256-tile kernel (variant 6 forces it, 0 = auto when the shape qualifies)
  const bool big_ok = EPI != EPI_PARTIAL_F32 && (a.N % BBN) == 0 && (a.K % TBK) == 0 && a.M >= 1024 && !a.split;
  if (variant == 6 && !big_ok) { set_error("gemm: the 256-tile kernel needs N %% 256 == 0, K %% 64 == 0, M >= 1024"); return WJ_E_INVALID; }
  if (variant == 6 || ((variant == 0 || variant == 1) && g_gemm_big && big_ok)) return launch_big<T, EPI>(a, s);
  dim3 grid(ceil_div(a.N, TBN), ceil_div(a.M, TBM), EPI == EPI_PARTIAL_F32 ? a.ksplit : a.nbatch);
  static const int tile_mode = [] {   // WJ_GEMM_TILE=reg|glds overrides the default staging path
    const char* e = getenv("WJ_GEMM_TILE");
    if (e && !strcmp(e, "reg")) return 1;
    if (e && !strcmp(e, "glds")) return 2;
    return 0;
  }();
  const int ktot = a.split ? 2 * a.K : a.K;
  const int kchunk = EPI == EPI_PARTIAL_F32 ? ktot / a.ksplit : ktot;
  bool glds = (kchunk % TBK) == 0 && (a.K % TBK) == 0 && (variant == 3 || (variant != 4 && tile_mode != 1));
  if (variant == 3 && (a.K % TBK)) { set_error("gemm: the LDS-DMA tile kernel needs K %% 64 == 0"); return WJ_E_INVALID; }
  if (a.split && !glds) { set_error("gemm: split activations need the LDS-DMA tile kernel (K %% 64 == 0)"); return WJ_E_INVALID; }
  if (glds) hipLaunchKernelGGL((gemm_h_tile_kernel<T, EPI, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((gemm_h_tile_kernel<T, EPI, false>), grid, dim3(256), 0, s, a);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <int EPI>
static int launch_epi(int dtype, const GemmArgs& a, hipStream_t s, int variant) {
  if (EPI == EPI_PARTIAL_F32) {
    if (!is16(dtype)) { set_error("gemm: split-K partial output is a 16-bit-path feature"); return WJ_E_INVALID; }
    if (a.ksplit < 1 || a.nbatch != 1 || ((a.split ? 2 * a.K : a.K) % (32 * a.ksplit))) {
      set_error("gemm: split-K needs nbatch == 1 and K %% (32 * ksplit) == 0 (K=%d ksplit=%d)", a.K, a.ksplit);
      return WJ_E_INVALID;
    }
  }
  if (dtype == WJ_F32) {
    dim3 grid(ceil_div(a.N, 64), ceil_div(a.M, 64), a.nbatch);
    hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, dim3(256), 0, s, a);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  }
  if (dtype == WJ_F16) return launch_epi16<f16_t, EPI>(a, s, variant);
  return launch_epi16<bf16_t, EPI>(a, s, variant);
}

int launch_gemm(int dtype, Epi epi, const GemmArgs& a, hipStream_t s, int variant) {
  const int kalign = is16(dtype) ? 8 : 4;
  if (a.K % kalign || a.lda % kalign || a.ldw % kalign || a.a_batch % kalign) {
    set_error("gemm: K/lda/ldw/a_batch must be multiples of %d elements (K=%d lda=%lld ldw=%lld)", kalign, a.K,
              (long long)a.lda, (long long)a.ldw);
    return WJ_E_INVALID;
  }
  if (epi != EPI_F32 && (a.N % 4)) { set_error("gemm: N must be a multiple of 4 for this epilogue"); return WJ_E_INVALID; }
  if (epi == EPI_VT && (a.M % 4)) { set_error("gemm: M must be a multiple of 4 for EPI_VT"); return WJ_E_INVALID; }
  if (a.split && (!is16(dtype) || a.lda < 2 * (int64_t)a.K || a.nbatch != 1)) {
    set_error("gemm: split activations are a 16-bit single-batch feature with lda >= 2 K");
    return WJ_E_INVALID;
  }
  if (a.split_out && (!is16(dtype) || (epi != EPI_T && epi != EPI_GELU_T) || a.ldc < 2 * (int64_t)a.N)) {
    set_error("gemm: split_out needs a 16-bit EPI_T / EPI_GELU_T output with ldc >= 2 N");
    return WJ_E_INVALID;
  }
  if (a.M <= 0 || a.N <= 0) return WJ_OK;
  if (dtype != WJ_F32 && dtype != WJ_BF16 && dtype != WJ_F16) { set_error("gemm: unknown dtype %d", dtype); return WJ_E_INVALID; }
  switch (epi) {
    case EPI_T: return launch_epi<EPI_T>(dtype, a, s, variant);
    case EPI_GELU_T: return launch_epi<EPI_GELU_T>(dtype, a, s, variant);
    case EPI_F32: return launch_epi<EPI_F32>(dtype, a, s, variant);
    case EPI_RESID_F32: return launch_epi<EPI_RESID_F32>(dtype, a, s, variant);
    case EPI_GELU_POS_F32: return launch_epi<EPI_GELU_POS_F32>(dtype, a, s, variant);
    case EPI_QK_HEADS: return launch_epi<EPI_QK_HEADS>(dtype, a, s, variant);
    case EPI_VT: return launch_epi<EPI_VT>(dtype, a, s, variant);
    case EPI_QKV_DEC: return launch_epi<EPI_QKV_DEC>(dtype, a, s, variant);
    case EPI_CKV: return launch_epi<EPI_CKV>(dtype, a, s, variant);
    case EPI_PARTIAL_F32: return launch_epi<EPI_PARTIAL_F32>(dtype, a, s, variant);
    default: set_error("gemm: unknown epilogue %d", (int)epi); return WJ_E_INVALID;
  }
}

}  // namespace wj
