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

#include <algorithm>

#include "kernels.hpp"

namespace wj {

// --------------------------------------------------------------------------------------------
// epilogues
// --------------------------------------------------------------------------------------------
// Flat rows (GemmArgs::seq_T > 0, the blocked-operand path): row m of the GEMM is position m % seq_T of window m / seq_T.
// One float multiply + at most one correction instead of an integer division (m < 2^24, checked by launch_gemm).
__device__ __forceinline__ void flat_zt(const GemmArgs& g, int& z, int& m) {
  if (g.seq_T) {
    int zz = (int)((float)m * g.inv_seq_T);
    int t = m - zz * g.seq_T;
    if (t < 0) { --zz; t += g.seq_T; }
    if (t >= g.seq_T) { ++zz; t -= g.seq_T; }
    z = zz;
    m = t;
  }
}
// element offset of (row m, column n) in the blocked layout [rows / 256][cols / 32][256][32] (GemmArgs::blk / out_blk)
__device__ __forceinline__ int64_t blk_off(int m, int n, int cols) {
  return (((int64_t)(m >> 8) * (cols >> 5) + (n >> 5)) << 13) + ((m & 255) << 5) + (n & 31);
}

template <typename T>
__device__ __forceinline__ void epi_vt_store(const GemmArgs& g, int z, int m, int n, float v[4]) {
  // v[i] belongs to (row m + i, column n), bias already added; m % 4 == 0; m < M (M % 4 == 0), n < N.
  flat_zt(g, z, m);
  const int h = n >> 6, dd = n & 63;
  st4(reinterpret_cast<T*>(g.out) + (((int64_t)z * g.H + h) * 64 + dd) * g.Tpad + m, v);
}

template <typename T>
__device__ __forceinline__ void epi_vt(const GemmArgs& g, int z, int m, int n, float v[4]) {
  const float b = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] += b;
  epi_vt_store<T>(g, z, m, n, v);
}

// epi_nm_store: v already carries the bias.  (EPI_F32 here is the float4 store: callers guarantee n + 3 < N.)
template <int EPI, typename T>
__device__ __forceinline__ void epi_nm_store(const GemmArgs& g, int z, int m, int n, float v[4]) {
  if constexpr (EPI == EPI_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (int64_t)z * g.c_batch + (int64_t)m * g.ldc + n) =
        make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (EPI == EPI_T || EPI == EPI_GELU_T) {
    if constexpr (EPI == EPI_GELU_T) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = gelu_for<T>(v[j]);
    }
    if (g.out_blk) { st4(reinterpret_cast<T*>(g.out) + blk_off(m, n, g.N), v); return; }
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
    flat_zt(g, z, m);
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

template <int EPI, typename T>
__device__ __forceinline__ void epi_nm(const GemmArgs& g, int z, int m, int n, float v[4]) {
  // v[j] belongs to (row m, column n + j); n % 4 == 0; caller guarantees m < M and n < N.
  if constexpr (EPI == EPI_PARTIAL_F32) {   // z = K-slice index
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + ((int64_t)z * g.M + m) * g.ldc + n) =
        make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (EPI == EPI_F32) {    // N tail safe
    float* o = reinterpret_cast<float*>(g.out) + (int64_t)z * g.c_batch + (int64_t)m * g.ldc + n;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < g.N) o[j] = v[j] + (g.bias ? g.bias[n + j] : 0.0f);
  } else {
    if (g.bias) {
      const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    epi_nm_store<EPI, T>(g, z, m, n, v);
  }
}

// address of the 8 consecutive output elements (m, n8 .. n8 + 7), n8 % 8 == 0, of the 2-byte row-major / head-split epilogues
template <int EPI, typename T>
__device__ __forceinline__ T* epi_addr8(const GemmArgs& g, int z, int m, int n, int posv) {
  if constexpr (EPI == EPI_T || EPI == EPI_GELU_T) {
    if (g.out_blk) return reinterpret_cast<T*>(g.out) + blk_off(m, n, g.N);
    return reinterpret_cast<T*>(g.out) + (int64_t)z * g.c_batch + (int64_t)m * g.ldc + n;
  } else if constexpr (EPI == EPI_QK_HEADS || EPI == EPI_CKV) {
    flat_zt(g, z, m);
    const int which = n >= g.D;
    const int nn = n - which * g.D;
    const int h = nn >> 6, dd = nn & 63;
    return reinterpret_cast<T*>(which ? g.out2 : g.out) + (((int64_t)z * g.H + h) * g.Tpad + m) * 64 + dd;
  } else {   // EPI_QKV_DEC
    const int which = n / g.D;
    const int nn = n - which * g.D;
    if (which == 0) return reinterpret_cast<T*>(g.out) + (int64_t)m * g.D + nn;
    const int h = nn >> 6, dd = nn & 63;
    const int pos = g.seq_tp ? m % g.seq_tp : posv;
    const int64_t crow = g.seq_tp ? m / g.seq_tp : m;
    return reinterpret_cast<T*>(which == 1 ? g.out2 : g.out3) + ((crow * g.H + h) * g.cache_len + pos) * 64 + dd;
  }
}

// v_permlane16_swap: lanes 16-31 / 48-63 of a <-> lanes 0-15 / 32-47 of b
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

__device__ __forceinline__ bool g_epi_wide_dev(const GemmArgs& g) { return g.epi_wide != 0; }

// Epilogue of an MFMA tile (RI x 4 fragments of 16 x 16 per wave, wave origin (mw, nw)).  The bias of the wave's four column
// groups is fetched ONCE, before the fragment loop: written as one load inside every fragment's bounds check, hipcc emits a
// branch + global_load + s_waitcnt vmcnt(0) per fragment -- 16 to 32 serialised L2 round trips at the end of every tile.
template <int EPI, typename T, int RI>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& g, int z, int mw, int nw, int lane, f32x4_t (&acc)[RI][4]) {
  if constexpr (EPI == EPI_SWIGLU_T) {
    // W rows interleaved [16 gate | 16 up]: fragments (0, 1) and (2, 3) of a wave are (gate, up) of the same 16 output columns and
    // a lane holds the same (row, 4 columns) in both -- silu(gate) * up is register arithmetic, the wave's 64 accumulator columns
    // become 32 output columns, stored 16 bytes per lane after the same v_permlane16_swap as the plain epilogue.  The gate / up
    // matrix (2 F columns per row, written and read back by a separate element-wise pass before) never reaches memory.
    if constexpr (sizeof(T) == 2) {
      const int q = lane >> 4, Nout = g.N >> 1;
#pragma unroll
      for (int i = 0; i < RI; ++i) {
        const int m = mw + i * 16 + (lane & 15);
        float va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ga = acc[i][0][e], gb = acc[i][2][e];
          va[e] = ga / (1.f + __expf(-ga)) * acc[i][1][e];
          vb[e] = gb / (1.f + __expf(-gb)) * acc[i][3][e];
        }
        const int n8 = (nw >> 1) + (q & 1) * 16 + (q >> 1) * 8;
        const bool in = m < g.M && n8 < Nout;
        T* o = reinterpret_cast<T*>(g.out) + (int64_t)min(m, g.M - 1) * g.ldc + min(n8, Nout - 8);
        uint32_t a0 = pack2<T>(va[0], va[1]), a1 = pack2<T>(va[2], va[3]);
        uint32_t c0 = pack2<T>(vb[0], vb[1]), c1 = pack2<T>(vb[2], vb[3]);
        swap16(a0, c0); swap16(a1, c1);
        if (g.split_out) {
          float ra[4], rb[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { ra[e] = va[e] - round_T<T>(va[e]); rb[e] = vb[e] - round_T<T>(vb[e]); }
          uint32_t l0 = pack2<T>(ra[0], ra[1]), l1 = pack2<T>(ra[2], ra[3]);
          uint32_t d0 = pack2<T>(rb[0], rb[1]), d1 = pack2<T>(rb[2], rb[3]);
          swap16(l0, d0); swap16(l1, d1);
          if (in) *reinterpret_cast<uint4*>(o + Nout) = make_uint4(l0, l1, d0, d1);
        }
        if (in) *reinterpret_cast<uint4*>(o) = make_uint4(a0, a1, c0, c1);
      }
    }
  } else if constexpr (EPI == EPI_PARTIAL_F32 || EPI == EPI_F32) {
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        const int m = mw + i * 16 + (lane & 15), n = nw + j * 16 + (lane >> 4) * 4;
        if (m < g.M && n < g.N) epi_nm<EPI, T>(g, z, m, n, v);
      }
  } else if constexpr (EPI == EPI_RESID_F32) {
    // x += acc + bias: the 16 residual loads of four fragment rows are issued before the first add / store (written as
    // load-add-store per fragment this epilogue is a chain of dependent memory round trips)
    const int nb = nw + (lane >> 4) * 4;
    float4 bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = min(nb + j * 16, g.N - 4);
      bias4[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float* xo = reinterpret_cast<float*>(g.out) + (int64_t)z * g.c_batch;
#pragma unroll
    for (int ih = 0; ih < RI; ih += 4) {
      float4 r[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = min(mw + (ih + i) * 16 + (lane & 15), g.M - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          r[i][j] = *reinterpret_cast<const float4*>(xo + (int64_t)m * g.ldc + min(nb + j * 16, g.N - 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mw + (ih + i) * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = nb + j * 16;
          if (m < g.M && n < g.N) {
            const f32x4_t a4 = acc[ih + i][j];
            float4 x = r[i][j];
            x.x += a4[0] + bias4[j].x; x.y += a4[1] + bias4[j].y; x.z += a4[2] + bias4[j].z; x.w += a4[3] + bias4[j].w;
            *reinterpret_cast<float4*>(xo + (int64_t)m * g.ldc + n) = x;
          }
        }
      }
    }
  } else if constexpr (EPI == EPI_VT) {
    float b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = min(nw + j * 16 + (lane & 15), g.N - 1);
      b1[j] = g.bias ? g.bias[n] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4] = {acc[i][j][0] + b1[j], acc[i][j][1] + b1[j], acc[i][j][2] + b1[j], acc[i][j][3] + b1[j]};
        const int m = mw + i * 16 + (lane >> 4) * 4, n = nw + j * 16 + (lane & 15);
        if (m < g.M && n < g.N) epi_vt_store<T>(g, z, m, n, v);
      }
  } else {
    float4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = min(nw + j * 16 + (lane >> 4) * 4, g.N - 4);     // N % 4 == 0 for these epilogues
      b4[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the bias goes onto every accumulator in straight-line code: inside the bounds-checked fragment blocks each use
    // costs an `s_waitcnt vmcnt(0)`, which on gfx9 also waits for the previous fragment's STORES to reach L2
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j][0] += b4[j].x; acc[i][j][1] += b4[j].y; acc[i][j][2] += b4[j].z; acc[i][j][3] += b4[j].w;
      }
    constexpr bool WIDE = sizeof(T) == 2 && (EPI == EPI_T || EPI == EPI_GELU_T || EPI == EPI_QK_HEADS || EPI == EPI_CKV ||
                                             EPI == EPI_QKV_DEC);
    if constexpr (WIDE) {
      // 16-byte stores.  A lane holds 4 consecutive columns of one row (8 bytes); the lanes 16 apart hold the next 4.
      // v_permlane16_swap on the packed words of two neighbouring fragments (odd 16-lane rows of the first operand <->
      // even rows of the second) leaves every lane with 8 consecutive columns: lane group q = lane >> 4 owns columns
      // (q >> 1) * 8 .. + 7 of fragment j + (q & 1).  The epilogue of a 256 x 256 tile is bound by the NUMBER of store
      // instructions (32 per wave as 8-byte stores), not by their bytes.
      if ((g.N & 7) == 0 && (g.ldc & 7) == 0 && g_epi_wide_dev(g)) {
        int posv = 0;
        if constexpr (EPI == EPI_QKV_DEC) posv = g.seq_tp ? 0 : *g.pos_ptr;
        const int q = lane >> 4;
#pragma unroll
        for (int i = 0; i < RI; ++i) {
          const int m = mw + i * 16 + (lane & 15);
#pragma unroll
          for (int jp = 0; jp < 4; jp += 2) {
            float va[4] = {acc[i][jp][0], acc[i][jp][1], acc[i][jp][2], acc[i][jp][3]};
            float vb[4] = {acc[i][jp + 1][0], acc[i][jp + 1][1], acc[i][jp + 1][2], acc[i][jp + 1][3]};
            if constexpr (EPI == EPI_GELU_T) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { va[e] = gelu_for<T>(va[e]); vb[e] = gelu_for<T>(vb[e]); }
            }
            const int n8 = nw + (jp + (q & 1)) * 16 + (q >> 1) * 8;
            const bool in = m < g.M && n8 < g.N;
            T* o = epi_addr8<EPI, T>(g, z, min(m, g.M - 1), min(n8, g.N - 8), posv);
            uint32_t a0 = pack2<T>(va[0], va[1]), a1 = pack2<T>(va[2], va[3]);
            uint32_t c0 = pack2<T>(vb[0], vb[1]), c1 = pack2<T>(vb[2], vb[3]);
            swap16(a0, c0); swap16(a1, c1);
            bool split_out = false;
            if constexpr (EPI == EPI_T || EPI == EPI_GELU_T) split_out = g.split_out != 0;
            if (split_out) {   // [hi | lo] rows: lo = T(v - hi), N elements further on
              float ra[4], rb[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) { ra[e] = va[e] - round_T<T>(va[e]); rb[e] = vb[e] - round_T<T>(vb[e]); }
              uint32_t l0 = pack2<T>(ra[0], ra[1]), l1 = pack2<T>(ra[2], ra[3]);
              uint32_t d0 = pack2<T>(rb[0], rb[1]), d1 = pack2<T>(rb[2], rb[3]);
              swap16(l0, d0); swap16(l1, d1);
              if (in) *reinterpret_cast<uint4*>(o + g.N) = make_uint4(l0, l1, d0, d1);
            }
            if (in) *reinterpret_cast<uint4*>(o) = make_uint4(a0, a1, c0, c1);
          }
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        const int m = mw + i * 16 + (lane & 15), n = nw + j * 16 + (lane >> 4) * 4;
        if (m < g.M && n < g.N) epi_nm_store<EPI, T>(g, z, m, n, v);
      }
  }
}

// --------------------------------------------------------------------------------------------
// bf16 MFMA, 128x128x64 tile
// --------------------------------------------------------------------------------------------
constexpr int TBM = 128, TBN = 128, TBK = 64;

// wj_tune("tile_l2_kb"): the 128-tile kernels walk, per XCD, groups of GM row panels column by column (see gemm_h_tile_kernel).  A
// group's A panels (128 rows x the k slice, twice that with split activations) must stay in the XCD's 4 MiB L2 while its columns
// stream by, or every column re-fetches them through the fabric: 0 = rounds 1-5's fixed GM = 8 (K = 5120 split: 21 MB per group),
// > 0 = as many panels as fit this many KiB (at least 1, at most 8)
int g_tile_l2_kb = 0;
static inline int tile_gm(int kchunk) {
  if (g_tile_l2_kb <= 0) return 8;
  const int64_t panel = (int64_t)TBM * kchunk * 2;
  return (int)std::max<int64_t>(1, std::min<int64_t>(8, (int64_t)g_tile_l2_kb * 1024 / std::max<int64_t>(1, panel)));
}

// Predicated 16-byte load: the address is always valid (callers clamp it) and the VALUE is
// selected.  Writing `p ? *ptr : zero` instead lets the compiler select between the global pointer
// and a stack slot, which turns every load into a flat_load plus a scratch store.
__device__ __forceinline__ uint4 ldg16_pred(const void* p, bool pred) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  v.x = pred ? v.x : 0u; v.y = pred ? v.y : 0u; v.z = pred ? v.z : 0u; v.w = pred ? v.w : 0u;
  return v;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * TBK + ((chunk ^ (row & 7)) << 3); }

// Workgroup barrier WITHOUT the fence of __syncthreads().  With LDS-DMA (global_load_lds) requests in flight hipcc turns
// the fence into `s_waitcnt vmcnt(0)` in front of the s_barrier, i.e. every barrier drains the whole prefetch queue
// (seen in the device assembly of the multi-stage kernels: the counted wait right before it was dead code).  The kernels
// that use this barrier order their LDS traffic themselves: a counted s_waitcnt vmcnt(N) by the waves that issued the
// DMA, then this barrier, then the ds_reads (MI355X: nothing else orders a ds_read behind a pending LDS-DMA), and an
// s_waitcnt lgkmcnt(0) by the readers before the barrier that hands a buffer back to the DMA.
__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 32 && N % 4 == 0, "unsupported count");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
}

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
  const int GM = g.gm;        // row panels per group: launch_epi16 sizes the group's A panels for the XCD's L2 (wj_tune tile_l2_kb)
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

  tile_epilogue<EPI, T, 4>(g, z, m0 + wm * 64, n0 + wn * 64, lane, acc);
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

  tile_epilogue<EPI, T, 8>(g, z, m0 + wm * 128, n0 + wn * 64, lane, acc);
}

// --------------------------------------------------------------------------------------------
// 16-bit MFMA, 256x256 tile, k-stages of 32, NS-deep LDS-DMA ring, two wave groups in ping-pong.
//
// Why: the kernel above keeps all eight waves in lockstep -- after the per-k-step barrier every wave first issues its 8
// LDS-DMA requests and 24 ds_reads, so the matrix pipe idles for that whole issue phase, and the `vmcnt(0)` in front
// of the barrier allows one k-step (64 KiB) in flight per CU: measured 0.82-1.0 PFLOP/s, the waves 40-50 % parked at
// the barrier.  Here
//   * a stage is 256 x 32 of A and of W (32 KiB, one MFMA k-block), the ring holds NS of them and NS-1 are in flight:
//     the wait in front of a barrier is `s_waitcnt vmcnt(4 (NS-2))`, never 0 inside the main loop;
//   * the waves with wm = 0 and wm = 1 (the two waves that share each SIMD) run one barrier apart: while one group
//     issues its ds_reads + LDS-DMA (MEM phase) the other one runs its 32 MFMAs on fragments already in registers.
//
// Barrier sequence B0, B1, ... after the prologue barrier P (every wave takes part in every barrier):
//   group 0:          MEM(0) B0 MFMA(0) B1 MEM(1) B2 MFMA(1) B3 ...            MEM(kt) ends at B(2kt)
//   group 1:   -      B0 MEM(0) B1 MFMA(0) B2 MEM(1) B3 MFMA(1) B4 ...         MEM(kt) ends at B(2kt+1)
//   MEM(kt):  ds_read the fragments of stage kt; request stage kt+NS-1 into the buffer of stage kt-1;
//             s_waitcnt vmcnt -> own requests of stage kt+1 have landed; s_waitcnt lgkmcnt(0) -> own reads retired.
// RAW: both groups have waited for their requests of stage kt+1 by B(2kt+1); the first reads of that stage come after
//      B(2kt+1) (group 0) and B(2kt+2) (group 1).  Stage 0: waited for before P.
// WAR: the buffer of stage kt-1 is read last in group 1's MEM(kt-1), retired before B(2kt-1); it is requested again in
//      MEM(kt): after B(2kt-1) (group 0) and after B(2kt) (group 1).
// 64-byte LDS rows; the 16-byte chunk c of row r sits in slot c ^ f((r >> 2) & 3), f = {0, 2, 3, 1}: conflict-free for
// the four 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32).  The accumulation order over k is
// the one of the kernels above (ascending blocks of 32), so the results are bit-identical to theirs.
// --------------------------------------------------------------------------------------------
constexpr int PBK = 32;
constexpr int PSTAGE = 2 * BBM * PBK;   // elements per stage: A[256][32] then W[256][32]

__device__ __forceinline__ int pp_f(int row) {
  const int q = (row >> 2) & 3;
  return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1);
}

// ABL (timing experiments only, results are wrong): bit 0 no LDS-DMA in the main loop, bit 1 no ds_reads in the main loop,
// bit 2 no MFMAs, bit 3 no barriers in the main loop
// PLACE: where a wave issues its 4 LDS-DMA requests of a stage: 0 = in its MEM phase after the ds_reads, 1 = in its MEM
// phase before them, 2 = between the MFMAs of its MFMA phase (one request per 8 MFMAs; the stage then lands one phase
// later, so the wait in MEM(kt) leaves NS-3 stages in flight instead of NS-2)
template <typename T, int EPI, int NS, int ABL = 0, int PLACE = 0>
__global__ __launch_bounds__(512) void gemm_h_big_pp_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_pp[];   // [stage][A|W][256][32]
  constexpr int D = NS - 1;                                          // stages in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  // staging: a wave request deposits 64 x 16 B = 16 rows of 64 B, lane-linear; the swizzle goes into the SOURCE address
  // (slot s = lane & 3 of row r fetches chunk s ^ f(r)).  Two A pieces and two W pieces per wave and stage.
  const bf16_t* ga[2];
  const bf16_t* gw[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (wave * 2 + p) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ pp_f(row);
    ga[p] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
    gw[p] = W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
  }
#define WJ_PP_ISSUE(buf)                                                                                      \
  _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga[p],                    \
        (__attribute__((address_space(3))) void*)(&lds_pp[(buf) * PSTAGE + (wave * 2 + p) * 16 * PBK]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw[p],                    \
        (__attribute__((address_space(3))) void*)(&lds_pp[(buf) * PSTAGE + BBM * PBK + (wave * 2 + p) * 16 * PBK]), 16, 0, 0); \
    ga[p] += PBK;                                                                                             \
    gw[p] += PBK;                                                                                             \
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row (lane & 15) of a 16-row block, chunk lane >> 4, swizzled (f depends on the lane only)
  const int frag = (lane & 15) * PBK + (((lane >> 4) ^ pp_f(lane & 15)) << 3);
  const int a_frag = wm * 128 * PBK + frag;
  const int w_frag = BBM * PBK + wn * 64 * PBK + frag;

  const int nk = g.K / PBK;      // launch guarantees nk >= NS
#pragma unroll
  for (int st = 0; st < D; ++st) { WJ_PP_ISSUE(st) }
  wait_vmcnt<4 * (D - 1)>();     // own requests of stage 0 have landed
  wg_barrier();                  // P: stage 0 visible to every wave
  if (wm == 1) wg_barrier();     // B0: group 1 runs one barrier behind group 0

  int cbuf = 0, ibuf = D;        // buffer of stage kt / of stage kt + D
  typename Vec8<T>::type af[8], wf[4];
  if constexpr ((ABL & 2) != 0) {
    const bf16_t* ls = &lds_pp[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[w_frag + j * 16 * PBK]);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[a_frag + i * 16 * PBK]);
  }
#define WJ_PP_MFMA(i, j)                                                   \
  if constexpr (EPI == EPI_VT) acc[i][j] = mfma16(af[i], wf[j], acc[i][j]); \
  else acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
#define WJ_PP_ONE(X, p, region)                                                                                  \
  {                                                                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X[p] + back),               \
        (__attribute__((address_space(3))) void*)(&lds_pp[ibuf * PSTAGE + (region) + (wave * 2 + p) * 16 * PBK]), 16, 0, 0); \
    X[p] += step;                                                                                                \
  }
  for (int kt = 0; kt < nk; ++kt) {
    // ---- MEM(kt)
    const bool more = kt + D < nk;
    if constexpr (PLACE == 1 && (ABL & 1) == 0) {
      if (more) { WJ_PP_ISSUE(ibuf) }
    }
    if constexpr ((ABL & 2) == 0) {
      const bf16_t* ls = &lds_pp[cbuf * PSTAGE];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[w_frag + j * 16 * PBK]);
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[a_frag + i * 16 * PBK]);
    }
    if constexpr (PLACE == 2) {
      wait_vmcnt<4 * (D - 2)>();   // stage kt+1 landed; the requests of MFMA(kt-D+2) .. MFMA(kt-1) stay in flight
    } else {
      if (more) {
        if constexpr (PLACE == 0 && (ABL & 1) == 0) { WJ_PP_ISSUE(ibuf) }
        wait_vmcnt<4 * (D - 1)>();   // stage kt+1 landed; stages kt+2 .. kt+D stay in flight
      } else {
        wait_vmcnt<0>();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr ((ABL & 8) == 0) wg_barrier(); else __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA(kt)
    __builtin_amdgcn_s_setprio(1);
    if constexpr ((ABL & 4) == 0) {
      if constexpr (PLACE == 2) {
        // no branch around the requests (two copies of the MFMA block make hipcc spill the accumulators): past the last
        // stage the wave re-requests its last block into a buffer nobody reads any more
        const int back = more ? 0 : -PBK, step = more ? PBK : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { WJ_PP_MFMA(i, j) }
          if (i == 1) WJ_PP_ONE(ga, 0, 0)
          if (i == 3) WJ_PP_ONE(gw, 0, BBM * PBK)
          if (i == 5) WJ_PP_ONE(ga, 1, 0)
          if (i == 7) WJ_PP_ONE(gw, 1, BBM * PBK)
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) { WJ_PP_MFMA(i, j) }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(wf[j]));
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr ((ABL & 8) == 0) wg_barrier(); else __builtin_amdgcn_sched_barrier(0);
    cbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
    ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
  }
#undef WJ_PP_MFMA
#undef WJ_PP_ONE
  if constexpr (PLACE == 2) wait_vmcnt<0>();   // the dummy requests of the tail must not outlive the workgroup's LDS
  if (wm == 0) wg_barrier();     // every wave has taken part in the same number of barriers
#undef WJ_PP_ISSUE

  tile_epilogue<EPI, T, 8>(g, z, m0 + wm * 128, n0 + wn * 64, lane, acc);
}


// --------------------------------------------------------------------------------------------
// Ping-pong 256-tile kernel over BLOCKED operands (round 4; GemmArgs::blk): the schedule, LDS image and accumulation order
// of gemm_h_big_pp_kernel, with A stored as [ceil(M / 256)][K / 32][256][32] and W as [N / 256][K / 32][256][32].
// One 32-wide stage of a tile is then 16 KiB contiguous per operand and every LDS-DMA wave request (16 rows x 64 B) is
// 1 KiB contiguous.  Measured with the matrix pipe running beside the DMA stream (scripts/gemm_probe.hip `dmapp2`, no
// ds_reads): row-major operands with 128-byte segments bound the loop at 1120-1320 TFLOP/s-equivalent, blocked operands
// with 4 requests per MEM phase at 1390-1480 on all three encoder shapes -- rows K elements apart put the 8 segments of a
// request on a few L2 channels (K = 5120: stride 10 KiB), a contiguous KiB spreads over all of them.
// Rows past M inside the last row block are never written by the producers and never stored by the epilogue (a row of C
// depends on its own row of A only).  Flat rows only (no batch dimension): the head-split epilogues take the window from
// the row index (GemmArgs::seq_T).
// --------------------------------------------------------------------------------------------
template <typename T, int EPI, int NS>
__global__ __launch_bounds__(512) void gemm_h_big_ppb_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_pb[];   // [stage][A|W][256][32]
  constexpr int D = NS - 1;                                          // stages in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nx = gridDim.x, ntiles = gridDim.x * gridDim.y;
  const int lin = blockIdx.y * nx + blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = lin & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  const int GM = g.gm;
  const int per_group = GM * nx, group = tile / per_group, first_m = group * GM;
  const int gsz = min(GM, (int)gridDim.y - first_m), in_group = tile - group * per_group;
  const int mt = first_m + in_group % gsz, nt = in_group / gsz;
  const int nk = g.K / PBK;      // launch guarantees nk >= NS
  constexpr int BLK = BBM * PBK; // elements of one block
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)mt * nk * BLK;
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W) + (int64_t)nt * nk * BLK;

  // a wave request deposits 16 rows of 64 B lane-linear; slot s = lane & 3 of row r fetches chunk s ^ f(r) of the same row
  const bf16_t* ga[2];
  const bf16_t* gw[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = (wave * 2 + p) * 16 + (lane >> 2);
    const int off = row * PBK + (((lane & 3) ^ pp_f(row)) << 3);
    ga[p] = A + off;
    gw[p] = W + off;
  }
#define WJ_PB_ISSUE(buf)                                                                                      \
  _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga[p],                    \
        (__attribute__((address_space(3))) void*)(&lds_pb[(buf) * PSTAGE + (wave * 2 + p) * 16 * PBK]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw[p],                    \
        (__attribute__((address_space(3))) void*)(&lds_pb[(buf) * PSTAGE + BBM * PBK + (wave * 2 + p) * 16 * PBK]), 16, 0, 0); \
    ga[p] += BLK;                                                                                             \
    gw[p] += BLK;                                                                                             \
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frag = (lane & 15) * PBK + (((lane >> 4) ^ pp_f(lane & 15)) << 3);
  const int a_frag = wm * 128 * PBK + frag;
  const int w_frag = BBM * PBK + wn * 64 * PBK + frag;

#pragma unroll
  for (int st = 0; st < D; ++st) { WJ_PB_ISSUE(st) }
  wait_vmcnt<4 * (D - 1)>();     // own requests of stage 0 have landed
  wg_barrier();                  // P: stage 0 visible to every wave
  if (wm == 1) wg_barrier();     // B0: group 1 runs one barrier behind group 0

  int cbuf = 0, ibuf = D;        // buffer of stage kt / of stage kt + D
  typename Vec8<T>::type af[8], wf[4];
  for (int kt = 0; kt < nk; ++kt) {
    // ---- MEM(kt)
    const bf16_t* ls = &lds_pb[cbuf * PSTAGE];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[w_frag + j * 16 * PBK]);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[a_frag + i * 16 * PBK]);
    if (kt + D < nk) {
      WJ_PB_ISSUE(ibuf)
      wait_vmcnt<4 * (D - 1)>();   // stage kt+1 landed; stages kt+2 .. kt+D stay in flight
    } else {
      wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();
    // ---- MFMA(kt)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (EPI == EPI_VT) acc[i][j] = mfma16(af[i], wf[j], acc[i][j]);
        else acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
      }
    __builtin_amdgcn_s_setprio(0);
    wg_barrier();
    cbuf = cbuf + 1 == NS ? 0 : cbuf + 1;
    ibuf = ibuf + 1 == NS ? 0 : ibuf + 1;
  }
  if (wm == 0) wg_barrier();     // every wave has taken part in the same number of barriers
#undef WJ_PB_ISSUE

  tile_epilogue<EPI, T, 8>(g, 0, mt * BBM + wm * 128, nt * BBN + wn * 64, lane, acc);
}

int g_ppb_ns = 4;   // wj_tune("ppb_ns"): ring stages of the blocked kernel (3 / 4 / 5)
int g_ppb_gm = 8;   // wj_tune("ppb_gm"): row tiles per group of the tile order (L2 reuse of the W panels)

template <typename T, int EPI, int NS>
static int launch_big_ppb_ns(const GemmArgs& a, hipStream_t s);

template <typename T, int EPI>
static int launch_big_ppb(const GemmArgs& a, hipStream_t s) {
  if constexpr (EPI == EPI_PARTIAL_F32 || EPI == EPI_QKV_DEC || EPI == EPI_GELU_POS_F32) {
    set_error("gemm: blocked operands are an encoder-path feature (no split-K / decode / conv epilogues)");
    return WJ_E_INVALID;
  } else {
    if (g_ppb_ns == 3) return launch_big_ppb_ns<T, EPI, 3>(a, s);
    if (g_ppb_ns == 5) return launch_big_ppb_ns<T, EPI, 5>(a, s);
    return launch_big_ppb_ns<T, EPI, 4>(a, s);
  }
}

template <typename T, int EPI, int NS>
static int launch_big_ppb_ns(const GemmArgs& a_in, hipStream_t s) {
  {
    GemmArgs a = a_in;
    a.gm = g_ppb_gm > 0 ? g_ppb_gm : 8;
    if ((a.N % BBN) || (a.K % PBK) || a.K / PBK < NS + 1 || a.nbatch != 1 || a.split || a.split_out) {
      set_error("gemm: blocked operands need N %% 256 == 0, K %% 32 == 0, K >= 160, one batch, no split activations (N=%d K=%d)",
                a.N, a.K);
      return WJ_E_INVALID;
    }
    constexpr size_t smem = (size_t)NS * PSTAGE * sizeof(bf16_t);   // 128 KiB
    static AttrOnce attr_set;
    if (attr_set.need()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_ppb_kernel<T, EPI, NS>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(%d KiB LDS): %s", (int)(smem >> 10), hipGetErrorString(e)); return WJ_E_HIP; }
      attr_set.done();
    }
    dim3 grid(a.N / BBN, ceil_div(a.M, BBM), 1);
    hipLaunchKernelGGL((gemm_h_big_ppb_kernel<T, EPI, NS>), grid, dim3(512), smem, s, a);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  }
}

// row-major [rows][ld] <-> blocked [ceil(rows / 256)][cols / 32][256][32] (2-byte elements, 16-byte granules); rows past
// `rows` of the last block are zero-filled by to_blocked
__global__ __launch_bounds__(256) void to_blocked_kernel(const uint4* __restrict__ src, int64_t ld8, int rows, int cols8,
                                                         uint4* __restrict__ dst, int back) {
  const int rows_pad = (rows + 255) & ~255;
  const int64_t total = (int64_t)rows_pad * cols8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / cols8), c8 = (int)(i - (int64_t)m * cols8);
    const int64_t b = ((((int64_t)(m >> 8) * (cols8 >> 2) + (c8 >> 2)) << 13) + ((m & 255) << 5) + ((c8 & 3) << 3)) >> 3;
    if (back) {
      if (m < rows) dst[(int64_t)m * ld8 + c8] = src[b];
    } else {
      dst[b] = m < rows ? src[(int64_t)m * ld8 + c8] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

int launch_to_blocked(const void* src, int64_t ld, int rows, int cols, void* dst, hipStream_t s) {
  if ((cols % 32) || (ld % 8) || rows <= 0) { set_error("to_blocked: cols %% 32 == 0 and ld %% 8 == 0 required"); return WJ_E_INVALID; }
  hipLaunchKernelGGL(to_blocked_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)src, ld / 8, rows, cols / 8, (uint4*)dst, 0);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}
int launch_from_blocked(const void* src, int rows, int cols, void* dst, int64_t ld, hipStream_t s) {
  if ((cols % 32) || (ld % 8) || rows <= 0) { set_error("from_blocked: cols %% 32 == 0 and ld %% 8 == 0 required"); return WJ_E_INVALID; }
  hipLaunchKernelGGL(to_blocked_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)src, ld / 8, rows, cols / 8, (uint4*)dst, 1);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

// --------------------------------------------------------------------------------------------
// Ping-pong over 64-wide k "pairs": the schedule of gemm_h_big_pp_kernel, but the LDS-DMA fetches 128-byte row segments
// (8 rows x 128 B per wave request) into two 64 KiB buffers laid out like gemm_h_big_kernel's.  Measured on MI355X with the
// DMA stream of this tile sweep alone (scripts/gemm_probe.hip `dma`): 64-byte segments top out at 11-14 TB/s whatever the
// ring depth or request order, 128-byte segments reach 14-19 TB/s -- the L2 serves a 64-byte request at the cost of a
// 128-byte one, and the 32-wide stages of the kernel above are bound by exactly that request rate.
//   stage kt = 2 p + h reads k-half h of pair p (buffer p & 1);  MEM(2p) also requests pair p+1 into the other buffer,
//   MEM(2p+1) ends with s_waitcnt vmcnt(0).
// WAR: the other buffer held pair p-1, read last in group 1's MEM(2p-1) (retired before B(4p-1)); MEM(2p) starts after
//      B(4p-1) (group 0) / B(4p) (group 1).   RAW: both groups have waited for pair p+1 by B(4p+3); it is first read in
//      MEM(2p+2), after B(4p+3) (group 0) / B(4p+4) (group 1).
// --------------------------------------------------------------------------------------------
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm_h_big_pp64_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_p6[];   // [buf][A|W][256][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  constexpr int STAGE = 2 * BBM * TBK;   // elements per pair buffer (A then W)

  const bf16_t* ga[4];
  const bf16_t* gw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (wave * 4 + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (row & 7);
    ga[q] = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
    gw[q] = W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
  }
#define WJ_P6_ISSUE(buf)                                                                                       \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga[q],                     \
        (__attribute__((address_space(3))) void*)(&lds_p6[(buf) * STAGE + (wave * 4 + q) * 8 * TBK]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw[q],                     \
        (__attribute__((address_space(3))) void*)(&lds_p6[(buf) * STAGE + BBM * TBK + (wave * 4 + q) * 8 * TBK]), 16, 0, 0); \
    ga[q] += TBK;                                                                                              \
    gw[q] += TBK;                                                                                              \
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment offsets: row (lane & 15) of a 16-row block, chunk h * 4 + (lane >> 4), XOR-swizzled with row & 7 = lane & 7
  const int rowoff = (lane & 15) * TBK;
  const int off_h0 = rowoff + ((((lane >> 4)) ^ (lane & 7)) << 3);
  const int off_h1 = rowoff + (((4 + (lane >> 4)) ^ (lane & 7)) << 3);
  const int a_base = wm * 128 * TBK, w_base = BBM * TBK + wn * 64 * TBK;

  // split activations (GemmArgs::split, round 4: the Qwen prompt / aligner passes at >= 8192 rows): A rows are [hi(K) | lo(K)],
  // the pair loop runs over 2 K columns of A and the W pointers wrap back to column 0 half way -- the k order (all of hi, then
  // all of lo) and therefore the bits are those of the 128-tile kernel's split mode
  const int np = (g.split ? 2 * g.K : g.K) / TBK;
  WJ_P6_ISSUE(0)
  wait_vmcnt<0>();
  wg_barrier();                  // P: pair 0 visible to every wave
  if (wm == 1) wg_barrier();     // B0: group 1 runs one barrier behind group 0

  typename Vec8<T>::type af[8], wf[4];
#define WJ_P6_READ(buf, off)                                                                                          \
  {                                                                                                                    \
    const bf16_t* ls = &lds_p6[(buf) * STAGE];                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                      \
      wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[w_base + j * 16 * TBK + (off)]);                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                      \
      af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[a_base + i * 16 * TBK + (off)]);                    \
  }
#define WJ_P6_MFMA()                                                         \
  __builtin_amdgcn_s_setprio(1);                                             \
  _Pragma("unroll") for (int i = 0; i < 8; ++i)                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                          \
      if constexpr (EPI == EPI_VT) acc[i][j] = mfma16(af[i], wf[j], acc[i][j]); \
      else acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);                      \
    }                                                                        \
  __builtin_amdgcn_s_setprio(0);

  for (int p = 0; p < np; ++p) {
    const int b = p & 1;
    // ---- MEM(2p): k-half 0 of pair p; request pair p+1
    WJ_P6_READ(b, off_h0)
    if (p + 1 < np) {
      if (g.split && p + 1 == (np >> 1)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gw[q] -= g.K;
      }
      WJ_P6_ISSUE(b ^ 1)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();
    WJ_P6_MFMA()
    wg_barrier();
    // ---- MEM(2p+1): k-half 1 of pair p; pair p+1 has landed
    WJ_P6_READ(b, off_h1)
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();
    WJ_P6_MFMA()
    wg_barrier();
  }
  if (wm == 0) wg_barrier();     // every wave has taken part in the same number of barriers
#undef WJ_P6_ISSUE
#undef WJ_P6_READ
#undef WJ_P6_MFMA

  tile_epilogue<EPI, T, 8>(g, z, m0 + wm * 128, n0 + wn * 64, lane, acc);
}

// --------------------------------------------------------------------------------------------
// Persistent form of gemm_h_big_pp64_kernel (EXPERIMENT, variant 87 of wj_k_gemm; measured 2-5 % slower): one workgroup per
// CU walks tiles L, L + G, L + 2G, ... and treats their k-pairs as ONE stream -- the last MEM(h0) phase of a tile already
// requests pair 0 of the next tile, so no tile but the first pays the LDS-DMA round trip of a prologue, and the epilogue
// stores of tile t drain while the first pair of tile t+1 is in flight.  Barrier sequence, RAW and WAR arguments are those
// of the kernel above with the pair index running across tiles (the epilogue sits at the start of a MEM phase and touches
// no LDS).  Same accumulation order => bit-identical results.
// --------------------------------------------------------------------------------------------
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm_h_big_pp64p_kernel(const GemmArgs g, int nx, int ny, int total) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds_p7[];   // [buf][A|W][256][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int G = gridDim.x, ntiles = nx * ny;
  constexpr int STAGE = 2 * BBM * TBK;
  const bf16_t* __restrict__ Abase = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* __restrict__ W = reinterpret_cast<const bf16_t*>(g.W);

  // tile L of the launch -> batch entry z and tile origin (same XCD-aware order as the non-persistent kernels, per z)
  auto origin = [&](int L, int& z, int& m0, int& n0) {
    z = L / ntiles;
    const int lin = L - z * ntiles;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = lin & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    constexpr int GM = 8;
    const int per_group = GM * nx, group = tile / per_group, first_m = group * GM;
    const int gsz = min(GM, ny - first_m), in_group = tile - group * per_group;
    m0 = (first_m + in_group % gsz) * BBM;
    n0 = (in_group / gsz) * BBN;
  };
  const bf16_t* ga[4];
  const bf16_t* gw[4];
  auto set_ptrs = [&](int z, int m0, int n0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (wave * 4 + q) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      ga[q] = Abase + (int64_t)z * g.a_batch + (int64_t)min(m0 + row, g.M - 1) * g.lda + c * 8;
      gw[q] = W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + c * 8;
    }
  };
#define WJ_P7_ISSUE(buf)                                                                                       \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga[q],                     \
        (__attribute__((address_space(3))) void*)(&lds_p7[(buf) * STAGE + (wave * 4 + q) * 8 * TBK]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw[q],                     \
        (__attribute__((address_space(3))) void*)(&lds_p7[(buf) * STAGE + BBM * TBK + (wave * 4 + q) * 8 * TBK]), 16, 0, 0); \
    ga[q] += TBK;                                                                                              \
    gw[q] += TBK;                                                                                              \
  }
  const int rowoff = (lane & 15) * TBK;
  const int off_h0 = rowoff + ((((lane >> 4)) ^ (lane & 7)) << 3);
  const int off_h1 = rowoff + (((4 + (lane >> 4)) ^ (lane & 7)) << 3);
  const int a_base = wm * 128 * TBK, w_base = BBM * TBK + wn * 64 * TBK;
  const int np = g.K / TBK;

  int L = blockIdx.x, z, m0, n0;
  origin(L, z, m0, n0);
  set_ptrs(z, m0, n0);
  WJ_P7_ISSUE(0)
  wait_vmcnt<0>();
  wg_barrier();                  // P
  if (wm == 1) wg_barrier();     // B0

  typename Vec8<T>::type af[8], wf[4];
  f32x4_t acc[8][4];
#define WJ_P7_READ(buf, off)                                                                                          \
  {                                                                                                                    \
    const bf16_t* ls = &lds_p7[(buf) * STAGE];                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                      \
      wf[j] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[w_base + j * 16 * TBK + (off)]);                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                      \
      af[i] = *reinterpret_cast<const typename Vec8<T>::type*>(&ls[a_base + i * 16 * TBK + (off)]);                    \
  }
#define WJ_P7_MFMA()                                                         \
  __builtin_amdgcn_s_setprio(1);                                             \
  _Pragma("unroll") for (int i = 0; i < 8; ++i)                              \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                          \
      if constexpr (EPI == EPI_VT) acc[i][j] = mfma16(af[i], wf[j], acc[i][j]); \
      else acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);                      \
    }                                                                        \
  __builtin_amdgcn_s_setprio(0);

  int b = 0;
  while (true) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int Ln = L + G;
    const bool has_next = Ln < total;
    int zn = 0, m0n = 0, n0n = 0;
    if (has_next) origin(Ln, zn, m0n, n0n);
    for (int p = 0; p < np; ++p) {
      // ---- MEM(h0): request the next pair of the stream (of this tile, or pair 0 of the next one)
      WJ_P7_READ(b, off_h0)
      if (p + 1 < np) {
        WJ_P7_ISSUE(b ^ 1)
      } else if (has_next) {
        set_ptrs(zn, m0n, n0n);
        WJ_P7_ISSUE(b ^ 1)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wg_barrier();
      WJ_P7_MFMA()
      wg_barrier();
      // ---- MEM(h1)
      WJ_P7_READ(b, off_h1)
      wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wg_barrier();
      WJ_P7_MFMA()
      wg_barrier();
      b ^= 1;
    }
    tile_epilogue<EPI, T, 8>(g, z, m0 + wm * 128, n0 + wn * 64, lane, acc);
    if (!has_next) break;
    L = Ln; z = zn; m0 = m0n; n0 = n0n;
  }
  if (wm == 0) wg_barrier();
#undef WJ_P7_ISSUE
#undef WJ_P7_READ
#undef WJ_P7_MFMA
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
  const int GM = g.gm;        // row panels per group: launch_epi16 sizes the group's A panels for the XCD's L2 (wj_tune tile_l2_kb)
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
    // stage kt visible to all; buffer of stage kt-1 is free (its ds_reads were consumed by the MFMAs of step kt-1).
    // NOT __syncthreads(): its fence drains vmcnt to 0 and with it the stages this loop keeps in flight.
    wg_barrier();
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

  tile_epilogue<EPI, T, 4>(g, z, m0 + wm * 64, n0 + wn * 64, lane, acc);
}

template <typename T, int EPI, int NS>
static int launch_ms_inst(const GemmArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)NS * 2 * TBM * TBK * sizeof(bf16_t);
  static AttrOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_tile_ms_kernel<T, EPI, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%d KiB LDS): %s", (int)(smem >> 10), hipGetErrorString(e)); return WJ_E_HIP; }
    attr_set.done();
  }
  dim3 grid(ceil_div(a.N, TBN), ceil_div(a.M, TBM), EPI == EPI_PARTIAL_F32 ? a.ksplit : a.nbatch);
  GemmArgs b = a;
  b.gm = tile_gm((a.split ? 2 * a.K : a.K) / (EPI == EPI_PARTIAL_F32 ? a.ksplit : 1));
  hipLaunchKernelGGL((gemm_h_tile_ms_kernel<T, EPI, NS>), grid, dim3(256), smem, s, b);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T, int EPI>
static int launch_ms(const GemmArgs& a, hipStream_t s, int ns) {
  if constexpr (EPI == EPI_T || EPI == EPI_GELU_T || EPI == EPI_F32 || EPI == EPI_RESID_F32 || EPI == EPI_QKV_DEC ||
                EPI == EPI_PARTIAL_F32 || EPI == EPI_SWIGLU_T) {
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

// wj_tune("gemm_big"): 0 disables the 256-tile kernels, 1 = lockstep 256-tile kernel, 2 = its issue-order-hinted build,
// 3 / 4 / 5 = the ping-pong kernel with that many 32-wide ring stages, 6 = the ping-pong kernel over 64-wide pairs (default:
// 987 / 1136 / 1068 / 973 TFLOP/s on the encoder's fc1 / fc2 / qk / out shapes against 933 / 1084 / 1016 / 947 for 1 and
// 954 / 1122 / 1043 / 969 for 3; all of them bit-identical, profiles/r02_gemm_probe.json)
int g_gemm_big = 6;
int g_epi_wide = 1;   // wj_tune("epi_wide"): 16-byte epilogue stores (GemmArgs::epi_wide)

template <typename T, int EPI>
static int launch_big(const GemmArgs& a, hipStream_t s) {
  if constexpr (EPI == EPI_PARTIAL_F32) {
    set_error("gemm: the 256-tile kernel has no split-K mode");
    return WJ_E_INVALID;
  } else {
    constexpr size_t smem = 2 * 2 * BBM * TBK * sizeof(bf16_t);   // 128 KiB
    static AttrOnce attr_set;
    if (attr_set.need()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_kernel<T, EPI, 0>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_kernel<T, EPI, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(128 KiB LDS): %s", hipGetErrorString(e)); return WJ_E_HIP; }
      attr_set.done();
    }
    dim3 grid(ceil_div(a.N, BBN), ceil_div(a.M, BBM), a.nbatch);
    if (g_gemm_big == 2) hipLaunchKernelGGL((gemm_h_big_kernel<T, EPI, 1>), grid, dim3(512), smem, s, a);
    else hipLaunchKernelGGL((gemm_h_big_kernel<T, EPI, 0>), grid, dim3(512), smem, s, a);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  }
}

template <typename T, int EPI, int NS>
static int launch_big_pp_inst(const GemmArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)NS * PSTAGE * sizeof(bf16_t);   // NS x 32 KiB
  static AttrOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_pp_kernel<T, EPI, NS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%d KiB LDS): %s", (int)(smem >> 10), hipGetErrorString(e)); return WJ_E_HIP; }
    attr_set.done();
  }
  dim3 grid(ceil_div(a.N, BBN), ceil_div(a.M, BBM), a.nbatch);
  hipLaunchKernelGGL((gemm_h_big_pp_kernel<T, EPI, NS>), grid, dim3(512), smem, s, a);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T, int EPI>
static int launch_big_pp64(const GemmArgs& a, hipStream_t s) {
  if constexpr (EPI == EPI_PARTIAL_F32) {
    set_error("gemm: the 256-tile kernels have no split-K mode");
    return WJ_E_INVALID;
  } else {
    constexpr size_t smem = 2 * 2 * BBM * TBK * sizeof(bf16_t);   // 128 KiB
    static AttrOnce attr_set;
    if (attr_set.need()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_pp64_kernel<T, EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(128 KiB LDS): %s", hipGetErrorString(e)); return WJ_E_HIP; }
      attr_set.done();
    }
    dim3 grid(ceil_div(a.N, BBN), ceil_div(a.M, BBM), a.nbatch);
    hipLaunchKernelGGL((gemm_h_big_pp64_kernel<T, EPI>), grid, dim3(512), smem, s, a);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  }
}

template <typename T, int EPI>
static int launch_big_pp64p(const GemmArgs& a, hipStream_t s) {
  if constexpr (EPI == EPI_PARTIAL_F32) {
    set_error("gemm: the 256-tile kernels have no split-K mode");
    return WJ_E_INVALID;
  } else {
    constexpr size_t smem = 2 * 2 * BBM * TBK * sizeof(bf16_t);   // 128 KiB
    static AttrOnce attr_set;
    static int n_cu = 0;
    if (attr_set.need()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_pp64p_kernel<T, EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      int dev = 0;
      if (e == hipSuccess) e = hipGetDevice(&dev);
      if (e == hipSuccess) e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
      if (e != hipSuccess || n_cu <= 0) { set_error("persistent GEMM setup: %s", hipGetErrorString(e)); return WJ_E_HIP; }
      attr_set.done();
    }
    const int nx = a.N / BBN, ny = ceil_div(a.M, BBM), total = nx * ny * a.nbatch;
    hipLaunchKernelGGL((gemm_h_big_pp64p_kernel<T, EPI>), dim3(std::min(total, n_cu)), dim3(512), smem, s, a, nx, ny, total);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  }
}

template <typename T, int EPI, int NS, int ABL, int PLACE = 0>
static int launch_big_pp_abl(const GemmArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)NS * PSTAGE * sizeof(bf16_t);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_big_pp_kernel<T, EPI, NS, ABL, PLACE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return WJ_E_HIP; }
  dim3 grid(ceil_div(a.N, BBN), ceil_div(a.M, BBM), a.nbatch);
  hipLaunchKernelGGL((gemm_h_big_pp_kernel<T, EPI, NS, ABL, PLACE>), grid, dim3(512), smem, s, a);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T, int EPI>
static int launch_big_pp(const GemmArgs& a, hipStream_t s, int ns) {
  if constexpr (EPI == EPI_PARTIAL_F32) {
    set_error("gemm: the 256-tile kernels have no split-K mode");
    return WJ_E_INVALID;
  } else {
    switch (ns) {
      case 3: return launch_big_pp_inst<T, EPI, 3>(a, s);
      case 5: return launch_big_pp_inst<T, EPI, 5>(a, s);
      default: return launch_big_pp_inst<T, EPI, 4>(a, s);
    }
  }
}


// --------------------------------------------------------------------------------------------
// MX-fp8 GEMM (round 4): C = A8 . W8^T with OCP e4m3 operands and one E8M0 scale per 32-element block on BOTH operands,
// on v_mfma_scale_f32_16x16x128_f8f6f4 (the only large-K low-precision MFMA of gfx950: 2x the fp16 rate, 5 PFLOP/s dense).
// Operand layout of the instruction, verified on the MI355X by scripts/mx_probe.hip (profiles/r04_mx_mfma_layout_probe.jsonl:
// the ISA text is not on this box): lane (r = lane & 15, g = lane >> 4) holds row r of its operand, registers 0-3 = bytes
// k = 16 g .. 16 g + 15 and registers 4-7 = bytes k = 64 + 16 g .. + 15 of the 128-wide step -- i.e. the two 16-byte chunks g
// and 4 + g of a 128-byte row, exactly the two fragment reads the 16-bit tile kernel makes for its two 32-wide halves -- and
// the scale VGPR of lane group g carries (byte 0) the scale of MX block g (k = 32 g .. 32 g + 31) of row r.
// D[lane][i] = sum_k first[4 g + i][k] second[r][k]: with the weights first a lane ends up with 4 consecutive output columns
// of row r, the orientation of every other tile kernel here, so the epilogues are shared.
// Tile 128 x 128, 4 waves, k step 128 bytes: the LDS image, the LDS-DMA staging and the XOR swizzle of gemm_h_tile_kernel with
// a row = 128 bytes of fp8 instead of 64 halves.  Scales come straight from L2 (4 bytes per row and step), one step ahead.
// --------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_mx8_tile_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) uint8_t lds8[2 * 2 * TBM * 128];  // [buf][A|W][128 rows][128 B] = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nx = gridDim.x, ntiles = gridDim.x * gridDim.y;
  const int lin = blockIdx.y * nx + blockIdx.x;
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = lin & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  const int GM = g.gm;        // row panels per group: launch_epi16 sizes the group's A panels for the XCD's L2 (wj_tune tile_l2_kb)
  const int per_group = GM * nx, group = tile / per_group, first_m = group * GM;
  const int gsz = min(GM, (int)gridDim.y - first_m), in_group = tile - group * per_group;
  const int m0 = (first_m + in_group % gsz) * TBM, n0 = (in_group / gsz) * TBN;
  const uint8_t* __restrict__ A = reinterpret_cast<const uint8_t*>(g.A);
  const uint8_t* __restrict__ W = reinterpret_cast<const uint8_t*>(g.W);
  const int nk = g.K / 128, nsc = g.K / 32;      // launch guarantees K % 128 == 0

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define WJ_MX_STAGE(buf, k0)                                                                       \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                  \
    const int r0 = (wave * 4 + q) * 8;                                                             \
    const int row = r0 + (lane >> 3);                                                              \
    const int c = (lane & 7) ^ (row & 7);                                                          \
    const uint8_t* ga = A + (int64_t)min(m0 + row, g.M - 1) * g.lda + (k0) + c * 16;               \
    const uint8_t* gw = W + (int64_t)min(n0 + row, g.N - 1) * g.ldw + (k0) + c * 16;               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,            \
        (__attribute__((address_space(3))) void*)(&lds8[((buf) * 2 + 0) * TBM * 128 + r0 * 128]), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,            \
        (__attribute__((address_space(3))) void*)(&lds8[((buf) * 2 + 1) * TBM * 128 + r0 * 128]), 16, 0, 0); \
  }
  // scale bytes of this lane's rows: A rows wm * 64 + i * 16 + r, W rows wn * 64 + j * 16 + r, MX block kt * 4 + g
  const int r = lane & 15, gq = lane >> 4;
  const uint8_t* sa_p[4];
  const uint8_t* sw_p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sa_p[i] = g.a_scale + (int64_t)min(m0 + wm * 64 + i * 16 + r, g.M - 1) * nsc + gq;
    sw_p[i] = g.w_scale + (int64_t)min(n0 + wn * 64 + i * 16 + r, g.N - 1) * nsc + gq;
  }
  int sa[4], sw[4], sa_n[4], sw_n[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sa[i] = sa_p[i][0]; sw[i] = sw_p[i][0]; }

  WJ_MX_STAGE(0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      WJ_MX_STAGE(cur ^ 1, (kt + 1) * 128)
#pragma unroll
      for (int i = 0; i < 4; ++i) { sa_n[i] = sa_p[i][(kt + 1) * 4]; sw_n[i] = sw_p[i][(kt + 1) * 4]; }
    }
    const uint8_t* la = &lds8[(cur * 2 + 0) * TBM * 128];
    const uint8_t* lb = &lds8[(cur * 2 + 1) * TBM * 128];
    i32x8_t af[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wm * 64 + i * 16 + r;
      const uint4 lo = *reinterpret_cast<const uint4*>(&la[row * 128 + ((gq ^ (row & 7)) << 4)]);
      const uint4 hi = *reinterpret_cast<const uint4*>(&la[row * 128 + (((4 + gq) ^ (row & 7)) << 4)]);
      af[i] = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wn * 64 + j * 16 + r;
      const uint4 lo = *reinterpret_cast<const uint4*>(&lb[row * 128 + ((gq ^ (row & 7)) << 4)]);
      const uint4 hi = *reinterpret_cast<const uint4*>(&lb[row * 128 + (((4 + gq) ^ (row & 7)) << 4)]);
      wf[j] = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (EPI == EPI_VT)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], wf[j], acc[i][j], 0, 0, 0, sa[i], 0, sw[j]);
        else
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af[i], acc[i][j], 0, 0, 0, sw[j], 0, sa[i]);
      }
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { sa[i] = sa_n[i]; sw[i] = sw_n[i]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of step kt+1 has landed
    __syncthreads();
  }
#undef WJ_MX_STAGE
  tile_epilogue<EPI, T, 4>(g, 0, m0 + wm * 64, n0 + wn * 64, lane, acc);
}

template <typename T, int EPI>
static int launch_mx8(const GemmArgs& a, hipStream_t s) {
  if constexpr (EPI == EPI_PARTIAL_F32 || EPI == EPI_QKV_DEC || EPI == EPI_GELU_POS_F32) {
    set_error("gemm: the MX-fp8 kernel carries the plain / residual / head-split epilogues only");
    return WJ_E_INVALID;
  } else {
    if ((a.K % 128) || a.nbatch != 1 || a.split || a.blk || !a.a_scale || !a.w_scale || a.lda < a.K || a.ldw < a.K) {
      set_error("gemm: MX-fp8 operands need K %% 128 == 0, one batch, both scale arrays and byte strides >= K (K=%d)", a.K);
      return WJ_E_INVALID;
    }
    dim3 grid(ceil_div(a.N, TBN), ceil_div(a.M, TBM), 1);
    hipLaunchKernelGGL((gemm_mx8_tile_kernel<T, EPI>), grid, dim3(256), 0, s, a);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  }
}

// One thread per 32-element block: amax -> E8M0 scale 2^(floor(log2 amax) - 8) (OCP MX, e4m3: emax = 8), elements scaled,
// clamped to +-448 and rounded to nearest even by v_cvt_pk_fp8_f32 (gfx950: OCP e4m3).  An all-zero block gets scale 2^-127.
template <typename S>
__global__ __launch_bounds__(256) void mx8_quantize_kernel(const S* __restrict__ src, int64_t ld, int rows, int K,
                                                           uint8_t* __restrict__ out8, uint8_t* __restrict__ scales) {
  const int nb = K / 32;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)rows * nb) return;
  const int row = (int)(idx / nb), b = (int)(idx - (int64_t)row * nb);
  const S* p = src + (int64_t)row * ld + b * 32;
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    if constexpr (sizeof(S) == 4) {
      const float4 x = *reinterpret_cast<const float4*>(p + i), y = *reinterpret_cast<const float4*>(p + i + 4);
      v[i] = x.x; v[i + 1] = x.y; v[i + 2] = x.z; v[i + 3] = x.w; v[i + 4] = y.x; v[i + 5] = y.y; v[i + 6] = y.z; v[i + 7] = y.w;
    } else {
      ld8(p + i, v + i);
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(v[i]));
  int e = -127;
  if (amax > 0.f) {
    e = ((__float_as_int(amax) >> 23) & 0xff) - 127 - 8;      // floor(log2 amax) - emax; subnormal inputs land at the clamp below
    e = max(-127, min(127, e));
  }
  scales[(int64_t)row * nb + b] = (uint8_t)(e + 127);
  const float inv = __int_as_float((127 - e) << 23 > 0 ? (127 - e) << 23 : 0x00400000);      // 2^-e (e <= 127 -> exponent field >= 0)
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float q[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) q[t] = fminf(448.f, fmaxf(-448.f, v[4 * i + t] * inv));
    int packed = 0;
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], packed, false);
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], packed, true);
    w[i] = (uint32_t)packed;
  }
  uint4* o = reinterpret_cast<uint4*>(out8 + (int64_t)row * K + b * 32);
  o[0] = make_uint4(w[0], w[1], w[2], w[3]);
  o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

int launch_mx8_quantize(int src_dtype, const void* src, int64_t ld, int rows, int K, uint8_t* out8, uint8_t* scales, hipStream_t s) {
  if ((K % 32) || (ld % 8) || rows <= 0) { set_error("mx8_quantize: K %% 32 == 0 and ld %% 8 == 0 required"); return WJ_E_INVALID; }
  const unsigned blocks = (unsigned)ceil_div64((int64_t)rows * (K / 32), 256);
  if (src_dtype == WJ_F32) hipLaunchKernelGGL((mx8_quantize_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)src, ld, rows, K, out8, scales);
  else if (src_dtype == WJ_F16) hipLaunchKernelGGL((mx8_quantize_kernel<f16_t>), dim3(blocks), dim3(256), 0, s, (const f16_t*)src, ld, rows, K, out8, scales);
  else hipLaunchKernelGGL((mx8_quantize_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, ld, rows, K, out8, scales);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
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
  if constexpr (EPI == EPI_SWIGLU_T) {      // the MFMA tile kernels only (their shared tile_epilogue carries the fused form)
    if (a.mx8 || a.blk || a.nbatch != 1 || (a.N % 64) || (a.ldc % 8) || (a.K % TBK)) {
      set_error("gemm: the SwiGLU epilogue needs row-major 16-bit operands, one batch, N %% 64 == 0, K %% 64 == 0, ldc %% 8 == 0");
      return WJ_E_INVALID;
    }
    if (variant >= 73 && variant <= 75) return launch_ms<T, EPI>(a, s, variant - 70);
    const bool big_ok = (a.N % BBN) == 0 && a.M >= 1024 && (!a.split || g_gemm_big == 6);
    if (variant == 0 && big_ok && g_gemm_big == 6) return launch_big_pp64<T, EPI>(a, s);
    if (variant == 0 && big_ok && g_gemm_big) return launch_big<T, EPI>(a, s);
    dim3 grid(ceil_div(a.N, TBN), ceil_div(a.M, TBM), 1);
    GemmArgs b = a;
    b.gm = tile_gm(a.split ? 2 * a.K : a.K);
    hipLaunchKernelGGL((gemm_h_tile_kernel<T, EPI, true>), grid, dim3(256), 0, s, b);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  } else {
  if (a.mx8) return launch_mx8<T, EPI>(a, s);
  if (a.blk) return launch_big_ppb<T, EPI>(a, s);
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
  const bool big_ok = EPI != EPI_PARTIAL_F32 && (a.N % BBN) == 0 && (a.K % TBK) == 0 && a.M >= 1024 && (!a.split || g_gemm_big == 6);
  if (variant == 6 && !big_ok) { set_error("gemm: the 256-tile kernel needs N %% 256 == 0, K %% 64 == 0, M >= 1024"); return WJ_E_INVALID; }
  // 83 / 84 / 85 force the ping-pong kernel with a 3 / 4 / 5 stage ring; wj_tune("gemm_big", 3 / 4 / 5) makes it the default
  const bool pp_ok = big_ok && !a.split && (a.K % PBK) == 0 && a.K / PBK >= 5;
  if (a.split && (variant == 6 || variant == 87)) { set_error("gemm: of the 256-tile kernels only the pairs kernel (86) takes split activations"); return WJ_E_INVALID; }
  if ((variant >= 100 && variant < 270) || variant == 93 || variant == 94) {   // timing ablations of the ping-pong kernel (wrong results by design): 100 + 10 ABL + NS
    if constexpr (EPI == EPI_T && Elem<T>::dtype == WJ_F16) {
      if (!pp_ok) { set_error("gemm: shape not supported by the ping-pong kernel"); return WJ_E_INVALID; }
      switch (variant) {
#define WJ_ABL(A_) case 100 + 10 * A_ + 3: return launch_big_pp_abl<T, EPI, 3, A_>(a, s);
        WJ_ABL(1) WJ_ABL(2) WJ_ABL(3) WJ_ABL(4) WJ_ABL(5) WJ_ABL(6) WJ_ABL(8) WJ_ABL(9) WJ_ABL(10) WJ_ABL(12) WJ_ABL(15)
#undef WJ_ABL
        case 93: return launch_big_pp_abl<T, EPI, 3, 0, 1>(a, s);    // experiments with the placement of the LDS-DMA requests
        case 94: return launch_big_pp_abl<T, EPI, 4, 0, 1>(a, s);    // (correct results)
        case 263: return launch_big_pp_abl<T, EPI, 3, 0, 2>(a, s);
        case 264: return launch_big_pp_abl<T, EPI, 4, 0, 2>(a, s);
        case 265: return launch_big_pp_abl<T, EPI, 5, 0, 2>(a, s);
        default: break;
      }
    }
    set_error("gemm: unknown ablation variant %d (float16 EPI_T only)", variant);
    return WJ_E_INVALID;
  }
  if (variant >= 83 && variant <= 85) {
    if (!pp_ok) { set_error("gemm: the ping-pong 256-tile kernel needs N %% 256 == 0, K %% 64 == 0, K >= 160, M >= 1024"); return WJ_E_INVALID; }
    return launch_big_pp<T, EPI>(a, s, variant - 80);
  }
  if (variant == 87) {   // persistent form of 86: a measured, rejected experiment, reachable from wj_k_gemm only
    if constexpr (EPI == EPI_T || EPI == EPI_GELU_T || EPI == EPI_F32) {
      if (!big_ok) { set_error("gemm: the 256-tile kernel needs N %% 256 == 0, K %% 64 == 0, M >= 1024"); return WJ_E_INVALID; }
      return launch_big_pp64p<T, EPI>(a, s);
    } else {
      set_error("gemm: variant 87 carries the plain epilogues only");
      return WJ_E_INVALID;
    }
  }
  if (variant == 86) {   // ping-pong over 64-wide pairs (128-byte DMA segments); wj_tune("gemm_big", 6)
    if (!big_ok) { set_error("gemm: the 256-tile kernel needs N %% 256 == 0, K %% 64 == 0, M >= 1024"); return WJ_E_INVALID; }
    return launch_big_pp64<T, EPI>(a, s);
  }
  if ((variant == 0 || variant == 1) && g_gemm_big == 6 && big_ok) return launch_big_pp64<T, EPI>(a, s);
  if ((variant == 0 || variant == 1) && g_gemm_big >= 3 && g_gemm_big <= 5 && pp_ok) return launch_big_pp<T, EPI>(a, s, g_gemm_big);
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
  GemmArgs b = a;
  b.gm = tile_gm(kchunk);
  if (glds) hipLaunchKernelGGL((gemm_h_tile_kernel<T, EPI, true>), grid, dim3(256), 0, s, b);
  else hipLaunchKernelGGL((gemm_h_tile_kernel<T, EPI, false>), grid, dim3(256), 0, s, b);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
  }
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
    if constexpr (EPI == EPI_SWIGLU_T) {
      set_error("gemm: the fused SwiGLU epilogue is a 16-bit-path feature");
      return WJ_E_INVALID;
    } else {
      dim3 grid(ceil_div(a.N, 64), ceil_div(a.M, 64), a.nbatch);
      hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, dim3(256), 0, s, a);
      WJ_LAUNCH_CHECK();
      return WJ_OK;
    }
  }
  if (dtype == WJ_F16) return launch_epi16<f16_t, EPI>(a, s, variant);
  return launch_epi16<bf16_t, EPI>(a, s, variant);
}

int launch_gemm(int dtype, Epi epi, const GemmArgs& a_in, hipStream_t s, int variant) {
  GemmArgs a = a_in;
  a.epi_wide = g_epi_wide;
  if (a.blk && !is16(dtype)) { set_error("gemm: blocked operands are a 16-bit feature"); return WJ_E_INVALID; }
  if (a.mx8 && !is16(dtype)) { set_error("gemm: MX-fp8 operands take a 16-bit output type"); return WJ_E_INVALID; }
  if (a.out_blk && (!a.blk || (epi != EPI_T && epi != EPI_GELU_T) || (a.N % 32))) {
    set_error("gemm: a blocked output needs blocked operands, an EPI_T / EPI_GELU_T epilogue and N %% 32 == 0");
    return WJ_E_INVALID;
  }
  if (a.seq_T) {
    if (a.seq_T < 4 || (a.seq_T % 4) || a.M >= (1 << 24) || a.nbatch != 1) {
      set_error("gemm: flat rows need seq_T %% 4 == 0, M < 2^24 and one batch (seq_T=%d M=%d)", a.seq_T, a.M);
      return WJ_E_INVALID;
    }
    a.inv_seq_T = 1.0f / (float)a.seq_T;
  }
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
  if (a.split_out && epi == EPI_SWIGLU_T) {
    if (!is16(dtype) || a.ldc < (int64_t)a.N) { set_error("gemm: split_out of the SwiGLU epilogue needs ldc >= N (two halves of N / 2)"); return WJ_E_INVALID; }
  } else
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
    case EPI_SWIGLU_T: return launch_epi<EPI_SWIGLU_T>(dtype, a, s, variant);
    default: set_error("gemm: unknown epilogue %d", (int)epi); return WJ_E_INVALID;
  }
}

}  // namespace wj
