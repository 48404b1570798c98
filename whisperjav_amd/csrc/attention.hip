// attention.hip -- encoder self-attention (flash style, MFMA bf16 / exact fp32) and decode-step
// attention (HBM-bound streaming of K/V shared by the beams of one window).
//
// Encoder layouts (written by the QKV GEMM epilogues, see gemm.hip):
//   Q, K : [B][H][Tpad][64]     Vt : [B][H][64][Tpad]     Tpad = round_up(T, 64), pad = zeros
// bf16 kernel, per workgroup = 128 query rows of one (b, h), 4 waves x 32 rows:
//   S^T = K . Q^T   (A = K fragment with a permuted key order, B = Q fragment held in VGPRs)
//   so that after the MFMA a lane already holds, for ITS query column, 8 *consecutive* keys per
//   32-key step -- exactly the B operand of  O^T += Vt . P^T  with no cross-lane movement, and
//   the online-softmax statistics (m, l, alpha) are lane-local per query.
#include <type_traits>

#include "kernels.hpp"

namespace wj {

template <typename T> struct Lane16;                       // scalar type of an MFMA operand element
template <> struct Lane16<bf16_t> { typedef __bf16 type; };
template <> struct Lane16<f16_t> { typedef _Float16 type; };

// max over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} in registers: v_permlane16_swap / v_permlane32_swap exchange 16- resp.
// 32-lane rows between two VGPRs (with both operands = x, lane l of the pair holds x[l] and x[l ^ 16] resp. x[l ^ 32]).
// __shfl_xor compiles to ds_bpermute_b32 here -- an LDS round trip with an s_waitcnt lgkmcnt(0) behind it, four of them in a
// row per key tile of the encoder attention (seen in the round-4 disassembly: ~400 cycles of exposed latency per tile and wave).
__device__ __forceinline__ float quad_row_max(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t v = __float_as_uint(x);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

__device__ __forceinline__ int aswz(int row, int chunk) { return row * 64 + ((chunk ^ (row & 7)) << 3); }
// K tile of the encoder attention: key k of the 64-key tile is stored at the LDS row the MFMA A-operand reads it from
// (block kb = 2 (k >> 5) + ((k >> 2) & 1), row 4 ((k >> 3) & 3) + (k & 3) of it), so a fragment read touches 16
// CONSECUTIVE rows like a GEMM operand; the XOR key folds in the block so the staging writes do not collide either
// (PMC before: 22 % of the kernel's LDS cycles were bank conflicts from the scattered key rows).
__device__ __forceinline__ int kperm(int k) { return 16 * (2 * (k >> 5) + ((k >> 2) & 1)) + 4 * ((k >> 3) & 3) + (k & 3); }
__device__ __forceinline__ int kswz(int prow, int chunk) { return prow * 64 + ((chunk ^ ((prow ^ (prow >> 4)) & 7)) << 3); }

template <typename E, int VAR>
__global__ __launch_bounds__(256) void attn_enc_h_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                         const bf16_t* __restrict__ Vt, E* __restrict__ out,
                                                         int T, int Tpad, int H, int out_blk) {
  typedef typename Vec8<E>::type vec8_t;       // the operand vector of this instantiation (bf16 or fp16 lanes)
  typedef typename Lane16<E>::type lane_t;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 2 * 64 * 64];  // [buf][K|Vt][64][64] = 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  // XCD-aware order: workgroup i runs on XCD i % 8 with a private L2.  Without the remap the ~12 query
  // blocks of one (window, head) are spread over all XCDs and each L2 fetches that head's K/V separately
  // (PMC: 4x the algorithmic bytes); with it each XCD walks whole heads and K/V is fetched once.
  const int nqb = gridDim.x, total = gridDim.x * gridDim.y * gridDim.z;
  const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7;
  const int id = (VAR & 1) ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3) : lin;
  const int qb = id % nqb, h = (id / nqb) % H, b = id / (nqb * H);
  const int q0 = qb * 128 + wave * 32;
  const int64_t bh = (int64_t)b * H + h;
  const bf16_t* Qp = Q + bh * Tpad * 64;
  const bf16_t* Kp = K + bh * Tpad * 64;
  const bf16_t* Vp = Vt + bh * 64 * Tpad;

  vec8_t qf[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[f][ks] = *reinterpret_cast<const vec8_t*>(Qp + (int64_t)(q0 + f * 16 + li) * 64 + ks * 32 + lg * 8);

  f32x4_t o[2][4];
  f32x4_t lsum[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};   // VAR & 8: row sums from the MFMA pipe
  vec8_t ones8;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones8[e] = (lane_t)1.0f;
  float m_run[2], l_run[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    m_run[f] = -INFINITY;
    l_run[f] = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) o[f][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const int nt = Tpad / 64;
  uint4 rk0, rk1, rv0, rv1;  // individually named: an indexed array would live in scratch
#define WJ_ALOAD1(i, kt)                                                                          \
  {                                                                                               \
    const int idx = tid + (i) * 256;                                                              \
    const int row = idx >> 3, ch = idx & 7;                                                       \
    rk##i = *reinterpret_cast<const uint4*>(Kp + (int64_t)((kt) * 64 + row) * 64 + ch * 8);       \
    rv##i = *reinterpret_cast<const uint4*>(Vp + (int64_t)row * Tpad + (kt) * 64 + ch * 8);       \
  }
#define WJ_ALOAD(kt) WJ_ALOAD1(0, kt) WJ_ALOAD1(1, kt)
#define WJ_ASTORE1(i, buf)                                                                        \
  {                                                                                               \
    const int idx = tid + (i) * 256;                                                              \
    const int row = idx >> 3, ch = idx & 7;                                                       \
    *reinterpret_cast<uint4*>(&lds[((buf) * 2 + 0) * 4096 + kswz(kperm(row), ch)]) = rk##i;       \
    *reinterpret_cast<uint4*>(&lds[((buf) * 2 + 1) * 4096 + aswz(row, ch)]) = rv##i;              \
  }
#define WJ_ASTORE(buf) WJ_ASTORE1(0, buf) WJ_ASTORE1(1, buf)

  WJ_ALOAD(0)
  WJ_ASTORE(0)
  __syncthreads();

  // One key tile.  MASKED = the tile may contain padding keys (with Tpad = round_up(T, 64) only the LAST tile can).  The
  // two instantiations are called from a loop over the full tiles and a loop over the rest: written as a branch on
  // `(kt + 1) * 64 > T` inside one loop body, hipcc if-converts the masking into 64 v_cmp / v_cndmask per tile on EVERY
  // tile (a quarter of the loop's VALU work, seen in the device assembly).
  auto key_tile = [&](const int kt, auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int cur = kt & 1;
    if (kt + 1 < nt) { WJ_ALOAD(kt + 1) }
    const bf16_t* lk = &lds[(cur * 2 + 0) * 4096];
    const bf16_t* lv = &lds[(cur * 2 + 1) * 4096];

    // ---- S^T = K . Q^T for 4 blocks of 16 (permuted) keys ----
    f32x4_t st[2][4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      // A-operand row li of block kb is key  32*(kb>>1) + 8*(li>>2) + 4*(kb&1) + (li&3)
      const int krow = 16 * kb + li;   // LDS row (see kperm)
      st[0][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      st[1][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const vec8_t kf = *reinterpret_cast<const vec8_t*>(&lk[kswz(krow, ks * 4 + lg)]);
        st[0][kb] = mfma16(kf, qf[0][ks], st[0][kb]);
        st[1][kb] = mfma16(kf, qf[1][ks], st[1][kb]);
      }
    }
    // lane (q = li, lg) now holds, for block kb, keys  kt*64 + 32*(kb>>1) + 8*lg + 4*(kb&1) + r
    vec8_t pf[2][2];
    if constexpr (MASKED) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 64 + 32 * (kb >> 1) + 8 * lg + 4 * (kb & 1) + r;
            if (key >= T) st[f][kb][r] = -INFINITY;
          }
    }
    if constexpr (VAR & 8) {
      // Lean softmax.  The running maximum is kept on the RAW scores and only moves when it is exceeded by more than
      // kSlack (in base-2 exponent units after scaling): probabilities may then reach 2^kSlack, which fp32 sums and
      // bf16 P hold without trouble, and the accumulator rescale (32 multiplies + its register traffic) runs on
      // a few tiles instead of every tile.  p = exp2(fma(s, c, -m c)): no separate scale pass.  Row sums come
      // out of the MFMA pipe (ones block appended to V^T) instead of 32 VALU adds per tile.
      constexpr float c2 = 0.125f * 1.44269504088896340736f;
      constexpr float kSlack = 8.0f;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[f][kb][r]);
        mx = quad_row_max(mx);
        const float mxs = mx * c2;
        if (mxs > m_run[f] + kSlack || m_run[f] == -INFINITY) {      // rare after the first tiles
          const float m_new = fmaxf(m_run[f], mxs);
          const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
          m_run[f] = m_new;
#pragma unroll
          for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[f][d][r] *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) lsum[f][r] *= alpha;
        }
        const float nm = -m_run[f];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          vec8_t v;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = (lane_t)__builtin_amdgcn_exp2f(fmaf(st[f][2 * s2][r], c2, nm));
            v[4 + r] = (lane_t)__builtin_amdgcn_exp2f(fmaf(st[f][2 * s2 + 1][r], c2, nm));
          }
          pf[f][s2] = v;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {     // row sums of the bf16 probabilities: ones . P^T
        lsum[0] = mfma16(ones8, pf[0][s2], lsum[0]);
        lsum[1] = mfma16(ones8, pf[1][s2], lsum[1]);
      }
    } else {
    // VAR & 2: softmax in base 2 on the raw v_exp_f32 (scale folds 1/sqrt(64) * log2 e); else natural exp
    constexpr float kScaleLog2 = (VAR & 2) ? 0.125f * 1.44269504088896340736f : 0.125f;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = st[f][kb][r] * kScaleLog2;
          st[f][kb][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[f], mx);
      const float alpha = (VAR & 2) ? __builtin_amdgcn_exp2f(m_run[f] - m_new) : __expf(m_run[f] - m_new);
      m_run[f] = m_new;
      float psum = 0.f;
      float p[4][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[kb][r] = (VAR & 2) ? __builtin_amdgcn_exp2f(st[f][kb][r] - m_new) : __expf(st[f][kb][r] - m_new);
          psum += p[kb][r];
        }
      l_run[f] = l_run[f] * alpha + psum;
      if (!(VAR & 4) || alpha != 1.0f) {   // VAR & 4: skip the O rescale when the running max did not move
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[f][d][r] *= alpha;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vec8_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = (lane_t)p[2 * s2][r];
          v[4 + r] = (lane_t)p[2 * s2 + 1][r];
        }
        pf[f][s2] = v;
      }
    }
    }
    // ---- O^T += Vt . P^T ----
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const vec8_t vf = *reinterpret_cast<const vec8_t*>(&lv[aswz(d * 16 + li, s2 * 4 + lg)]);
        o[0][d] = mfma16(vf, pf[0][s2], o[0][d]);
        o[1][d] = mfma16(vf, pf[1][s2], o[1][d]);
      }
    if (kt + 1 < nt) { WJ_ASTORE(cur ^ 1) }
    __syncthreads();
  };
  const int nfull = min(nt, T / 64);          // tiles made of real keys only
  for (int kt = 0; kt < nfull; ++kt) key_tile(kt, std::false_type{});
  for (int kt = nfull; kt < nt; ++kt) key_tile(kt, std::true_type{});
#undef WJ_ALOAD
#undef WJ_ASTORE
#undef WJ_ALOAD1
#undef WJ_ASTORE1

#pragma unroll
  for (int f = 0; f < 2; ++f) {
    float l;
    if constexpr (VAR & 8) {
      l = lsum[f][0];                      // every row of the ones-block result holds the full sum of column li
    } else {
      l = l_run[f];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    const int t = q0 + f * 16 + li;
    if (t < T) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float v[4] = {o[f][d][0] * inv, o[f][d][1] * inv, o[f][d][2] * inv, o[f][d][3] * inv};
        if (out_blk) {   // blocked GEMM-operand layout [rows / 256][D / 32][256][32] (GemmArgs::blk): the out-projection's A
          const int row = b * T + t, c = h * 64 + d * 16 + lg * 4;
          st4(out + (((int64_t)(row >> 8) * (H * 2) + (c >> 5)) << 13) + ((row & 255) << 5) + (c & 31), v);
        } else {
          st4(out + ((int64_t)b * T + t) * (H * 64) + h * 64 + d * 16 + lg * 4, v);
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
// Software-pipelined form (round 4; wj_tune attn_enc_variant bit 4): the same tile arithmetic in the same order as the lean
// variant of attn_enc_h_kernel above (bit-identical output), but the S^T = K.Q^T MFMAs of key tile t+1 are issued in the same
// straight-line block as the softmax VALU work of tile t, on a second score register set, so ONE wave keeps the matrix pipe busy
// while its exponentials run (the loop above serialises 16 MFMAs -> ~150 VALU -> 20 MFMAs per wave and relies on other waves to
// fill the gaps; with 166 VGPRs only three fit a SIMD and the measured matrix-pipe utilisation is 38 %).  K tiles live in a
// ring of three LDS slots (tile t+1 must be resident while tile t is being consumed), V tiles in two.
//   iteration t:  global loads of K(t+2), V(t+1) -> registers;  S_next = K(t+1).Q^T  ||  P = softmax(S_cur);
//                 O^T += V(t)^T.P^T;  registers -> LDS;  barrier;  S_cur <- S_next.
// WAR: slot of K(t+2) held K(t-1), last read in iteration t-2; slot of V(t+1) held V(t-1), last read in iteration t-1 -- both
// behind the barrier that ended iteration t-1.  RAW: stored before the barrier of iteration t, read in iteration t+1.
// --------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void attn_enc_p_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                         const bf16_t* __restrict__ Vt, E* __restrict__ out,
                                                         int T, int Tpad, int H, int out_blk) {
  typedef typename Vec8<E>::type vec8_t;
  typedef typename Lane16<E>::type lane_t;
  __shared__ __attribute__((aligned(16))) bf16_t lds[5 * 4096];  // K ring [3][64][64], then V [2][64][64] = 40 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int nqb = gridDim.x, total = gridDim.x * gridDim.y * gridDim.z;
  const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
  const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7;
  const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  const int qb = id % nqb, h = (id / nqb) % H, b = id / (nqb * H);
  const int q0 = qb * 128 + wave * 32;
  const int64_t bh = (int64_t)b * H + h;
  const bf16_t* Qp = Q + bh * Tpad * 64;
  const bf16_t* Kp = K + bh * Tpad * 64;
  const bf16_t* Vp = Vt + bh * 64 * Tpad;

  vec8_t qf[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[f][ks] = *reinterpret_cast<const vec8_t*>(Qp + (int64_t)(q0 + f * 16 + li) * 64 + ks * 32 + lg * 8);
  f32x4_t o[2][4];
  f32x4_t lsum[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  vec8_t ones8;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones8[e] = (lane_t)1.0f;
  float m_run[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int d = 0; d < 4; ++d) o[f][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = Tpad / 64;
  const int row0 = tid >> 3, ch0 = tid & 7;            // this thread's two 16-byte granules of a 64 x 64 tile: rows row0, row0 + 32
  uint4 rk0, rk1, rv0, rv1;
#define WJ_PK_LOAD(kt)                                                                                   \
  {                                                                                                      \
    rk0 = *reinterpret_cast<const uint4*>(Kp + (int64_t)((kt) * 64 + row0) * 64 + ch0 * 8);              \
    rk1 = *reinterpret_cast<const uint4*>(Kp + (int64_t)((kt) * 64 + row0 + 32) * 64 + ch0 * 8);         \
  }
#define WJ_PV_LOAD(kt)                                                                                   \
  {                                                                                                      \
    rv0 = *reinterpret_cast<const uint4*>(Vp + (int64_t)row0 * Tpad + (kt) * 64 + ch0 * 8);              \
    rv1 = *reinterpret_cast<const uint4*>(Vp + (int64_t)(row0 + 32) * Tpad + (kt) * 64 + ch0 * 8);       \
  }
#define WJ_PK_STORE(slot)                                                                                \
  {                                                                                                      \
    *reinterpret_cast<uint4*>(&lds[(slot) * 4096 + kswz(kperm(row0), ch0)]) = rk0;                        \
    *reinterpret_cast<uint4*>(&lds[(slot) * 4096 + kswz(kperm(row0 + 32), ch0)]) = rk1;                   \
  }
#define WJ_PV_STORE(slot)                                                                                \
  {                                                                                                      \
    *reinterpret_cast<uint4*>(&lds[(3 + (slot)) * 4096 + aswz(row0, ch0)]) = rv0;                         \
    *reinterpret_cast<uint4*>(&lds[(3 + (slot)) * 4096 + aswz(row0 + 32, ch0)]) = rv1;                    \
  }
  // S^T = K . Q^T of the K tile in ring slot `slot`
  auto scores = [&](int slot, f32x4_t (&st)[2][4]) __attribute__((always_inline)) {
    const bf16_t* lk = &lds[slot * 4096];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int krow = 16 * kb + li;
      st[0][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      st[1][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const vec8_t kf = *reinterpret_cast<const vec8_t*>(&lk[kswz(krow, ks * 4 + lg)]);
        st[0][kb] = mfma16(kf, qf[0][ks], st[0][kb]);
        st[1][kb] = mfma16(kf, qf[1][ks], st[1][kb]);
      }
    }
  };
  WJ_PK_LOAD(0)
  WJ_PV_LOAD(0)
  WJ_PK_STORE(0)
  WJ_PV_STORE(0)
  if (nt > 1) {
    WJ_PK_LOAD(1)
    WJ_PK_STORE(1)
  }
  __syncthreads();
  f32x4_t sa[2][4], sb[2][4];
  scores(0, sa);

  // one key tile: `cur` = scores of tile kt (consumed), `nxt` = scores of tile kt + 1 (produced)
  auto key_tile = [&](const int kt, f32x4_t (&cur)[2][4], f32x4_t (&nxt)[2][4], auto masked_tag) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    if (kt + 2 < nt) { WJ_PK_LOAD(kt + 2) }
    if (kt + 1 < nt) { WJ_PV_LOAD(kt + 1) }
    const bf16_t* lv = &lds[(3 + (kt & 1)) * 4096];
    vec8_t pf[2][2];
    if constexpr (MASKED) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 64 + 32 * (kb >> 1) + 8 * lg + 4 * (kb & 1) + r;
            if (key >= T) cur[f][kb][r] = -INFINITY;
          }
    }
    constexpr float c2 = 0.125f * 1.44269504088896340736f;
    constexpr float kSlack = 8.0f;
    // running maximum and the (rare) accumulator rescale of both query fragments first: the branches end here
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, cur[f][kb][r]);
      mx = quad_row_max(mx);
      const float mxs = mx * c2;
      if (mxs > m_run[f] + kSlack || m_run[f] == -INFINITY) {
        const float m_new = fmaxf(m_run[f], mxs);
        const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
        m_run[f] = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[f][d][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) lsum[f][r] *= alpha;
      }
    }
    // ONE straight-line block: the 16 score MFMAs of the NEXT tile (unconditional -- past the last tile they read a stale ring
    // slot and their result is dropped -- so that no branch separates them from the VALU work) beside the 32 fma + 32 exp2 +
    // 16 cvt_pk of THIS tile's probabilities; the group barriers below ask the scheduler for 1 ds_read : 2 MFMA : 10 VALU
    scores((kt + 1) % 3, nxt);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float nm = -m_run[f];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        vec8_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = (lane_t)__builtin_amdgcn_exp2f(fmaf(cur[f][2 * s2][r], c2, nm));
          v[4 + r] = (lane_t)__builtin_amdgcn_exp2f(fmaf(cur[f][2 * s2 + 1][r], c2, nm));
        }
        pf[f][s2] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS_READ: one K fragment
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // the two MFMAs it feeds
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);    // 10 of the 80 VALU instructions
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      lsum[0] = mfma16(ones8, pf[0][s2], lsum[0]);
      lsum[1] = mfma16(ones8, pf[1][s2], lsum[1]);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const vec8_t vf = *reinterpret_cast<const vec8_t*>(&lv[aswz(d * 16 + li, s2 * 4 + lg)]);
        o[0][d] = mfma16(vf, pf[0][s2], o[0][d]);
        o[1][d] = mfma16(vf, pf[1][s2], o[1][d]);
      }
    if (kt + 2 < nt) { WJ_PK_STORE((kt + 2) % 3) }
    if (kt + 1 < nt) { WJ_PV_STORE((kt + 1) & 1) }
    __syncthreads();
  };
  const int nfull = min(nt, T / 64);
  int kt = 0;
  for (; kt + 1 < nfull; kt += 2) {
    key_tile(kt, sa, sb, std::false_type{});
    key_tile(kt + 1, sb, sa, std::false_type{});
  }
  // the rest, one by one with static register roles: an odd unmasked tile, then the tiles that may hold padding keys
  for (; kt < nt; ++kt) {
    const bool masked = kt >= nfull;
    if ((kt & 1) == 0) { if (masked) key_tile(kt, sa, sb, std::true_type{}); else key_tile(kt, sa, sb, std::false_type{}); }
    else { if (masked) key_tile(kt, sb, sa, std::true_type{}); else key_tile(kt, sb, sa, std::false_type{}); }
  }
#undef WJ_PK_LOAD
#undef WJ_PV_LOAD
#undef WJ_PK_STORE
#undef WJ_PV_STORE

#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const float inv = 1.0f / lsum[f][0];
    const int t = q0 + f * 16 + li;
    if (t < T) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float v[4] = {o[f][d][0] * inv, o[f][d][1] * inv, o[f][d][2] * inv, o[f][d][3] * inv};
        if (out_blk) {
          const int row = b * T + t, c = h * 64 + d * 16 + lg * 4;
          st4(out + (((int64_t)(row >> 8) * (H * 2) + (c >> 5)) << 13) + ((row & 255) << 5) + (c & 31), v);
        } else {
          st4(out + ((int64_t)b * T + t) * (H * 64) + h * 64 + d * 16 + lg * 4, v);
        }
      }
    }
  }
}

// Exact fp32 encoder attention: one query row per lane, K / Vt tiles broadcast from LDS.
__global__ __launch_bounds__(64) void attn_enc_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ Vt, float* __restrict__ out, int T,
                                                          int Tpad, int H) {
  __shared__ __attribute__((aligned(16))) float ks[32][64];
  __shared__ __attribute__((aligned(16))) float vs[64][32];
  const int lane = threadIdx.x;
  const int b = blockIdx.z, h = blockIdx.y;
  const int t = blockIdx.x * 64 + lane;
  const int64_t bh = (int64_t)b * H + h;
  const float* Qp = Q + bh * Tpad * 64;
  const float* Kp = K + bh * Tpad * 64;
  const float* Vp = Vt + bh * 64 * Tpad;
  float q[64], o[64];
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const float4 v = *reinterpret_cast<const float4*>(Qp + (int64_t)t * 64 + d);
    q[d] = v.x * 0.125f; q[d + 1] = v.y * 0.125f; q[d + 2] = v.z * 0.125f; q[d + 3] = v.w * 0.125f;
    o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < T; k0 += 32) {
    __syncthreads();
    for (int i = lane; i < 32 * 16; i += 64) {  // K tile: 32 keys x 64 dims, float4 granules
      const int r = i >> 4, c4 = (i & 15) * 4;
      *reinterpret_cast<float4*>(&ks[r][c4]) = *reinterpret_cast<const float4*>(Kp + (int64_t)(k0 + r) * 64 + c4);
    }
    for (int i = lane; i < 64 * 8; i += 64) {   // Vt tile: 64 dims x 32 keys
      const int r = i >> 3, c4 = (i & 7) * 4;
      *reinterpret_cast<float4*>(&vs[r][c4]) = *reinterpret_cast<const float4*>(Vp + (int64_t)r * Tpad + k0 + c4);
    }
    __syncthreads();
    const int kn = min(32, T - k0);
    for (int j = 0; j < kn; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) s = fmaf(q[d], ks[j][d], s);
      const float m_new = fmaxf(m, s);
      const float alpha = expf(m - m_new);
      const float p = expf(s - m_new);
      l = l * alpha + p;
      m = m_new;
#pragma unroll
      for (int d = 0; d < 64; ++d) o[d] = fmaf(p, vs[d][j], o[d] * alpha);
    }
  }
  if (t < T) {
    const float inv = 1.0f / l;
    float* op = out + ((int64_t)b * T + t) * (H * 64) + h * 64;
#pragma unroll
    for (int d = 0; d < 64; d += 4)
      *reinterpret_cast<float4*>(op + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
  }
}

int g_attn_enc_variant = 9;   // wj_tune("attn_enc_variant"): bit0 XCD remap, bit1 base-2 softmax, bit2 lazy rescale, bit3 lean softmax

int launch_attention_enc(int dtype, const void* Q, const void* K, const void* Vt, void* out, int B, int T, int Tpad,
                         int H, hipStream_t s, int out_blk) {
  if (Tpad % 128 || Tpad < T) { set_error("attention_enc: Tpad must be a multiple of 128 and >= T"); return WJ_E_INVALID; }
  if (out_blk && dtype == WJ_F32) { set_error("attention_enc: a blocked output is a 16-bit feature"); return WJ_E_INVALID; }
  if (dtype == WJ_F32) {
    dim3 grid(Tpad / 64, H, B);
    hipLaunchKernelGGL(attn_enc_f32_kernel, grid, dim3(64), 0, s, (const float*)Q, (const float*)K, (const float*)Vt,
                       (float*)out, T, Tpad, H);
  } else {
    dim3 grid(Tpad / 128, H, B);
#define WJ_ATTN(V)                                                                                                 \
  do {                                                                                                             \
    if (dtype == WJ_F16)                                                                                           \
      hipLaunchKernelGGL((attn_enc_h_kernel<f16_t, V>), grid, dim3(256), 0, s, (const bf16_t*)Q, (const bf16_t*)K, \
                         (const bf16_t*)Vt, (f16_t*)out, T, Tpad, H, out_blk);                                     \
    else                                                                                                           \
      hipLaunchKernelGGL((attn_enc_h_kernel<bf16_t, V>), grid, dim3(256), 0, s, (const bf16_t*)Q, (const bf16_t*)K, \
                         (const bf16_t*)Vt, (bf16_t*)out, T, Tpad, H, out_blk);                                    \
  } while (0)
    if (g_attn_enc_variant & 16) {      // software-pipelined kernel (lean softmax, XCD remap: the arithmetic of variant 9)
      if (dtype == WJ_F16)
        hipLaunchKernelGGL((attn_enc_p_kernel<f16_t>), grid, dim3(256), 0, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, (f16_t*)out, T, Tpad, H, out_blk);
      else
        hipLaunchKernelGGL((attn_enc_p_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)out, T, Tpad, H, out_blk);
      WJ_LAUNCH_CHECK();
      return WJ_OK;
    }
    switch (g_attn_enc_variant & 15) {
      case 0: WJ_ATTN(0); break;
      case 1: WJ_ATTN(1); break;
      case 2: WJ_ATTN(2); break;
      case 3: WJ_ATTN(3); break;
      case 4: WJ_ATTN(4); break;
      case 5: WJ_ATTN(5); break;
      case 6: WJ_ATTN(6); break;
      case 7: WJ_ATTN(7); break;
      case 8: WJ_ATTN(8); break;
      default: WJ_ATTN(9); break;
    }
#undef WJ_ATTN
  }
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

// --------------------------------------------------------------------------------------------
// decode-step attention: NB query rows (the beams of one window, or 1 row for self-attention)
// share one K/V set; HBM-bound.  lane = (key-in-octet kk, 8-wide feature chunk c): a wave reads
// 8 keys x 128 B = 1 KiB per instruction, fully coalesced.  Two passes (scores -> LDS, exact
// max; then exp / P.V), fp32 throughout.
// --------------------------------------------------------------------------------------------
template <typename T, int NB, int NW, bool SELF>
__global__ __launch_bounds__(NW * 64) void attn_dec_kernel(const DecAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 3, c = lane & 7;
  const int h = blockIdx.x, gi = blockIdx.y;
  const int n_keys = (SELF && a.seq_tp > 0) ? gi % a.seq_tp + 1 : (a.n_keys_ptr ? (*a.n_keys_ptr + 1) : a.n_keys);
  const int kpad = (((SELF && a.seq_tp > 0) ? a.seq_tp : n_keys) + 7) & ~7;
  float* sc = smem;                         // [NB][kpad]
  float* red_m = sc + NB * kpad;            // [NW][NB]
  float* red_l = red_m + NW * NB;           // [NW][NB]
  float* red_o = red_l + NW * NB;           // [NW][NB][64]
  const int D = a.H * 64;
  const T* Kb = reinterpret_cast<const T*>(a.K);
  const T* Vb = reinterpret_cast<const T*>(a.V);
  const int grp = (!SELF && a.group_of) ? a.group_of[gi] : gi;
  const int32_t* rmap = (SELF && a.row_map) ? a.row_map + (int64_t)gi * a.kv_stride : nullptr;

  float qf[NB][8];
  float knew[8], vnew[8];          // fused mode: the newest position's k / v (this lane's 8-feature chunk)
  bool fused = false;
  if constexpr (SELF && NB == 1 && sizeof(T) == 2) fused = a.slab != nullptr;
  if (fused) {
    const int64_t col = h * 64 + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { qf[0][e] = 0.f; knew[e] = 0.f; vnew[e] = 0.f; }
    for (int sl = 0; sl < a.slab_ks; ++sl) {
      const float* sp = a.slab + ((int64_t)sl * a.slab_rows + gi) * a.slab_ld + col;
      float t0[8], t1[8], t2[8];
      ld8(sp, t0);
      ld8(sp + D, t1);
      ld8(sp + 2 * D, t2);
#pragma unroll
      for (int e = 0; e < 8; ++e) { qf[0][e] += t0[e]; knew[e] += t1[e]; vnew[e] += t2[e]; }
    }
    {   // slices first, bias last: the order of the stand-alone reduce kernel (bit-identical results)
      float b0[8], b1[8], b2[8];
      ld8(a.slab_bias + col, b0);
      ld8(a.slab_bias + D + col, b1);
      ld8(a.slab_bias + 2 * D + col, b2);
#pragma unroll
      for (int e = 0; e < 8; ++e) { qf[0][e] += b0[e]; knew[e] += b1[e]; vnew[e] += b2[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {   // the projection epilogue rounds q, k, v to the cache dtype
      qf[0][e] = round_T<T>(qf[0][e]) * 0.125f;
      knew[e] = round_T<T>(knew[e]);
      vnew[e] = round_T<T>(vnew[e]);
    }
    if (kk == 0) {                  // append to the cache for the steps to come
      const int64_t off = (((int64_t)(a.row_base + gi) * a.H + h) * a.kv_stride + (n_keys - 1)) * 64 + c * 8;
      float k4[4] = {knew[0], knew[1], knew[2], knew[3]}, k5[4] = {knew[4], knew[5], knew[6], knew[7]};
      float v4[4] = {vnew[0], vnew[1], vnew[2], vnew[3]}, v5[4] = {vnew[4], vnew[5], vnew[6], vnew[7]};
      T* kc = const_cast<T*>(Kb) + off;
      T* vc = const_cast<T*>(Vb) + off;
      st4(kc, k4); st4(kc + 4, k5);
      st4(vc, v4); st4(vc + 4, v5);
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      ld8(reinterpret_cast<const T*>(a.q) + (int64_t)(gi * NB + b) * D + h * 64 + c * 8, qf[b]);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[b][e] *= 0.125f;
    }
  }
  // fused mode reads positions < n_old from the cache and takes the newest one from registers
  const int n_old = fused ? n_keys - 1 : n_keys;

  // Addresses are clamped to the last valid key instead of predicating the loads (a predicated
  // load makes the compiler select between a global and a stack pointer -> flat loads); U key octets
  // are fetched per wave before any of them is consumed so that enough bytes are in flight to cover
  // HBM latency.
  constexpr int U = 4;
  float lmax[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) lmax[b] = -INFINITY;

  for (int j0 = wave * 8; j0 < n_old; j0 += NW * 8 * U) {
    float kv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = min(j0 + u * NW * 8 + kk, n_old - 1);
      const int64_t prow = SELF ? (a.seq_tp > 0 ? gi / a.seq_tp : (rmap ? rmap[j] : gi)) : grp;
      ld8(Kb + ((prow * a.H + h) * a.kv_stride + j) * 64 + c * 8, kv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * NW * 8 + kk;
      const bool valid = j < n_old;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qf[b][e], kv[u][e], s);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (valid) {
          if (c == 0) sc[b * kpad + j] = s;
          lmax[b] = fmaxf(lmax[b], s);
        }
      }
    }
  }
  float s_new = 0.f;
  if (fused) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s_new = fmaf(qf[0][e], knew[e], s_new);
    s_new += __shfl_xor(s_new, 1, 64);
    s_new += __shfl_xor(s_new, 2, 64);
    s_new += __shfl_xor(s_new, 4, 64);
    lmax[0] = fmaxf(lmax[0], s_new);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float v = lmax[b];
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    if (lane == 0) red_m[wave * NB + b] = v;
  }
  __syncthreads();
  if constexpr (!SELF) {
    if (a.dump) {   // alignment heads hand their scaled scores to the word-timestamp pass
      const int sel = a.dump_sel[h];
      if (sel >= 0) {
        for (int b = 0; b < NB; ++b) {
          const int64_t drow = a.dump_chunks > 0 ? gi / a.dump_chunks : a.dump_row_base + gi * NB + b;
          const int pos = a.dump_chunks > 0 ? (gi % a.dump_chunks) * NB + b : *a.dump_pos_ptr;
          float* dst = a.dump + (((drow * a.dump_nsel + sel) * a.dump_tmax + pos) * n_keys);
          for (int j = tid; j < n_keys; j += NW * 64) dst[j] = sc[b * kpad + j];
        }
      }
    }
  }
  float mx[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float v = red_m[b];
#pragma unroll
    for (int w = 1; w < NW; ++w) v = fmaxf(v, red_m[w * NB + b]);
    mx[b] = v;
  }

  float lsum[NB], o[NB][8];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    lsum[b] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[b][e] = 0.f;
  }
  if (fused && kk == 0 && wave == 0) {
    const float p = expf(s_new - mx[0]);
    lsum[0] = p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[0][e] = p * vnew[e];
  }
  for (int j0 = wave * 8; j0 < n_old; j0 += NW * 8 * U) {
    float vv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = min(j0 + u * NW * 8 + kk, n_old - 1);
      const int64_t prow = SELF ? (a.seq_tp > 0 ? gi / a.seq_tp : (rmap ? rmap[j] : gi)) : grp;
      ld8(Vb + ((prow * a.H + h) * a.kv_stride + j) * 64 + c * 8, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * NW * 8 + kk;
      const bool valid = j < n_old;
      const int jc = min(j, n_old - 1);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float p = valid ? expf(sc[b * kpad + jc] - mx[b]) : 0.f;
        lsum[b] += p;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[b][e] = fmaf(p, vv[u][e], o[b][e]);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float l = lsum[b];
    l += __shfl_xor(l, 8, 64);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = o[b][e];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (kk == 0) red_o[(wave * NB + b) * 64 + c * 8 + e] = v;
    }
    if (lane == 0) red_l[wave * NB + b] = l;
  }
  __syncthreads();
  for (int idx = tid; idx < NB * 64; idx += NW * 64) {
    const int b = idx >> 6, d = idx & 63;
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      num += red_o[(w * NB + b) * 64 + d];
      den += red_l[w * NB + b];
    }
    T* op = reinterpret_cast<T*>(a.out) + (int64_t)(gi * NB + b) * D * (a.out_split ? 2 : 1) + h * 64 + d;
    if constexpr (sizeof(T) == 2) {
      if (a.out_split) { st_split<T>(op, D, num / den); continue; }     // [hi(D) | lo(D)] rows for a split-activation GEMM
    }
    Elem<T>::st(op, num / den);
  }
}

// --------------------------------------------------------------------------------------------
// decode-step CROSS attention on the matrix cores (bf16).  The VALU kernel above needs ~25 vector
// instructions per KiB of K/V (bf16->fp32 unpack, FMAs, shuffles, a redundant exp per lane), which
// at 8 TB/s is as expensive as the HBM stream itself; here a 16-byte load IS an MFMA operand:
//   pass 1  S[16 keys][beams]   = K tile [16 keys x 64]  . Q^T   (A = K rows as stored, 2 MFMAs / KiB pair)
//   softmax scores -> LDS (fp32), exact max, P = exp(S - max) -> LDS once per key (bf16) + fp32 row sums
//   pass 2  O^T[16 d][beams]    = V^T tile [16 d x 32 keys] . P^T (wave w owns output features 16w..16w+15:
//           it streams 16 contiguous V^T rows and needs no cross-wave reduction)
// V is therefore kept TRANSPOSED per head ([H][64][vt_stride], zero padded) -- written that way by the
// cross-K/V GEMM epilogue.  Beams of a window are the MFMA's N columns (up to 16 for free).
// --------------------------------------------------------------------------------------------
template <typename E> __device__ __forceinline__ uint32_t scale2(uint32_t pair, float sc) {
  float lo, hi;
  unpack2<E>(pair, lo, hi);
  return pack2<E>(lo * sc, hi * sc);
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// K / V are read exactly once per launch and never again before 30 GB of other traffic has passed: NT = non-temporal
// loads keep them from displacing the step's reusable lines (weights slices, q, partial slabs) in L2 / MALL
template <bool NT>
__device__ __forceinline__ uint4 ldg_stream(const bf16_t* p) {
  if constexpr (NT) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
  } else {
    return *reinterpret_cast<const uint4*>(p);
  }
}

template <typename E, int U1, int U2, bool NT>
__global__ __launch_bounds__(256) void attn_cross_mfma_kernel(const DecAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int h = blockIdx.x, gi = blockIdx.y;
  const int NB = a.nb, n_keys = a.n_keys, kpad = a.vt_stride;
  float* sc = smem;                                              // [NB][kpad] scores
  E* pl = reinterpret_cast<E*>(sc + NB * kpad);                  // [NB][kpad] probabilities
  float* red_m = reinterpret_cast<float*>(pl + NB * kpad);       // [4][16]
  float* red_l = red_m + 64;                                     // [4][16]
  const int D = a.H * 64;
  const int grp = a.group_of ? a.group_of[gi] : gi;
  const bf16_t* Kh = reinterpret_cast<const bf16_t*>(a.K) + ((int64_t)grp * a.H + h) * (int64_t)a.kv_stride * 64;
  const bf16_t* Vh = reinterpret_cast<const bf16_t*>(a.V) + (((int64_t)grp * a.H + h) * 64 + wave * 16 + li) * (int64_t)kpad;

  // B operand of pass 1: Q^T, column = beam (zero beyond NB), pre-scaled by 1/8 (exact in bf16)
  uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
  if (li < NB && a.slab) {
    // q = bf16(bias + sum of the projection's K-slices), same rounding point as the unfused epilogue
    const int64_t col = h * 64 + lg * 8;
    float s0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < a.slab_ks; ++sl) {
      const float* sp = a.slab + ((int64_t)sl * a.slab_rows + (gi * NB + li)) * a.slab_ld + col;
      float t0[8], t1[8];
      ld8(sp, t0);
      ld8(sp + 32, t1);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0[e] += t0[e]; s1[e] += t1[e]; }
    }
    {   // slices first, bias last: the order of the stand-alone reduce kernel (bit-identical results)
      float b0[8], b1[8];
      ld8(a.slab_bias + col, b0);
      ld8(a.slab_bias + col + 32, b1);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0[e] += b0[e]; s1[e] += b1[e]; }
    }
    auto pack = [](float lo, float hi) { return pack2<E>(lo, hi); };
    q0 = make_uint4(pack(s0[0], s0[1]), pack(s0[2], s0[3]), pack(s0[4], s0[5]), pack(s0[6], s0[7]));
    q1 = make_uint4(pack(s1[0], s1[1]), pack(s1[2], s1[3]), pack(s1[4], s1[5]), pack(s1[6], s1[7]));
  } else if (li < NB) {
    const bf16_t* qp = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)(gi * NB + li) * D + h * 64 + lg * 8;
    q0 = *reinterpret_cast<const uint4*>(qp);
    q1 = *reinterpret_cast<const uint4*>(qp + 32);
  }
  if (li < NB) {
    q0.x = scale2<E>(q0.x, 0.125f); q0.y = scale2<E>(q0.y, 0.125f); q0.z = scale2<E>(q0.z, 0.125f); q0.w = scale2<E>(q0.w, 0.125f);
    q1.x = scale2<E>(q1.x, 0.125f); q1.y = scale2<E>(q1.y, 0.125f); q1.z = scale2<E>(q1.z, 0.125f); q1.w = scale2<E>(q1.w, 0.125f);
  }
  const typename Vec8<E>::type qb0 = as_vec8<E>(q0), qb1 = as_vec8<E>(q1);

  // ---- pass 1: scores.  Tile t (16 keys) belongs to wave t % 4; U1 tiles are fetched before any is used.
  const int n_tiles = (n_keys + 15) >> 4;
  float lmax = -INFINITY;
  for (int t0 = wave; t0 < n_tiles; t0 += 4 * U1) {
    uint4 ka[U1][2];
#pragma unroll
    for (int u = 0; u < U1; ++u) {
      const int key = min((t0 + 4 * u) * 16 + li, n_keys - 1);     // clamped, never predicated
      const bf16_t* kp = Kh + (int64_t)key * 64 + lg * 8;
      ka[u][0] = ldg_stream<NT>(kp);
      ka[u][1] = ldg_stream<NT>(kp + 32);
    }
#pragma unroll
    for (int u = 0; u < U1; ++u) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma16(as_vec8<E>(ka[u][0]), qb0, acc);
      acc = mfma16(as_vec8<E>(ka[u][1]), qb1, acc);
      const int kbase = (t0 + 4 * u) * 16 + lg * 4;
      if (li < NB) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kbase + r < n_keys) {
            sc[li * kpad + kbase + r] = acc[r];
            lmax = fmaxf(lmax, acc[r]);
          }
      }
    }
  }
  lmax = fmaxf(lmax, __shfl_xor(lmax, 16, 64));
  lmax = fmaxf(lmax, __shfl_xor(lmax, 32, 64));
  if (lg == 0) red_m[wave * 16 + li] = lmax;
  __syncthreads();

  if (a.dump) {   // alignment heads hand their scaled scores to the word-timestamp pass
    const int sel = a.dump_sel[h];
    if (sel >= 0) {
      for (int b = 0; b < NB; ++b) {
        const int64_t drow = a.dump_chunks > 0 ? gi / a.dump_chunks : a.dump_row_base + gi * NB + b;
        const int pos = a.dump_chunks > 0 ? (gi % a.dump_chunks) * NB + b : *a.dump_pos_ptr;
        float* dst = a.dump + (((drow * a.dump_nsel + sel) * a.dump_tmax + pos) * n_keys);
        for (int j = tid; j < n_keys; j += 256) dst[j] = sc[b * kpad + j];
      }
    }
  }

  // ---- probabilities, once per (beam, key)
  for (int b = 0; b < NB; ++b) {
    const float mx = fmaxf(fmaxf(red_m[b], red_m[16 + b]), fmaxf(red_m[32 + b], red_m[48 + b]));
    float ls = 0.f;
    for (int j = tid; j < kpad; j += 256) {
      const float p = j < n_keys ? __builtin_amdgcn_exp2f((sc[b * kpad + j] - mx) * 1.4426950408889634f) : 0.f;
      Elem<E>::st(pl + b * kpad + j, p);
      ls += p;
    }
    ls = wave_sum(ls);
    if (lane == 0) red_l[wave * 16 + b] = ls;
  }
  __syncthreads();

  // ---- pass 2: O^T = V^T . P^T, 32 keys per MFMA, U2 chunks in flight
  const int n_chunks = (n_keys + 31) >> 5;
  f32x4_t o = {0.f, 0.f, 0.f, 0.f};
  const E* prow = pl + li * kpad + lg * 8;
  for (int c0 = 0; c0 < n_chunks; c0 += U2) {
    uint4 va[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      const int c = min(c0 + u, n_chunks - 1);
      va[u] = ldg_stream<NT>(Vh + c * 32 + lg * 8);
    }
#pragma unroll
    for (int u = 0; u < U2; ++u) {
      if (c0 + u < n_chunks) {
        uint4 pb = make_uint4(0, 0, 0, 0);
        if (li < NB) pb = *reinterpret_cast<const uint4*>(prow + (c0 + u) * 32);
        o = mfma16(as_vec8<E>(va[u]), as_vec8<E>(pb), o);
      }
    }
  }
  if (li < NB) {
    const float l = (red_l[li] + red_l[16 + li]) + (red_l[32 + li] + red_l[48 + li]);
    const float inv = 1.0f / l;
    float v[4] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
    E* op = reinterpret_cast<E*>(a.out) + (int64_t)(gi * NB + li) * D * (a.out_split ? 2 : 1) + h * 64 + wave * 16 + lg * 4;
    if (a.out_split) st4_split<E>(op, D, v);      // [hi(D) | lo(D)] rows for a split-activation GEMM
    else st4(op, v);
  }
}

int g_dec_cross_nt = 0;  // wj_tune("dec_cross_nt"): non-temporal K/V loads in the MFMA cross-attention kernel
int g_dec_cross_u = 0;   // wj_tune("dec_cross_u"): loads in flight per wave in the MFMA cross-attention kernel

template <typename E>
static int launch_cross_mfma(const DecAttnArgs& a, hipStream_t s) {
  if (a.nb < 1 || a.nb > 16) { set_error("attention_dec: %d query rows per window (1..16)", a.nb); return WJ_E_INVALID; }
  if (a.vt_stride < a.n_keys || (a.vt_stride & 31)) { set_error("attention_dec: vt_stride %d must be a multiple of 32 >= n_keys %d", a.vt_stride, a.n_keys); return WJ_E_INVALID; }
  const size_t smem = (size_t)a.nb * a.vt_stride * 6 + 128 * sizeof(float);
  const dim3 grid(a.H, a.G), block(256);
  const int sel = (g_dec_cross_u & 3) | (g_dec_cross_nt ? 4 : 0);
  switch (sel) {
    case 1: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 2, 4, false>), grid, block, smem, s, a); break;
    case 2: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 6, 12, false>), grid, block, smem, s, a); break;
    case 3: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 8, 16, false>), grid, block, smem, s, a); break;
    case 4: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 4, 8, true>), grid, block, smem, s, a); break;
    case 5: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 2, 4, true>), grid, block, smem, s, a); break;
    case 6: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 6, 12, true>), grid, block, smem, s, a); break;
    case 7: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 8, 16, true>), grid, block, smem, s, a); break;
    default: hipLaunchKernelGGL((attn_cross_mfma_kernel<E, 4, 8, false>), grid, block, smem, s, a); break;
  }
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T, int NB, int NW, bool SELF>
static int launch_dec_inst(const DecAttnArgs& a, int kmax, hipStream_t s) {
  const int kpad = (kmax + 7) & ~7;
  const size_t smem = sizeof(float) * ((size_t)NB * kpad + 2 * NW * NB + (size_t)NW * NB * 64);
  hipLaunchKernelGGL((attn_dec_kernel<T, NB, NW, SELF>), dim3(a.H, a.G), dim3(NW * 64), smem, s, a);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

template <typename T>
static int launch_dec_T(const DecAttnArgs& a, hipStream_t s) {
  const bool self = a.n_keys_ptr != nullptr || a.seq_tp > 0;
  if (self) {
    if (a.nb != 1) { set_error("attention_dec: self attention expects nb == 1"); return WJ_E_INVALID; }
    return launch_dec_inst<T, 1, 1, true>(a, a.seq_tp > 0 ? a.seq_tp : a.kv_stride, s);
  }
  switch (a.nb) {
    case 1: return launch_dec_inst<T, 1, 4, false>(a, a.n_keys, s);
    case 2: return launch_dec_inst<T, 2, 4, false>(a, a.n_keys, s);
    case 3: return launch_dec_inst<T, 3, 4, false>(a, a.n_keys, s);
    case 4: return launch_dec_inst<T, 4, 4, false>(a, a.n_keys, s);
    case 5: return launch_dec_inst<T, 5, 4, false>(a, a.n_keys, s);
    case 6: return launch_dec_inst<T, 6, 4, false>(a, a.n_keys, s);
    case 8: return launch_dec_inst<T, 8, 4, false>(a, a.n_keys, s);
    default: set_error("attention_dec: unsupported beam count %d (1-6, 8)", a.nb); return WJ_E_INVALID;
  }
}

int launch_attention_dec(int dtype, const DecAttnArgs& a, hipStream_t s) {
  if (a.G <= 0) return WJ_OK;
  if (a.vt_stride > 0) {
    if (!is16(dtype) || a.n_keys_ptr || a.seq_tp) { set_error("attention_dec: transposed V is the 16-bit cross-attention layout"); return WJ_E_INVALID; }
    return dtype == WJ_F16 ? launch_cross_mfma<f16_t>(a, s) : launch_cross_mfma<bf16_t>(a, s);
  }
  if (dtype == WJ_F32) {
    if (a.out_split) { set_error("attention_dec: split (hi | lo) output rows exist for the 16-bit compute types only"); return WJ_E_INVALID; }
    return launch_dec_T<float>(a, s);
  }
  return dtype == WJ_F16 ? launch_dec_T<f16_t>(a, s) : launch_dec_T<bf16_t>(a, s);
}

}  // namespace wj
