// vadgraph.hip -- executor of a lowered TorchScript VAD window scorer (whisperjav_amd/vad_graph.py) on gfx950.
//
// Replaces: the per-window TorchScript forward of the reference's DEFAULT segmenter network, silero-v3.1 / v4.0
// (torch.hub archive, /root/reference/whisperjav/modules/speech_segmentation/backends/silero.py:197-206, called from the
// archive's get_speech_timestamps at :258-273: model(chunk, 16000) on consecutive 1536-sample windows, LSTM state carried in
// the module between calls).  The host side walks the archive's graph and emits a small instruction stream whose LSTM
// instructions cut it into STAGES; tensors that live inside a stage sit in a per-window ARENA laid out by liveness (~53 KB for
// the silero-shaped graphs), tensors that cross a stage or that an LSTM touches in a small per-window EXCHANGE area in HBM.
//
// Round 6: the executor is three launches per call instead of ~55 --
//   * vadg_stage_kernel<true>: ONE launch per stage, one workgroup per window, the arena in LDS.  The window's samples are
//     gathered straight from the recording into LDS, the stage's instructions run back to back with a workgroup barrier
//     between them, nothing but the stage's results (the LSTM input: 7 x 64 floats; the probability) goes back to HBM.
//     Wide strided convolutions (the conv-STFT: 258 channels x 256 taps, hop 64) run with the wavefront's lanes on the OUTPUT
//     CHANNELS over a tap-major copy of the weights (coalesced 256-byte rows from L2) and the window read from LDS as
//     broadcast ds_read_b128 -- 5 output frames per lane in registers; everything else is one output element per thread with
//     the lanes along time (conflict-free LDS rows).  Reductions over an axis split every output over a lane group and finish
//     with wavefront shuffles.
//   * vadg_lstm64_kernel<L>: one workgroup per stream, sequential over the stream's windows and time steps.  Thread (unit j,
//     quarter q) of a layer keeps 4 gates x 32 weights of [W_ih | W_hh] in REGISTERS for the whole stream, reads its 32 inputs
//     from LDS as broadcast b128 loads, the quarters meet in two lane shuffles, (h, c) never leave registers / LDS, the layers
//     run one time step apart in the same tick (one barrier per tick), the next steps' inputs are prefetched from the
//     exchange area two ticks ahead.
// Arenas beyond the LDS, and LSTM geometries other than hidden 64 / input <= 64 / <= 2 layers, fall back to round 5's scheme
// (mode 1 of wj_vadg_create forces it: the cross-check of the fused kernels): vadg_stage_kernel<false> runs ONE instruction
// per launch over an arena in HBM, vadg_lstm_kernel reads its weights through L2.
//
// All arithmetic is float32 (fmaf chains, expf / tanhf / log1pf from the device library): the bar is 1e-5 on the window
// probabilities against the same archive executed by torch.jit on the CPU (tests/test_gpu_vad_graph.py).
//
// Instruction encoding: see OP_* / the "view" layout in whisperjav_amd/vad_graph.py (mirrored below; wj_vadg_create
// validates every offset against the arena / exchange / constant / state sizes before anything is launched).
#include <stdlib.h>

#include <algorithm>
#include <initializer_list>
#include <vector>

#include "common.hpp"

namespace wj {

constexpr int kMaxDims = 4;
constexpr int kViewWords = 2 + 2 * kMaxDims;
constexpr int kConvTB = 5;          // output positions a lane of the channel-lanes convolution keeps in registers
constexpr int kStageThreads = 512;  // of a fused stage's workgroup: the instructions are latency-bound chains, 2 workgroups x 8 wavefronts per CU hide them
constexpr int kConvPF = 8;          // tap quads its weight stream runs ahead of the arithmetic
enum { OP_EW = 1, OP_CONV1D = 2, OP_PAD = 3, OP_MEAN = 4, OP_LINEAR = 5, OP_LSTM = 6 };
enum { SP_ARENA = 0, SP_CONST = 1, SP_STATE = 2, SP_XCHG = 3 };
enum { EW_COPY = 0, EW_ADD, EW_SUB, EW_MUL, EW_DIV, EW_RELU, EW_SIGMOID, EW_TANH, EW_EXP, EW_LOG1P, EW_SQRT, EW_ABS, EW_NEG,
       EW_POW_SCALAR, EW_ADD_SCALAR, EW_MUL_SCALAR, EW_FMA, EW_CLAMP, EW_LEAKY_RELU, EW_LOG, EW_RSUB_SCALAR, EW_SILU, EW_HARDTANH,
       EW_COUNT };

struct View {
  int32_t space, offset;
  int32_t shape[kMaxDims];
  int32_t stride[kMaxDims];
};

// `flat`: 0 = strided indexing, 1 = every operand is addressed by the flat element index (or is one broadcast scalar)
struct EwArgs { int32_t fn, nin; float p0, p1; View out, in[3]; int32_t flat, in_mode[3]; };
// chan_lanes: output channels [0, cout_full) run with the lanes on the channels over the tap-major weights at wt_off
// ws_*: arena scratch of the instruction (weights staged with row stride ws_stride, then the bias), -1 = none
struct ConvArgs {
  View out, in;
  int32_t w_off, b_off, cout, cin, k, t, tout, stride, padding, dilation, groups, ws_space, ws_off, ws_stride;
  int32_t cout_full, wt_off, vec4;
};
struct PadArgs { View out, in; int32_t left, right, mode; float value; };
struct MeanArgs { View out, in; int32_t r, rstride; float inv; };
struct LinearArgs { int32_t out_space, out_off, in_space, in_off, w_off, b_off, rows, nin, nout; };
struct LstmArgs {
  int32_t y_space, y_off, x_space, x_off, st_t, st_f, t, nin, hidden, layers, hn_space, hn_off, cn_space, cn_off;
  int32_t blob[12];   // per layer: W_ih^T [in][4H], W_hh^T [H][4H], b_ih + b_hh [4H] (offsets into the constants)
  int32_t h_slot, c_slot, need_hc;
  int32_t fast_off;   // >= 0: register-resident kernel, its weight image [layer][gate * 32 + k][256 threads] in the constants
};
struct Instr {
  int op;
  int items;          // work items of the instruction (threads of a stage = the largest, rounded to wavefronts)
  union { EwArgs ew; ConvArgs conv; PadArgs pad; MeanArgs mean; LinearArgs lin; LstmArgs lstm; };
};
static_assert(sizeof(Instr) % 4 == 0, "Instr is copied dword by dword");

struct Segment { int32_t first, count, stream; };   // a stream's windows inside one slab: first window, how many, state row

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// The three per-window memories of an instruction.  `arena` is an LDS pointer (address space 3) in the fused stage kernel and
// an HBM pointer in the per-instruction fall-back; the accessors branch on the (wavefront-uniform) space, and because the
// two sides of the branch are different address spaces the compiler cannot fold them into one flat access: arena traffic of
// the fused kernel is ds_read / ds_write.
typedef const __attribute__((address_space(4))) float cst_f32;      // read-only for the whole launch: wavefront-uniform loads become s_load
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x4_t lds_f32x4;
template <bool FUSED> struct ArenaPtr { typedef float* type; };
template <> struct ArenaPtr<true> { typedef lds_f32* type; };
template <bool FUSED> struct Mem {
  typename ArenaPtr<FUSED>::type arena;
  const float* consts;
  float* xchg;
};

template <bool F> __device__ __forceinline__ float ld(const Mem<F>& m, int space, int64_t idx) {
  if (space == SP_ARENA) return m.arena[idx];
  if (space == SP_XCHG) return m.xchg[idx];
  return m.consts[idx];
}
template <bool F> __device__ __forceinline__ void st(const Mem<F>& m, int space, int64_t idx, float v) {
  if (space == SP_ARENA) m.arena[idx] = v;
  else m.xchg[idx] = v;
}
__device__ __forceinline__ f32x4_t ld4(const lds_f32* p) { return *reinterpret_cast<const lds_f32x4*>(p); }
__device__ __forceinline__ f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }

__device__ __forceinline__ int64_t view_index(const View& v, int i0, int i1, int i2, int i3) {
  return (int64_t)v.offset + (int64_t)i0 * v.stride[0] + (int64_t)i1 * v.stride[1] + (int64_t)i2 * v.stride[2] + (int64_t)i3 * v.stride[3];
}

// N elements of one function: the (wavefront-uniform) switch runs once per round, not once per element
template <int N> __device__ __forceinline__ void ew_apply(int fn, const float (&x)[N], const float (&y)[N], const float (&z)[N], float p0, float p1, float (&r)[N]) {
#define WJ_EW(expr) _Pragma("unroll") for (int u = 0; u < N; ++u) r[u] = (expr); break
  switch (fn) {
    case EW_COPY: WJ_EW(x[u]);
    case EW_ADD: WJ_EW(x[u] + y[u]);
    case EW_SUB: WJ_EW(x[u] - y[u]);
    case EW_MUL: WJ_EW(x[u] * y[u]);
    case EW_DIV: WJ_EW(x[u] / y[u]);
    case EW_RELU: WJ_EW(fmaxf(x[u], 0.f));
    case EW_SIGMOID: WJ_EW(sigmoidf_(x[u]));
    case EW_TANH: WJ_EW(tanhf(x[u]));
    case EW_EXP: WJ_EW(expf(x[u]));
    case EW_LOG1P: WJ_EW(log1pf(x[u]));
    case EW_SQRT: WJ_EW(sqrtf(x[u]));
    case EW_ABS: WJ_EW(fabsf(x[u]));
    case EW_NEG: WJ_EW(-x[u]);
    case EW_POW_SCALAR: WJ_EW(powf(x[u], p0));
    case EW_ADD_SCALAR: WJ_EW(x[u] + p0);
    case EW_MUL_SCALAR: WJ_EW(x[u] * p0);
    case EW_FMA: WJ_EW(x[u] * y[u] + z[u]);      // contracted to one fma by the compiler, as torch's fused affine is not: |d| <= 1 ulp
    case EW_CLAMP: WJ_EW(fminf(fmaxf(x[u], p0), p1));
    case EW_LEAKY_RELU: WJ_EW(x[u] > 0.f ? x[u] : x[u] * p0);
    case EW_LOG: WJ_EW(logf(x[u]));
    case EW_RSUB_SCALAR: WJ_EW(p0 - x[u] * p1);
    case EW_SILU: WJ_EW(x[u] * sigmoidf_(x[u]));
    default: WJ_EW(fminf(fmaxf(x[u], p0), p1));   // EW_HARDTANH
  }
#undef WJ_EW
}

template <bool F> __device__ __forceinline__ void run_ew(const EwArgs& a, const Mem<F>& m, int tid, int nthreads) {
  const int d1 = a.out.shape[1], d2 = a.out.shape[2], d3 = a.out.shape[3];
  const int total = a.out.shape[0] * d1 * d2 * d3;
  if (a.flat) {
    // operands addressed by the flat element index; broadcast scalars are fetched once.  Four independent elements per thread and
    // round: their loads are in flight together.  (In-place results: an element is read and written by the same thread through
    // the identical view, loads first.)  The all-arena case has no space branches between its loads.
    float sc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < a.nin && a.in_mode[i] == 2) sc[i] = ld(m, a.in[i].space, a.in[i].offset);
    bool arena_only = a.out.space == SP_ARENA;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < a.nin && a.in_mode[i] == 1 && a.in[i].space != SP_ARENA) arena_only = false;
    for (int e0 = tid; e0 < total; e0 += 4 * nthreads) {
      float v[3][4], r[4];
      if (arena_only) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            v[i][u] = (i < a.nin && a.in_mode[i] == 1) ? m.arena[a.in[i].offset + min(e0 + u * nthreads, total - 1)] : sc[i];
      } else {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int u = 0; u < 4; ++u)
            v[i][u] = (i < a.nin && a.in_mode[i] == 1) ? ld(m, a.in[i].space, a.in[i].offset + min(e0 + u * nthreads, total - 1)) : sc[i];
      }
      ew_apply<4>(a.fn, v[0], v[1], v[2], a.p0, a.p1, r);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nthreads;
        if (e < total) {
          if (arena_only) m.arena[a.out.offset + e] = r[u];
          else st(m, a.out.space, a.out.offset + e, r[u]);
        }
      }
    }
    return;
  }
  for (int e = tid; e < total; e += nthreads) {
    const int i3 = e % d3, r3 = e / d3;
    const int i2 = r3 % d2, r2 = r3 / d2;
    const int i1 = r2 % d1, i0 = r2 / d1;
    const float x = ld(m, a.in[0].space, view_index(a.in[0], i0, i1, i2, i3));
    float y = 0.f, z = 0.f;
    if (a.nin > 1) y = ld(m, a.in[1].space, view_index(a.in[1], i0, i1, i2, i3));
    if (a.nin > 2) z = ld(m, a.in[2].space, view_index(a.in[2], i0, i1, i2, i3));
    const float xs[1] = {x}, ys[1] = {y}, zs[1] = {z};
    float r[1];
    ew_apply<1>(a.fn, xs, ys, zs, a.p0, a.p1, r);
    st(m, a.out.space, view_index(a.out, i0, i1, i2, i3), r[0]);
  }
}

// conv1d.  Channels [0, cout_full): lanes on the output channel (consecutive lanes read consecutive floats of the tap-major
// weights), kConvTB output positions per lane in registers, the input row read as wavefront-uniform (broadcast) loads.
// Channels [cout_full, cout): one output element per thread, lanes along the output positions.
template <bool F> __device__ __forceinline__ void run_conv(const ConvArgs& a, const Mem<F>& m, int tid, int nthreads) {
  const int sc = a.in.stride[2], st_ = a.in.stride[3];     // [1][1][C][T] after the leading-1 padding of a [1, C, T] view
  const int cg = a.cin / a.groups, og = a.cout / a.groups;
  const int in_space = a.in.space, out_space = a.out.space;
  if (a.cout_full > 0) {
    const int n_tb = (a.tout + kConvTB - 1) / kConvTB;
    const int items = n_tb * a.cout_full;                   // cout_full is a multiple of 64: a wavefront shares its position block
    const float* __restrict__ wt = m.consts + a.wt_off;
    for (int it = tid; it < items; it += nthreads) {
      const int tb = it / a.cout_full, co = it - tb * a.cout_full;
      const int to0 = tb * kConvTB;
      float acc[kConvTB];
#pragma unroll
      for (int j = 0; j < kConvTB; ++j) acc[j] = 0.f;
      // tap-major weights in blocks of four taps: element (row r = ci * k + kk, channel co) at ((r / 4) * cout + co) * 4 + r % 4,
      // so a lane fetches four taps of its channel in one 16-byte load and a wavefront 1 KiB contiguous
      if (a.vec4 && in_space == SP_ARENA) {
        // padding 0, dilation 1, unit input stride, taps a multiple of 4 and every row start 16-byte aligned (checked at create)
        // the weight vectors run kConvPF tap-quads ahead of the arithmetic in a register ring: an L2 round trip (~600 cycles) is
        // covered by kConvPF x (5 broadcast ds_read_b128 + 20 FMA) instead of sitting on every step
        for (int ci = 0; ci < cg; ++ci) {
          const auto xr = m.arena + a.in.offset + (int64_t)ci * sc;
          const float* wr = wt + ((int64_t)(ci * a.k / 4) * a.cout + co) * 4;
          const int nq = a.k >> 2;
          f32x4_t ring[kConvPF];
#pragma unroll
          for (int u = 0; u < kConvPF; ++u) ring[u] = u < nq ? *reinterpret_cast<const f32x4_t*>(wr + (int64_t)(4 * u) * a.cout) : f32x4_t{0.f, 0.f, 0.f, 0.f};
          for (int q0 = 0; q0 < nq; q0 += kConvPF) {
#pragma unroll
            for (int u = 0; u < kConvPF; ++u) {
              const int qi = q0 + u;
              const f32x4_t wv = ring[u];
              if (qi + kConvPF < nq) ring[u] = *reinterpret_cast<const f32x4_t*>(wr + (int64_t)(4 * (qi + kConvPF)) * a.cout);
              if (qi < nq) {
#pragma unroll
                for (int j = 0; j < kConvTB; ++j) {
                  const int to = min(to0 + j, a.tout - 1);
                  const f32x4_t xv = ld4(xr + to * a.stride + 4 * qi);
                  acc[j] = fmaf(wv.x, xv.x, acc[j]);
                  acc[j] = fmaf(wv.y, xv.y, acc[j]);
                  acc[j] = fmaf(wv.z, xv.z, acc[j]);
                  acc[j] = fmaf(wv.w, xv.w, acc[j]);
                }
              }
            }
          }
        }
      } else {
        for (int ci = 0; ci < cg; ++ci) {
          const int64_t xrow = (int64_t)a.in.offset + (int64_t)ci * sc;
          for (int kk = 0; kk < a.k; ++kk) {
            const int r = ci * a.k + kk;
            const float w = wt[((int64_t)(r >> 2) * a.cout + co) * 4 + (r & 3)];
#pragma unroll
            for (int j = 0; j < kConvTB; ++j) {
              const int to = min(to0 + j, a.tout - 1);
              const int ti = to * a.stride - a.padding + kk * a.dilation;
              if (ti >= 0 && ti < a.t) acc[j] = fmaf(w, ld(m, in_space, xrow + (int64_t)ti * st_), acc[j]);
            }
          }
        }
      }
      const float b = a.b_off >= 0 ? m.consts[a.b_off + co] : 0.f;
#pragma unroll
      for (int j = 0; j < kConvTB; ++j)
        if (to0 + j < a.tout) st(m, out_space, (int64_t)a.out.offset + (int64_t)co * a.out.stride[2] + (int64_t)(to0 + j) * a.out.stride[3], acc[j] + b);
    }
  }
  // the other channels: one output element per thread, lanes along the output positions (conflict-free input rows).  Fused
  // executor with a scratch: the weights and the bias are first staged in LDS by one coalesced sweep of the whole workgroup
  // (rows at an odd stride: the few channels a wavefront spans sit in different banks), so the accumulation chains read LDS
  // only; otherwise they come through L2.
  const int row = cg * a.k;
  const int rest = (a.cout - a.cout_full) * a.tout;
  if constexpr (F) {
    if (a.ws_space == SP_ARENA) {
      const auto ws = m.arena + a.ws_off;
      const float* __restrict__ src = m.consts + a.w_off;
      const int nw = a.cout * row;
      for (int i = tid; i < nw; i += nthreads) {
        const int co = i / row;
        ws[co * a.ws_stride + (i - co * row)] = src[i];
      }
      const auto wb = ws + a.cout * a.ws_stride;
      for (int i = tid; i < a.cout; i += nthreads) wb[i] = a.b_off >= 0 ? m.consts[a.b_off + i] : 0.f;
      __syncthreads();
      const auto xin = m.arena + a.in.offset;        // scratch implies an arena input only when in_space says so: checked below
      for (int e = tid; e < rest; e += nthreads) {
        const int co = a.cout_full + e / a.tout, to = e % a.tout;
        const int g = co / og;
        const auto wr = ws + co * a.ws_stride;
        const int t0 = to * a.stride - a.padding;
        float acc = wb[co];
        if (in_space == SP_ARENA) {
          if (a.k == 1) {
            const bool ok = t0 >= 0 && t0 < a.t;
            const auto xp = xin + (int64_t)(g * cg) * sc + (int64_t)(ok ? t0 : 0) * st_;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int ci = 0;
            for (; ci + 4 <= cg; ci += 4) {
              a0 = fmaf(wr[ci], xp[(int64_t)ci * sc], a0);
              a1 = fmaf(wr[ci + 1], xp[(int64_t)(ci + 1) * sc], a1);
              a2 = fmaf(wr[ci + 2], xp[(int64_t)(ci + 2) * sc], a2);
              a3 = fmaf(wr[ci + 3], xp[(int64_t)(ci + 3) * sc], a3);
            }
            for (; ci < cg; ++ci) a0 = fmaf(wr[ci], xp[(int64_t)ci * sc], a0);
            acc += ok ? (a0 + a1) + (a2 + a3) : 0.f;
          } else {
            for (int ci = 0; ci < cg; ++ci) {
              const auto xr = xin + (int64_t)(g * cg + ci) * sc;
#pragma unroll 4
              for (int kk = 0; kk < a.k; ++kk) {
                const int ti = t0 + kk * a.dilation;
                const bool ok = ti >= 0 && ti < a.t;
                const float xv = xr[(int64_t)(ok ? ti : 0) * st_];
                acc = fmaf(wr[ci * a.k + kk], ok ? xv : 0.f, acc);
              }
            }
          }
        } else {
          for (int ci = 0; ci < cg; ++ci) {
            const int64_t xrow = (int64_t)a.in.offset + (int64_t)(g * cg + ci) * sc;
            for (int kk = 0; kk < a.k; ++kk) {
              const int ti = t0 + kk * a.dilation;
              if (ti >= 0 && ti < a.t) acc = fmaf(wr[ci * a.k + kk], ld(m, in_space, xrow + (int64_t)ti * st_), acc);
            }
          }
        }
        st(m, out_space, (int64_t)a.out.offset + (int64_t)co * a.out.stride[2] + (int64_t)to * a.out.stride[3], acc);
      }
      return;
    }
  }
  for (int e = tid; e < rest; e += nthreads) {
    const int co = a.cout_full + e / a.tout, to = e % a.tout;
    const int g = co / og;
    const float* wt = m.consts + a.w_off + (int64_t)co * row;
    float acc = 0.f;
    const int t0 = to * a.stride - a.padding;
    for (int ci = 0; ci < cg; ++ci) {
      const int64_t xrow = (int64_t)a.in.offset + (int64_t)(g * cg + ci) * sc;
      for (int kk = 0; kk < a.k; ++kk) {
        const int ti = t0 + kk * a.dilation;
        if (ti >= 0 && ti < a.t) acc = fmaf(wt[ci * a.k + kk], ld(m, in_space, xrow + (int64_t)ti * st_), acc);
      }
    }
    if (a.b_off >= 0) acc += m.consts[a.b_off + co];
    st(m, out_space, (int64_t)a.out.offset + (int64_t)co * a.out.stride[2] + (int64_t)to * a.out.stride[3], acc);
  }
}

template <bool F> __device__ __forceinline__ void run_pad(const PadArgs& a, const Mem<F>& m, int tid, int nthreads) {
  const int d1 = a.out.shape[1], d2 = a.out.shape[2], d3 = a.out.shape[3], t = a.in.shape[3];
  const int total = a.out.shape[0] * d1 * d2 * d3;
  for (int e = tid; e < total; e += nthreads) {
    const int i3 = e % d3, r3 = e / d3;
    const int i2 = r3 % d2, r2 = r3 / d2;
    const int i1 = r2 % d1, i0 = r2 / d1;
    int j = i3 - a.left;
    float v;
    if (j >= 0 && j < t) {
      v = ld(m, a.in.space, view_index(a.in, i0, i1, i2, j));
    } else if (a.mode == 0) {
      v = a.value;
    } else {
      if (a.mode == 1) j = j < 0 ? -j : 2 * (t - 1) - j;       // reflect (edge not repeated)
      else j = j < 0 ? 0 : t - 1;                               // replicate
      v = ld(m, a.in.space, view_index(a.in, i0, i1, i2, j));
    }
    st(m, a.out.space, view_index(a.out, i0, i1, i2, i3), v);
  }
}

// mean over one axis: every output is shared by a group of G lanes (a power of two, inside one wavefront) that take the
// reduced axis G apart and meet in xor-shuffles.  `nthreads` is a multiple of 64 and every lane runs every round.
template <bool F> __device__ __forceinline__ void run_mean(const MeanArgs& a, const Mem<F>& m, int tid, int nthreads) {
  const int d1 = a.out.shape[1], d2 = a.out.shape[2], d3 = a.out.shape[3];
  const int n_out = a.out.shape[0] * d1 * d2 * d3;
  int G = 1;
  if (a.r >= 16)
    while (G < 64 && 2 * G * n_out <= nthreads && 4 * G <= a.r) G *= 2;
  const int items = n_out * G;
  for (int base = 0; base < items; base += nthreads) {
    const int it = base + tid;
    const bool valid = it < items;
    const int e = valid ? it / G : 0, part = it % G;
    const int i3 = e % d3, r3 = e / d3;
    const int i2 = r3 % d2, r2 = r3 / d2;
    const int i1 = r2 % d1, i0 = r2 / d1;
    const int64_t p = view_index(a.in, i0, i1, i2, i3);
    float acc = 0.f;
    if (valid)
      for (int i = part; i < a.r; i += G) acc += ld(m, a.in.space, p + (int64_t)i * a.rstride);
    for (int o = 1; o < G; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (valid && part == 0) st(m, a.out.space, view_index(a.out, i0, i1, i2, i3), acc * a.inv);
  }
}

template <bool F> __device__ __forceinline__ void run_linear(const LinearArgs& a, const Mem<F>& m, int tid, int nthreads) {
  const int total = a.rows * a.nout;
  for (int e = tid; e < total; e += nthreads) {
    const int r = e / a.nout, o = e % a.nout;
    const int64_t x = (int64_t)a.in_off + (int64_t)r * a.nin;
    const float* wt = m.consts + a.w_off + (int64_t)o * a.nin;
    float acc = 0.f;
    for (int i = 0; i < a.nin; ++i) acc = fmaf(ld(m, a.in_space, x + i), wt[i], acc);
    if (a.b_off >= 0) acc += m.consts[a.b_off + o];
    st(m, a.out_space, (int64_t)a.out_off + e, acc);
  }
}

template <bool F> __device__ __forceinline__ void run_instr(const Instr& in, const Mem<F>& m, int tid, int nthreads) {
  switch (in.op) {
    case OP_EW: run_ew(in.ew, m, tid, nthreads); break;
    case OP_CONV1D: run_conv(in.conv, m, tid, nthreads); break;
    case OP_PAD: run_pad(in.pad, m, tid, nthreads); break;
    case OP_MEAN: run_mean(in.mean, m, tid, nthreads); break;
    case OP_LINEAR: run_linear(in.lin, m, tid, nthreads); break;
    default: break;
  }
}

struct StageArgs {
  const Instr* __restrict__ prog;       // the whole program, in device memory
  int first, count;        // this launch's instructions
  const float* consts;
  float* xchg;             // [windows][xchg_stride]
  int64_t xchg_stride;
  float* arena;            // fall-back only: [windows][arena_stride] in HBM
  int64_t arena_stride;
  // gather (the first stage of a call) / scatter (the last one)
  const float* pcm;
  const int64_t* src;
  const int32_t* valid;
  int in_space, in_off, window;
  float* probs;
  int out_space, out_off;
  long long* clocks;       // diagnostic (WJ_VADG_CLOCKS=1): s_memtime after every instruction of workgroups 0 and grid / 2
};

// FUSED: grid.x = window, the arena in dynamic LDS, instructions [first, first + count) with a barrier between them.
// !FUSED: grid = (blocks, window), ONE instruction (count == 1) or the gather / scatter alone (count == 0) over an HBM arena.
template <bool FUSED>
__global__ __launch_bounds__(kStageThreads) void vadg_stage_kernel(StageArgs s) {
  extern __shared__ __align__(16) float lds_arena[];
  const int w = FUSED ? blockIdx.x : blockIdx.y;
  Mem<FUSED> m;
  if constexpr (FUSED) m.arena = (lds_f32*)lds_arena;
  else m.arena = s.arena + (int64_t)w * s.arena_stride;
  m.consts = s.consts;
  m.xchg = s.xchg + (int64_t)w * s.xchg_stride;
  const int tid = FUSED ? threadIdx.x : blockIdx.x * blockDim.x + threadIdx.x;
  const int nthreads = FUSED ? blockDim.x : gridDim.x * blockDim.x;
  if (s.pcm) {     // the window's samples: zero past the stream's end, as utils_vad.get_speech_timestamps pads the last window
    const int64_t src = s.src[w];
    const int n = s.valid[w];
    for (int i = tid; i < s.window; i += nthreads) st(m, s.in_space, s.in_off + i, i < n ? s.pcm[src + i] : 0.f);
    if (FUSED) __syncthreads();
  }
  const bool stamp = FUSED && s.clocks && tid == 0 && (w == 0 || w == (int)(gridDim.x / 2));
  long long* ck = s.clocks + (w == 0 ? 0 : 256);
  if (stamp) ck[0] = clock64();
  for (int i = 0; i < s.count; ++i) {
    // the instruction's arguments through the constant address space: wavefront-uniform, so they arrive as s_load into SGPRs
    union { Instr in; int32_t w[sizeof(Instr) / 4]; } u;
    const __attribute__((address_space(4))) int32_t* src = (const __attribute__((address_space(4))) int32_t*)(s.prog + s.first + i);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(Instr) / 4); ++k) u.w[k] = src[k];
    run_instr(u.in, m, tid, nthreads);
    if (FUSED) __syncthreads();
    if (stamp && i < 254) ck[1 + i] = clock64();
  }
  if (s.probs && tid == 0) s.probs[w] = ld(m, s.out_space, s.out_off);
}

__global__ __launch_bounds__(256) void vadg_state_init_kernel(float* state, const float* init, int state_floats, int n_streams) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (int64_t)state_floats * n_streams) state[i] = init[i % state_floats];
}

// ---- LSTM, hidden 64, input <= 64, L <= 2 layers: weights in registers ---------------------------------------------------------
// 256 threads per layer.  Thread (wave wv, lane = 4 * jj + q): hidden unit j = 16 * wv + jj, quarter q of the 128-wide
// concatenated input [x or the layer below's h (64, zero padded) | own h (64)].  Level v of V holds: v = 0 the step's input,
// v = l + 1 the h of layer l; two parities (a tick reads what the previous tick wrote); level stride 80 floats so that the four
// quarter slices of a b128 read sit in distinct banks.  Layer l runs step k - l in tick k.
constexpr int kLstmLevel = 80;

template <int L>
__global__ __launch_bounds__(256 * L) void vadg_lstm64_kernel(LstmArgs a, const float* __restrict__ consts, float* xchg, int64_t xchg_stride,
                                                               float* state, int state_floats, const Segment* __restrict__ segs) {
  __shared__ __align__(16) float V[2][(L + 1) * kLstmLevel];
  const Segment seg = segs[blockIdx.x];
  const int tid = threadIdx.x, l = tid >> 8, tl = tid & 255, wv = tl >> 6, lane = tl & 63, jj = lane >> 2, q = lane & 3;
  const int j = wv * 16 + jj;
  float w[4][32];
  {
    const float* src = consts + a.fast_off + (int64_t)l * 128 * 256 + tl;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int k = 0; k < 32; ++k) w[g][k] = src[(g * 32 + k) * 256];
  }
  float bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias[g] = consts[a.blob[3 * l + 2] + g * 64 + j];
  float* st_row = state + (int64_t)seg.stream * state_floats;
  float c = st_row[a.c_slot + l * 64 + j];
  float h = st_row[a.h_slot + l * 64 + j];
  const int in_off = q < 2 ? l * kLstmLevel + 32 * q : (l + 1) * kLstmLevel + 32 * (q - 2);
  const int T = a.t;
  const int total = seg.count * T;
  const bool loader = (l == 0 && q == 0);
  const bool live_in = j < a.nin;
  // loader cursor: the step whose input is fetched next
  int lw = 0, lt = 0;
  auto fetch = [&]() -> float {
    float v = 0.f;
    if (lw < seg.count) {
      if (live_in) v = xchg[(int64_t)(seg.first + lw) * xchg_stride + a.x_off + (int64_t)lt * a.st_t + (int64_t)j * a.st_f];
      if (++lt == T) { lt = 0; ++lw; }
    }
    return v;
  };
  float xr = 0.f;
  if (q == 0) V[(l + 1) & 1][(l + 1) * kLstmLevel + j] = h;        // own h where the layer's first tick (k = l) reads it: parity (l - 1) & 1
  if (loader) {
    V[1][j] = fetch();            // step 0, read in tick 0 from parity (0 - 1) & 1
    xr = fetch();                 // step 1
  }
  __syncthreads();
  // this layer's cursor
  int cw = 0, ct = 0;
  for (int k = 0; k < total + L - 1; ++k) {
    const int s = k - l;
    const int pr = (k + 1) & 1, pw = k & 1;
    float xn = 0.f;
    if (loader) xn = fetch();       // step k + 2
    if (s >= 0 && s < total) {
      const float* in = &V[pr][in_off];
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 v = *reinterpret_cast<const float4*>(in + 4 * k4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc[g] = fmaf(w[g][4 * k4], v.x, acc[g]);
          acc[g] = fmaf(w[g][4 * k4 + 1], v.y, acc[g]);
          acc[g] = fmaf(w[g][4 * k4 + 2], v.z, acc[g]);
          acc[g] = fmaf(w[g][4 * k4 + 3], v.w, acc[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        acc[g] += __shfl_xor(acc[g], 1, 64);
        acc[g] += __shfl_xor(acc[g], 2, 64);
        acc[g] += bias[g];
      }
      const float ig = sigmoidf_(acc[0]), fg = sigmoidf_(acc[1]), gg = tanhf(acc[2]), og = sigmoidf_(acc[3]);
      c = fg * c + ig * gg;
      h = og * tanhf(c);
      if (q == 0) {
        V[pw][(l + 1) * kLstmLevel + j] = h;
        float* xw = xchg + (int64_t)(seg.first + cw) * xchg_stride;
        if (l == L - 1) xw[a.y_off + ct * 64 + j] = h;
        if (a.need_hc && ct == T - 1) {
          xw[a.hn_off + l * 64 + j] = h;
          xw[a.cn_off + l * 64 + j] = c;
        }
      }
      if (++ct == T) { ct = 0; ++cw; }
    }
    if (loader) {
      V[pw][j] = xr;        // step k + 1, read in tick k + 1 from parity k & 1
      xr = xn;
    }
    __syncthreads();
  }
  if (q == 0) {
    st_row[a.h_slot + l * 64 + j] = h;
    st_row[a.c_slot + l * 64 + j] = c;
  }
}

// The general LSTM (any hidden <= 128, input <= 512, <= 4 layers): one workgroup per stream segment, thread j < 4H owns gate
// row j (weights input-major: the lanes of a wavefront read consecutive floats, served by L2); h / c of every layer live in
// LDS for the whole segment and go back to the stream's state row at its end.
__global__ __launch_bounds__(512) void vadg_lstm_kernel(LstmArgs a, const float* __restrict__ consts, float* xchg, int64_t xchg_stride, float* state,
                                                        int state_floats, const Segment* __restrict__ segs) {
  __shared__ float s_x[512];
  __shared__ float s_h[4][128];
  __shared__ float s_c[4][128];
  __shared__ float s_g[512];
  const Segment seg = segs[blockIdx.x];
  const int tid = threadIdx.x, H = a.hidden, G = 4 * H;
  float* st = state + (int64_t)seg.stream * state_floats;
  for (int i = tid; i < a.layers * H; i += blockDim.x) {
    s_h[i / H][i % H] = st[a.h_slot + i];
    s_c[i / H][i % H] = st[a.c_slot + i];
  }
  __syncthreads();
  for (int wi = 0; wi < seg.count; ++wi) {
    float* xw = xchg + (int64_t)(seg.first + wi) * xchg_stride;
    const float* xs = (a.x_space == SP_CONST ? consts : xw) + a.x_off;
    for (int t = 0; t < a.t; ++t) {
      for (int i = tid; i < a.nin; i += blockDim.x) s_x[i] = xs[(int64_t)t * a.st_t + (int64_t)i * a.st_f];
      __syncthreads();
      for (int l = 0; l < a.layers; ++l) {
        const int nin = l ? H : a.nin;
        const float* in = l ? s_h[l - 1] : s_x;
        if (tid < G) {
          const float* wih = consts + a.blob[3 * l] + tid;
          const float* whh = consts + a.blob[3 * l + 1] + tid;
          float acc = consts[a.blob[3 * l + 2] + tid];
          for (int i = 0; i < nin; ++i) acc = fmaf(wih[(int64_t)i * G], in[i], acc);
          for (int i = 0; i < H; ++i) acc = fmaf(whh[(int64_t)i * G], s_h[l][i], acc);
          s_g[tid] = acc;
        }
        __syncthreads();
        if (tid < H) {
          const float ig = sigmoidf_(s_g[tid]), fg = sigmoidf_(s_g[H + tid]);
          const float gg = tanhf(s_g[2 * H + tid]), og = sigmoidf_(s_g[3 * H + tid]);
          const float c = fg * s_c[l][tid] + ig * gg;
          s_c[l][tid] = c;
          s_h[l][tid] = og * tanhf(c);
        }
        __syncthreads();
      }
      if (tid < H) xw[a.y_off + (int64_t)t * H + tid] = s_h[a.layers - 1][tid];
    }
    if (a.need_hc)
      for (int i = tid; i < a.layers * H; i += blockDim.x) {      // the window's final (h, c): the graph reads them
        xw[a.hn_off + i] = s_h[i / H][i % H];
        xw[a.cn_off + i] = s_c[i / H][i % H];
      }
  }
  __syncthreads();
  for (int i = tid; i < a.layers * H; i += blockDim.x) {
    st[a.h_slot + i] = s_h[i / H][i % H];
    st[a.c_slot + i] = s_c[i / H][i % H];
  }
}

}  // namespace wj

using namespace wj;

struct VadgStage { int first = 0, count = 0, threads = 64; };

struct wj_vadg {
  wj_ctx* ctx = nullptr;
  std::vector<Instr> prog;
  std::vector<float> consts_host;   // the program's constants + derived images (tap-major conv weights, register LSTM image)
  Instr* prog_dev = nullptr;
  float* consts = nullptr;
  float* state_init = nullptr;
  int64_t n_consts = 0;              // of the caller's blob (what the program's offsets are validated against)
  int state_floats = 0;
  int64_t arena_floats = 0, xchg_floats = 0;
  int in_space = 0, in_off = 0, out_space = 0, out_off = 0, window = 0, max_windows = 0;
  bool fused = false;
  std::vector<VadgStage> stages;     // stage i runs before LSTM i (the LSTM instructions in program order are lstm_at)
  std::vector<int> lstm_at;
  // grow-only per-call memory
  float* xchg = nullptr;             // [cap_windows][xchg_floats]
  float* arena = nullptr;            // fall-back only: [cap_windows][arena_floats]
  void* tables = nullptr;            // src int64[cap], valid int32[cap], Segment[cap]
  int64_t cap_windows = 0;
  float* state = nullptr;            // [streams][state_floats]
  int64_t state_rows = 0;
};

namespace {

int64_t view_extent(const View& v) {     // one past the largest element index the view touches, relative to its space
  int64_t hi = v.offset;
  for (int d = 0; d < kMaxDims; ++d) {
    if (v.shape[d] < 1 || v.stride[d] < 0) return -1;
    hi += (int64_t)(v.shape[d] - 1) * v.stride[d];
  }
  return hi + 1;
}

int64_t space_limit(const wj_vadg* h, int space) {
  return space == SP_ARENA ? h->arena_floats : space == SP_XCHG ? h->xchg_floats : space == SP_CONST ? h->n_consts : -1;
}

int check_view(const wj_vadg* h, const View& v, bool is_output, const char* what, int idx) {
  WJ_REQUIRE(v.space == SP_ARENA || v.space == SP_XCHG || (v.space == SP_CONST && !is_output), "wj_vadg_create: instruction %d: %s lives in space %d", idx,
             what, v.space);
  const int64_t lim = space_limit(h, v.space);
  const int64_t ext = view_extent(v);
  WJ_REQUIRE(v.offset >= 0 && ext > 0 && ext <= lim, "wj_vadg_create: instruction %d: %s reaches float %lld of %lld", idx, what, (long long)ext,
             (long long)lim);
  return WJ_OK;
}

int check_range(const wj_vadg* h, int space, int64_t off, int64_t n, bool is_output, const char* what, int idx) {
  WJ_REQUIRE(space == SP_ARENA || space == SP_XCHG || (space == SP_CONST && !is_output), "wj_vadg_create: instruction %d: %s lives in space %d", idx, what,
             space);
  WJ_REQUIRE(off >= 0 && n >= 1 && off + n <= space_limit(h, space), "wj_vadg_create: instruction %d: %s reaches float %lld of %lld", idx, what,
             (long long)(off + n), (long long)space_limit(h, space));
  return WJ_OK;
}

void read_view(const int32_t* w, View* v) {
  v->space = w[0]; v->offset = w[1];
  for (int d = 0; d < kMaxDims; ++d) { v->shape[d] = w[2 + d]; v->stride[d] = w[2 + kMaxDims + d]; }
}

float w2f(int32_t w) { float f; memcpy(&f, &w, 4); return f; }

inline int out_numel(const View& v) { return v.shape[0] * v.shape[1] * v.shape[2] * v.shape[3]; }

// 1: the view walks its elements in flat order (contiguous strides for `shape`, size-1 axes ignored); 2: one broadcast scalar
int flat_mode(const View& v, const int32_t* shape) {
  bool contig = true, scalar = true;
  int64_t acc = 1;
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (shape[d] == 1) continue;
    if (v.stride[d] != acc) contig = false;
    if (v.stride[d] != 0) scalar = false;
    acc *= shape[d];
  }
  return contig ? 1 : scalar ? 2 : 0;
}

#define WJ_TRYV(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

int parse_program(wj_vadg* h, const int32_t* words, int n_words, int n_instr) {
  int pos = 0;
  for (int idx = 0; idx < n_instr; ++idx) {
    WJ_REQUIRE(pos + 2 <= n_words, "wj_vadg_create: program ends inside instruction %d", idx);
    const int op = words[pos], n = words[pos + 1];
    WJ_REQUIRE(n >= 2 && pos + n <= n_words, "wj_vadg_create: instruction %d has a bad length %d", idx, n);
    const int32_t* a = words + pos + 2;
    const int na = n - 2;
    Instr in;
    memset(&in, 0, sizeof(in));
    in.op = op;
    if (op == OP_EW) {
      WJ_REQUIRE(na == 4 + 4 * kViewWords, "wj_vadg_create: instruction %d: element-wise payload of %d words", idx, na);
      in.ew.fn = a[0]; in.ew.nin = a[1]; in.ew.p0 = w2f(a[2]); in.ew.p1 = w2f(a[3]);
      WJ_REQUIRE(in.ew.fn >= 0 && in.ew.fn < EW_COUNT && in.ew.nin >= 1 && in.ew.nin <= 3, "wj_vadg_create: instruction %d: element-wise function %d / %d operands", idx, in.ew.fn, in.ew.nin);
      read_view(a + 4, &in.ew.out);
      WJ_TRYV(check_view(h, in.ew.out, true, "the result", idx));
      in.ew.flat = flat_mode(in.ew.out, in.ew.out.shape) == 1;
      for (int i = 0; i < in.ew.nin; ++i) {
        read_view(a + 4 + kViewWords * (1 + i), &in.ew.in[i]);
        for (int d = 0; d < kMaxDims; ++d) in.ew.in[i].shape[d] = in.ew.out.shape[d];     // operands are indexed with the RESULT's shape
        WJ_TRYV(check_view(h, in.ew.in[i], false, "an operand", idx));
        in.ew.in_mode[i] = flat_mode(in.ew.in[i], in.ew.out.shape);
        if (!in.ew.in_mode[i]) in.ew.flat = 0;
      }
      in.items = out_numel(in.ew.out);
    } else if (op == OP_CONV1D) {
      WJ_REQUIRE(na == 2 * kViewWords + 14, "wj_vadg_create: instruction %d: conv1d payload of %d words", idx, na);
      read_view(a, &in.conv.out); read_view(a + kViewWords, &in.conv.in);
      const int32_t* p = a + 2 * kViewWords;
      in.conv.w_off = p[0]; in.conv.b_off = p[1]; in.conv.cout = p[2]; in.conv.cin = p[3]; in.conv.k = p[4]; in.conv.t = p[5]; in.conv.tout = p[6];
      in.conv.stride = p[7]; in.conv.padding = p[8]; in.conv.dilation = p[9]; in.conv.groups = p[10];
      in.conv.ws_space = p[11]; in.conv.ws_off = p[12]; in.conv.ws_stride = p[13];
      WJ_TRYV(check_view(h, in.conv.out, true, "the result", idx));
      WJ_TRYV(check_view(h, in.conv.in, false, "the input", idx));
      ConvArgs& c = in.conv;
      WJ_REQUIRE(c.groups >= 1 && c.cin % c.groups == 0 && c.cout % c.groups == 0 && c.k >= 1 && c.stride >= 1 && c.dilation >= 1 && c.padding >= 0 &&
                 c.in.shape[0] == 1 && c.in.shape[1] == 1 && c.out.shape[0] == 1 && c.out.shape[1] == 1 &&
                 c.in.shape[2] == c.cin && c.in.shape[3] == c.t && c.out.shape[2] == c.cout && c.out.shape[3] == c.tout &&
                 c.tout == (c.t + 2 * c.padding - c.dilation * (c.k - 1) - 1) / c.stride + 1, "wj_vadg_create: instruction %d: inconsistent conv1d geometry", idx);
      WJ_REQUIRE(c.w_off >= 0 && (int64_t)c.w_off + (int64_t)c.cout * (c.cin / c.groups) * c.k <= h->n_consts && c.b_off >= -1 && (int64_t)c.b_off + c.cout <= h->n_consts,
                 "wj_vadg_create: instruction %d: conv1d weights outside the constants", idx);
      if (c.ws_space != -1) {
        const int64_t row = (int64_t)(c.cin / c.groups) * c.k;
        WJ_REQUIRE(c.ws_space == SP_ARENA && c.ws_stride >= row, "wj_vadg_create: instruction %d: conv1d scratch in space %d with row stride %d", idx, c.ws_space, c.ws_stride);
        WJ_TRYV(check_range(h, c.ws_space, c.ws_off, (int64_t)c.cout * c.ws_stride + c.cout, true, "the weight scratch", idx));
      }
      // wide strided convolutions (the conv-STFT): lanes on the output channels over a tap-major copy of the weights
      c.cout_full = 0; c.wt_off = -1; c.vec4 = 0;
      if (c.ws_space == -1 && c.groups == 1 && c.cout >= 64 && (int64_t)c.cin * c.k >= 32 && c.stride > 1) {
        c.cout_full = c.cout / 64 * 64;
        const int64_t rows = (int64_t)c.cin * c.k, rows4 = (rows + 3) / 4;
        c.wt_off = (int32_t)h->consts_host.size();
        h->consts_host.resize(h->consts_host.size() + (size_t)(rows4 * c.cout * 4), 0.f);
        for (int co = 0; co < c.cout; ++co)
          for (int64_t r = 0; r < rows; ++r)
            h->consts_host[(size_t)c.wt_off + (size_t)(((r >> 2) * c.cout + co) * 4 + (r & 3))] = h->consts_host[(size_t)c.w_off + (size_t)(co * rows + r)];
        c.vec4 = c.in.space == SP_ARENA && c.padding == 0 && c.dilation == 1 && c.in.stride[3] == 1 && c.k % 4 == 0 && c.in.offset % 4 == 0 && c.in.stride[2] % 4 == 0 &&
                 c.stride % 4 == 0;
      }
      const int n_tb = (c.tout + kConvTB - 1) / kConvTB;
      in.items = std::max(n_tb * c.cout_full, (c.cout - c.cout_full) * c.tout);
    } else if (op == OP_PAD) {
      WJ_REQUIRE(na == 2 * kViewWords + 4, "wj_vadg_create: instruction %d: pad payload of %d words", idx, na);
      read_view(a, &in.pad.out); read_view(a + kViewWords, &in.pad.in);
      in.pad.left = a[2 * kViewWords]; in.pad.right = a[2 * kViewWords + 1]; in.pad.mode = a[2 * kViewWords + 2]; in.pad.value = w2f(a[2 * kViewWords + 3]);
      WJ_TRYV(check_view(h, in.pad.out, true, "the result", idx));
      WJ_TRYV(check_view(h, in.pad.in, false, "the input", idx));
      const int t = in.pad.in.shape[3];
      WJ_REQUIRE(in.pad.left >= 0 && in.pad.right >= 0 && in.pad.mode >= 0 && in.pad.mode <= 2 && in.pad.out.shape[3] == t + in.pad.left + in.pad.right &&
                 in.pad.in.shape[0] == in.pad.out.shape[0] && in.pad.in.shape[1] == in.pad.out.shape[1] && in.pad.in.shape[2] == in.pad.out.shape[2] &&
                 (in.pad.mode != 1 || (in.pad.left < t && in.pad.right < t)), "wj_vadg_create: instruction %d: inconsistent padding", idx);
      in.items = out_numel(in.pad.out);
    } else if (op == OP_MEAN) {
      WJ_REQUIRE(na == 2 * kViewWords + 3, "wj_vadg_create: instruction %d: mean payload of %d words", idx, na);
      read_view(a, &in.mean.out); read_view(a + kViewWords, &in.mean.in);
      in.mean.r = a[2 * kViewWords]; in.mean.rstride = a[2 * kViewWords + 1]; in.mean.inv = w2f(a[2 * kViewWords + 2]);
      WJ_TRYV(check_view(h, in.mean.out, true, "the result", idx));
      WJ_REQUIRE(in.mean.r >= 1 && in.mean.rstride >= 0, "wj_vadg_create: instruction %d: bad reduction", idx);
      for (int d = 0; d < kMaxDims; ++d)
        WJ_REQUIRE(in.mean.in.shape[d] == in.mean.out.shape[d], "wj_vadg_create: instruction %d: the reduction's operand and result shapes differ", idx);
      WJ_TRYV(check_view(h, in.mean.in, false, "the input", idx));     // the operand view with the reduced axis folded into its extent
      WJ_REQUIRE(view_extent(in.mean.in) + (int64_t)(in.mean.r - 1) * in.mean.rstride <= space_limit(h, in.mean.in.space),
                 "wj_vadg_create: instruction %d: reduction reads past its space", idx);
      in.items = out_numel(in.mean.out) * (in.mean.r >= 16 ? 8 : 1);
    } else if (op == OP_LINEAR) {
      WJ_REQUIRE(na == 9, "wj_vadg_create: instruction %d: linear payload of %d words", idx, na);
      LinearArgs& l = in.lin;
      l.out_space = a[0]; l.out_off = a[1]; l.in_space = a[2]; l.in_off = a[3]; l.w_off = a[4]; l.b_off = a[5]; l.rows = a[6]; l.nin = a[7]; l.nout = a[8];
      WJ_REQUIRE(l.rows >= 1 && l.nin >= 1 && l.nout >= 1 && l.w_off >= 0 && (int64_t)l.w_off + (int64_t)l.nout * l.nin <= h->n_consts && l.b_off >= -1 &&
                 (int64_t)l.b_off + l.nout <= h->n_consts, "wj_vadg_create: instruction %d: linear weights out of range", idx);
      WJ_TRYV(check_range(h, l.out_space, l.out_off, (int64_t)l.rows * l.nout, true, "the result", idx));
      WJ_TRYV(check_range(h, l.in_space, l.in_off, (int64_t)l.rows * l.nin, false, "the input", idx));
      in.items = l.rows * l.nout;
    } else if (op == OP_LSTM) {
      WJ_REQUIRE(na == 29, "wj_vadg_create: instruction %d: lstm payload of %d words", idx, na);
      LstmArgs& l = in.lstm;
      l.y_space = a[0]; l.y_off = a[1]; l.x_space = a[2]; l.x_off = a[3]; l.st_t = a[4]; l.st_f = a[5]; l.t = a[6]; l.nin = a[7]; l.hidden = a[8]; l.layers = a[9];
      l.hn_space = a[10]; l.hn_off = a[11]; l.cn_space = a[12]; l.cn_off = a[13];
      for (int i = 0; i < 12; ++i) l.blob[i] = a[14 + i];
      l.h_slot = a[26]; l.c_slot = a[27]; l.need_hc = a[28];
      WJ_REQUIRE(l.hidden >= 1 && l.hidden <= 128 && l.nin >= 1 && l.nin <= 512 && l.layers >= 1 && l.layers <= 4 && l.t >= 1 && l.st_t >= 0 && l.st_f >= 0,
                 "wj_vadg_create: instruction %d: LSTM geometry (input %d, hidden %d, layers %d) outside 512 / 128 / 4", idx, l.nin, l.hidden, l.layers);
      WJ_REQUIRE(l.y_space == SP_XCHG && l.hn_space == SP_XCHG && l.cn_space == SP_XCHG && (l.x_space == SP_XCHG || l.x_space == SP_CONST),
                 "wj_vadg_create: instruction %d: the LSTM's operands must live in the exchange area", idx);
      WJ_REQUIRE(l.x_off >= 0 && (int64_t)l.x_off + (int64_t)(l.t - 1) * l.st_t + (int64_t)(l.nin - 1) * l.st_f < space_limit(h, l.x_space),
                 "wj_vadg_create: instruction %d: LSTM input out of range", idx);
      const int64_t LH = (int64_t)l.layers * l.hidden;
      WJ_REQUIRE(l.y_off >= 0 && (int64_t)l.y_off + (int64_t)l.t * l.hidden <= h->xchg_floats && l.hn_off >= 0 && l.hn_off + LH <= h->xchg_floats && l.cn_off >= 0 &&
                 l.cn_off + LH <= h->xchg_floats, "wj_vadg_create: instruction %d: LSTM outputs out of range", idx);
      WJ_REQUIRE(l.h_slot >= 0 && l.c_slot >= 0 && l.h_slot + LH <= h->state_floats && l.c_slot + LH <= h->state_floats && (l.h_slot + LH <= l.c_slot || l.c_slot + LH <= l.h_slot),
                 "wj_vadg_create: instruction %d: LSTM state slots (%d, %d) outside the %d state floats", idx, l.h_slot, l.c_slot, h->state_floats);
      for (int ly = 0; ly < l.layers; ++ly) {
        const int64_t nin = ly ? l.hidden : l.nin, G = 4 * (int64_t)l.hidden;
        WJ_REQUIRE(l.blob[3 * ly] >= 0 && l.blob[3 * ly] + nin * G <= h->n_consts && l.blob[3 * ly + 1] >= 0 && l.blob[3 * ly + 1] + l.hidden * G <= h->n_consts &&
                   l.blob[3 * ly + 2] >= 0 && l.blob[3 * ly + 2] + G <= h->n_consts, "wj_vadg_create: instruction %d: LSTM layer %d weights outside the constants", idx, ly);
      }
      l.fast_off = -1;
      if (l.hidden == 64 && l.nin <= 64 && l.layers <= 2 && l.x_space == SP_XCHG) {
        // register image: [layer][gate g * 32 + k][thread tl], thread tl = (wave wv, lane 4 jj + q) holds row g * 64 + j (j = 16 wv + jj),
        // columns 32 q + k of [W_ih (64, zero padded) | W_hh (64)]
        l.fast_off = (int32_t)h->consts_host.size();
        h->consts_host.resize(h->consts_host.size() + (size_t)l.layers * 128 * 256, 0.f);
        for (int ly = 0; ly < l.layers; ++ly) {
          const int nin = ly ? 64 : l.nin;
          const float* wih = h->consts_host.data() + l.blob[3 * ly];        // [nin][256]
          const float* whh = h->consts_host.data() + l.blob[3 * ly + 1];    // [64][256]
          for (int tl = 0; tl < 256; ++tl) {
            const int wv = tl >> 6, lane = tl & 63, jj = lane >> 2, q = lane & 3, j = wv * 16 + jj;
            for (int g = 0; g < 4; ++g)
              for (int k = 0; k < 32; ++k) {
                const int col = 32 * q + k, row = g * 64 + j;
                const float v = col < 64 ? (col < nin ? wih[(size_t)col * 256 + row] : 0.f) : whh[(size_t)(col - 64) * 256 + row];
                h->consts_host[(size_t)l.fast_off + ((size_t)ly * 128 + (size_t)(g * 32 + k)) * 256 + tl] = v;
              }
          }
        }
      }
      in.items = 0;
    } else {
      set_error("wj_vadg_create: instruction %d has unknown opcode %d", idx, op);
      return WJ_E_INVALID;
    }
    h->prog.push_back(in);
    pos += n;
  }
  WJ_REQUIRE(pos == n_words, "wj_vadg_create: %d trailing words after %d instructions", n_words - pos, n_instr);
  return WJ_OK;
}

AttrOnce g_stage_attr;

int ensure_capacity(wj_vadg* h, int64_t windows, hipStream_t s) {
  if (windows <= h->cap_windows) return WJ_OK;
  WJ_HIP(hipStreamSynchronize(s));
  for (void* p : {(void*)h->xchg, (void*)h->arena, h->tables})
    if (p) (void)hipFree(p);
  h->xchg = nullptr; h->arena = nullptr; h->tables = nullptr; h->cap_windows = 0;
  WJ_HIP(hipMalloc(&h->xchg, sizeof(float) * (size_t)h->xchg_floats * (size_t)windows));
  if (!h->fused) WJ_HIP(hipMalloc(&h->arena, sizeof(float) * (size_t)h->arena_floats * (size_t)windows));
  WJ_HIP(hipMalloc(&h->tables, (sizeof(int64_t) + sizeof(int32_t) + sizeof(Segment)) * (size_t)windows));
  h->cap_windows = windows;
  return WJ_OK;
}

}  // namespace

extern "C" {

int wj_vadg_create(wj_ctx* ctx, const int32_t* words, int n_words, int n_instr, const float* consts_host, int64_t n_consts,
                   const float* state_init_host, int state_floats, int64_t arena_floats, int64_t xchg_floats, int input_space, int input_offset,
                   int output_space, int output_offset, int window, int max_windows, int mode, wj_vadg** out) {
  WJ_REQUIRE(ctx && words && consts_host && out, "wj_vadg_create: NULL argument");
  WJ_REQUIRE(n_words > 0 && n_instr > 0 && n_consts > 0 && state_floats >= 0 && (state_floats == 0 || state_init_host) && window >= 1 && max_windows >= 1,
             "wj_vadg_create: empty program or bad sizes");
  WJ_REQUIRE(mode >= 0 && mode <= 2, "wj_vadg_create: mode %d (0 = fused when the arena fits the LDS, 1 = one launch per instruction, 2 = fused or fail)", mode);
  WJ_REQUIRE(arena_floats >= 1 && xchg_floats >= 1 && arena_floats < ((int64_t)1 << 30) && xchg_floats < ((int64_t)1 << 30) && n_consts < ((int64_t)1 << 30),
             "wj_vadg_create: arena, exchange area or constants too large for 32-bit offsets");
  WJ_REQUIRE((input_space == SP_ARENA || input_space == SP_XCHG) && input_offset >= 0 &&
             (int64_t)input_offset + window <= (input_space == SP_ARENA ? arena_floats : xchg_floats),
             "wj_vadg_create: the window's samples lie outside their space");
  WJ_REQUIRE((output_space == SP_ARENA || output_space == SP_XCHG) && output_offset >= 0 &&
             output_offset < (output_space == SP_ARENA ? arena_floats : xchg_floats), "wj_vadg_create: the probability lies outside its space");
  WJ_HIP(hipSetDevice(ctx->device));
  wj_vadg* h = new wj_vadg();
  h->ctx = ctx; h->n_consts = n_consts; h->state_floats = state_floats; h->arena_floats = arena_floats; h->xchg_floats = xchg_floats;
  h->in_space = input_space; h->in_off = input_offset; h->out_space = output_space; h->out_off = output_offset;
  h->window = window;
  h->consts_host.assign(consts_host, consts_host + n_consts);
  h->consts_host.resize((size_t)((n_consts + 3) / 4 * 4), 0.f);
  int rc = parse_program(h, words, n_words, n_instr);
  const size_t lds_bytes = sizeof(float) * (size_t)arena_floats;
  if (!rc) {
    h->fused = mode != 1 && lds_bytes <= 160 * 1024;
    if (mode == 1)         // the cross-check configuration: round 5's general LSTM kernel as well
      for (Instr& in : h->prog)
        if (in.op == OP_LSTM) in.lstm.fast_off = -1;
    if (mode == 2 && !h->fused) {
      set_error("wj_vadg_create: the arena (%zu bytes per window) does not fit the 160 KiB of LDS", lds_bytes);
      rc = WJ_E_INVALID;
    }
  }
  if (!rc) {
    // stages: the instructions between the LSTMs
    VadgStage cur;
    cur.first = 0;
    for (int i = 0; i <= (int)h->prog.size(); ++i) {
      if (i == (int)h->prog.size() || h->prog[i].op == OP_LSTM) {
        cur.count = i - cur.first;
        int items = 64;
        for (int k = cur.first; k < i; ++k) items = std::max(items, h->prog[k].items);
        cur.threads = std::min(kStageThreads, (items + 63) / 64 * 64);
        h->stages.push_back(cur);
        if (i < (int)h->prog.size()) h->lstm_at.push_back(i);
        cur.first = i + 1;
      }
    }
    h->stages.front().threads = std::max(h->stages.front().threads, std::min(kStageThreads, (window + 63) / 64 * 64));
    // windows per launch group: the fall-back's grid.y and 4 GiB of HBM arenas; 2 GiB of exchange areas either way
    int64_t cap = std::min<int64_t>(max_windows, ((int64_t)2 << 30) / (int64_t)(sizeof(float) * xchg_floats));
    if (!h->fused) cap = std::min<int64_t>(std::min<int64_t>(cap, 65535), ((int64_t)4 << 30) / (int64_t)(sizeof(float) * arena_floats));
    h->max_windows = (int)std::max<int64_t>(1, cap);
  }
  hipError_t e = hipSuccess;
  if (!rc) {
    const size_t nc = h->consts_host.size();
    e = hipMalloc(&h->consts, sizeof(float) * nc);
    if (e == hipSuccess) e = hipMemcpy(h->consts, h->consts_host.data(), sizeof(float) * nc, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&h->prog_dev, sizeof(Instr) * h->prog.size());
    if (e == hipSuccess) e = hipMemcpy(h->prog_dev, h->prog.data(), sizeof(Instr) * h->prog.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && state_floats) e = hipMalloc(&h->state_init, sizeof(float) * state_floats);
    if (e == hipSuccess && state_floats) e = hipMemcpy(h->state_init, state_init_host, sizeof(float) * state_floats, hipMemcpyHostToDevice);
    if (e == hipSuccess && h->fused && g_stage_attr.need()) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vadg_stage_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) g_stage_attr.done();
    }
    if (e != hipSuccess) { set_error("wj_vadg_create: %s", hipGetErrorString(e)); rc = WJ_E_HIP; }
    std::vector<float>().swap(h->consts_host);
  }
  if (rc) { wj_vadg_free(h); return rc; }
  *out = h;
  return WJ_OK;
}

int wj_vadg_info(wj_vadg* h, int32_t* out8) {
  WJ_REQUIRE(h && out8, "wj_vadg_info: NULL argument");
  bool fast = !h->lstm_at.empty();
  for (int i : h->lstm_at) fast = fast && h->prog[i].lstm.fast_off >= 0;
  out8[0] = h->fused ? 1 : 0;
  out8[1] = h->fused ? (int32_t)(sizeof(float) * h->arena_floats) : 0;
  out8[2] = (int32_t)h->stages.size();
  out8[3] = fast ? 1 : 0;
  out8[4] = h->max_windows;
  out8[5] = (int32_t)h->lstm_at.size();
  out8[6] = out8[7] = 0;
  return WJ_OK;
}

int wj_vadg_free(wj_vadg* h) {
  if (!h) return WJ_OK;
  (void)hipSetDevice(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->stream);
  for (void* p : {(void*)h->consts, (void*)h->prog_dev, (void*)h->state_init, (void*)h->xchg, (void*)h->arena, (void*)h->state, h->tables})
    if (p) (void)hipFree(p);
  delete h;
  return WJ_OK;
}

int wj_vadg_scores(wj_vadg* h, const float* pcm_dev, const int64_t* offsets_host, const int64_t* prob_offsets_host, int n_streams,
                   float* probs_dev, void* stream) {
  WJ_REQUIRE(h && pcm_dev && offsets_host && prob_offsets_host && probs_dev, "wj_vadg_scores: NULL argument");
  WJ_REQUIRE(n_streams >= 1, "wj_vadg_scores: no streams");
  wj_ctx* ctx = h->ctx;
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->pick(stream);
  const int W = h->window;
  for (int i = 0; i < n_streams; ++i) {
    const int64_t n = offsets_host[i + 1] - offsets_host[i];
    WJ_REQUIRE(n >= 0, "wj_vadg_scores: negative stream length");
    WJ_REQUIRE(prob_offsets_host[i + 1] - prob_offsets_host[i] == (n + W - 1) / W, "wj_vadg_scores: stream %d of %lld samples needs %lld probabilities, the table has %lld",
               i, (long long)n, (long long)((n + W - 1) / W), (long long)(prob_offsets_host[i + 1] - prob_offsets_host[i]));
  }
  const int64_t total = prob_offsets_host[n_streams] - prob_offsets_host[0];
  if (total == 0) return WJ_OK;
  if (h->state_floats && n_streams > h->state_rows) {
    if (h->state) { WJ_HIP(hipStreamSynchronize(s)); (void)hipFree(h->state); h->state = nullptr; h->state_rows = 0; }
    WJ_HIP(hipMalloc(&h->state, sizeof(float) * (size_t)h->state_floats * n_streams));
    h->state_rows = n_streams;
  }
  if (h->state_floats) {
    const int64_t n = (int64_t)h->state_floats * n_streams;
    hipLaunchKernelGGL(vadg_state_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->state, h->state_init, h->state_floats, n_streams);
    WJ_LAUNCH_CHECK();
  }
  const int64_t slab = std::min<int64_t>(h->max_windows, total);
  WJ_TRYV(ensure_capacity(h, slab, s));
  int64_t* d_src = reinterpret_cast<int64_t*>(h->tables);
  int32_t* d_valid = reinterpret_cast<int32_t*>(d_src + h->cap_windows);
  Segment* d_seg = reinterpret_cast<Segment*>(d_valid + h->cap_windows);
  std::vector<int64_t> src((size_t)slab);
  std::vector<int32_t> valid((size_t)slab);
  std::vector<Segment> segs;
  // slabs of consecutive windows in (stream, window) order: a stream may straddle slabs, its state row carries it over
  int stream_i = 0;
  int64_t win_in_stream = 0;
  for (int64_t g0 = 0; g0 < total; g0 += slab) {
    const int n_win = (int)std::min<int64_t>(slab, total - g0);
    segs.clear();
    for (int k = 0; k < n_win; ++k) {
      while (win_in_stream >= prob_offsets_host[stream_i + 1] - prob_offsets_host[stream_i]) { ++stream_i; win_in_stream = 0; }
      const int64_t len = offsets_host[stream_i + 1] - offsets_host[stream_i];
      src[k] = offsets_host[stream_i] + win_in_stream * W;
      valid[k] = (int32_t)std::min<int64_t>(W, len - win_in_stream * W);
      if (segs.empty() || segs.back().stream != stream_i) segs.push_back(Segment{k, 0, stream_i});
      ++segs.back().count;
      ++win_in_stream;
    }
    WJ_HIP(hipMemcpyAsync(d_src, src.data(), sizeof(int64_t) * n_win, hipMemcpyHostToDevice, s));
    WJ_HIP(hipMemcpyAsync(d_valid, valid.data(), sizeof(int32_t) * n_win, hipMemcpyHostToDevice, s));
    WJ_HIP(hipMemcpyAsync(d_seg, segs.data(), sizeof(Segment) * segs.size(), hipMemcpyHostToDevice, s));
    WJ_HIP(hipStreamSynchronize(s));      // the host vectors are reused by the next slab
    StageArgs base;
    memset(&base, 0, sizeof(base));
    base.prog = h->prog_dev; base.consts = h->consts; base.xchg = h->xchg; base.xchg_stride = h->xchg_floats; base.arena = h->arena;
    base.arena_stride = h->arena_floats; base.window = W; base.in_space = h->in_space; base.in_off = h->in_off; base.out_space = h->out_space; base.out_off = h->out_off;
    for (size_t si = 0; si < h->stages.size(); ++si) {
      const VadgStage& stg = h->stages[si];
      const bool is_first = si == 0, is_last = si + 1 == h->stages.size();
      StageArgs a = base;
      if (h->fused) {
        if (stg.count || is_first || is_last) {
          a.first = stg.first; a.count = stg.count;
          if (is_first) { a.pcm = pcm_dev; a.src = d_src; a.valid = d_valid; }
          if (is_last) a.probs = probs_dev + prob_offsets_host[0] + g0;
          static const bool want_clocks = getenv("WJ_VADG_CLOCKS") != nullptr;
          long long* d_ck = nullptr;
          if (want_clocks && hipMalloc(&d_ck, sizeof(long long) * 512) == hipSuccess) (void)hipMemsetAsync(d_ck, 0, sizeof(long long) * 512, s);
          a.clocks = d_ck;
          hipLaunchKernelGGL(vadg_stage_kernel<true>, dim3((unsigned)n_win), dim3((unsigned)stg.threads), sizeof(float) * (size_t)h->arena_floats, s, a);
          WJ_LAUNCH_CHECK();
          if (d_ck) {       // per-instruction cycles of two workgroups, on stderr
            std::vector<long long> ck(512);
            WJ_HIP(hipStreamSynchronize(s));
            WJ_HIP(hipMemcpy(ck.data(), d_ck, sizeof(long long) * 512, hipMemcpyDeviceToHost));
            (void)hipFree(d_ck);
            for (int b = 0; b < 2; ++b) {
              fprintf(stderr, "[vadg clocks] stage %zu (%d windows, %d threads) workgroup %s:", si, n_win, stg.threads, b ? "mid" : "0");
              for (int i = 0; i < std::min(stg.count, 254); ++i) fprintf(stderr, " %d:%lld", stg.first + i, ck[256 * b + 1 + i] - ck[256 * b + i]);
              fprintf(stderr, " total %lld\n", ck[256 * b + std::min(stg.count, 254)] - ck[256 * b]);
            }
          }
        }
      } else {
        if (is_first) {
          StageArgs g = a;
          g.pcm = pcm_dev; g.src = d_src; g.valid = d_valid;
          hipLaunchKernelGGL(vadg_stage_kernel<false>, dim3((unsigned)std::max(1, std::min(64, (W + 255) / 256)), (unsigned)n_win), dim3(256), 0, s, g);
          WJ_LAUNCH_CHECK();
        }
        for (int k = 0; k < stg.count; ++k) {
          StageArgs one = a;
          one.first = stg.first + k; one.count = 1;
          const int items = std::max(1, h->prog[stg.first + k].items);
          hipLaunchKernelGGL(vadg_stage_kernel<false>, dim3((unsigned)std::max(1, std::min(64, (items + 255) / 256)), (unsigned)n_win), dim3(256), 0, s, one);
          WJ_LAUNCH_CHECK();
        }
        if (is_last) {
          StageArgs g = a;
          g.probs = probs_dev + prob_offsets_host[0] + g0;
          hipLaunchKernelGGL(vadg_stage_kernel<false>, dim3(1, (unsigned)n_win), dim3(64), 0, s, g);
          WJ_LAUNCH_CHECK();
        }
      }
      if (!is_last) {
        const LstmArgs& l = h->prog[h->lstm_at[si]].lstm;
        if (l.fast_off >= 0 && l.layers == 1)
          hipLaunchKernelGGL(vadg_lstm64_kernel<1>, dim3((unsigned)segs.size()), dim3(256), 0, s, l, h->consts, h->xchg, h->xchg_floats, h->state, h->state_floats, d_seg);
        else if (l.fast_off >= 0)
          hipLaunchKernelGGL(vadg_lstm64_kernel<2>, dim3((unsigned)segs.size()), dim3(512), 0, s, l, h->consts, h->xchg, h->xchg_floats, h->state, h->state_floats, d_seg);
        else
          hipLaunchKernelGGL(vadg_lstm_kernel, dim3((unsigned)segs.size()), dim3((unsigned)std::max(64, (4 * l.hidden + 63) / 64 * 64)), 0, s, l, h->consts, h->xchg,
                             h->xchg_floats, h->state, h->state_floats, d_seg);
        WJ_LAUNCH_CHECK();
      }
    }
  }
  return WJ_OK;
}

}  // extern "C"
