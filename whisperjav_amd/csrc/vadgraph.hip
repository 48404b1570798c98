// vadgraph.hip -- executor of a lowered TorchScript VAD window scorer (whisperjav_amd/vad_graph.py) on gfx950.
//
// Replaces: the per-window TorchScript forward of the reference's DEFAULT segmenter network, silero-v3.1 / v4.0
// (torch.hub archive, /root/reference/whisperjav/modules/speech_segmentation/backends/silero.py:197-206, called from the
// archive's get_speech_timestamps at :258-273: model(chunk, 16000) on consecutive 1536-sample windows, LSTM state carried in
// the module between calls).  The host side walks the archive's graph and emits a small instruction stream; this file runs
// it: every window of every stream (scene) of a call owns one float32 arena, the stateless instructions (conv-STFT, log
// spectrum normalisation, depthwise / pointwise conv blocks, decoder) run as launches batched over ALL windows, the LSTM
// instruction runs one workgroup per stream sequentially over that stream's windows with (h, c) in LDS -- exactly the
// dependency structure of the reference's loop, without its per-window launch chain.
//
// All arithmetic is float32 (fmaf chains, expf / tanhf / log1pf from the device library): the bar is 1e-5 on the window
// probabilities against the same archive executed by torch.jit on the CPU (tests/test_gpu_vad_graph.py).
//
// Instruction encoding: see OP_* / the "view" layout in whisperjav_amd/vad_graph.py (mirrored below; wj_vadg_create
// validates every offset against the arena / constant / state sizes before anything is launched).
#include <algorithm>
#include <initializer_list>
#include <vector>

#include "common.hpp"

namespace wj {

constexpr int kMaxDims = 4;
constexpr int kViewWords = 2 + 2 * kMaxDims;
enum { OP_EW = 1, OP_CONV1D = 2, OP_PAD = 3, OP_MEAN = 4, OP_LINEAR = 5, OP_LSTM = 6 };
enum { SP_ARENA = 0, SP_CONST = 1, SP_STATE = 2 };
enum { EW_COPY = 0, EW_ADD, EW_SUB, EW_MUL, EW_DIV, EW_RELU, EW_SIGMOID, EW_TANH, EW_EXP, EW_LOG1P, EW_SQRT, EW_ABS, EW_NEG,
       EW_POW_SCALAR, EW_ADD_SCALAR, EW_MUL_SCALAR, EW_FMA, EW_CLAMP, EW_LEAKY_RELU, EW_LOG, EW_RSUB_SCALAR, EW_SILU, EW_HARDTANH,
       EW_COUNT };

struct View {
  int32_t space, offset;
  int32_t shape[kMaxDims];
  int32_t stride[kMaxDims];
};

struct EwArgs { int32_t fn, nin; float p0, p1; View out, in[3]; };
struct ConvArgs { View out, in; int32_t w_off, b_off, cout, cin, k, t, tout, stride, padding, dilation, groups; };
struct PadArgs { View out, in; int32_t left, right, mode; float value; };
struct MeanArgs { View out, in; int32_t r, rstride; float inv; };
struct LinearArgs { int32_t out_off, in_off, w_off, b_off, rows, nin, nout; };
struct LstmArgs {
  int32_t y_off, x_space, x_off, st_t, st_f, t, nin, hidden, layers, hn_off, cn_off;
  int32_t blob[12];   // per layer: W_ih^T [in][4H], W_hh^T [H][4H], b_ih + b_hh [4H] (offsets into the constants)
  int32_t h_slot, c_slot;
};
struct Instr {
  int op;
  union { EwArgs ew; ConvArgs conv; PadArgs pad; MeanArgs mean; LinearArgs lin; LstmArgs lstm; };
  Instr() { memset(this, 0, sizeof(*this)); }
};

struct Segment { int32_t first, count, stream; };   // a stream's windows inside one slab: first arena, how many, state row

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ const float* view_base(const View& v, const float* arena_w, const float* consts) {
  return (v.space == SP_CONST ? consts : arena_w) + v.offset;
}

__device__ __forceinline__ int64_t view_index(const View& v, int i0, int i1, int i2, int i3) {
  return (int64_t)i0 * v.stride[0] + (int64_t)i1 * v.stride[1] + (int64_t)i2 * v.stride[2] + (int64_t)i3 * v.stride[3];
}

// windows of a stream start from the recording: arena[in_off + i] = pcm[src + i] (zero past the stream's end)
__global__ __launch_bounds__(256) void vadg_gather_kernel(const float* __restrict__ pcm, const int64_t* __restrict__ src,
                                                          const int32_t* __restrict__ valid, float* arena, int64_t arena_stride,
                                                          int in_off, int window) {
  const int w = blockIdx.y;
  float* dst = arena + (int64_t)w * arena_stride + in_off;
  const int64_t s = src[w];
  const int n = valid[w];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < window; i += gridDim.x * 256) dst[i] = i < n ? pcm[s + i] : 0.f;
}

__global__ __launch_bounds__(256) void vadg_scatter_kernel(const float* arena, int64_t arena_stride, int out_off, float* probs, int n) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w < n) probs[w] = arena[(int64_t)w * arena_stride + out_off];
}

__global__ __launch_bounds__(256) void vadg_state_init_kernel(float* state, const float* init, int state_floats, int n_streams) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (int64_t)state_floats * n_streams) state[i] = init[i % state_floats];
}

__global__ __launch_bounds__(256) void vadg_ew_kernel(EwArgs a, float* arena, const float* __restrict__ consts, int64_t arena_stride,
                                                      int per_window) {
  const int w = blockIdx.y;
  float* arena_w = arena + (int64_t)w * arena_stride;
  const int d1 = a.out.shape[1], d2 = a.out.shape[2], d3 = a.out.shape[3];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < per_window; e += gridDim.x * 256) {
    const int i3 = e % d3, r3 = e / d3;
    const int i2 = r3 % d2, r2 = r3 / d2;
    const int i1 = r2 % d1, i0 = r2 / d1;
    const float x = view_base(a.in[0], arena_w, consts)[view_index(a.in[0], i0, i1, i2, i3)];
    float y = 0.f, z = 0.f;
    if (a.nin > 1) y = view_base(a.in[1], arena_w, consts)[view_index(a.in[1], i0, i1, i2, i3)];
    if (a.nin > 2) z = view_base(a.in[2], arena_w, consts)[view_index(a.in[2], i0, i1, i2, i3)];
    float r;
    switch (a.fn) {
      case EW_COPY: r = x; break;
      case EW_ADD: r = x + y; break;
      case EW_SUB: r = x - y; break;
      case EW_MUL: r = x * y; break;
      case EW_DIV: r = x / y; break;
      case EW_RELU: r = fmaxf(x, 0.f); break;
      case EW_SIGMOID: r = sigmoidf_(x); break;
      case EW_TANH: r = tanhf(x); break;
      case EW_EXP: r = expf(x); break;
      case EW_LOG1P: r = log1pf(x); break;
      case EW_SQRT: r = sqrtf(x); break;
      case EW_ABS: r = fabsf(x); break;
      case EW_NEG: r = -x; break;
      case EW_POW_SCALAR: r = powf(x, a.p0); break;
      case EW_ADD_SCALAR: r = x + a.p0; break;
      case EW_MUL_SCALAR: r = x * a.p0; break;
      case EW_FMA: r = x * y + z; break;      // contracted to one fma by the compiler, as torch's fused affine is not: |d| <= 1 ulp
      case EW_CLAMP: r = fminf(fmaxf(x, a.p0), a.p1); break;
      case EW_LEAKY_RELU: r = x > 0.f ? x : x * a.p0; break;
      case EW_LOG: r = logf(x); break;
      case EW_RSUB_SCALAR: r = a.p0 - x * a.p1; break;
      case EW_SILU: r = x * sigmoidf_(x); break;
      default: r = fminf(fmaxf(x, a.p0), a.p1); break;   // EW_HARDTANH
    }
    arena_w[a.out.offset + view_index(a.out, i0, i1, i2, i3)] = r;
  }
}

// one thread per (output channel, output position); lanes run along the positions (contiguous output, broadcast weights)
__global__ __launch_bounds__(256) void vadg_conv1d_kernel(ConvArgs a, float* arena, const float* __restrict__ consts, int64_t arena_stride) {
  const int w = blockIdx.y;
  float* arena_w = arena + (int64_t)w * arena_stride;
  const float* x = view_base(a.in, arena_w, consts);
  const int sc = a.in.stride[2], st = a.in.stride[3];     // [1][1][C][T] after the leading-1 padding of a [1, C, T] view
  const int cg = a.cin / a.groups, og = a.cout / a.groups;
  const int total = a.cout * a.tout;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int co = e / a.tout, to = e % a.tout;
    const int g = co / og;
    const float* wt = consts + a.w_off + (int64_t)co * cg * a.k;
    float acc = 0.f;
    const int t0 = to * a.stride - a.padding;
    for (int ci = 0; ci < cg; ++ci) {
      const float* xr = x + (int64_t)(g * cg + ci) * sc;
      for (int kk = 0; kk < a.k; ++kk) {
        const int ti = t0 + kk * a.dilation;
        if (ti >= 0 && ti < a.t) acc = fmaf(wt[ci * a.k + kk], xr[(int64_t)ti * st], acc);
      }
    }
    if (a.b_off >= 0) acc += consts[a.b_off + co];
    arena_w[a.out.offset + (int64_t)co * a.out.stride[2] + (int64_t)to * a.out.stride[3]] = acc;
  }
}

__global__ __launch_bounds__(256) void vadg_pad_kernel(PadArgs a, float* arena, const float* __restrict__ consts, int64_t arena_stride,
                                                       int per_window) {
  const int w = blockIdx.y;
  float* arena_w = arena + (int64_t)w * arena_stride;
  const float* x = view_base(a.in, arena_w, consts);
  const int d1 = a.out.shape[1], d2 = a.out.shape[2], d3 = a.out.shape[3], t = a.in.shape[3];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < per_window; e += gridDim.x * 256) {
    const int i3 = e % d3, r3 = e / d3;
    const int i2 = r3 % d2, r2 = r3 / d2;
    const int i1 = r2 % d1, i0 = r2 / d1;
    int j = i3 - a.left;
    float v;
    if (j >= 0 && j < t) {
      v = x[view_index(a.in, i0, i1, i2, j)];
    } else if (a.mode == 0) {
      v = a.value;
    } else {
      if (a.mode == 1) j = j < 0 ? -j : 2 * (t - 1) - j;       // reflect (edge not repeated)
      else j = j < 0 ? 0 : t - 1;                               // replicate
      v = x[view_index(a.in, i0, i1, i2, j)];
    }
    arena_w[a.out.offset + view_index(a.out, i0, i1, i2, i3)] = v;
  }
}

__global__ __launch_bounds__(256) void vadg_mean_kernel(MeanArgs a, float* arena, const float* __restrict__ consts, int64_t arena_stride,
                                                        int per_window) {
  const int w = blockIdx.y;
  float* arena_w = arena + (int64_t)w * arena_stride;
  const float* x = view_base(a.in, arena_w, consts);
  const int d1 = a.out.shape[1], d2 = a.out.shape[2], d3 = a.out.shape[3];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < per_window; e += gridDim.x * 256) {
    const int i3 = e % d3, r3 = e / d3;
    const int i2 = r3 % d2, r2 = r3 / d2;
    const int i1 = r2 % d1, i0 = r2 / d1;
    const float* p = x + view_index(a.in, i0, i1, i2, i3);
    float acc = 0.f;
    for (int i = 0; i < a.r; ++i) acc += p[(int64_t)i * a.rstride];
    arena_w[a.out.offset + view_index(a.out, i0, i1, i2, i3)] = acc * a.inv;
  }
}

__global__ __launch_bounds__(256) void vadg_linear_kernel(LinearArgs a, float* arena, const float* __restrict__ consts, int64_t arena_stride) {
  const int w = blockIdx.y;
  float* arena_w = arena + (int64_t)w * arena_stride;
  const int total = a.rows * a.nout;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int r = e / a.nout, o = e % a.nout;
    const float* x = arena_w + a.in_off + (int64_t)r * a.nin;
    const float* wt = consts + a.w_off + (int64_t)o * a.nin;
    float acc = 0.f;
    for (int i = 0; i < a.nin; ++i) acc = fmaf(x[i], wt[i], acc);
    if (a.b_off >= 0) acc += consts[a.b_off + o];
    arena_w[a.out_off + e] = acc;
  }
}

// One workgroup per stream segment: the stream's windows in order, the window's time steps in order, the layers in order.
// Thread j < 4H owns gate row j (weights input-major: the lanes of a wavefront read consecutive floats, served by L2);
// h / c of every layer live in LDS for the whole segment and go back to the stream's state row at its end.
__global__ __launch_bounds__(512) void vadg_lstm_kernel(LstmArgs a, float* arena, const float* __restrict__ consts, float* state,
                                                        int state_floats, int64_t arena_stride, const Segment* __restrict__ segs) {
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
    float* arena_w = arena + (int64_t)(seg.first + wi) * arena_stride;
    const float* xs = (a.x_space == SP_CONST ? consts : arena_w) + a.x_off;
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
      if (tid < H) arena_w[a.y_off + (int64_t)t * H + tid] = s_h[a.layers - 1][tid];
    }
    for (int i = tid; i < a.layers * H; i += blockDim.x) {      // the window's final (h, c): the graph may read them
      arena_w[a.hn_off + i] = s_h[i / H][i % H];
      arena_w[a.cn_off + i] = s_c[i / H][i % H];
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

struct wj_vadg {
  wj_ctx* ctx = nullptr;
  std::vector<Instr> prog;
  float* consts = nullptr;
  float* state_init = nullptr;
  int64_t n_consts = 0;
  int state_floats = 0;
  int64_t arena_floats = 0;
  int in_off = 0, out_off = 0, window = 0, max_windows = 0;
  float* arena = nullptr;        // [max_windows][arena_floats]
  float* state = nullptr;        // [streams][state_floats], grown on demand
  int64_t state_rows = 0;
  void* tables = nullptr;        // per slab: src int64[max_windows], valid int32[max_windows], Segment[max_windows]
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

int check_view(const wj_vadg* h, const View& v, bool is_output, const char* what, int idx) {
  WJ_REQUIRE(v.space == SP_ARENA || (v.space == SP_CONST && !is_output), "wj_vadg_create: instruction %d: %s lives in space %d", idx, what, v.space);
  const int64_t lim = v.space == SP_ARENA ? h->arena_floats : h->n_consts;
  const int64_t ext = view_extent(v);
  WJ_REQUIRE(v.offset >= 0 && ext > 0 && ext <= lim, "wj_vadg_create: instruction %d: %s reaches float %lld of %lld", idx, what, (long long)ext,
             (long long)lim);
  return WJ_OK;
}

void read_view(const int32_t* w, View* v) {
  v->space = w[0]; v->offset = w[1];
  for (int d = 0; d < kMaxDims; ++d) { v->shape[d] = w[2 + d]; v->stride[d] = w[2 + kMaxDims + d]; }
}

float w2f(int32_t w) { float f; memcpy(&f, &w, 4); return f; }

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
    in.op = op;
    if (op == OP_EW) {
      WJ_REQUIRE(na == 4 + 4 * kViewWords, "wj_vadg_create: instruction %d: element-wise payload of %d words", idx, na);
      in.ew.fn = a[0]; in.ew.nin = a[1]; in.ew.p0 = w2f(a[2]); in.ew.p1 = w2f(a[3]);
      WJ_REQUIRE(in.ew.fn >= 0 && in.ew.fn < EW_COUNT && in.ew.nin >= 1 && in.ew.nin <= 3, "wj_vadg_create: instruction %d: element-wise function %d / %d operands", idx, in.ew.fn, in.ew.nin);
      read_view(a + 4, &in.ew.out);
      WJ_TRYV(check_view(h, in.ew.out, true, "the result", idx));
      for (int i = 0; i < in.ew.nin; ++i) {
        read_view(a + 4 + kViewWords * (1 + i), &in.ew.in[i]);
        WJ_TRYV(check_view(h, in.ew.in[i], false, "an operand", idx));
        for (int d = 0; d < kMaxDims; ++d) in.ew.in[i].shape[d] = in.ew.out.shape[d];
      }
    } else if (op == OP_CONV1D) {
      WJ_REQUIRE(na == 2 * kViewWords + 11, "wj_vadg_create: instruction %d: conv1d payload of %d words", idx, na);
      read_view(a, &in.conv.out); read_view(a + kViewWords, &in.conv.in);
      const int32_t* p = a + 2 * kViewWords;
      in.conv.w_off = p[0]; in.conv.b_off = p[1]; in.conv.cout = p[2]; in.conv.cin = p[3]; in.conv.k = p[4]; in.conv.t = p[5]; in.conv.tout = p[6];
      in.conv.stride = p[7]; in.conv.padding = p[8]; in.conv.dilation = p[9]; in.conv.groups = p[10];
      WJ_TRYV(check_view(h, in.conv.out, true, "the result", idx));
      WJ_TRYV(check_view(h, in.conv.in, false, "the input", idx));
      const ConvArgs& c = in.conv;
      WJ_REQUIRE(c.groups >= 1 && c.cin % c.groups == 0 && c.cout % c.groups == 0 && c.k >= 1 && c.stride >= 1 && c.dilation >= 1 && c.padding >= 0 &&
                 c.in.shape[2] == c.cin && c.in.shape[3] == c.t && c.out.shape[2] == c.cout && c.out.shape[3] == c.tout &&
                 c.tout == (c.t + 2 * c.padding - c.dilation * (c.k - 1) - 1) / c.stride + 1, "wj_vadg_create: instruction %d: inconsistent conv1d geometry", idx);
      WJ_REQUIRE(c.w_off >= 0 && (int64_t)c.w_off + (int64_t)c.cout * (c.cin / c.groups) * c.k <= h->n_consts && c.b_off >= -1 && (int64_t)c.b_off + c.cout <= h->n_consts,
                 "wj_vadg_create: instruction %d: conv1d weights outside the constants", idx);
    } else if (op == OP_PAD) {
      WJ_REQUIRE(na == 2 * kViewWords + 4, "wj_vadg_create: instruction %d: pad payload of %d words", idx, na);
      read_view(a, &in.pad.out); read_view(a + kViewWords, &in.pad.in);
      in.pad.left = a[2 * kViewWords]; in.pad.right = a[2 * kViewWords + 1]; in.pad.mode = a[2 * kViewWords + 2]; in.pad.value = w2f(a[2 * kViewWords + 3]);
      WJ_TRYV(check_view(h, in.pad.out, true, "the result", idx));
      WJ_TRYV(check_view(h, in.pad.in, false, "the input", idx));
      const int t = in.pad.in.shape[3];
      WJ_REQUIRE(in.pad.left >= 0 && in.pad.right >= 0 && in.pad.mode >= 0 && in.pad.mode <= 2 && in.pad.out.shape[3] == t + in.pad.left + in.pad.right &&
                 (in.pad.mode != 1 || (in.pad.left < t && in.pad.right < t)), "wj_vadg_create: instruction %d: inconsistent padding", idx);
    } else if (op == OP_MEAN) {
      WJ_REQUIRE(na == 2 * kViewWords + 3, "wj_vadg_create: instruction %d: mean payload of %d words", idx, na);
      read_view(a, &in.mean.out); read_view(a + kViewWords, &in.mean.in);
      in.mean.r = a[2 * kViewWords]; in.mean.rstride = a[2 * kViewWords + 1]; in.mean.inv = w2f(a[2 * kViewWords + 2]);
      WJ_TRYV(check_view(h, in.mean.out, true, "the result", idx));
      WJ_REQUIRE(in.mean.r >= 1 && in.mean.rstride >= 0, "wj_vadg_create: instruction %d: bad reduction", idx);
      View whole = in.mean.in;     // the operand view with the reduced axis folded into its extent
      WJ_TRYV(check_view(h, whole, false, "the input", idx));
      const int64_t lim = whole.space == SP_ARENA ? h->arena_floats : h->n_consts;
      WJ_REQUIRE(view_extent(whole) + (int64_t)(in.mean.r - 1) * in.mean.rstride <= lim, "wj_vadg_create: instruction %d: reduction reads past its space", idx);
    } else if (op == OP_LINEAR) {
      WJ_REQUIRE(na == 7, "wj_vadg_create: instruction %d: linear payload of %d words", idx, na);
      LinearArgs& l = in.lin;
      l.out_off = a[0]; l.in_off = a[1]; l.w_off = a[2]; l.b_off = a[3]; l.rows = a[4]; l.nin = a[5]; l.nout = a[6];
      WJ_REQUIRE(l.rows >= 1 && l.nin >= 1 && l.nout >= 1 && l.out_off >= 0 && l.in_off >= 0 && (int64_t)l.out_off + (int64_t)l.rows * l.nout <= h->arena_floats &&
                 (int64_t)l.in_off + (int64_t)l.rows * l.nin <= h->arena_floats && l.w_off >= 0 && (int64_t)l.w_off + (int64_t)l.nout * l.nin <= h->n_consts &&
                 l.b_off >= -1 && (int64_t)l.b_off + l.nout <= h->n_consts, "wj_vadg_create: instruction %d: linear operands out of range", idx);
    } else if (op == OP_LSTM) {
      WJ_REQUIRE(na == 25, "wj_vadg_create: instruction %d: lstm payload of %d words", idx, na);
      LstmArgs& l = in.lstm;
      l.y_off = a[0]; l.x_space = a[1]; l.x_off = a[2]; l.st_t = a[3]; l.st_f = a[4]; l.t = a[5]; l.nin = a[6]; l.hidden = a[7]; l.layers = a[8];
      l.hn_off = a[9]; l.cn_off = a[10];
      for (int i = 0; i < 12; ++i) l.blob[i] = a[11 + i];
      l.h_slot = a[23]; l.c_slot = a[24];
      WJ_REQUIRE(l.hidden >= 1 && l.hidden <= 128 && l.nin >= 1 && l.nin <= 512 && l.layers >= 1 && l.layers <= 4 && l.t >= 1 && l.st_t >= 0 && l.st_f >= 0,
                 "wj_vadg_create: instruction %d: LSTM geometry (input %d, hidden %d, layers %d) outside 512 / 128 / 4", idx, l.nin, l.hidden, l.layers);
      const int64_t lim = l.x_space == SP_ARENA ? h->arena_floats : h->n_consts;
      WJ_REQUIRE((l.x_space == SP_ARENA || l.x_space == SP_CONST) && l.x_off >= 0 && (int64_t)l.x_off + (int64_t)(l.t - 1) * l.st_t + (int64_t)(l.nin - 1) * l.st_f < lim,
                 "wj_vadg_create: instruction %d: LSTM input out of range", idx);
      const int64_t LH = (int64_t)l.layers * l.hidden;
      WJ_REQUIRE(l.y_off >= 0 && (int64_t)l.y_off + (int64_t)l.t * l.hidden <= h->arena_floats && l.hn_off >= 0 && l.hn_off + LH <= h->arena_floats && l.cn_off >= 0 &&
                 l.cn_off + LH <= h->arena_floats, "wj_vadg_create: instruction %d: LSTM outputs out of range", idx);
      WJ_REQUIRE(l.h_slot >= 0 && l.c_slot >= 0 && l.h_slot + LH <= h->state_floats && l.c_slot + LH <= h->state_floats && (l.h_slot + LH <= l.c_slot || l.c_slot + LH <= l.h_slot),
                 "wj_vadg_create: instruction %d: LSTM state slots (%d, %d) outside the %d state floats", idx, l.h_slot, l.c_slot, h->state_floats);
      for (int ly = 0; ly < l.layers; ++ly) {
        const int64_t nin = ly ? l.hidden : l.nin, G = 4 * (int64_t)l.hidden;
        WJ_REQUIRE(l.blob[3 * ly] >= 0 && l.blob[3 * ly] + nin * G <= h->n_consts && l.blob[3 * ly + 1] >= 0 && l.blob[3 * ly + 1] + l.hidden * G <= h->n_consts &&
                   l.blob[3 * ly + 2] >= 0 && l.blob[3 * ly + 2] + G <= h->n_consts, "wj_vadg_create: instruction %d: LSTM layer %d weights outside the constants", idx, ly);
      }
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

inline int out_numel(const View& v) { return v.shape[0] * v.shape[1] * v.shape[2] * v.shape[3]; }
inline dim3 grid_for(int per_window, int n_win) { return dim3((unsigned)std::max(1, std::min(64, (per_window + 255) / 256)), (unsigned)n_win); }

}  // namespace

extern "C" {

int wj_vadg_create(wj_ctx* ctx, const int32_t* words, int n_words, int n_instr, const float* consts_host, int64_t n_consts,
                   const float* state_init_host, int state_floats, int64_t arena_floats, int input_offset, int output_offset, int window,
                   int max_windows, wj_vadg** out) {
  WJ_REQUIRE(ctx && words && consts_host && out, "wj_vadg_create: NULL argument");
  WJ_REQUIRE(n_words > 0 && n_instr > 0 && n_consts > 0 && state_floats >= 0 && (state_floats == 0 || state_init_host) && window >= 1 && max_windows >= 1,
             "wj_vadg_create: empty program or bad sizes");
  WJ_REQUIRE(arena_floats >= window && input_offset >= 0 && (int64_t)input_offset + window <= arena_floats && output_offset >= 0 && output_offset < arena_floats,
             "wj_vadg_create: input / output outside the %lld-float arena", (long long)arena_floats);
  WJ_REQUIRE(arena_floats < ((int64_t)1 << 31) && n_consts < ((int64_t)1 << 31), "wj_vadg_create: arena or constants too large for 32-bit offsets");
  WJ_HIP(hipSetDevice(ctx->device));
  wj_vadg* h = new wj_vadg();
  h->ctx = ctx; h->n_consts = n_consts; h->state_floats = state_floats; h->arena_floats = arena_floats; h->in_off = input_offset; h->out_off = output_offset;
  h->window = window; h->max_windows = max_windows;
  int rc = parse_program(h, words, n_words, n_instr);
  hipError_t e = hipSuccess;
  if (!rc) {
    e = hipMalloc(&h->consts, sizeof(float) * n_consts);
    if (e == hipSuccess) e = hipMemcpy(h->consts, consts_host, sizeof(float) * n_consts, hipMemcpyHostToDevice);
    if (e == hipSuccess && state_floats) e = hipMalloc(&h->state_init, sizeof(float) * state_floats);
    if (e == hipSuccess && state_floats) e = hipMemcpy(h->state_init, state_init_host, sizeof(float) * state_floats, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&h->arena, sizeof(float) * (size_t)arena_floats * max_windows);
    if (e == hipSuccess) e = hipMalloc(&h->tables, (sizeof(int64_t) + sizeof(int32_t) + sizeof(Segment)) * (size_t)max_windows);
    if (e != hipSuccess) { set_error("wj_vadg_create: %s", hipGetErrorString(e)); rc = WJ_E_HIP; }
  }
  if (rc) { wj_vadg_free(h); return rc; }
  *out = h;
  return WJ_OK;
}

int wj_vadg_free(wj_vadg* h) {
  if (!h) return WJ_OK;
  (void)hipSetDevice(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->stream);
  for (void* p : {(void*)h->consts, (void*)h->state_init, (void*)h->arena, (void*)h->state, h->tables})
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
  int64_t* d_src = reinterpret_cast<int64_t*>(h->tables);
  int32_t* d_valid = reinterpret_cast<int32_t*>(d_src + h->max_windows);
  Segment* d_seg = reinterpret_cast<Segment*>(d_valid + h->max_windows);
  std::vector<int64_t> src(h->max_windows);
  std::vector<int32_t> valid(h->max_windows);
  std::vector<Segment> segs;
  // slabs of consecutive windows in (stream, window) order: a stream may straddle slabs, its state row carries it over
  int stream_i = 0;
  int64_t win_in_stream = 0;
  for (int64_t g0 = 0; g0 < total; g0 += h->max_windows) {
    const int n_win = (int)std::min<int64_t>(h->max_windows, total - g0);
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
    hipLaunchKernelGGL(vadg_gather_kernel, grid_for(W, n_win), dim3(256), 0, s, pcm_dev, d_src, d_valid, h->arena, h->arena_floats, h->in_off, W);
    WJ_LAUNCH_CHECK();
    for (const Instr& in : h->prog) {
      switch (in.op) {
        case OP_EW: {
          const int n = out_numel(in.ew.out);
          hipLaunchKernelGGL(vadg_ew_kernel, grid_for(n, n_win), dim3(256), 0, s, in.ew, h->arena, h->consts, h->arena_floats, n);
          break;
        }
        case OP_CONV1D:
          hipLaunchKernelGGL(vadg_conv1d_kernel, grid_for(in.conv.cout * in.conv.tout, n_win), dim3(256), 0, s, in.conv, h->arena, h->consts, h->arena_floats);
          break;
        case OP_PAD: {
          const int n = out_numel(in.pad.out);
          hipLaunchKernelGGL(vadg_pad_kernel, grid_for(n, n_win), dim3(256), 0, s, in.pad, h->arena, h->consts, h->arena_floats, n);
          break;
        }
        case OP_MEAN: {
          const int n = out_numel(in.mean.out);
          hipLaunchKernelGGL(vadg_mean_kernel, grid_for(n, n_win), dim3(256), 0, s, in.mean, h->arena, h->consts, h->arena_floats, n);
          break;
        }
        case OP_LINEAR:
          hipLaunchKernelGGL(vadg_linear_kernel, grid_for(in.lin.rows * in.lin.nout, n_win), dim3(256), 0, s, in.lin, h->arena, h->consts, h->arena_floats);
          break;
        default: {   // OP_LSTM
          const int threads = std::max(64, (4 * in.lstm.hidden + 63) / 64 * 64);
          hipLaunchKernelGGL(vadg_lstm_kernel, dim3((unsigned)segs.size()), dim3(threads), 0, s, in.lstm, h->arena, h->consts, h->state, h->state_floats,
                             h->arena_floats, d_seg);
          break;
        }
      }
      WJ_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(vadg_scatter_kernel, dim3((unsigned)((n_win + 255) / 256)), dim3(256), 0, s, h->arena, h->arena_floats, h->out_off,
                       probs_dev + prob_offsets_host[0] + g0, n_win);
    WJ_LAUNCH_CHECK();
  }
  return WJ_OK;
}

}  // extern "C"
