// vad.hip -- Silero-VAD (v5 / v6 architecture, 16 kHz) window scorer on gfx950.
//
// Replaces the per-window TorchScript forward that silero_vad.get_speech_timestamps drives from the
// host (reference call sites: whisperjav/modules/speech_segmentation/backends/silero_v6.py:205-210,
// silero.py:269-273): context(64) + chunk(512) -> reflect-pad(64) -> conv-STFT(256, hop 128)
// magnitude [129 x 4] -> 4 x (Conv1d k3 + ReLU; strides 1,2,2,1) -> LSTMCell(128) -> ReLU ->
// Conv1d(128 -> 1) -> sigmoid.
//
// The recurrence makes one stream strictly sequential (31.25 windows per audio second), so the
// throughput comes from running every scene of a file concurrently: one workgroup per stream,
// all 256 lanes cooperate on each matrix-vector product, the ~1.2 MB of fp32 weights are stored
// input-major ("[in][out]") so lanes (= output channels) read consecutive addresses out of L2, the
// activations and the LSTM state live in LDS, and each stream's state is reset at its first window
// exactly like upstream's reset_states() per get_speech_timestamps call.
#include <vector>

#include "kernels.hpp"

struct wj_vad {
  wj_ctx* ctx = nullptr;
  float* w = nullptr;  // device copy of the weight blob
  int64_t n = 0;
};

namespace wj {

// blob layout (floats), see whisperjav_amd/vad_weights.py
constexpr int64_t V_STFT = 0;                         // [256 taps][258 ch]
constexpr int64_t V_C1W = V_STFT + 256 * 258;         // [129][3][128]
constexpr int64_t V_C1B = V_C1W + 129 * 3 * 128;      // [128]
constexpr int64_t V_C2W = V_C1B + 128;                // [128][3][64]
constexpr int64_t V_C2B = V_C2W + 128 * 3 * 64;       // [64]
constexpr int64_t V_C3W = V_C2B + 64;                 // [64][3][64]
constexpr int64_t V_C3B = V_C3W + 64 * 3 * 64;        // [64]
constexpr int64_t V_C4W = V_C3B + 64;                 // [64][3][128]
constexpr int64_t V_C4B = V_C4W + 64 * 3 * 128;       // [128]
constexpr int64_t V_WIH = V_C4B + 128;                // [128][512]
constexpr int64_t V_WHH = V_WIH + 128 * 512;          // [128][512]
constexpr int64_t V_LB = V_WHH + 128 * 512;           // [512]  (b_ih + b_hh)
constexpr int64_t V_OW = V_LB + 512;                  // [128]
constexpr int64_t V_OB = V_OW + 128;                  // [1]
constexpr int64_t V_TOTAL = V_OB + 1;

struct VadStream {
  int64_t offset, n, prob_offset;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void vad_kernel(const float* __restrict__ pcm, const VadStream* __restrict__ streams,
                                                  const float* __restrict__ w, float* __restrict__ probs) {
  __shared__ float s_x[640];         // context + chunk + reflect pad
  __shared__ float s_ft[258 * 4];    // conv-STFT output
  __shared__ float s_mag[129 * 6];   // [c][t+1], t = -1..4 zero padded
  __shared__ float s_a1[128 * 6];    // conv1 out [o][t+1], zero padded
  __shared__ float s_p[256];         // partial sums
  __shared__ float s_a2[64 * 4];     // conv2 out [o][t+1] (t = 0,1), zero padded
  __shared__ float s_a3[64 * 3];     // conv3 out [o][t+1] (t = 0), zero padded
  __shared__ float s_a4[128];        // conv4 out = LSTM input
  __shared__ float s_h[128], s_c[128], s_g[512];
  const int tid = threadIdx.x;
  const VadStream st = streams[blockIdx.x];
  const float* x = pcm + st.offset;
  const int n_win = (int)((st.n + 511) / 512);
  if (tid < 128) { s_h[tid] = 0.f; s_c[tid] = 0.f; }
  for (int i = tid; i < 129 * 6; i += 256) s_mag[i] = 0.f;
  for (int i = tid; i < 128 * 6; i += 256) s_a1[i] = 0.f;
  for (int i = tid; i < 64 * 4; i += 256) s_a2[i] = 0.f;
  if (tid < 64 * 3) s_a3[tid] = 0.f;
  __syncthreads();

  for (int win = 0; win < n_win; ++win) {
    const int64_t start = (int64_t)win * 512;
    // ---- input: 64 samples of context (zeros for the first window), 512 new (zero padded), reflect 64
    for (int i = tid; i < 576; i += 256) {
      const int64_t p = start - 64 + i;
      s_x[i] = (p >= 0 && p < st.n) ? x[p] : 0.f;
    }
    __syncthreads();
    if (tid < 64) s_x[576 + tid] = s_x[574 - tid];
    __syncthreads();
    // ---- conv-STFT: 258 channels x 4 frames, 256 taps, stride 128
    for (int c = tid; c < 258; c += 256) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const float* wb = w + V_STFT + c;
#pragma unroll 4
      for (int j = 0; j < 256; ++j) {
        const float wv = wb[(int64_t)j * 258];
        a0 = fmaf(wv, s_x[j], a0);
        a1 = fmaf(wv, s_x[128 + j], a1);
        a2 = fmaf(wv, s_x[256 + j], a2);
        a3 = fmaf(wv, s_x[384 + j], a3);
      }
      s_ft[c * 4 + 0] = a0; s_ft[c * 4 + 1] = a1; s_ft[c * 4 + 2] = a2; s_ft[c * 4 + 3] = a3;
    }
    __syncthreads();
    for (int i = tid; i < 129 * 4; i += 256) {
      const int c = i >> 2, t = i & 3;
      const float re = s_ft[c * 4 + t], im = s_ft[(129 + c) * 4 + t];
      s_mag[c * 6 + t + 1] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    // ---- conv1: 129 -> 128, k3 s1 p1, T 4 -> 4 ; thread = (o, pair of output times)
    {
      const int o = tid & 127, tp = (tid >> 7) * 2;
      float a0 = w[V_C1B + o], a1 = a0;
      const float* wc = w + V_C1W + o;
      for (int c = 0; c < 129; ++c) {
        const float* m = &s_mag[c * 6 + tp];  // m[0] is time tp-1
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float wv = wc[(int64_t)(c * 3 + k) * 128];
          a0 = fmaf(wv, m[k], a0);
          a1 = fmaf(wv, m[k + 1], a1);
        }
      }
      s_a1[o * 6 + tp + 1] = fmaxf(a0, 0.f);
      s_a1[o * 6 + tp + 2] = fmaxf(a1, 0.f);
    }
    __syncthreads();
    // ---- conv2: 128 -> 64, k3 s2 p1, T 4 -> 2 ; thread = (o, t, half of the input channels)
    {
      const int o = tid & 63, t = (tid >> 6) & 1, half = tid >> 7;
      float a = 0.f;
      const float* wc = w + V_C2W + o;
      for (int c = half * 64; c < half * 64 + 64; ++c) {
        const float* m = &s_a1[c * 6 + 2 * t];  // times 2t-1, 2t, 2t+1
#pragma unroll
        for (int k = 0; k < 3; ++k) a = fmaf(wc[(int64_t)(c * 3 + k) * 64], m[k], a);
      }
      s_p[tid] = a;
    }
    __syncthreads();
    if (tid < 128) {
      const int o = tid & 63, t = tid >> 6;
      s_a2[o * 4 + t + 1] = fmaxf(s_p[tid] + s_p[tid + 128] + w[V_C2B + o], 0.f);
    }
    __syncthreads();
    // ---- conv3: 64 -> 64, k3 s2 p1, T 2 -> 1 ; thread = (o, quarter of the input channels)
    {
      const int o = tid & 63, part = tid >> 6;
      float a = 0.f;
      const float* wc = w + V_C3W + o;
      for (int c = part * 16; c < part * 16 + 16; ++c) {
        const float* m = &s_a2[c * 4];  // times -1, 0, 1
#pragma unroll
        for (int k = 0; k < 3; ++k) a = fmaf(wc[(int64_t)(c * 3 + k) * 64], m[k], a);
      }
      s_p[tid] = a;
    }
    __syncthreads();
    if (tid < 64)
      s_a3[tid * 3 + 1] = fmaxf(s_p[tid] + s_p[tid + 64] + s_p[tid + 128] + s_p[tid + 192] + w[V_C3B + tid], 0.f);
    __syncthreads();
    // ---- conv4: 64 -> 128, k3 s1 p1, T 1 -> 1 ; thread = (o, half of the input channels)
    {
      const int o = tid & 127, half = tid >> 7;
      float a = 0.f;
      const float* wc = w + V_C4W + o;
      for (int c = half * 32; c < half * 32 + 32; ++c) {
        const float* m = &s_a3[c * 3];
#pragma unroll
        for (int k = 0; k < 3; ++k) a = fmaf(wc[(int64_t)(c * 3 + k) * 128], m[k], a);
      }
      s_p[tid] = a;
    }
    __syncthreads();
    if (tid < 128) s_a4[tid] = fmaxf(s_p[tid] + s_p[tid + 128] + w[V_C4B + tid], 0.f);
    __syncthreads();
    // ---- LSTM cell: gates i, f, g, o (PyTorch order), 512 rows, thread owns rows tid and tid+256
    {
      float g0 = w[V_LB + tid], g1 = w[V_LB + 256 + tid];
      const float* wi = w + V_WIH + tid;
      const float* wh = w + V_WHH + tid;
#pragma unroll 4
      for (int j = 0; j < 128; ++j) {
        const float xv = s_a4[j], hv = s_h[j];
        g0 = fmaf(wi[(int64_t)j * 512], xv, g0);
        g1 = fmaf(wi[(int64_t)j * 512 + 256], xv, g1);
        g0 = fmaf(wh[(int64_t)j * 512], hv, g0);
        g1 = fmaf(wh[(int64_t)j * 512 + 256], hv, g1);
      }
      s_g[tid] = g0;
      s_g[tid + 256] = g1;
    }
    __syncthreads();
    if (tid < 128) {
      const float ig = sigmoidf_(s_g[tid]), fg = sigmoidf_(s_g[128 + tid]);
      const float gg = tanhf(s_g[256 + tid]), og = sigmoidf_(s_g[384 + tid]);
      const float c = fg * s_c[tid] + ig * gg;
      s_c[tid] = c;
      s_h[tid] = og * tanhf(c);
    }
    __syncthreads();
    // ---- head: sigmoid(w . relu(h) + b), one wavefront reduction
    if (tid < 64) {
      float a = w[V_OW + tid] * fmaxf(s_h[tid], 0.f) + w[V_OW + 64 + tid] * fmaxf(s_h[64 + tid], 0.f);
      a = wave_sum(a);
      if (tid == 0) probs[st.prob_offset + win] = sigmoidf_(a + w[V_OB]);
    }
    __syncthreads();
  }
}

}  // namespace wj

using namespace wj;

extern "C" {

int wj_vad_create(wj_ctx* ctx, const float* weights_host, int64_t n_floats, wj_vad** out) {
  WJ_REQUIRE(ctx && weights_host && out, "wj_vad_create: NULL argument");
  WJ_REQUIRE(n_floats == V_TOTAL, "wj_vad_create: weight blob has %lld floats, expected %lld", (long long)n_floats,
             (long long)V_TOTAL);
  WJ_HIP(hipSetDevice(ctx->device));
  wj_vad* v = new wj_vad();
  v->ctx = ctx;
  v->n = n_floats;
  hipError_t e = hipMalloc(&v->w, sizeof(float) * n_floats);
  if (e == hipSuccess) e = hipMemcpy(v->w, weights_host, sizeof(float) * n_floats, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("wj_vad_create: %s", hipGetErrorString(e));
    if (v->w) (void)hipFree(v->w);
    delete v;
    return WJ_E_HIP;
  }
  *out = v;
  return WJ_OK;
}

int wj_vad_free(wj_vad* v) {
  if (!v) return WJ_OK;
  (void)hipSetDevice(v->ctx->device);
  (void)hipStreamSynchronize(v->ctx->stream);
  if (v->w) (void)hipFree(v->w);
  delete v;
  return WJ_OK;
}

int wj_vad_scores(wj_vad* v, const float* pcm_dev, const int64_t* offsets_host, const int64_t* prob_offsets_host,
                  int n_streams, float* probs_dev, void* stream) {
  WJ_REQUIRE(v && pcm_dev && offsets_host && prob_offsets_host && probs_dev, "wj_vad_scores: NULL argument");
  WJ_REQUIRE(n_streams >= 1, "wj_vad_scores: no streams");
  wj_ctx* ctx = v->ctx;
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->pick(stream);
  std::vector<VadStream> hs(n_streams);
  for (int i = 0; i < n_streams; ++i) {
    hs[i].offset = offsets_host[i];
    hs[i].n = offsets_host[i + 1] - offsets_host[i];
    hs[i].prob_offset = prob_offsets_host[i];
    WJ_REQUIRE(hs[i].n >= 0, "wj_vad_scores: negative stream length");
  }
  int rc = ctx->ensure_scratch(sizeof(VadStream) * n_streams);
  if (rc) return rc;
  WJ_HIP(hipMemcpyAsync(ctx->scratch, hs.data(), sizeof(VadStream) * n_streams, hipMemcpyHostToDevice, s));
  WJ_HIP(hipStreamSynchronize(s));
  hipLaunchKernelGGL(vad_kernel, dim3(n_streams), dim3(256), 0, s, pcm_dev, reinterpret_cast<const VadStream*>(ctx->scratch),
                     v->w, probs_dev);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

}  // extern "C"
