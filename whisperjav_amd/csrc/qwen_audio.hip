// qwen_audio.hip -- the audio tower of Qwen3-ASR (SURVEY.md 8f-3 / BASELINE cfg5): first correct path.
//
// Replaces (reference, un-vendored): the audio encoder inside the ``qwen_asr`` package's model, loaded at
// whisperjav/modules/qwen_asr.py:545-636 and run per VAD group from :638-757.  Architecture as published and as restated
// in oracle/qwen3_ref.py (``audio_tokens``; pinned against transformers.models.qwen3_asr):
//   mel [128][frames] cut into chunks of 100 frames -> three 3x3 stride-2 convolutions over (frequency x time) with GELU
//   (1 -> 480 -> 480 -> 480 channels; 128 x 100 -> 64 x 50 -> 32 x 25 -> 16 x 13) -> linear over (channel, frequency) to
//   d_model -> + a 13-position sinusoid table per chunk -> padding tokens dropped (tokens of all chunks of all clips
//   PACKED) -> pre-LN transformer layers whose self-attention runs inside windows of 8 chunks' worth of tokens -> ln_post
//   -> projector (linear, GELU, linear) to the decoder width.
// The convolutions run on the matrix cores as GEMMs over gathered patches (im2col kernels below + gemm.hip): channels-last
// activations make a patch row 9 contiguous runs of C elements; the last convolution's rows are ordered (chunk, time,
// frequency) so that its output IS the [chunk x time][frequency x channel] operand of the following linear (the linear's
// columns are permuted to that order when the blob is packed).  Everything else reuses the library: LayerNorm, the GEMM
// epilogues (bias, GELU, fp32 residual read-modify-write).  The windowed attention is a plain one-wave-per-(token, head)
// kernel: windows hold <= 104 tokens, the tower is a few per cent of the model's work.
#include <algorithm>
#include <climits>
#include <vector>

#include "kernels.hpp"

namespace wj {
int g_qwen_tower_split = 2;   // wj_tune("qwen_tower_split"), read at wj_qwen_audio_create (float16 towers): 1 = the inputs of every transformer / projector
                              // GEMM, the mel patches, the third convolution's output and the attention's QUERIES travel as [hi | lo] fp16 pairs
                              // (embedding error 4.6e-4 -> 1.06e-4 of their range on the device; oracle study scripts/precision_qwen_tower.py);
                              // 2 (default) = the attention's keys and values as well (-> 6.0e-5) for +45 ms per 1800-clip cfg5 step.  End to end
                              // at the published geometry (decoder mode 5): 4.0e-4 per token at level 2, 7.0e-4 / 8.9e-4 (two clips) at level 1,
                              // profiles/r05_parity_diag_qwen_tower_levels.jsonl; 0 = plain
int g_qwen_conv_kpad = 0;   // wj_tune("qwen_conv_kpad"), read at wj_qwen_audio_create: pad the 3x3 convolutions' K = 9 C to a multiple of 64 so their
                            // GEMMs take the LDS-DMA tile kernel.  Measured (scripts/qwen_tower_kpad_ab.py, 512 clips of 4 s, published tower):
                            // 61.2 ms against 61.7 ms, identical output -- the stem is bound by the patch matrices' HBM traffic, not by operand
                            // staging.  Off; kept as an A/B switch
}
using namespace wj;

namespace {

constexpr int CHUNK = 100, TOK = 13;      // mel frames per chunk; tokens a full chunk yields

// rows (chunk, f, t) of the first convolution: 9 taps of the single input channel, padded to 16 columns
template <typename T>
__global__ __launch_bounds__(256) void im2col_mel_kernel(const float* __restrict__ mel, const int32_t* __restrict__ chunk_clip,
                                                         const int32_t* __restrict__ chunk_f0, T* __restrict__ out, int n_mels,
                                                         int frames_max, int Fo, int To, int64_t n_rows, int split) {
  const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int k = threadIdx.x & 15;
  if (r >= n_rows) return;
  const int c = (int)(r / (Fo * To)), rem = (int)(r - (int64_t)c * Fo * To), f = rem / To, t = rem - f * To;
  float v = 0.f;
  if (k < 9) {
    const int mf = 2 * f - 1 + k / 3, tt = 2 * t - 1 + k % 3;
    if (mf >= 0 && mf < n_mels && tt >= 0 && tt < CHUNK) {
      const int fr = chunk_f0[c] + tt;
      if (fr < frames_max) v = mel[((int64_t)chunk_clip[c] * n_mels + mf) * frames_max + fr];
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (split) { st_split<T>(out + r * 32 + k, 16, v); return; }      // rows [hi(16) | lo(16)] against conv1's weights written twice
  }
  Elem<T>::st(out + r * 16 + k, v);
}

// 3x3 stride-2 patches of a channels-last activation [chunk][Fi][Ti][C] -> rows of 9 * C elements (tap-major).  Row order:
// t_major = 0: (chunk, f, t); 1: (chunk, t, f).
template <typename T>
__global__ __launch_bounds__(256) void im2col_cl_kernel(const T* __restrict__ in, T* __restrict__ out, int Fi, int Ti, int Fo, int To,
                                                        int C, int t_major, int ld) {     // ld >= 9 C: row stride, the tail is zeroed
  const int64_t r = blockIdx.x;
  const int c = (int)(r / (Fo * To)), rem = (int)(r - (int64_t)c * Fo * To);
  const int f = t_major ? rem % Fo : rem / To, t = t_major ? rem / Fo : rem % To;
  T* o = out + r * ld;
  constexpr int V = 16 / (int)sizeof(T);
  if (C % V == 0 && ld % V == 0) {        // 16-byte copies (C = 480 in the published tower: 60 per tap); rows and taps start 16-byte aligned
    const int cv = C / V;
    uint4* ov = reinterpret_cast<uint4*>(o);
    for (int i = threadIdx.x; i < 9 * cv; i += 256) {
      const int tap = i / cv, j = i - tap * cv;
      const int fi = 2 * f - 1 + tap / 3, ti = 2 * t - 1 + tap % 3;
      const bool ok = fi >= 0 && fi < Fi && ti >= 0 && ti < Ti;
      uint4 v = uint4{0u, 0u, 0u, 0u};
      if (ok) v = reinterpret_cast<const uint4*>(in + (((int64_t)c * Fi + fi) * Ti + ti) * C)[j];
      ov[i] = v;
    }
    for (int i = 9 * cv + threadIdx.x; i < ld / V; i += 256) ov[i] = uint4{0u, 0u, 0u, 0u};
    return;
  }
  for (int ch = 9 * C + threadIdx.x; ch < ld; ch += 256) o[ch] = (T)0;
  for (int tap = 0; tap < 9; ++tap) {
    const int fi = 2 * f - 1 + tap / 3, ti = 2 * t - 1 + tap % 3;
    const bool ok = fi >= 0 && fi < Fi && ti >= 0 && ti < Ti;
    const T* src = in + (((int64_t)c * Fi + (ok ? fi : 0)) * Ti + (ok ? ti : 0)) * C;
    for (int ch = threadIdx.x; ch < C; ch += 256) o[(int64_t)tap * C + ch] = ok ? src[ch] : (T)0;
  }
}

// packed token i = (chunk, t): x[i] = conv_out[chunk * 13 + t] + pos[t]
__global__ __launch_bounds__(256) void pos_select_kernel(const float* __restrict__ y, const float* __restrict__ pos,
                                                         const int32_t* __restrict__ tok_src, float* __restrict__ x, int D) {
  const int i = blockIdx.x, src = tok_src[i], t = src % TOK;
  for (int c = threadIdx.x; c < D; c += 256) x[(int64_t)i * D + c] = y[(int64_t)src * D + c] + pos[(int64_t)t * D + c];
}

// Non-causal attention inside [lo, hi) of the packed token axis, head_dim 64.  One wave per (QB consecutive query tokens,
// head): the QB queries share every K / V load (consecutive tokens almost always share their window; at a window edge the
// wave walks the union of the two windows and masks per query), which divides the L2 traffic that bounded the one-query form
// (26 KB of K / V re-read per query and head) by QB.  A lane is (kq = lane >> 3, dq = lane & 7): one load instruction
// covers 8 keys x 128 B, lane (kq, dq) holding dims 8 dq .. 8 dq + 7 of key base + kq; the 8 lanes of a key reduce the dot
// product with three exchanges; the softmax weights stay in the score registers for the value pass (see gqa_attn_kernel).
template <typename T, int QB>
__global__ __launch_bounds__(64) void win_attn_kernel(const T* __restrict__ qkv, const int32_t* __restrict__ win_lo,
                                                      const int32_t* __restrict__ win_hi, T* __restrict__ out, int D, int N, int split) {
  // split (bit 0): qkv rows are [hi(3D) | lo(3D)] (the QKV GEMM's split_out) and the output rows [hi(D) | lo(D)]; the query is
  // read as hi + lo (one extra load per query).  Bit 1: keys and values as hi + lo too -- twice the K / V loads, which are what this
  // kernel is bound by (2.1x its time, profiles/r05_rocprofv3_kernel_stats_cfg5.csv); wj_tune qwen_tower_split 2
  const int64_t ldq = (int64_t)(split ? 6 : 3) * D, lo_off = split ? 3 * D : 0, kv_lo = (split & 2) ? 3 * D : 0;
  split &= 1;
  constexpr int STEPS = 8;
  const int i0 = blockIdx.x * QB, h = blockIdx.y, lane = threadIdx.x, kq = lane >> 3, dq = lane & 7;
  int lo[QB], hi[QB], ulo = INT_MAX, uhi = 0;
  float qv[QB][8], acc[QB][8], run_max[QB], run_sum[QB];
#pragma unroll
  for (int b = 0; b < QB; ++b) {
    const int i = min(i0 + b, N - 1);
    lo[b] = win_lo[i]; hi[b] = win_hi[i];
    ulo = min(ulo, lo[b]); uhi = max(uhi, hi[b]);
    ld8(qkv + (int64_t)i * ldq + h * 64 + dq * 8, qv[b]);
    if (lo_off) {
      float t8[8];
      ld8(qkv + (int64_t)i * ldq + lo_off + h * 64 + dq * 8, t8);
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[b][e] += t8[e];
    }
    run_max[b] = -INFINITY; run_sum[b] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
  }
  const T* Kb = qkv + D + h * 64 + dq * 8;
  const T* Vb = qkv + 2 * D + h * 64 + dq * 8;
  for (int base = ulo; base < uhi; base += 8 * STEPS) {
    float s[QB][STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int key = base + st * 8 + kq;
      if (base + st * 8 < uhi) {                    // wave-uniform
        float kv[8];
        ld8(Kb + (int64_t)min(key, uhi - 1) * ldq, kv);
        if (kv_lo) {
          float t8[8];
          ld8(Kb + (int64_t)min(key, uhi - 1) * ldq + kv_lo, t8);
#pragma unroll
          for (int e = 0; e < 8; ++e) kv[e] += t8[e];
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
          float d = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) d = fmaf(qv[b][e], kv[e], d);
          d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
          s[b][st] = (key >= lo[b] && key < hi[b]) ? d * 0.125f : -INFINITY;
        }
      } else {
#pragma unroll
        for (int b = 0; b < QB; ++b) s[b][st] = -INFINITY;
      }
    }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      float cm = s[b][0];
#pragma unroll
      for (int st = 1; st < STEPS; ++st) cm = fmaxf(cm, s[b][st]);
      cm = fmaxf(cm, __shfl_xor(cm, 8, 64)); cm = fmaxf(cm, __shfl_xor(cm, 16, 64)); cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      const float new_max = fmaxf(run_max[b], cm);
      const float ref = new_max == -INFINITY ? 0.f : new_max;      // no key of this query seen yet: every weight below is exp(-inf) = 0
      const float corr = expf(run_max[b] - ref);
      float ps = 0.f;
#pragma unroll
      for (int st = 0; st < STEPS; ++st) { s[b][st] = expf(s[b][st] - ref); ps += s[b][st]; }
      ps += __shfl_xor(ps, 8, 64); ps += __shfl_xor(ps, 16, 64); ps += __shfl_xor(ps, 32, 64);
      run_sum[b] = run_sum[b] * corr + ps;
      run_max[b] = new_max;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[b][e] *= corr;
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if (base + st * 8 < uhi) {
        float vv[8];
        ld8(Vb + (int64_t)min(base + st * 8 + kq, uhi - 1) * ldq, vv);
        if (kv_lo) {
          float t8[8];
          ld8(Vb + (int64_t)min(base + st * 8 + kq, uhi - 1) * ldq + kv_lo, t8);
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] += t8[e];
        }
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[b][e] = fmaf(s[b][st], vv[e], acc[b][e]);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < QB; ++b) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[b][e];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      acc[b][e] = v / run_sum[b];
    }
    if (kq == 0 && i0 + b < N) {
      if (split) {      // rows [hi(D) | lo(D)]: the out-projection reads them as split activations
        T* dst = out + (int64_t)(i0 + b) * 2 * D + h * 64 + dq * 8;
        st4_split(dst, D, acc[b]); st4_split(dst + 4, D, acc[b] + 4);
      } else {
        T* dst = out + (int64_t)(i0 + b) * D + h * 64 + dq * 8;
        st4(dst, acc[b]); st4(dst + 4, acc[b] + 4);
      }
    }
  }
}

}  // namespace

struct wj_qwen_audio {
  wj_ctx* ctx = nullptr;
  wj_qwen_audio_dims d{};
  int dtype = WJ_F16;
  size_t esz = 2;
  const char* blob = nullptr;
  std::vector<int64_t> off;
  int max_chunks = 0;
  std::vector<void*> allocs;
  void *col = nullptr, *a1 = nullptr, *a2 = nullptr, *a3 = nullptr;   // im2col buffer, activations of the three convolutions
  // 16-bit types: K = 9 C of the second and third convolution (4320 in the published tower) padded to a multiple of 64 so that their
  // GEMMs qualify for the LDS-DMA tile kernels: zero columns in a device copy of the two weight matrices and in the patch rows
  int split = 0;             // float16: [hi | lo] activations into every transformer / projector GEMM (g_qwen_tower_split)
  int kp = 0;                // padded K, 0 = off
  void *w2p = nullptr, *w3p = nullptr;   // [C][kp]
  // split towers: conv1's weights [C][16] written twice per row ([C][32]) and conv_out's [D][16][C] as [D][16][2C], so that rows
  // holding [hi | lo] pairs (the first patch matrix; the third convolution's output) multiply as plain GEMMs of twice the K
  void *w1d = nullptr, *wcod = nullptr;
  float* y = nullptr;        // f32 [chunks * 13][D]   conv_out
  float* x = nullptr;        // f32 [tokens][D]        residual stream
  void *h = nullptr, *qkv = nullptr, *attn = nullptr, *ff = nullptr;
  int32_t *chunk_clip = nullptr, *chunk_f0 = nullptr, *tok_src = nullptr, *win_lo = nullptr, *win_hi = nullptr;
  const void* W(int i) const { return blob + off[i]; }
  const float* F(int i) const { return reinterpret_cast<const float*>(blob + off[i]); }
  int layer_base(int l) const { return WJ_QA_N_GLOBAL + l * WJ_QAL_N; }
};

namespace {
#define WJ_TRYA(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define TPA(T, p) reinterpret_cast<T*>(p)

int aalloc(wj_qwen_audio* m, void** p, size_t bytes) {
  bytes = align_up(bytes ? bytes : 256, 256);
  WJ_HIP(hipMalloc(p, bytes));
  m->allocs.push_back(*p);
  return WJ_OK;
}

int post_cnn(int n) {
  for (int i = 0; i < 3; ++i) n = n > 0 ? (n - 1) / 2 + 1 : 0;
  return n;
}
}  // namespace

extern "C" {

int wj_qwen_audio_free(wj_qwen_audio* m) {
  if (!m) return WJ_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
  return WJ_OK;
}

int wj_qwen_audio_create(wj_ctx* ctx, const wj_qwen_audio_dims* dims, int dtype, const void* blob_dev, size_t blob_bytes,
                         const int64_t* offsets_host, int n_offsets, int max_chunks, wj_qwen_audio** out) {
  WJ_REQUIRE(ctx && dims && blob_dev && offsets_host && out, "wj_qwen_audio_create: NULL argument");
  const wj_qwen_audio_dims& d = *dims;
  WJ_REQUIRE(d.n_mels == 128 && d.n_window == 50, "wj_qwen_audio_create: 128 mel bins and 100-frame chunks (n_window 50) expected");
  WJ_REQUIRE(d.d_model % 64 == 0 && d.d_model / 64 == d.n_head && d.d_model <= 1280, "wj_qwen_audio_create: head_dim 64 and d_model <= 1280 expected");
  WJ_REQUIRE(d.conv_hidden % 8 == 0 && d.ffn % 8 == 0 && d.out_dim % 8 == 0 && d.n_layer >= 1 && d.n_window_infer % 100 == 0 && d.n_window_infer >= 100,
             "wj_qwen_audio_create: bad dimensions");
  WJ_REQUIRE(dtype == WJ_F32 || dtype == WJ_F16 || dtype == WJ_BF16, "wj_qwen_audio_create: unknown dtype %d", dtype);
  WJ_REQUIRE(n_offsets == WJ_QA_N_GLOBAL + d.n_layer * WJ_QAL_N, "wj_qwen_audio_create: %d tensor offsets expected, got %d",
             WJ_QA_N_GLOBAL + d.n_layer * WJ_QAL_N, n_offsets);
  WJ_REQUIRE(max_chunks >= 1, "wj_qwen_audio_create: max_chunks >= 1");
  for (int i = 0; i < n_offsets; ++i)
    WJ_REQUIRE(offsets_host[i] >= 0 && (size_t)offsets_host[i] < blob_bytes && offsets_host[i] % 16 == 0, "wj_qwen_audio_create: bad offset %d", i);
  WJ_HIP(hipSetDevice(ctx->device));
  wj_qwen_audio* m = new wj_qwen_audio();
  m->ctx = ctx; m->d = d; m->dtype = dtype; m->esz = dtype_size(dtype);
  m->blob = reinterpret_cast<const char*>(blob_dev);
  m->off.assign(offsets_host, offsets_host + n_offsets);
  m->max_chunks = max_chunks;
  const size_t e = m->esz, C = d.conv_hidden, NC = max_chunks, N = NC * TOK, D = d.d_model;
  int rc = 0;
#define AA(field, bytes) do { if (!rc) rc = aalloc(m, reinterpret_cast<void**>(&m->field), (bytes)); } while (0)
  {
    const size_t k9 = 9 * C, kpad = (k9 + 63) / 64 * 64;
    m->kp = (g_qwen_conv_kpad && dtype != WJ_F32 && kpad != k9) ? (int)kpad : 0;
  }
  // split GEMMs above 512 rows run on the LDS-DMA tile kernel, which needs K % 64 == 0 for every [hi | lo] input: towers of other
  // geometries keep plain float16 activations (as wj_qwen_create gates its own split mode)
  const bool split_ok = d.d_model % 64 == 0 && d.ffn % 64 == 0;      // (the published tower: 1024 / 4096; its conv_hidden 480 only feeds K = 16 x 2 x 480 = 15360)
  m->split = (dtype == WJ_F16 && g_qwen_tower_split && split_ok) ? (g_qwen_tower_split >= 2 ? 3 : 1) : 0;
  const size_t sp = m->split ? 2 : 1;
  const size_t Kc = m->kp ? (size_t)m->kp : 9 * C;
  AA(col, NC * 32 * 25 * Kc * e);                    // the largest patch matrix (second convolution)
  if (m->kp) { AA(w2p, C * Kc * e); AA(w3p, C * Kc * e); }
  AA(a1, NC * 64 * 50 * C * e); AA(a2, NC * 32 * 25 * C * e); AA(a3, NC * TOK * 16 * C * e * sp);
  if (m->split) { AA(w1d, C * 32 * e); AA(wcod, D * 16 * 2 * C * e); }
  AA(y, N * D * sizeof(float)); AA(x, N * D * sizeof(float));
  AA(h, N * std::max<size_t>(D, d.out_dim) * e * sp); AA(qkv, N * 3 * D * e * sp); AA(attn, N * D * e * sp); AA(ff, N * (size_t)d.ffn * e * sp);
  AA(chunk_clip, NC * 4); AA(chunk_f0, NC * 4); AA(tok_src, N * 4); AA(win_lo, N * 4); AA(win_hi, N * 4);
#undef AA
  if (!rc && m->kp) {
    const int widx[2] = {WJ_QA_CONV2_W, WJ_QA_CONV3_W};
    void* dst[2] = {m->w2p, m->w3p};
    hipError_t he = hipSuccess;
    for (int i = 0; i < 2 && he == hipSuccess; ++i) {
      he = hipMemsetAsync(dst[i], 0, C * Kc * e, ctx->stream);
      if (he == hipSuccess)
        he = hipMemcpy2DAsync(dst[i], Kc * e, m->W(widx[i]), 9 * C * e, 9 * C * e, C, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    if (he != hipSuccess) { set_error("wj_qwen_audio_create: padding the convolution weights failed: %s", hipGetErrorString(he)); rc = WJ_E_HIP; }
  }
  if (!rc && m->split) {
    hipError_t he = hipSuccess;
    for (int half = 0; half < 2 && he == hipSuccess; ++half) {
      he = hipMemcpy2DAsync(reinterpret_cast<char*>(m->w1d) + half * 16 * e, 32 * e, m->W(WJ_QA_CONV1_W), 16 * e, 16 * e, C, hipMemcpyDeviceToDevice, ctx->stream);
      if (he == hipSuccess)
        he = hipMemcpy2DAsync(reinterpret_cast<char*>(m->wcod) + half * C * e, 2 * C * e, m->W(WJ_QA_CONVOUT_W), C * e, C * e, D * 16, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    if (he != hipSuccess) { set_error("wj_qwen_audio_create: duplicating the split-path weights failed: %s", hipGetErrorString(he)); rc = WJ_E_HIP; }
  }
  if (rc) { wj_qwen_audio_free(m); return rc; }
  *out = m;
  return WJ_OK;
}

int wj_qwen_audio_tokens(int n_frames) {
  int n = (n_frames / CHUNK) * TOK;
  return n + post_cnn(n_frames % CHUNK);
}

int wj_qwen_audio_encode(wj_qwen_audio* m, const float* mel_dev, int n_clips, int frames_max, const int32_t* n_frames_host,
                         float* out_dev, int32_t* n_tokens_out_host, void* stream) {
  WJ_REQUIRE(m && mel_dev && n_frames_host && out_dev && n_tokens_out_host, "wj_qwen_audio_encode: NULL argument");
  WJ_REQUIRE(n_clips >= 1 && frames_max >= 1, "wj_qwen_audio_encode: empty batch");
  const wj_qwen_audio_dims& d = m->d;
  const int D = d.d_model, C = d.conv_hidden, H = d.n_head, dt = m->dtype;
  std::vector<int32_t> cclip, cf0, tsrc, wlo, whi;
  const int win = TOK * (d.n_window_infer / CHUNK);
  for (int c = 0; c < n_clips; ++c) {
    const int nf = n_frames_host[c];
    WJ_REQUIRE(nf >= 1 && nf <= frames_max, "wj_qwen_audio_encode: clip %d has %d frames (buffer %d)", c, nf, frames_max);
    const int first_tok = (int)tsrc.size();
    for (int f0 = 0; f0 < nf; f0 += CHUNK) {
      const int chunk = (int)cclip.size(), n_tok = post_cnn(std::min(CHUNK, nf - f0));
      cclip.push_back(c); cf0.push_back(f0);
      for (int t = 0; t < n_tok; ++t) tsrc.push_back(chunk * TOK + t);
    }
    const int n_tok = (int)tsrc.size() - first_tok;
    n_tokens_out_host[c] = n_tok;
    for (int t = 0; t < n_tok; ++t) {      // attention windows restart with every clip
      const int lo = first_tok + t / win * win;
      wlo.push_back(lo); whi.push_back(std::min(lo + win, first_tok + n_tok));
    }
  }
  const int NC = (int)cclip.size(), N = (int)tsrc.size();
  WJ_REQUIRE(NC <= m->max_chunks, "wj_qwen_audio_encode: %d chunks of 1 s in the batch (max_chunks %d)", NC, m->max_chunks);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  WJ_HIP(hipMemcpyAsync(m->chunk_clip, cclip.data(), 4 * (size_t)NC, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->chunk_f0, cf0.data(), 4 * (size_t)NC, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->tok_src, tsrc.data(), 4 * (size_t)N, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->win_lo, wlo.data(), 4 * (size_t)N, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->win_hi, whi.data(), 4 * (size_t)N, hipMemcpyHostToDevice, s));
  WJ_HIP(hipStreamSynchronize(s));       // the host vectors die at return
  auto gemm = [&](Epi epi, const void* A, int64_t lda, int wi, int bi, int M, int Nn, int K, void* out, int64_t ldc,
                  const void* w_override = nullptr, int split_in = 0, int split_out = 0) -> int {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = w_override ? w_override : m->W(wi); g.ldw = K; g.bias = bi >= 0 ? m->F(bi) : nullptr; g.M = M; g.N = Nn; g.K = K; g.out = out; g.ldc = ldc;
    g.split = split_in; g.split_out = split_out;
    return launch_gemm(dt, epi, g, s, 0);
  };
  const int sp = m->split ? 1 : 0, spm = sp ? 2 : 1, sp_attn = m->split;
  // ---- convolution stem -------------------------------------------------------------------------------------------
  {
    const int64_t rows = (int64_t)NC * 64 * 50;
    const dim3 grid((unsigned)ceil_div64(rows, 16));
    if (dt == WJ_F32) hipLaunchKernelGGL((im2col_mel_kernel<float>), grid, dim3(256), 0, s, mel_dev, m->chunk_clip, m->chunk_f0, TPA(float, m->col), d.n_mels, frames_max, 64, 50, rows, 0);
    else if (dt == WJ_F16) hipLaunchKernelGGL((im2col_mel_kernel<f16_t>), grid, dim3(256), 0, s, mel_dev, m->chunk_clip, m->chunk_f0, TPA(f16_t, m->col), d.n_mels, frames_max, 64, 50, rows, sp);
    else hipLaunchKernelGGL((im2col_mel_kernel<bf16_t>), grid, dim3(256), 0, s, mel_dev, m->chunk_clip, m->chunk_f0, TPA(bf16_t, m->col), d.n_mels, frames_max, 64, 50, rows, 0);
    WJ_LAUNCH_CHECK();
    if (sp) WJ_TRYA(gemm(EPI_GELU_T, m->col, 32, WJ_QA_CONV1_W, WJ_QA_CONV1_B, (int)rows, C, 32, m->a1, C, m->w1d));     // the mel patches as [hi | lo]
    else WJ_TRYA(gemm(EPI_GELU_T, m->col, 16, WJ_QA_CONV1_W, WJ_QA_CONV1_B, (int)rows, C, 16, m->a1, C));
  }
  auto conv = [&](const void* in, int Fi, int Ti, int Fo, int To, int t_major, int wi, int bi, void* out, int split_out = 0) -> int {
    const int64_t rows = (int64_t)NC * Fo * To;
    const int K = m->kp ? m->kp : 9 * C;
    if (dt == WJ_F32) hipLaunchKernelGGL((im2col_cl_kernel<float>), dim3((unsigned)rows), dim3(256), 0, s, TPA(const float, in), TPA(float, m->col), Fi, Ti, Fo, To, C, t_major, K);
    else if (dt == WJ_F16) hipLaunchKernelGGL((im2col_cl_kernel<f16_t>), dim3((unsigned)rows), dim3(256), 0, s, TPA(const f16_t, in), TPA(f16_t, m->col), Fi, Ti, Fo, To, C, t_major, K);
    else hipLaunchKernelGGL((im2col_cl_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, s, TPA(const bf16_t, in), TPA(bf16_t, m->col), Fi, Ti, Fo, To, C, t_major, K);
    WJ_LAUNCH_CHECK();
    const void* wp = !m->kp ? nullptr : (wi == WJ_QA_CONV2_W ? m->w2p : m->w3p);
    return gemm(EPI_GELU_T, m->col, K, wi, bi, (int)rows, C, K, out, (int64_t)C * (split_out ? 2 : 1), wp, 0, split_out);
  };
  WJ_TRYA(conv(m->a1, 64, 50, 32, 25, 0, WJ_QA_CONV2_W, WJ_QA_CONV2_B, m->a2));
  WJ_TRYA(conv(m->a2, 32, 25, 16, TOK, 1, WJ_QA_CONV3_W, WJ_QA_CONV3_B, m->a3, sp));      // rows (chunk, t, f); split: [hi(C) | lo(C)] per row
  if (sp) WJ_TRYA(gemm(EPI_F32, m->a3, 32 * C, WJ_QA_CONVOUT_W, -1, NC * TOK, D, 32 * C, m->y, D, m->wcod));
  else WJ_TRYA(gemm(EPI_F32, m->a3, 16 * C, WJ_QA_CONVOUT_W, -1, NC * TOK, D, 16 * C, m->y, D));
  hipLaunchKernelGGL(pos_select_kernel, dim3(N), dim3(256), 0, s, m->y, m->F(WJ_QA_POS), m->tok_src, m->x, D);
  WJ_LAUNCH_CHECK();
  // ---- transformer layers over the packed tokens ------------------------------------------------------------------
  for (int l = 0; l < d.n_layer; ++l) {
    const int b0 = m->layer_base(l);
    WJ_TRYA(launch_layernorm(dt, m->x, m->F(b0 + WJ_QAL_LN1_W), m->F(b0 + WJ_QAL_LN1_B), m->h, N, D, s, sp));
    WJ_TRYA(gemm(EPI_T, m->h, (int64_t)D * spm, b0 + WJ_QAL_QKV_W, b0 + WJ_QAL_QKV_B, N, 3 * D, D, m->qkv, (int64_t)3 * D * spm, nullptr, sp, sp));
    {
      constexpr int QB = 4;
      const dim3 grid(ceil_div(N, QB), H);
      if (dt == WJ_F32) hipLaunchKernelGGL((win_attn_kernel<float, QB>), grid, dim3(64), 0, s, TPA(const float, m->qkv), m->win_lo, m->win_hi, TPA(float, m->attn), D, N, 0);
      else if (dt == WJ_F16) hipLaunchKernelGGL((win_attn_kernel<f16_t, QB>), grid, dim3(64), 0, s, TPA(const f16_t, m->qkv), m->win_lo, m->win_hi, TPA(f16_t, m->attn), D, N, sp_attn);
      else hipLaunchKernelGGL((win_attn_kernel<bf16_t, QB>), grid, dim3(64), 0, s, TPA(const bf16_t, m->qkv), m->win_lo, m->win_hi, TPA(bf16_t, m->attn), D, N, 0);
    }
    WJ_LAUNCH_CHECK();
    WJ_TRYA(gemm(EPI_RESID_F32, m->attn, (int64_t)D * spm, b0 + WJ_QAL_OUT_W, b0 + WJ_QAL_OUT_B, N, D, D, m->x, D, nullptr, sp));
    WJ_TRYA(launch_layernorm(dt, m->x, m->F(b0 + WJ_QAL_LN2_W), m->F(b0 + WJ_QAL_LN2_B), m->h, N, D, s, sp));
    WJ_TRYA(gemm(EPI_GELU_T, m->h, (int64_t)D * spm, b0 + WJ_QAL_FC1_W, b0 + WJ_QAL_FC1_B, N, d.ffn, D, m->ff, (int64_t)d.ffn * spm, nullptr, sp, sp));
    WJ_TRYA(gemm(EPI_RESID_F32, m->ff, (int64_t)d.ffn * spm, b0 + WJ_QAL_FC2_W, b0 + WJ_QAL_FC2_B, N, D, d.ffn, m->x, D, nullptr, sp));
  }
  WJ_TRYA(launch_layernorm(dt, m->x, m->F(WJ_QA_LNPOST_W), m->F(WJ_QA_LNPOST_B), m->h, N, D, s, sp));
  WJ_TRYA(gemm(EPI_GELU_T, m->h, (int64_t)D * spm, WJ_QA_PROJ1_W, WJ_QA_PROJ1_B, N, D, D, m->attn, (int64_t)D * spm, nullptr, sp, sp));
  WJ_TRYA(gemm(EPI_F32, m->attn, (int64_t)D * spm, WJ_QA_PROJ2_W, WJ_QA_PROJ2_B, N, d.out_dim, D, out_dev, d.out_dim, nullptr, sp));
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

}  // extern "C"
