// logmel.hip -- Whisper log-mel front end on gfx950 (fp32, HBM-light, LDS-staged).
//
// Replaces faster_whisper.feature_extractor.FeatureExtractor.__call__ (NumPy on the host, entered
// from whisperjav/modules/faster_whisper_pro_asr.py:819) and whisper.audio.log_mel_spectrogram
// (whisperjav/modules/whisper_pro_asr.py:433).  Two launches per batch of clips:
//
//   power  : one workgroup = 16 consecutive frames of one clip.  The 2800 samples they span are
//            read once, coalesced, into LDS (reflect / zero padding resolved per sample), folded
//            with the periodic Hann window into the even/odd parts
//                e[n] = w[n](s[n] + s[400-n]),  o[n] = w[n](s[n] - s[400-n]),  n = 1..199
//            (w is symmetric, w[0] = 0, w[200] = 1) so the real DFT needs 199 instead of 400 terms:
//                Re X_k = sum e[n] cos(2 pi k n / 400) + (-1)^k s[200],  Im X_k = -sum o[n] sin(...)
//            one lane per frequency bin, twiddles from a 400-entry LDS table, frame data broadcast
//            from LDS as float4.  |X|^2 goes back to LDS, the Slaney filter bank (sparse rows,
//            fp32 weights identical to the reference's) is applied, log10 is taken and the clip
//            maximum is accumulated with an order-independent atomic max -> bit-reproducible.
//   final  : clamp to (clip max - 8), (x + 4) / 4, zero padding of the frame axis.
#include <math.h>

#include <vector>

#include "kernels.hpp"

namespace wj {

constexpr int NFFT = 400, HOP = 160, NBINS = 201, FPB = 16;  // frames per block
constexpr int SPAN = (FPB - 1) * HOP + NFFT;                   // 2800 samples

struct MelTables {
  float* twiddle = nullptr;  // [800]: cos[400], sin[400]
  float* hann = nullptr;     // [400]
  float* filt[2] = {nullptr, nullptr};   // dense [n_mels][201] for 80 / 128
  int* range[2] = {nullptr, nullptr};    // [n_mels][2] first / one-past-last non-zero bin
  bool ready = false;
};
static MelTables g_tables[16];  // per device ordinal

static void mel_filterbank(int n_mels, std::vector<float>& filt, std::vector<int>& range) {
  // Slaney mel scale + Slaney area normalisation (== librosa.filters.mel == whisper's mel_filters.npz)
  const double sr = 16000.0;
  const double lin_step = 200.0 / 3.0, knee_hz = 1000.0, knee_mel = knee_hz / lin_step;
  const double log_step = log(6.4) / 27.0;
  const double top_mel = knee_mel + log((sr / 2.0) / knee_hz) / log_step;
  std::vector<double> hz(n_mels + 2);
  const double step = top_mel / (n_mels + 1);
  for (int i = 0; i < n_mels + 2; ++i) {
    double mel = (i == n_mels + 1) ? top_mel : i * step;
    hz[i] = mel >= knee_mel ? knee_hz * exp(log_step * (mel - knee_mel)) : lin_step * mel;
  }
  filt.assign((size_t)n_mels * NBINS, 0.f);
  range.assign((size_t)n_mels * 2, 0);
  for (int m = 0; m < n_mels; ++m) {
    const double enorm = 2.0 / (hz[m + 2] - hz[m]);
    int lo = NBINS, hi = 0;
    for (int k = 0; k < NBINS; ++k) {
      const double f = k * sr / NFFT;
      const double rising = (f - hz[m]) / (hz[m + 1] - hz[m]);
      const double falling = (hz[m + 2] - f) / (hz[m + 2] - hz[m + 1]);
      double v = fmin(rising, falling);
      if (v < 0.0) v = 0.0;
      const float w = (float)(v * enorm);
      filt[(size_t)m * NBINS + k] = w;
      if (w != 0.f) { lo = k < lo ? k : lo; hi = k + 1; }
    }
    if (lo > hi) lo = hi = 0;
    range[m * 2] = lo;
    range[m * 2 + 1] = hi;
  }
}

static int ensure_tables(wj_ctx* ctx) {
  MelTables& t = g_tables[ctx->device & 15];
  if (t.ready) return WJ_OK;
  std::vector<float> tw(800), hann(400);
  for (int i = 0; i < 400; ++i) {
    const double a = 2.0 * M_PI * i / 400.0;
    tw[i] = (float)cos(a);
    tw[400 + i] = (float)sin(a);
    hann[i] = (float)(0.5 - 0.5 * cos(a));
  }
  WJ_HIP(hipMalloc(&t.twiddle, 800 * sizeof(float)));
  WJ_HIP(hipMalloc(&t.hann, 400 * sizeof(float)));
  WJ_HIP(hipMemcpy(t.twiddle, tw.data(), 800 * sizeof(float), hipMemcpyHostToDevice));
  WJ_HIP(hipMemcpy(t.hann, hann.data(), 400 * sizeof(float), hipMemcpyHostToDevice));
  const int sizes[2] = {80, 128};
  for (int s = 0; s < 2; ++s) {
    std::vector<float> f;
    std::vector<int> r;
    mel_filterbank(sizes[s], f, r);
    WJ_HIP(hipMalloc(&t.filt[s], f.size() * sizeof(float)));
    WJ_HIP(hipMalloc(&t.range[s], r.size() * sizeof(int)));
    WJ_HIP(hipMemcpy(t.filt[s], f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
    WJ_HIP(hipMemcpy(t.range[s], r.data(), r.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  t.ready = true;
  return WJ_OK;
}

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

struct ClipDesc {
  int64_t offset;    // first sample in pcm
  int64_t n;         // samples in the clip
  int64_t n_pad;     // n + zero padding (160 FW / 480000 OW / 0 RAW)
  int32_t n_frames;  // n_pad / 160
  int32_t pad_;
};

__global__ __launch_bounds__(256) void logmel_power_kernel(const float* __restrict__ pcm, const ClipDesc* __restrict__ clips,
                                                           const float* __restrict__ twiddle, const float* __restrict__ hann,
                                                           const float* __restrict__ filt, const int* __restrict__ frange,
                                                           int n_mels, int out_frames, float* __restrict__ out,
                                                           float* __restrict__ clip_max) {
  // LDS: samples[2800] | eo[200][2][16] (later reused as power[16][201]) | tw[800]
  __shared__ __attribute__((aligned(16))) float s_samp[SPAN];
  __shared__ __attribute__((aligned(16))) float s_eo[200 * 2 * FPB];
  __shared__ float s_tw[800];
  __shared__ float s_mid[FPB];
  __shared__ float s_wmax[4];
  const int tid = threadIdx.x;
  const int clip = blockIdx.y;
  const ClipDesc cd = clips[clip];
  const int f0 = blockIdx.x * FPB;
  if (f0 >= cd.n_frames) return;
  const float* x = pcm + cd.offset;

  for (int i = tid; i < 800; i += 256) s_tw[i] = twiddle[i];
  // stage the 2800-sample span: padded index p = f0*160 - 200 + i, reflect at both ends of [0, n_pad)
  for (int i = tid; i < SPAN; i += 256) {
    int64_t p = (int64_t)f0 * HOP - NFFT / 2 + i;
    if (p < 0) p = -p;
    if (p >= cd.n_pad) p = 2 * (cd.n_pad - 1) - p;
    s_samp[i] = (p >= 0 && p < cd.n) ? x[p] : 0.f;
  }
  __syncthreads();
  // fold with the window: eo[n][0][f] = e, eo[n][1][f] = o   (n = 1..199 stored at n)
  for (int i = tid; i < 200 * FPB; i += 256) {
    const int n = i / FPB, f = i % FPB;
    float e = 0.f, o = 0.f;
    if (n > 0) {
      const float w = hann[n];
      const float a = s_samp[f * HOP + n], b = s_samp[f * HOP + NFFT - n];
      e = w * a + w * b;
      o = w * a - w * b;
    }
    s_eo[(n * 2 + 0) * FPB + f] = e;
    s_eo[(n * 2 + 1) * FPB + f] = o;
  }
  if (tid < FPB) s_mid[tid] = s_samp[tid * HOP + 200];  // hann[200] == 1
  __syncthreads();

  float re[FPB], im[FPB];
  const int k = tid;
  if (k < NBINS) {
    const float sign = (k & 1) ? -1.f : 1.f;
#pragma unroll
    for (int f = 0; f < FPB; ++f) { re[f] = sign * s_mid[f]; im[f] = 0.f; }
    int idx = 0;
    for (int n = 1; n < 200; ++n) {
      idx += k;
      if (idx >= NFFT) idx -= NFFT;
      const float c = s_tw[idx], s = s_tw[400 + idx];
      const float4* ep = reinterpret_cast<const float4*>(&s_eo[(n * 2 + 0) * FPB]);
      const float4* op = reinterpret_cast<const float4*>(&s_eo[(n * 2 + 1) * FPB]);
#pragma unroll
      for (int q = 0; q < FPB / 4; ++q) {
        const float4 e4 = ep[q], o4 = op[q];
        re[q * 4 + 0] = fmaf(e4.x, c, re[q * 4 + 0]); im[q * 4 + 0] = fmaf(o4.x, s, im[q * 4 + 0]);
        re[q * 4 + 1] = fmaf(e4.y, c, re[q * 4 + 1]); im[q * 4 + 1] = fmaf(o4.y, s, im[q * 4 + 1]);
        re[q * 4 + 2] = fmaf(e4.z, c, re[q * 4 + 2]); im[q * 4 + 2] = fmaf(o4.z, s, im[q * 4 + 2]);
        re[q * 4 + 3] = fmaf(e4.w, c, re[q * 4 + 3]); im[q * 4 + 3] = fmaf(o4.w, s, im[q * 4 + 3]);
      }
    }
  }
  __syncthreads();  // everyone is done reading s_eo; reuse it as power[f][k]
  if (k < NBINS) {
#pragma unroll
    for (int f = 0; f < FPB; ++f) s_eo[f * NBINS + k] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();

  float lmax = -INFINITY;
  for (int i = tid; i < n_mels * FPB; i += 256) {
    const int m = i / FPB, f = i % FPB;
    const int frame = f0 + f;
    if (frame >= cd.n_frames) continue;
    const int lo = frange[m * 2], hi = frange[m * 2 + 1];
    const float* fw = filt + (int64_t)m * NBINS;
    const float* pw = &s_eo[f * NBINS];
    float acc = 0.f;
    for (int kk = lo; kk < hi; ++kk) acc = fmaf(fw[kk], pw[kk], acc);
    const float lv = log10f(fmaxf(acc, 1e-10f));
    lmax = fmaxf(lmax, lv);
    if (frame < out_frames) out[((int64_t)clip * n_mels + m) * out_frames + frame] = lv;
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0) s_wmax[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) {
    const float v = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]));
    atomic_max_float(&clip_max[clip], v);
  }
}

__global__ __launch_bounds__(256) void logmel_final_kernel(const ClipDesc* __restrict__ clips, const float* __restrict__ clip_max,
                                                           int n_mels, int out_frames, float* __restrict__ out) {
  const int clip = blockIdx.y;
  const int nf = clips[clip].n_frames;
  const float floor_v = clip_max[clip] - 8.0f;
  const int64_t total = (int64_t)n_mels * out_frames;
  float* o = out + (int64_t)clip * total;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int frame = (int)(i % out_frames);
    o[i] = frame < nf ? (fmaxf(o[i], floor_v) + 4.0f) / 4.0f : 0.0f;
  }
}

__global__ void fill_kernel(float* p, float v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

int logmel_run(wj_ctx* ctx, const float* pcm, const int64_t* offsets_host, int n_clips, int n_mels, int mode,
               int out_frames, float* out, hipStream_t s) {
  WJ_REQUIRE(n_mels == 80 || n_mels == 128, "logmel: n_mels must be 80 or 128 (got %d)", n_mels);
  WJ_REQUIRE(mode == WJ_MEL_FW || mode == WJ_MEL_OW || mode == WJ_MEL_RAW, "logmel: unknown mode %d", mode);
  WJ_REQUIRE(n_clips > 0 && out_frames > 0, "logmel: empty batch");
  int rc = ensure_tables(ctx);
  if (rc) return rc;
  const MelTables& t = g_tables[ctx->device & 15];
  const int fi = n_mels == 80 ? 0 : 1;
  std::vector<ClipDesc> cds(n_clips);
  int max_frames = 0;
  for (int i = 0; i < n_clips; ++i) {
    ClipDesc& c = cds[i];
    c.offset = offsets_host[i];
    c.n = offsets_host[i + 1] - offsets_host[i];
    WJ_REQUIRE(c.n > NFFT / 2, "logmel: clip %d has %lld samples; need more than %d", i, (long long)c.n, NFFT / 2);
    c.n_pad = c.n + (mode == WJ_MEL_FW ? 160 : mode == WJ_MEL_OW ? 480000 : 0);
    c.n_frames = (int32_t)(c.n_pad / HOP);
    c.pad_ = 0;
    max_frames = c.n_frames > max_frames ? c.n_frames : max_frames;
  }
  const size_t desc_bytes = align_up(sizeof(ClipDesc) * n_clips, 256);
  const size_t need = desc_bytes + align_up(sizeof(float) * n_clips, 256);
  rc = ctx->ensure_scratch(need);
  if (rc) return rc;
  ClipDesc* d_clips = reinterpret_cast<ClipDesc*>(ctx->scratch);
  float* d_max = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->scratch) + desc_bytes);
  WJ_HIP(hipMemcpyAsync(d_clips, cds.data(), sizeof(ClipDesc) * n_clips, hipMemcpyHostToDevice, s));
  // the descriptor vector dies at return: make the copy complete before that (pageable memcpy is
  // already synchronous w.r.t. the host buffer, the sync keeps this robust for pinned callers)
  WJ_HIP(hipStreamSynchronize(s));
  hipLaunchKernelGGL(fill_kernel, dim3(ceil_div(n_clips, 256)), dim3(256), 0, s, d_max, -INFINITY, n_clips);
  WJ_LAUNCH_CHECK();
  dim3 grid(ceil_div(max_frames, FPB), n_clips);
  hipLaunchKernelGGL(logmel_power_kernel, grid, dim3(256), 0, s, pcm, d_clips, t.twiddle, t.hann, t.filt[fi],
                     t.range[fi], n_mels, out_frames, out, d_max);
  WJ_LAUNCH_CHECK();
  const int fblocks = (int)min((int64_t)1024, ceil_div64((int64_t)n_mels * out_frames, 256));
  hipLaunchKernelGGL(logmel_final_kernel, dim3(fblocks, n_clips), dim3(256), 0, s, d_clips, d_max, n_mels, out_frames,
                     out);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

}  // namespace wj

// --------------------------------------------------------------------------------------------
// Scene detection front end: sum of squares of the PCM16-quantised samples per analysis frame.
// The reference quantises with (audio * 32767).astype(int16) (auditok_backend.py:385) and auditok gates on
// 20 log10(sqrt(mean(x^2))): the integer sums are exact, so the host reproduces auditok's float64 energies bit for
// bit from them.  One workgroup per frame (frames are 50 ms: 800 samples at 16 kHz).
// --------------------------------------------------------------------------------------------
namespace wj {

__global__ __launch_bounds__(256) void frame_sumsq_kernel(const float* __restrict__ pcm, const int64_t* __restrict__ off,
                                                          const int32_t* __restrict__ len, long long* __restrict__ out) {
  __shared__ long long red[4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = pcm + off[f];
  const int n = len[f];
  long long acc = 0;
  for (int i = tid; i < n; i += 256) {
    const int q = (int)(short)(int)(x[i] * 32767.0f);       // float32 product, truncation toward zero, int16 wrap
    acc += (long long)q * q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (tid == 0) out[f] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace wj

extern "C" int wj_frame_sumsq(wj_ctx* ctx, const float* pcm_dev, int64_t n_samples, const int64_t* frame_off_host,
                              const int32_t* frame_len_host, int64_t n_frames, int64_t* sums_out_host, void* stream) {
  using namespace wj;
  WJ_REQUIRE(ctx && pcm_dev && frame_off_host && frame_len_host && sums_out_host, "wj_frame_sumsq: NULL argument");
  if (n_frames <= 0) return WJ_OK;
  for (int64_t i = 0; i < n_frames; ++i)
    WJ_REQUIRE(frame_len_host[i] >= 1 && frame_off_host[i] >= 0 && frame_off_host[i] + frame_len_host[i] <= n_samples,
               "wj_frame_sumsq: frame %lld outside the clip", (long long)i);
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->pick(stream);
  const size_t b_off = align_up(sizeof(int64_t) * (size_t)n_frames, 256), b_len = align_up(sizeof(int32_t) * (size_t)n_frames, 256);
  if (int rc = ctx->ensure_scratch(2 * b_off + b_len)) return rc;
  char* base = reinterpret_cast<char*>(ctx->scratch);
  int64_t* d_off = reinterpret_cast<int64_t*>(base);
  long long* d_out = reinterpret_cast<long long*>(base + b_off);
  int32_t* d_len = reinterpret_cast<int32_t*>(base + 2 * b_off);
  WJ_HIP(hipMemcpyAsync(d_off, frame_off_host, sizeof(int64_t) * (size_t)n_frames, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(d_len, frame_len_host, sizeof(int32_t) * (size_t)n_frames, hipMemcpyHostToDevice, s));
  for (int64_t f0 = 0; f0 < n_frames; f0 += 1 << 30) {
    const unsigned n = (unsigned)std::min<int64_t>(n_frames - f0, 1 << 30);
    hipLaunchKernelGGL(frame_sumsq_kernel, dim3(n), dim3(256), 0, s, pcm_dev, d_off + f0, d_len + f0, d_out + f0);
    WJ_LAUNCH_CHECK();
  }
  WJ_HIP(hipMemcpyAsync(sums_out_host, d_out, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}
