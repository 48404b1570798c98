// sampler.hip -- vocabulary-wide passes of the decode step: Whisper's logit filters
// (SuppressTokens, SuppressBlank, ApplyTimestampRules), log-softmax, greedy pick, no-speech
// probability and masked top-k for host-driven beam search.  One workgroup per hypothesis row;
// everything is fp32 and deterministic (ties break to the lowest token id).
//
// The filter semantics restate whisper/decoding.py (openai-whisper 20250625: SuppressBlank,
// SuppressTokens, ApplyTimestampRules, GreedyDecoder.update) which CTranslate2 4.7.1 mirrors for
// faster-whisper (reference call sites: whisperjav/modules/faster_whisper_pro_asr.py:819,
// whisperjav/modules/whisper_pro_asr.py:433).
#include "kernels.hpp"

namespace wj {

struct ArgMax {
  float v;
  int i;
};
__device__ __forceinline__ ArgMax amax(ArgMax a, ArgMax b) {
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
__device__ __forceinline__ ArgMax wave_amax(ArgMax a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b;
    b.v = __shfl_xor(a.v, o, 64);
    b.i = __shfl_xor(a.i, o, 64);
    a = amax(a, b);
  }
  return a;
}

struct RowRules {
  int first, ts_rules, last_ts, penult_ts, ts_floor;
};

__device__ __forceinline__ bool token_allowed(int v, const RowRules& rr, const wj_decode_opts& o) {
  if (o.suppress_mask_dev && o.suppress_mask_dev[v]) return false;
  if (rr.first && o.suppress_blank && (v == o.blank || v == o.eot)) return false;
  if (rr.ts_rules) {
    if (v == o.no_timestamps) return false;
    if (rr.last_ts) {
      if (rr.penult_ts) {
        if (v >= o.timestamp_begin) return false;   // a pair just closed: text (or EOT) must follow
      } else {
        if (v < o.eot) return false;                // an opening timestamp must be closed (or EOT)
      }
    }
    if (rr.ts_floor >= 0 && v >= o.timestamp_begin && v < rr.ts_floor) return false;
    if (rr.first) {
      if (v < o.timestamp_begin) return false;
      if (o.max_initial_timestamp_index >= 0 && v > o.timestamp_begin + o.max_initial_timestamp_index) return false;
    }
  }
  return true;
}

// counter-based uniform in (0,1) for (seed, row, step, token): stateless, so sampling is reproducible and
// independent of launch geometry
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t row, uint32_t step, uint32_t v) {
  auto mix = [](uint32_t h) { h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16; return h; };
  uint32_t h = mix(seed ^ (row * 0x9E3779B1u));
  h = mix(h + step * 0x85EBCA77u);
  h = mix(h ^ (v * 0xC2B2AE3Du));
  return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

constexpr int SB = 1024;  // sampler block size (one workgroup per row sweeps the 51.9k-entry vocabulary)

__global__ __launch_bounds__(SB) void greedy_sample_kernel(const GreedyArgs a) {
  __shared__ float s_f[4][SB / 64];
  __shared__ int s_i[2][SB / 64];
  __shared__ RowRules s_rr;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = *a.pos_ptr + 1;  // tokens in the history (prompt + sampled so far)
  int32_t* tok = a.tokens + (int64_t)r * a.tok_stride;
  const wj_decode_opts& o = a.opts;
  if (a.finished[r]) {
    if (tid == 0) {
      tok[len] = o.eot;
      if (a.token_logprob) a.token_logprob[(int64_t)r * a.tok_stride + len] = 0.f;
    }
    return;
  }
  const float rep = o.repetition_penalty;
  const int ngram = o.no_repeat_ngram_size;
  if ((rep > 0.f && rep != 1.f) || ngram > 0) {
    // ctranslate2 RepetitionPenalty then NoRepeatNgram, in place on this row's logits (they are recomputed
    // every step); the processor sequence is [last prompt token] + generated = tok[s0 .. len)
    float* xw = a.logits + (int64_t)r * a.ldl;
    const int s0 = a.sample_begin - 1, L = len - s0;
    if (rep > 0.f && rep != 1.f) {
      for (int j = tid; j < L; j += SB) {
        const int t = tok[s0 + j];
        bool first = true;
        for (int i = 0; i < j; ++i)
          if (tok[s0 + i] == t) { first = false; break; }
        if (first) { const float v = xw[t]; xw[t] = v < 0.f ? v * rep : v / rep; }
      }
      __syncthreads();
    }
    if (ngram > 0 && L >= ngram) {
      for (int i = tid; i <= L - ngram; i += SB) {
        bool match = true;
        for (int k = 0; k < ngram - 1; ++k)
          if (tok[s0 + i + k] != tok[len - (ngram - 1) + k]) { match = false; break; }
        if (match) xw[tok[s0 + i + ngram - 1]] = -INFINITY;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    RowRules rr;
    const int ns = len - a.sample_begin;
    rr.first = ns == 0;
    rr.ts_rules = !o.without_timestamps;
    rr.last_ts = ns >= 1 && tok[len - 1] >= o.timestamp_begin;
    rr.penult_ts = ns < 2 || tok[len - 2] >= o.timestamp_begin;
    int ts_last = -1;
    for (int i = a.sample_begin; i < len; ++i)
      if (tok[i] >= o.timestamp_begin) ts_last = tok[i];
    rr.ts_floor = -1;
    if (ts_last >= 0) rr.ts_floor = (rr.last_ts && !rr.penult_ts) ? ts_last : ts_last + 1;
    s_rr = rr;
  }
  __syncthreads();
  const RowRules rr = s_rr;
  const float* x = a.logits + (int64_t)r * a.ldl;

  // pass A: maxima
  ArgMax best_all = {-INFINITY, 0x7fffffff}, best_ts = {-INFINITY, 0x7fffffff};
  float max_text = -INFINITY;
  for (int v = tid; v < a.V; v += SB) {
    if (!token_allowed(v, rr, o)) continue;
    const float xv = x[v];
    best_all = amax(best_all, ArgMax{xv, v});
    if (v >= o.timestamp_begin) best_ts = amax(best_ts, ArgMax{xv, v});
    else max_text = fmaxf(max_text, xv);
  }
  best_all = wave_amax(best_all);
  best_ts = wave_amax(best_ts);
  max_text = wave_max(max_text);
  if (lane == 0) {
    s_f[0][wave] = best_all.v; s_i[0][wave] = best_all.i;
    s_f[1][wave] = best_ts.v;  s_i[1][wave] = best_ts.i;
    s_f[2][wave] = max_text;
  }
  __syncthreads();
  best_all = ArgMax{s_f[0][0], s_i[0][0]};
  best_ts = ArgMax{s_f[1][0], s_i[1][0]};
  max_text = s_f[2][0];
#pragma unroll
  for (int w = 1; w < SB / 64; ++w) {
    best_all = amax(best_all, ArgMax{s_f[0][w], s_i[0][w]});
    best_ts = amax(best_ts, ArgMax{s_f[1][w], s_i[1][w]});
    max_text = fmaxf(max_text, s_f[2][w]);
  }
  __syncthreads();

  // pass B: partition sums relative to the global max
  float sum_all = 0.f, sum_ts = 0.f;
  for (int v = tid; v < a.V; v += SB) {
    if (!token_allowed(v, rr, o)) continue;
    const float e = expf(x[v] - best_all.v);
    sum_all += e;
    if (v >= o.timestamp_begin) sum_ts += e;
  }
  sum_all = wave_sum(sum_all);
  sum_ts = wave_sum(sum_ts);
  if (lane == 0) { s_f[0][wave] = sum_all; s_f[1][wave] = sum_ts; }
  __syncthreads();
  __shared__ float s_pick[3];   // norm, ts_only, (unused)
  if (tid == 0) {
    sum_all = 0.f; sum_ts = 0.f;
    for (int w = 0; w < SB / 64; ++w) { sum_all += s_f[0][w]; sum_ts += s_f[1][w]; }
    const float lse = best_all.v + logf(sum_all);
    float norm = lse, ts_only = 0.f;
    if (rr.ts_rules && sum_ts > 0.f) {
      // "if the probability mass on timestamps exceeds every single text token, emit a timestamp"
      const float ts_lp = best_all.v + logf(sum_ts) - lse;
      const float text_lp = max_text - lse;
      if (ts_lp > text_lp) { ts_only = 1.f; norm = best_all.v + logf(sum_ts); }   // renormalised over timestamps only
    }
    s_pick[0] = norm;
    s_pick[1] = ts_only;
  }
  __syncthreads();
  const float norm = s_pick[0];
  const bool ts_only = s_pick[1] != 0.f;
  ArgMax pick = ts_only ? best_ts : best_all;      // temperature 0: arg-max of what is left
  if (a.temperature > 0.f) {
    // Gumbel-max draw from softmax(filtered logits / T); the reported log-prob is the UNSCALED one
    // (whisper.decoding.GreedyDecoder.update)
    const float inv_t = 1.0f / a.temperature;
    ArgMax best = {-INFINITY, 0x7fffffff};
    for (int v = tid; v < a.V; v += SB) {
      if (ts_only && v < o.timestamp_begin) continue;
      if (!token_allowed(v, rr, o)) continue;
      const float u = uniform01(a.seed, (uint32_t)(a.row_offset + r), (uint32_t)len, (uint32_t)v);
      best = amax(best, ArgMax{x[v] * inv_t - logf(-logf(u)), v});
    }
    best = wave_amax(best);
    __syncthreads();
    if (lane == 0) { s_f[0][wave] = best.v; s_i[0][wave] = best.i; }
    __syncthreads();
    pick = ArgMax{s_f[0][0], s_i[0][0]};
    for (int w = 1; w < SB / 64; ++w) pick = amax(pick, ArgMax{s_f[0][w], s_i[0][w]});
  }
  if (tid == 0) {
    const int token = pick.i;
    const float lp = x[token] - norm;
    tok[len] = token;
    a.sum_logprob[r] += lp;
    if (a.token_logprob) a.token_logprob[(int64_t)r * a.tok_stride + len] = lp;
    if (token == o.eot) a.finished[r] = 1;
  }
}

int launch_greedy_sample(const GreedyArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(greedy_sample_kernel, dim3(a.R), dim3(SB), 0, s, a);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

__global__ __launch_bounds__(SB) void no_speech_kernel(const float* __restrict__ logits, int64_t ldl, int V, int ns_id,
                                                       float* __restrict__ out) {
  __shared__ float s_f[SB / 64];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)r * ldl;
  float mx = -INFINITY;
  for (int v = tid; v < V; v += SB) mx = fmaxf(mx, x[v]);
  mx = wave_max(mx);
  if (lane == 0) s_f[wave] = mx;
  __syncthreads();
  mx = s_f[0];
  for (int w = 1; w < SB / 64; ++w) mx = fmaxf(mx, s_f[w]);
  __syncthreads();
  float sum = 0.f;
  for (int v = tid; v < V; v += SB) sum += expf(x[v] - mx);
  sum = wave_sum(sum);
  if (lane == 0) s_f[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    sum = 0.f;
    for (int w = 0; w < SB / 64; ++w) sum += s_f[w];
    out[r] = expf(x[ns_id] - mx) / sum;
  }
}

int launch_no_speech_prob(const float* logits, int64_t ldl, int R, int V, int no_speech_id, float* out, hipStream_t s) {
  hipLaunchKernelGGL(no_speech_kernel, dim3(R), dim3(SB), 0, s, logits, ldl, V, no_speech_id, out);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

// masked log-softmax + top-k by k rounds of block arg-max (k <= 16), lowest id wins ties
__global__ __launch_bounds__(SB) void topk_logprob_kernel(const float* __restrict__ logits, int64_t ldl, int V, int k,
                                                          const uint8_t* __restrict__ ban, int32_t* __restrict__ ids,
                                                          float* __restrict__ logprobs, float* __restrict__ lse_out) {
  __shared__ float s_f[SB / 64];
  __shared__ int s_i[SB / 64];
  __shared__ int s_sel[16];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = logits + (int64_t)r * ldl;
  const uint8_t* bm = ban ? ban + (int64_t)r * V : nullptr;
  float mx = -INFINITY;
  for (int v = tid; v < V; v += SB)
    if (!bm || !bm[v]) mx = fmaxf(mx, x[v]);
  mx = wave_max(mx);
  if (lane == 0) s_f[wave] = mx;
  __syncthreads();
  mx = s_f[0];
  for (int w = 1; w < SB / 64; ++w) mx = fmaxf(mx, s_f[w]);
  __syncthreads();
  float sum = 0.f;
  for (int v = tid; v < V; v += SB)
    if (!bm || !bm[v]) sum += expf(x[v] - mx);
  sum = wave_sum(sum);
  if (lane == 0) s_f[wave] = sum;
  __syncthreads();
  sum = 0.f;
  for (int w = 0; w < SB / 64; ++w) sum += s_f[w];
  const float lse = mx + logf(sum);
  __syncthreads();
  if (tid == 0 && lse_out) lse_out[r] = lse;
  for (int round = 0; round < k; ++round) {
    ArgMax best = {-INFINITY, 0x7fffffff};
    for (int v = tid; v < V; v += SB) {
      if (bm && bm[v]) continue;
      bool taken = false;
      for (int q = 0; q < round; ++q) taken |= (s_sel[q] == v);
      if (!taken) best = amax(best, ArgMax{x[v], v});
    }
    best = wave_amax(best);
    if (lane == 0) { s_f[wave] = best.v; s_i[wave] = best.i; }
    __syncthreads();
    if (tid == 0) {
      ArgMax b = {s_f[0], s_i[0]};
      for (int w = 1; w < SB / 64; ++w) b = amax(b, ArgMax{s_f[w], s_i[w]});
      s_sel[round] = b.i;
      ids[(int64_t)r * k + round] = b.i == 0x7fffffff ? -1 : b.i;
      logprobs[(int64_t)r * k + round] = b.v - lse;
    }
    __syncthreads();
  }
}

int launch_topk_logprob(const float* logits, int64_t ldl, int R, int V, int k, const uint8_t* ban, int32_t* ids,
                        float* logprobs, float* lse, hipStream_t s) {
  if (k < 1 || k > 16) { set_error("topk: k must be in 1..16"); return WJ_E_INVALID; }
  hipLaunchKernelGGL(topk_logprob_kernel, dim3(R), dim3(SB), 0, s, logits, ldl, V, k, ban, ids, logprobs, lse);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

// --------------------------------------------------------------------------------------------
// Beam-search scoring: CTranslate2-style logits processors + Whisper timestamp rules + masked
// log-softmax + top-k, one workgroup per hypothesis row.  The host (whisperjav_amd/search.py) owns
// the beam bookkeeping and sends, per row: the timestamp-rule state, the tokens banned this step
// (no-repeat-ngram, -1 padded) and the distinct tokens subject to the repetition penalty.
// The logits buffer is modified in place (it is recomputed every step).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SB) void topk_rules_kernel(float* __restrict__ logits, int64_t ldl, int V, int k,
                                                        const wj_decode_opts o, const int32_t* __restrict__ row_rules,
                                                        const int32_t* __restrict__ ban, int maxb,
                                                        const int32_t* __restrict__ pen, int maxp, float penalty,
                                                        int32_t* __restrict__ ids, float* __restrict__ logprobs) {
  __shared__ float s_f[4][SB / 64];
  __shared__ int s_i[2][SB / 64];
  __shared__ int s_sel[16];
  __shared__ float s_stat[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* x = logits + (int64_t)r * ldl;
  if (penalty != 1.0f)
    for (int i = tid; i < maxp; i += SB) {
      const int t = pen[(int64_t)r * maxp + i];
      if (t >= 0 && t < V) { const float v = x[t]; x[t] = v < 0.f ? v * penalty : v / penalty; }
    }
  __syncthreads();
  for (int i = tid; i < maxb; i += SB) {
    const int t = ban[(int64_t)r * maxb + i];
    if (t >= 0 && t < V) x[t] = -INFINITY;
  }
  __syncthreads();
  RowRules rr;
  rr.first = row_rules[r * 4 + 0];
  rr.last_ts = row_rules[r * 4 + 1];
  rr.penult_ts = row_rules[r * 4 + 2];
  rr.ts_floor = row_rules[r * 4 + 3];
  rr.ts_rules = !o.without_timestamps;

  ArgMax best_all = {-INFINITY, 0x7fffffff};
  float max_text = -INFINITY;
  for (int v = tid; v < V; v += SB) {
    if (!token_allowed(v, rr, o)) continue;
    const float xv = x[v];
    best_all = amax(best_all, ArgMax{xv, v});
    if (v < o.timestamp_begin) max_text = fmaxf(max_text, xv);
  }
  best_all = wave_amax(best_all);
  max_text = wave_max(max_text);
  if (lane == 0) { s_f[0][wave] = best_all.v; s_i[0][wave] = best_all.i; s_f[2][wave] = max_text; }
  __syncthreads();
  best_all = ArgMax{s_f[0][0], s_i[0][0]};
  max_text = s_f[2][0];
  for (int w = 1; w < SB / 64; ++w) {
    best_all = amax(best_all, ArgMax{s_f[0][w], s_i[0][w]});
    max_text = fmaxf(max_text, s_f[2][w]);
  }
  __syncthreads();
  float sum_all = 0.f, sum_ts = 0.f;
  for (int v = tid; v < V; v += SB) {
    if (!token_allowed(v, rr, o)) continue;
    const float e = expf(x[v] - best_all.v);
    sum_all += e;
    if (v >= o.timestamp_begin) sum_ts += e;
  }
  sum_all = wave_sum(sum_all);
  sum_ts = wave_sum(sum_ts);
  if (lane == 0) { s_f[0][wave] = sum_all; s_f[1][wave] = sum_ts; }
  __syncthreads();
  if (tid == 0) {
    sum_all = 0.f; sum_ts = 0.f;
    for (int w = 0; w < SB / 64; ++w) { sum_all += s_f[0][w]; sum_ts += s_f[1][w]; }
    const float lse = best_all.v + logf(sum_all);
    float ts_only = 0.f, norm = lse;
    if (rr.ts_rules && sum_ts > 0.f) {
      const float ts_lp = best_all.v + logf(sum_ts) - lse;
      if (ts_lp > max_text - lse) { ts_only = 1.f; norm = best_all.v + logf(sum_ts); }
    }
    s_stat[0] = norm;
    s_stat[1] = ts_only;
  }
  __syncthreads();
  const float norm = s_stat[0];
  const bool ts_only = s_stat[1] != 0.f;
  for (int round = 0; round < k; ++round) {
    ArgMax best = {-INFINITY, 0x7fffffff};
    for (int v = tid; v < V; v += SB) {
      if (ts_only && v < o.timestamp_begin) continue;
      if (!token_allowed(v, rr, o)) continue;
      bool taken = false;
      for (int q = 0; q < round; ++q) taken |= (s_sel[q] == v);
      if (!taken) best = amax(best, ArgMax{x[v], v});
    }
    best = wave_amax(best);
    if (lane == 0) { s_f[0][wave] = best.v; s_i[0][wave] = best.i; }
    __syncthreads();
    if (tid == 0) {
      ArgMax b = {s_f[0][0], s_i[0][0]};
      for (int w = 1; w < SB / 64; ++w) b = amax(b, ArgMax{s_f[0][w], s_i[0][w]});
      s_sel[round] = b.i;
      const bool none = b.i == 0x7fffffff || b.v == -INFINITY;
      ids[(int64_t)r * k + round] = none ? -1 : b.i;
      logprobs[(int64_t)r * k + round] = none ? -INFINITY : b.v - norm;
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------------
// Device-resident beam search (CTranslate2 semantics, the host restatement is whisperjav_amd/search.py
// beam_search; reference entry: WhisperModel.generate(beam_size > 1) behind faster_whisper_pro_asr.py:819).
// Per step:  beam_topk  (one workgroup per hypothesis row: logits processors + Whisper rules computed from
// the row's device token history, masked log-softmax, top 2K)  ->  beam_merge  (one workgroup per window:
// best 2K of beam x vocab by (score desc, beam asc, token asc), EOT / last-step candidates move to the
// window's finished list and are refilled from the secondary candidates, the K survivors' histories are
// gathered from their parents).  Nothing goes to the host until the windows are done.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SB) void beam_topk_kernel(const BeamArgs a) {
  __shared__ float s_f[4][SB / 64];
  __shared__ int s_i[2][SB / 64];
  __shared__ int s_sel[16];
  __shared__ float s_stat[4];
  __shared__ RowRules s_rr;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const wj_decode_opts& o = a.opts;
  const int k = 2 * a.K, V = a.V;
  const int len = *a.pos_ptr + 1;
  const int32_t* tok = a.hist_in + (int64_t)r * a.tok_stride;
  float* x = a.logits + (int64_t)r * a.ldl;
  const float rep = o.repetition_penalty;
  const int ngram = o.no_repeat_ngram_size;
  if ((rep > 0.f && rep != 1.f) || ngram > 0) {      // same prepass as greedy_sample_kernel
    const int s0 = a.sample_begin - 1, L = len - s0;
    if (rep > 0.f && rep != 1.f) {
      for (int j = tid; j < L; j += SB) {
        const int t = tok[s0 + j];
        bool first = true;
        for (int i = 0; i < j; ++i)
          if (tok[s0 + i] == t) { first = false; break; }
        if (first) { const float v = x[t]; x[t] = v < 0.f ? v * rep : v / rep; }
      }
      __syncthreads();
    }
    if (ngram > 0 && L >= ngram) {
      for (int i = tid; i <= L - ngram; i += SB) {
        bool match = true;
        for (int q = 0; q < ngram - 1; ++q)
          if (tok[s0 + i + q] != tok[len - (ngram - 1) + q]) { match = false; break; }
        if (match) x[tok[s0 + i + ngram - 1]] = -INFINITY;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    RowRules rr;
    const int ns = len - a.sample_begin;
    rr.first = ns == 0;
    rr.ts_rules = !o.without_timestamps;
    rr.last_ts = ns >= 1 && tok[len - 1] >= o.timestamp_begin;
    rr.penult_ts = ns < 2 || tok[len - 2] >= o.timestamp_begin;
    int ts_last = -1;
    for (int i = a.sample_begin; i < len; ++i)
      if (tok[i] >= o.timestamp_begin) ts_last = tok[i];
    rr.ts_floor = -1;
    if (ts_last >= 0) rr.ts_floor = (rr.last_ts && !rr.penult_ts) ? ts_last : ts_last + 1;
    s_rr = rr;
  }
  __syncthreads();
  const RowRules rr = s_rr;

  ArgMax best_all = {-INFINITY, 0x7fffffff};
  float max_text = -INFINITY;
  for (int v = tid; v < V; v += SB) {
    if (!token_allowed(v, rr, o)) continue;
    const float xv = x[v];
    best_all = amax(best_all, ArgMax{xv, v});
    if (v < o.timestamp_begin) max_text = fmaxf(max_text, xv);
  }
  best_all = wave_amax(best_all);
  max_text = wave_max(max_text);
  if (lane == 0) { s_f[0][wave] = best_all.v; s_i[0][wave] = best_all.i; s_f[2][wave] = max_text; }
  __syncthreads();
  best_all = ArgMax{s_f[0][0], s_i[0][0]};
  max_text = s_f[2][0];
  for (int w = 1; w < SB / 64; ++w) {
    best_all = amax(best_all, ArgMax{s_f[0][w], s_i[0][w]});
    max_text = fmaxf(max_text, s_f[2][w]);
  }
  __syncthreads();
  float sum_all = 0.f, sum_ts = 0.f;
  for (int v = tid; v < V; v += SB) {
    if (!token_allowed(v, rr, o)) continue;
    const float e = expf(x[v] - best_all.v);
    sum_all += e;
    if (v >= o.timestamp_begin) sum_ts += e;
  }
  sum_all = wave_sum(sum_all);
  sum_ts = wave_sum(sum_ts);
  if (lane == 0) { s_f[0][wave] = sum_all; s_f[1][wave] = sum_ts; }
  __syncthreads();
  if (tid == 0) {
    sum_all = 0.f; sum_ts = 0.f;
    for (int w = 0; w < SB / 64; ++w) { sum_all += s_f[0][w]; sum_ts += s_f[1][w]; }
    const float lse = best_all.v + logf(sum_all);
    float ts_only = 0.f, norm = lse;
    if (rr.ts_rules && sum_ts > 0.f) {
      const float ts_lp = best_all.v + logf(sum_ts) - lse;
      if (ts_lp > max_text - lse) { ts_only = 1.f; norm = best_all.v + logf(sum_ts); }
    }
    s_stat[0] = norm;
    s_stat[1] = ts_only;
  }
  __syncthreads();
  const float norm = s_stat[0];
  const bool ts_only = s_stat[1] != 0.f;
  for (int round = 0; round < k; ++round) {
    ArgMax best = {-INFINITY, 0x7fffffff};
    for (int v = tid; v < V; v += SB) {
      if (ts_only && v < o.timestamp_begin) continue;
      if (!token_allowed(v, rr, o)) continue;
      bool taken = false;
      for (int q = 0; q < round; ++q) taken |= (s_sel[q] == v);
      if (!taken) best = amax(best, ArgMax{x[v], v});
    }
    best = wave_amax(best);
    if (lane == 0) { s_f[0][wave] = best.v; s_i[0][wave] = best.i; }
    __syncthreads();
    if (tid == 0) {
      ArgMax b = {s_f[0][0], s_i[0][0]};
      for (int w = 1; w < SB / 64; ++w) b = amax(b, ArgMax{s_f[0][w], s_i[0][w]});
      s_sel[round] = b.i;
      const bool none = b.i == 0x7fffffff || b.v == -INFINITY;
      a.cand_ids[(int64_t)r * 16 + round] = none ? -1 : b.i;
      a.cand_lp[(int64_t)r * 16 + round] = none ? -INFINITY : b.v - norm;
    }
    __syncthreads();
  }
}

// Register-resident form of beam_topk_kernel (the default): the row's filtered logits are read ONCE -- NV values per
// thread, all loads issued up front -- and the max / log-sum-exp / top-2K rounds run on registers.  The sweep above makes
// 2 + 2K dependent passes over the 51.9 k logits (rocprofv3, 120-min balanced step: 3.3 ms per launch at 1920 rows =
// 7.8 % of the step); this one is a single coalesced read.  Same summation order, same tie-breaking: identical results.
template <int NV>
__global__ __launch_bounds__(SB) void beam_topk_reg_kernel(const BeamArgs a) {
  __shared__ float s_f[4][SB / 64];
  __shared__ int s_i[2][SB / 64];
  __shared__ int s_win;
  __shared__ float s_stat[4];
  __shared__ RowRules s_rr;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const wj_decode_opts& o = a.opts;
  const int k = 2 * a.K, V = a.V;
  const int len = *a.pos_ptr + 1;
  const int32_t* tok = a.hist_in + (int64_t)r * a.tok_stride;
  float* x = a.logits + (int64_t)r * a.ldl;
  const float rep = o.repetition_penalty;
  const int ngram = o.no_repeat_ngram_size;
  if ((rep > 0.f && rep != 1.f) || ngram > 0) {      // same prepass as beam_topk_kernel / greedy_sample_kernel
    const int s0 = a.sample_begin - 1, L = len - s0;
    if (rep > 0.f && rep != 1.f) {
      for (int j = tid; j < L; j += SB) {
        const int t = tok[s0 + j];
        bool first = true;
        for (int i = 0; i < j; ++i)
          if (tok[s0 + i] == t) { first = false; break; }
        if (first) { const float v = x[t]; x[t] = v < 0.f ? v * rep : v / rep; }
      }
      __syncthreads();
    }
    if (ngram > 0 && L >= ngram) {
      for (int i = tid; i <= L - ngram; i += SB) {
        bool match = true;
        for (int q = 0; q < ngram - 1; ++q)
          if (tok[s0 + i + q] != tok[len - (ngram - 1) + q]) { match = false; break; }
        if (match) x[tok[s0 + i + ngram - 1]] = -INFINITY;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    RowRules rr;
    const int ns = len - a.sample_begin;
    rr.first = ns == 0;
    rr.ts_rules = !o.without_timestamps;
    rr.last_ts = ns >= 1 && tok[len - 1] >= o.timestamp_begin;
    rr.penult_ts = ns < 2 || tok[len - 2] >= o.timestamp_begin;
    int ts_last = -1;
    for (int i = a.sample_begin; i < len; ++i)
      if (tok[i] >= o.timestamp_begin) ts_last = tok[i];
    rr.ts_floor = -1;
    if (ts_last >= 0) rr.ts_floor = (rr.last_ts && !rr.penult_ts) ? ts_last : ts_last + 1;
    s_rr = rr;
  }
  __syncthreads();
  const RowRules rr = s_rr;

  // the one read of the row: value j of this thread is token j * SB + tid
  float val[NV];
  uint64_t live = 0;                      // bit j: token allowed by the rules (candidates come from these only)
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = j * SB + tid, vc = min(v, V - 1);
    val[j] = x[vc];
    if (v < V && token_allowed(vc, rr, o)) live |= 1ull << j;
  }
  ArgMax best_all = {-INFINITY, 0x7fffffff};
  float max_text = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if ((live >> j) & 1) {
      const int v = j * SB + tid;
      best_all = amax(best_all, ArgMax{val[j], v});
      if (v < o.timestamp_begin) max_text = fmaxf(max_text, val[j]);
    }
  }
  best_all = wave_amax(best_all);
  max_text = wave_max(max_text);
  if (lane == 0) { s_f[0][wave] = best_all.v; s_i[0][wave] = best_all.i; s_f[2][wave] = max_text; }
  __syncthreads();
  best_all = ArgMax{s_f[0][0], s_i[0][0]};
  max_text = s_f[2][0];
  for (int w = 1; w < SB / 64; ++w) {
    best_all = amax(best_all, ArgMax{s_f[0][w], s_i[0][w]});
    max_text = fmaxf(max_text, s_f[2][w]);
  }
  __syncthreads();
  float sum_all = 0.f, sum_ts = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if ((live >> j) & 1) {
      const float e = expf(val[j] - best_all.v);
      sum_all += e;
      if (j * SB + tid >= o.timestamp_begin) sum_ts += e;
    }
  }
  sum_all = wave_sum(sum_all);
  sum_ts = wave_sum(sum_ts);
  if (lane == 0) { s_f[0][wave] = sum_all; s_f[1][wave] = sum_ts; }
  __syncthreads();
  if (tid == 0) {
    sum_all = 0.f; sum_ts = 0.f;
    for (int w = 0; w < SB / 64; ++w) { sum_all += s_f[0][w]; sum_ts += s_f[1][w]; }
    const float lse = best_all.v + logf(sum_all);
    float ts_only = 0.f, norm = lse;
    if (rr.ts_rules && sum_ts > 0.f) {
      const float ts_lp = best_all.v + logf(sum_ts) - lse;
      if (ts_lp > max_text - lse) { ts_only = 1.f; norm = best_all.v + logf(sum_ts); }
    }
    s_stat[0] = norm;
    s_stat[1] = ts_only;
  }
  __syncthreads();
  const float norm = s_stat[0];
  if (s_stat[1] != 0.f) {                 // timestamps only: the text tokens leave the candidate set
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j * SB + tid < o.timestamp_begin) live &= ~(1ull << j);
  }
  for (int round = 0; round < k; ++round) {
    ArgMax best = {-INFINITY, 0x7fffffff};
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if ((live >> j) & 1) best = amax(best, ArgMax{val[j], j * SB + tid});
    best = wave_amax(best);
    if (lane == 0) { s_f[0][wave] = best.v; s_i[0][wave] = best.i; }
    __syncthreads();
    if (tid == 0) {
      ArgMax b = {s_f[0][0], s_i[0][0]};
      for (int w = 1; w < SB / 64; ++w) b = amax(b, ArgMax{s_f[0][w], s_i[0][w]});
      s_win = b.i;
      const bool none = b.i == 0x7fffffff || b.v == -INFINITY;
      a.cand_ids[(int64_t)r * 16 + round] = none ? -1 : b.i;
      a.cand_lp[(int64_t)r * 16 + round] = none ? -INFINITY : b.v - norm;
    }
    __syncthreads();
    const int win = s_win;
    if (win != 0x7fffffff && (win & (SB - 1)) == tid) live &= ~(1ull << (win / SB));     // taken
  }
}

__global__ __launch_bounds__(128) void beam_merge_kernel(const BeamArgs a) {
  __shared__ float c_score[128];
  __shared__ int c_beam[128], c_tok[128];
  __shared__ float s_score[16];          // sorted best 2K
  __shared__ int s_beam[16], s_tok[16];
  __shared__ float old_score[8];
  __shared__ int n_parent[8], n_feed[8];
  __shared__ float n_score[8];
  __shared__ int fin_b[8], fin_tok[8], fin_slot[8];
  __shared__ float fin_sc[8];
  __shared__ int s_counts[3];            // candidates, finished this step, sorted length
  const int w = blockIdx.x, tid = threadIdx.x;
  const int ow = a.win_ids ? a.win_ids[w] : w;      // index into the per-window result arrays (done, finished lists)
  const int K = a.K, nc = 2 * K, eot = a.opts.eot;
  const int pos = *a.pos_ptr, len = pos + 1;
  const bool last_step = (len - a.sample_begin) == a.max_new - 1;
  const bool done = a.done[ow] != 0;
  if (tid < K) old_score[tid] = a.score[w * K + tid];
  __syncthreads();
  // ---- candidates: beam-major, token order as the top-k kernel emitted them
  const int total = K * nc;
  for (int i = tid; i < 128; i += 128) {
    float sc = -INFINITY; int b = 0, t = -1;
    if (i < total) {
      b = i / nc;
      const int j = i - b * nc;
      t = a.cand_ids[(int64_t)(w * K + b) * 16 + j];
      sc = (old_score[b] == -INFINITY || t < 0) ? -INFINITY : old_score[b] + a.cand_lp[(int64_t)(w * K + b) * 16 + j];
      if (old_score[b] == -INFINITY) t = -1;
    }
    c_score[i] = sc; c_beam[i] = b; c_tok[i] = t;
  }
  __syncthreads();
  // ---- rank = number of valid candidates that precede (score desc, beam asc, token asc): a strict total order
  {
    const int i = tid;
    if (i < total && c_tok[i] >= 0) {
      int rank = 0;
      for (int j = 0; j < total; ++j) {
        if (j == i || c_tok[j] < 0) continue;
        const bool before = c_score[j] > c_score[i] ||
                            (c_score[j] == c_score[i] && (c_beam[j] < c_beam[i] || (c_beam[j] == c_beam[i] && c_tok[j] < c_tok[i])));
        rank += before;
      }
      if (rank < nc) { s_score[rank] = c_score[i]; s_beam[rank] = c_beam[i]; s_tok[rank] = c_tok[i]; }
    }
    if (tid == 0) {
      int n = 0;
      for (int j = 0; j < total; ++j) n += c_tok[j] >= 0;
      s_counts[0] = n;
      s_counts[2] = n < nc ? n : nc;
    }
  }
  __syncthreads();
  // ---- selection (sequential, as the reference loop)
  if (tid == 0) {
    const int n = s_counts[2];
    int secondary = K, nfin = 0, nn = 0;
    int fcount = a.fin_count[ow];
    if (!done) {
      for (int k = 0; k < (K < n ? K : n); ++k) {
        float sc = s_score[k]; int b = s_beam[k], t = s_tok[k];
        float usc = sc; int ub = b, ut = t;
        if (t == eot || last_step) {
          fin_b[nfin] = b; fin_tok[nfin] = (t == eot) ? -1 : t; fin_sc[nfin] = sc; fin_slot[nfin] = fcount + nfin;
          ++nfin;
          for (int j = secondary; j < n; ++j)
            if (s_tok[j] != eot) { usc = s_score[j]; ub = s_beam[j]; ut = s_tok[j]; secondary = j + 1; break; }
        }
        n_parent[nn] = ub; n_feed[nn] = ut; n_score[nn] = usc;
        ++nn;
      }
    }
    for (; nn < K; ++nn) {     // dead beams (or a finished window): keep the row, feed EOT
      n_parent[nn] = done ? nn : nn; n_feed[nn] = eot; n_score[nn] = done ? old_score[nn] : -INFINITY;
    }
    s_counts[1] = nfin;
    if (!done) {
      a.fin_count[ow] = fcount + nfin;
      if (last_step || fcount + nfin >= a.max_candidates) { a.done[ow] = 1; atomicAdd(a.n_done, 1); }
    }
  }
  __syncthreads();
  // ---- finished hypotheses: score + generated tokens (history of the parent beam [+ the token])
  const int nfin = s_counts[1];
  for (int f = 0; f < nfin; ++f) {
    const int slot = fin_slot[f];
    if (slot >= a.fin_cap) continue;                                   // cannot happen: cap = max_candidates + K
    const int32_t* src = a.hist_in + (int64_t)(w * K + fin_b[f]) * a.tok_stride + a.sample_begin;
    int32_t* dst = a.fin_tokens + ((int64_t)ow * a.fin_cap + slot) * a.tok_stride;
    const int ng = len - a.sample_begin;
    for (int j = tid; j < ng; j += 128) dst[j] = src[j];
    if (tid == 0) {
      int n = ng;
      if (fin_tok[f] >= 0) dst[n++] = fin_tok[f];
      a.fin_len[(int64_t)ow * a.fin_cap + slot] = n;
      a.fin_score[(int64_t)ow * a.fin_cap + slot] = fin_sc[f];
    }
    if (a.cum_in) {
      const float* csrc = a.cum_in + (int64_t)(w * K + fin_b[f]) * a.tok_stride + a.sample_begin;
      float* cdst = a.fin_cum + ((int64_t)ow * a.fin_cap + slot) * a.tok_stride;
      for (int j = tid; j < ng; j += 128) cdst[j] = csrc[j];
      if (tid == 0) { cdst[ng] = fin_sc[f]; if (fin_tok[f] >= 0) cdst[ng + 1] = fin_sc[f]; }
    }
  }
  // ---- survivors: history gather + new token, scores, parents
  for (int k = 0; k < K; ++k) {
    const int32_t* src = a.hist_in + (int64_t)(w * K + n_parent[k]) * a.tok_stride;
    int32_t* dst = a.hist_out + (int64_t)(w * K + k) * a.tok_stride;
    for (int j = tid; j < len; j += 128) dst[j] = src[j];
    if (a.cum_in) {
      const float* csrc = a.cum_in + (int64_t)(w * K + n_parent[k]) * a.tok_stride;
      float* cdst = a.cum_out + (int64_t)(w * K + k) * a.tok_stride;
      for (int j = tid; j < len; j += 128) cdst[j] = csrc[j];
      if (tid == 0) cdst[len] = n_score[k];
    }
    if (tid == 0) {
      dst[len] = n_feed[k];
      a.score[w * K + k] = n_score[k];
      a.parent[w * K + k] = w * K + n_parent[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// openai-whisper's BeamSearchDecoder.update (whisper/decoding.py; fidelity mode, reached from
// whisperjav/modules/whisper_pro_asr.py:433), one workgroup per window:
//   every beam proposes its top (K + 1) tokens; candidates are keyed by their token SEQUENCE (beams holding the same
//   history -- all of them at the first step, copies made when fewer than K candidates survive -- count once); in
//   order of decreasing cumulative log-prob a candidate ending in EOT joins this step's finished set, any other one
//   becomes a beam, until K beams exist; this step's finished sequences are then admitted best first while the window
//   holds fewer than round(K * patience) of them.  All K beams stay alive (no -inf beams, no refill from a second
//   candidate list: that is CTranslate2's rule, beam_merge_kernel above).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void beam_merge_ow_kernel(const BeamArgs a) {
  __shared__ float c_score[80];
  __shared__ int c_beam[80], c_tok[80], c_order[80];
  __shared__ float old_score[8];
  __shared__ int cls[8];
  __shared__ int n_parent[8], n_feed[8];
  __shared__ float n_score[8];
  __shared__ int fin_b[16], fin_slot[16];
  __shared__ float fin_sc[16];
  __shared__ int s_counts[2];            // candidates, finished this step
  const int w = blockIdx.x, tid = threadIdx.x;
  const int ow = a.win_ids ? a.win_ids[w] : w;
  const int K = a.K, nk = K + 1, eot = a.opts.eot;
  const int pos = *a.pos_ptr, len = pos + 1;
  const bool done = a.done[ow] != 0;
  if (tid < K) old_score[tid] = a.score[w * K + tid];
  // ---- beams that hold the same token sequence form one class (its first member stands for it)
  if (tid < K) {
    int c = tid;
    const int32_t* mine = a.hist_in + (int64_t)(w * K + tid) * a.tok_stride;
    for (int b = 0; b < tid; ++b) {
      const int32_t* other = a.hist_in + (int64_t)(w * K + b) * a.tok_stride;
      bool same = true;
      for (int j = len - 1; j >= a.sample_begin; --j)        // histories differ near their end, if at all
        if (mine[j] != other[j]) { same = false; break; }
      if (same) { c = b; break; }
    }
    cls[tid] = c;
  }
  __syncthreads();
  const int total = K * nk;
  if (tid < 80) {
    float sc = -INFINITY; int b = 0, t = -1;
    if (tid < total) {
      b = tid / nk;
      const int j = tid - b * nk;
      t = a.cand_ids[(int64_t)(w * K + b) * 16 + j];
      const float lp = a.cand_lp[(int64_t)(w * K + b) * 16 + j];
      if (cls[b] != b || t < 0 || lp == -INFINITY) t = -1;
      else sc = old_score[b] + lp;
    }
    c_score[tid] = sc; c_beam[tid] = b; c_tok[tid] = t;
  }
  __syncthreads();
  // ---- total order: score descending, then first insertion (beam, rank within the beam's top-k)
  if (tid < total && c_tok[tid] >= 0) {
    int rank = 0;
    for (int j = 0; j < total; ++j) {
      if (j == tid || c_tok[j] < 0) continue;
      rank += (c_score[j] > c_score[tid] || (c_score[j] == c_score[tid] && j < tid)) ? 1 : 0;
    }
    c_order[rank] = tid;
  }
  if (tid == 0) {
    int n = 0;
    for (int j = 0; j < total; ++j) n += c_tok[j] >= 0;
    s_counts[0] = n;
  }
  __syncthreads();
  if (tid == 0) {
    const int n = s_counts[0];
    int nn = 0, nfin = 0;
    int fcount = a.fin_count[ow];
    if (!done) {
      for (int i = 0; i < n && nn < K; ++i) {
        const int c = c_order[i];
        if (c_tok[c] == eot) {
          if (fcount + nfin < a.max_candidates && nfin < 16) {      // admitted best first while there is room
            fin_b[nfin] = c_beam[c]; fin_sc[nfin] = c_score[c]; fin_slot[nfin] = fcount + nfin;
            ++nfin;
          }
        } else {
          n_parent[nn] = c_beam[c]; n_feed[nn] = c_tok[c]; n_score[nn] = c_score[c];
          ++nn;
        }
      }
      if (nn == 0) { n_parent[0] = 0; n_feed[0] = eot; n_score[0] = -INFINITY; nn = 1; }   // everything masked: keep a row
      for (; nn < K; ++nn) { n_parent[nn] = n_parent[nn - 1]; n_feed[nn] = n_feed[nn - 1]; n_score[nn] = n_score[nn - 1]; }
      a.fin_count[ow] = fcount + nfin;
      if (fcount + nfin >= a.max_candidates) { a.done[ow] = 1; atomicAdd(a.n_done, 1); }
    } else {
      for (; nn < K; ++nn) { n_parent[nn] = nn; n_feed[nn] = eot; n_score[nn] = old_score[nn]; }
    }
    s_counts[1] = nfin;
  }
  __syncthreads();
  const int nfin = s_counts[1];
  for (int f = 0; f < nfin; ++f) {         // finished: the parent's generated tokens (EOT itself is not stored)
    const int slot = fin_slot[f];
    if (slot >= a.fin_cap) continue;
    const int32_t* src = a.hist_in + (int64_t)(w * K + fin_b[f]) * a.tok_stride + a.sample_begin;
    int32_t* dst = a.fin_tokens + ((int64_t)ow * a.fin_cap + slot) * a.tok_stride;
    const int ng = len - a.sample_begin;
    for (int j = tid; j < ng; j += 128) dst[j] = src[j];
    if (tid == 0) {
      a.fin_len[(int64_t)ow * a.fin_cap + slot] = ng;
      a.fin_score[(int64_t)ow * a.fin_cap + slot] = fin_sc[f];
    }
    if (a.cum_in) {
      const float* csrc = a.cum_in + (int64_t)(w * K + fin_b[f]) * a.tok_stride + a.sample_begin;
      float* cdst = a.fin_cum + ((int64_t)ow * a.fin_cap + slot) * a.tok_stride;
      for (int j = tid; j < ng; j += 128) cdst[j] = csrc[j];
      if (tid == 0) cdst[ng] = fin_sc[f];
    }
  }
  for (int k = 0; k < K; ++k) {
    const int32_t* src = a.hist_in + (int64_t)(w * K + n_parent[k]) * a.tok_stride;
    int32_t* dst = a.hist_out + (int64_t)(w * K + k) * a.tok_stride;
    for (int j = tid; j < len; j += 128) dst[j] = src[j];
    if (a.cum_in) {
      const float* csrc = a.cum_in + (int64_t)(w * K + n_parent[k]) * a.tok_stride;
      float* cdst = a.cum_out + (int64_t)(w * K + k) * a.tok_stride;
      for (int j = tid; j < len; j += 128) cdst[j] = csrc[j];
      if (tid == 0) cdst[len] = n_score[k];
    }
    if (tid == 0) {
      dst[len] = n_feed[k];
      a.score[w * K + k] = n_score[k];
      a.parent[w * K + k] = w * K + n_parent[k];
    }
  }
}

int g_beam_topk_reg = 1;   // wj_tune("beam_topk_reg"): 0 = the multi-pass sweep (A/B, cross-check)

int launch_beam_step(const BeamArgs& a, int R, int B, hipStream_t s) {
  if (a.K < 1 || a.K > 8) { set_error("beam search: beam size %d outside 1..8", a.K); return WJ_E_INVALID; }
  // register-resident top-2K (one read of the logits) for vocabularies up to 64 values per thread; the sweep otherwise
  if (g_beam_topk_reg && a.V <= 51 * SB) hipLaunchKernelGGL(beam_topk_reg_kernel<51>, dim3(R), dim3(SB), 0, s, a);
  else if (g_beam_topk_reg && a.V <= 64 * SB) hipLaunchKernelGGL(beam_topk_reg_kernel<64>, dim3(R), dim3(SB), 0, s, a);
  else hipLaunchKernelGGL(beam_topk_kernel, dim3(R), dim3(SB), 0, s, a);
  WJ_LAUNCH_CHECK();
  if (a.flavor == 1) hipLaunchKernelGGL(beam_merge_ow_kernel, dim3(B), dim3(128), 0, s, a);
  else hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(128), 0, s, a);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

int launch_topk_rules(float* logits, int64_t ldl, int R, int V, int k, const wj_decode_opts& o, const int32_t* row_rules,
                      const int32_t* ban, int maxb, const int32_t* pen, int maxp, float penalty, int32_t* ids,
                      float* logprobs, hipStream_t s) {
  if (k < 1 || k > 16) { set_error("topk_rules: k must be in 1..16"); return WJ_E_INVALID; }
  hipLaunchKernelGGL(topk_rules_kernel, dim3(R), dim3(SB), 0, s, logits, ldl, V, k, o, row_rules, ban, maxb, pen, maxp,
                     penalty, ids, logprobs);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

}  // namespace wj
