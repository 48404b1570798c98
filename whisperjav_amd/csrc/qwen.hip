// qwen.hip -- Qwen3 text decoder (the LLM of Qwen3-ASR, SURVEY.md 8f-3 / BASELINE cfg5): first correct path.
//
// Replaces (reference, un-vendored): the ``qwen_asr`` package's generate() behind
// whisperjav/modules/qwen_asr.py:638-757 (checkpoints named at :192-193).  Architecture as published and as restated in
// oracle/qwen3_ref.py (pinned against transformers.models.qwen3_asr): RMSNorm, fused QKV projection, per-head RMSNorm on q / k,
// rotary position embedding (rotate-half), grouped-query causal attention (head_dim 128), SwiGLU MLP, tied LM head.
// The audio embeddings enter as rows of the prompt's embedding matrix (wj_qwen_prefill takes EMBEDDINGS, not token ids), so
// the same entry points serve text-only and audio-conditioned prompts.
//
// Layout: sequences are PACKED -- row m belongs to sequence row_seq[m] at position row_pos[m] -- so ragged prompts cost no
// padding; K / V caches are [layer][sequence][kv_head][position][128] in the compute type; the residual stream is fp32.
// GEMMs are the library's (gemm.hip: MFMA tile / skinny kernels, exact fp32 kernel) through launch_gemm; the kernels here
// are the glue a decoder-only LLM adds: RMSNorm, q/k-norm + RoPE + cache append, GQA attention, SwiGLU, row gather, advance.
// What bounds them at scale (not yet measured: this is the parity slice): the decode step streams 3.4 GB of fp16 weights
// for the 1.7 B model, i.e. it is HBM-bound on the GEMMs, not on these kernels.
#include <algorithm>
#include <vector>

#include "kernels.hpp"

namespace wj {
int g_qwen_split_act = 5;   // wj_tune("qwen_split_act"), read at wj_qwen_create: which GEMM inputs of the float16 decoder travel as [hi | lo] pairs.
                            // 1 = LM head; 2 = + o_proj / down_proj inputs and the qkv / gate-up outputs (round 4's default: 1.06e-3 per token at
                            // the published geometry); 3 = + both RMSNorm outputs (every projection input); 4 / 5 = 2 + only the q/k/v input /
                            // only the gate/up input.  Round 5 ablation at the published geometry (profiles/r05_parity_diag_qwen_split_ablation.jsonl,
                            // decoder alone / end to end): 2: 1.06e-3 / 1.06e-3, 4: 1.60e-3 / 8.3e-4, 3: 6.5e-4 / 9.3e-4, 5: 3.8e-4 / 4.0e-4 -- the
                            // gate/up input is the rounding point that matters (K = hidden into 2 x ffn columns feeding a product of two
                            // activations); the q/k/v input is inside the noise.  5 is the default: inside the 1e-3 bar at 1/4 of mode 3's cost
int g_qwen_fuse_swiglu = 1;  // wj_tune("qwen_fuse_swiglu"): SwiGLU in the gate-up GEMM's epilogue (0 = the separate element-wise pass, for A/B)
int g_qwen_prompt_mfma = 1;  // wj_tune("qwen_prompt_mfma"): prompts of the 16-bit types take the MFMA tile attention (0 = the one-row-per-wave kernel)
int g_qwen_splitk = 4;       // wj_tune("qwen_splitk"): o_proj / down_proj of a 16-bit pass of 65 .. max_seqs rows are cut into this many K slices whose
                            // raw sums the following RMSNorm adds into the residual stream (1 = the projections add themselves)
int g_qwen_compact_pct = 15; // wj_tune("qwen_compact_pct"): re-pack the decode batch at a poll when this share of its rows has ended (0 = never)
}
using namespace wj;

namespace {

constexpr int HD = 128;   // head_dim of every published Qwen3 size

// One workgroup per row: 16-byte reads, the row stays in registers between the sum of squares and the scaling when it has
// at most 2048 columns (every published size), LDS carries the four wave sums.
// Fused split-K consumer (wj_tune "qwen_splitk"): when `slab` is given, the projection that precedes this norm (o_proj / down_proj) left
// `ks` raw fp32 K-slices [ks][M][D] instead of adding into the residual stream; the row is first completed, x += sum_s slab[s] in
// slice order (deterministic), written back, and then normalised.  out == NULL: only the completion (after the last layer).
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, const float* __restrict__ w, T* __restrict__ out,
                                                      int M, int D, float eps, int split = 0, const float* __restrict__ slab = nullptr,
                                                      int ks = 0) {
  __shared__ float part[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (int64_t)row * D;
  float4 keep[2];
  float ss = 0.f;
  int it = 0;
  for (int c = tid * 4; c < D; c += 1024, ++it) {
    float4 v = *reinterpret_cast<const float4*>(xr + c);
    if (slab) {
      for (int sI = 0; sI < ks; ++sI) {
        const float4 p = *reinterpret_cast<const float4*>(slab + ((int64_t)sI * M + row) * D + c);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
      }
      *reinterpret_cast<float4*>(const_cast<float*>(xr) + c) = v;      // read back below when the row has more than 2048 columns: same thread, same address
    }
    if (it < 2) keep[it] = v;
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (!out) return;
  ss = wave_sum(ss);
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  const float r = rsqrtf((part[0] + part[1] + part[2] + part[3]) / (float)D + eps);
  T* o = out + (int64_t)row * D * (split ? 2 : 1);        // split: [hi(D) | lo(D)] rows (GemmArgs::split)
  it = 0;
  for (int c = tid * 4; c < D; c += 1024, ++it) {
    const float4 v = it < 2 ? keep[it] : *reinterpret_cast<const float4*>(xr + c);
    const float4 g = *reinterpret_cast<const float4*>(w + c);
    const float y[4] = {g.x * (v.x * r), g.y * (v.y * r), g.z * (v.z * r), g.w * (v.w * r)};
    if constexpr (sizeof(T) == 2) {
      if (split) { st4_split<T>(o + c, D, y); continue; }
    }
    st4(o + c, y);
  }
}

// One wave per (row, head slot): slots 0..H-1 = query heads, H..H+KV-1 = key heads, H+KV.. = value heads of the fused
// projection.  q / k: RMSNorm over the 128 dims, then RoPE in the rotate-half convention (lane i owns dims i and i + 64,
// the two halves of one rotation pair); k and v go to the caches at (sequence, position).
// rope_tab [position][lane] = (cos, sin) of position * theta^(-2 lane / 128), built once per model by rope_table_kernel with
// the expressions this kernel used to evaluate per (row, head slot) -- same values, one 8-byte load instead of exp2f + sincosf.
// Four head slots per workgroup (one per wave): a quarter of the workgroups of the one-wave form.
__global__ __launch_bounds__(64) void rope_table_kernel(float2* __restrict__ tab, int n_pos, float log2_theta) {
  const int pos = blockIdx.x, lane = threadIdx.x;
  if (pos >= n_pos) return;
  const float inv_freq = exp2f(-(float)(2 * lane) / (float)HD * log2_theta);
  float sn, cs;
  sincosf((float)pos * inv_freq, &sn, &cs);
  tab[(int64_t)pos * 64 + lane] = float2{cs, sn};
}

template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const T* __restrict__ qkv, const float* __restrict__ q_w,
                                                           const float* __restrict__ k_w, const int32_t* __restrict__ row_seq,
                                                           const int32_t* __restrict__ row_pos, T* __restrict__ q_out,
                                                           T* __restrict__ kc, T* __restrict__ vc, int H, int KV, int ctx,
                                                           const float2* __restrict__ rope_tab, float eps, int split_in) {
  const int m = blockIdx.x, slot = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slot >= H + 2 * KV) return;
  const int W = (H + 2 * KV) * HD;
  const T* src = qkv + (int64_t)m * W * (split_in ? 2 : 1) + (int64_t)slot * HD;
  float a = Elem<T>::ld(src + lane), b = Elem<T>::ld(src + lane + 64);
  if (split_in) {      // the projection stored [hi(W) | lo(W)] rows (GemmArgs::split_out): the fp32 accumulator to ~22 bits
    a += Elem<T>::ld(src + W + lane);
    b += Elem<T>::ld(src + W + lane + 64);
  }
  const int b_seq = row_seq[m], pos = row_pos[m];
  if (slot >= H + KV) {      // value head: straight to the cache
    T* dst = vc + (((int64_t)b_seq * KV + (slot - H - KV)) * ctx + pos) * HD;
    Elem<T>::st(dst + lane, a); Elem<T>::st(dst + lane + 64, b);
    return;
  }
  const float* w = slot < H ? q_w : k_w;
  const float r = rsqrtf(wave_sum(a * a + b * b) / (float)HD + eps);
  a = w[lane] * (a * r); b = w[lane + 64] * (b * r);
  const float2 t2 = rope_tab[(int64_t)pos * 64 + lane];
  const float cs = t2.x, sn = t2.y;
  const float ra = a * cs - b * sn, rb = b * cs + a * sn;
  T* dst = slot < H ? q_out + ((int64_t)m * H + slot) * HD : kc + (((int64_t)b_seq * KV + (slot - H)) * ctx + pos) * HD;
  Elem<T>::st(dst + lane, ra); Elem<T>::st(dst + lane + 64, rb);
}

// Causal grouped-query attention, one wave per (row, KV head): the G = H / KV query heads that share the KV head are
// scored together, so every K and V row is read ONCE per group (the kernel is bound by those reads: 512 B per key per KV
// head).  A lane is (kq = lane >> 4, dq = lane & 15): one load instruction covers 4 keys x 256 B -- lane (kq, dq) holds
// dims 8 dq .. 8 dq + 7 of key base + kq -- i.e. 1 KiB contiguous; the 16 lanes of a key reduce the dot product with four
// row-local exchanges.  Keys are taken 64 at a time (16 such loads in flight), an online softmax carries (max, sum)
// across chunks; the weights stay in the registers that held the scores (all 16 lanes of key kq hold that key's weight),
// so the value pass needs no broadcast: acc[g][8] += p[g][key] * V[key][8 dq ..], summed over the four kq rows at the end.
template <typename T, int G>
__global__ __launch_bounds__(64) void gqa_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc,
                                                      const int32_t* __restrict__ row_seq, const int32_t* __restrict__ row_pos,
                                                      T* __restrict__ out, int H, int KV, int ctx, int split) {
  constexpr int STEPS = 16;
  const int m = blockIdx.x, kvh = blockIdx.y, lane = threadIdx.x, kq = lane >> 4, dq = lane & 15;
  const int b_seq = row_seq[m], n_keys = row_pos[m] + 1;
  const int64_t kvoff = ((int64_t)b_seq * KV + kvh) * (int64_t)ctx * HD + dq * 8;
  const T* Kb = kc + kvoff;
  const T* Vb = vc + kvoff;
  float qv[G][8], acc[G][8], run_max[G], run_sum[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    ld8(q + ((int64_t)m * H + kvh * G + g) * HD + dq * 8, qv[g]);
    run_max[g] = -INFINITY; run_sum[g] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
  }
  const float scale = rsqrtf((float)HD);
  for (int base = 0; base < n_keys; base += 4 * STEPS) {
    float s[G][STEPS];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int key = base + st * 4 + kq;
      if (base + st * 4 < n_keys) {                 // wave-uniform
        float kv[8];
        ld8(Kb + (int64_t)min(key, n_keys - 1) * HD, kv);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float d = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) d = fmaf(qv[g][e], kv[e], d);
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) d += __shfl_xor(d, o, 64);
          s[g][st] = key < n_keys ? d * scale : -INFINITY;
        }
      } else {
#pragma unroll
        for (int g = 0; g < G; ++g) s[g][st] = -INFINITY;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float cm = s[g][0];
#pragma unroll
      for (int st = 1; st < STEPS; ++st) cm = fmaxf(cm, s[g][st]);
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));       // key `base` is always valid, so the chunk maximum is finite
      const float new_max = fmaxf(run_max[g], cm);
      const float corr = expf(run_max[g] - new_max);  // exp(-inf) = 0 on the first chunk
      float ps = 0.f;
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        s[g][st] = expf(s[g][st] - new_max);          // masked keys: exp(-inf) = 0
        ps += s[g][st];
      }
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      run_sum[g] = run_sum[g] * corr + ps;
      run_max[g] = new_max;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[g][e] *= corr;
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      if (base + st * 4 < n_keys) {
        float vv[8];
        ld8(Vb + (int64_t)min(base + st * 4 + kq, n_keys - 1) * HD, vv);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(s[g][st], vv[e], acc[g][e]);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc[g][e] += __shfl_xor(acc[g][e], 16, 64);
      acc[g][e] += __shfl_xor(acc[g][e], 32, 64);
    }
    if (kq == 0) {
      const float inv = 1.f / run_sum[g];
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = acc[g][e] * inv;
      T* dst = out + ((int64_t)m * H * (split ? 2 : 1) + kvh * G + g) * HD + dq * 8;
      if constexpr (sizeof(T) == 2) {
        if (split) { st4_split<T>(dst, (int64_t)H * HD, o); st4_split<T>(dst + 4, (int64_t)H * HD, o + 4); continue; }
      }
      st4(dst, o); st4(dst + 4, o + 4);
    }
  }
}

// gate / up columns arrive INTERLEAVED in blocks of 16 (the weight rows are packed [16 gate | 16 up], qwen.engine_tensors): column
// c of the activation = silu(gu[32 (c >> 4) + (c & 15)]) * gu[... + 16].  The 16-bit types run this inside the gate-up GEMM's
// epilogue (EPI_SWIGLU_T) whenever an MFMA tile kernel serves the pass; this kernel remains for float32, MX-fp8 and the few-row
// decode batches on the skinny kernels.
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ gu, T* __restrict__ out, int M, int F, int split) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * F) return;
  const int64_t m = i / F, c = i - m * F;
  const int64_t cg = ((c >> 4) << 5) + (c & 15);
  float g, u;
  if (split) {         // gate / up arrive as [hi(2F) | lo(2F)] rows as well
    const T* r = gu + m * 4 * F;
    g = Elem<T>::ld(r + cg) + Elem<T>::ld(r + 2 * F + cg);
    u = Elem<T>::ld(r + cg + 16) + Elem<T>::ld(r + 2 * F + cg + 16);
  } else {
    g = Elem<T>::ld(gu + m * 2 * F + cg);
    u = Elem<T>::ld(gu + m * 2 * F + cg + 16);
  }
  const float y = g / (1.f + expf(-g)) * u;
  if constexpr (sizeof(T) == 2) {
    if (split) { st_split<T>(out + m * 2 * F + c, F, y); return; }     // [hi(F) | lo(F)] rows
  }
  Elem<T>::st(out + i, y);
}

template <typename T>
__global__ __launch_bounds__(256) void embed_rows_kernel(const T* __restrict__ emb, const int32_t* __restrict__ tokens,
                                                         float* __restrict__ out, int D) {
  const int m = blockIdx.x;
  const T* e = emb + (int64_t)tokens[m] * D;
  for (int c = threadIdx.x; c < D; c += 256) out[(int64_t)m * D + c] = Elem<T>::ld(e + c);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ rows,
                                                          float* __restrict__ out, int D) {
  const int m = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += 256) out[(int64_t)m * D + c] = x[(int64_t)rows[m] * D + c];
}

// After the logits of a decode step: record the arg-max token and its log-prob, stop a sequence at an EOS id or at ITS
// token budget lim[b] (the reference scales max_new_tokens with the clip's duration, qwen_asr.py:414-437), advance its
// position.  `next_tok` feeds the embedding lookup of the following step.  With a repetition penalty the accepted token
// joins the sequence's set of seen ids (kept duplicate-free, so the penalty kernel touches every logit once).
__global__ void advance_kernel(const int32_t* __restrict__ top_id, const float* __restrict__ top_lp, const int32_t* __restrict__ eos,
                               int n_eos, int32_t* __restrict__ finished, int32_t* __restrict__ n_out, int32_t* __restrict__ row_pos,
                               int32_t* __restrict__ next_tok, int32_t* __restrict__ tokens_out, float* __restrict__ lp_out,
                               int max_new, int n_rows, int ctx, int first, const int32_t* __restrict__ lim,
                               int32_t* __restrict__ seen, int32_t* __restrict__ seen_n, int seen_cap,
                               const int32_t* __restrict__ row_seq) {
  // Row m of the (possibly compacted) decode batch belongs to sequence b = row_seq[m]: positions, fed tokens and the head's
  // arg-max are per ROW; flags, counters, budgets, outputs and the penalty's id set are per SEQUENCE (round 4: finished
  // sequences leave the batch at the polls, the survivors' rows are re-packed)
  const int mrow = blockIdx.x * 64 + threadIdx.x;
  if (mrow >= n_rows) return;
  const int b = row_seq[mrow];
  if (!first) row_pos[mrow] = min(row_pos[mrow] + 1, ctx - 1);       // the token fed this step now occupies its position
  if (finished[b]) return;
  const int t = top_id[mrow];
  const int n = n_out[b];
  lp_out[(int64_t)b * (max_new + 1) + n] = top_lp[mrow];
  bool stop = false;
  for (int e = 0; e < n_eos; ++e) stop |= t == eos[e];
  if (stop || n >= lim[b]) { finished[b] = 1; return; }
  tokens_out[(int64_t)b * max_new + n] = t;
  n_out[b] = n + 1;
  next_tok[mrow] = t;
  if (seen) {
    int32_t* mine = seen + (int64_t)b * seen_cap;
    const int cnt = seen_n[b];
    bool known = false;
    for (int i = 0; i < cnt; ++i) known |= mine[i] == t;
    if (!known && cnt < seen_cap) { mine[cnt] = t; seen_n[b] = cnt + 1; }
  }
}

// batch compaction: row r of the shrunk batch continues old row src[r]
__global__ void compact_decode_rows_kernel(const int32_t* __restrict__ src, int n, const int32_t* __restrict__ seq_in,
                                           const int32_t* __restrict__ pos_in, const int32_t* __restrict__ tok_in,
                                           int32_t* __restrict__ seq_out, int32_t* __restrict__ pos_out, int32_t* __restrict__ tok_out) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= n) return;
  const int o = src[r];
  seq_out[r] = seq_in[o]; pos_out[r] = pos_in[o]; tok_out[r] = tok_in[o];
}

// transformers' RepetitionPenaltyLogitsProcessor on the rows of a decode step: every id of the sequence so far (prompt
// and generated; here a duplicate-free list) has its logit divided by the penalty when positive, multiplied when negative.
__global__ __launch_bounds__(256) void rep_penalty_kernel(float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ seen,
                                                          const int32_t* __restrict__ seen_n, int seen_cap, float penalty,
                                                          const int32_t* __restrict__ finished, const int32_t* __restrict__ row_seq) {
  const int b = row_seq[blockIdx.x];        // logits row = batch row, id set = the row's sequence
  if (finished[b]) return;
  const int n = seen_n[b];
  float* row = logits + (int64_t)blockIdx.x * ldl;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int t = seen[(int64_t)b * seen_cap + i];
    const float l = row[t];
    row[t] = l < 0.f ? l * penalty : l / penalty;
  }
}

// --------------------------------------------------------------------------------------------
// Prompt attention on the matrix cores (round 4; wj_tune "qwen_prompt_mfma", 16-bit types).  A prefill or classification pass
// presents every sequence from position 0, so the causal product of a sequence is a lower-triangular tile problem: one
// workgroup = 128 consecutive query positions of one (sequence, query head), wave w owns 32 of them, the key/value rows of the
// sequence's KV head stream through LDS 64 at a time, double-buffered.  The arithmetic is the encoder attention's
// (attention.hip: S^T = K.Q^T with the key rows permuted so that the probabilities leave the MFMA already in B-operand order,
// lean online softmax with a slack-bounded running maximum, row sums from a ones block, O^T += V^T.P^T), widened to 128 head
// dims: 32 + 32 + 4 MFMAs per wave and key tile.  Two things differ:
//   * V lives in the cache as [key][128]; the PV product wants V^T fragments (8 consecutive KEYS of one dim per lane).  The tile
//     is transposed on its way into LDS: a lane loads 16 B = 8 dims of ITS key (lane = key), and stores them as 8 two-byte LDS
//     writes into rows d, column key -- the 64 lanes of a store hit one 128-byte LDS row, conflict-free.
//   * causality: a key tile is FULL for a wave when its last key <= the wave's first query, MASKED (key > query -> -inf) when it
//     straddles the diagonal, and skipped (loads and barriers only) when it lies beyond the wave's last query.
// Cache rows past the sequence end are never trusted: K rows are clamped to the last real key (their scores are masked by the
// select), V rows are zeroed while staging (0 * p, not NaN * 0).  Work items (sequence slot, first query, first row, length) are
// built on the host by the pass that owns the rows.
// --------------------------------------------------------------------------------------------
template <typename T> struct QLane;
template <> struct QLane<bf16_t> { typedef __bf16 type; };
template <> struct QLane<f16_t> { typedef _Float16 type; };

__device__ __forceinline__ int pk_perm(int k) { return 16 * (2 * (k >> 5) + ((k >> 2) & 1)) + 4 * ((k >> 3) & 3) + (k & 3); }
__device__ __forceinline__ int pk_swz(int prow, int chunk) { return prow * 128 + ((chunk ^ (prow & 15)) << 3); }     // K tile [64][128]
__device__ __forceinline__ int pv_swz(int d, int chunk) { return d * 64 + ((chunk ^ (d & 7)) << 3); }               // V^T tile [128][64]

__device__ __forceinline__ float pa_quad_max(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const uint32_t v = __float_as_uint(x);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <typename T>
__global__ __launch_bounds__(256, 2) void prompt_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc,
                                                          const int4* __restrict__ work, T* __restrict__ out, int H, int KV, int ctx,
                                                          int split) {
  typedef typename Vec8<T>::type vec8_t;
  typedef typename QLane<T>::type lane_t;
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 2 * 8192];   // [buf][K | V^T][8192] = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int4 wk = work[blockIdx.x];
  const int b_seq = wk.x, qb0 = wk.y, row0 = wk.z, n = wk.w;
  const int h = blockIdx.y, kvh = h / (H / KV);
  const int q0 = qb0 + wave * 32;                       // this wave's first query position
  const int64_t kvoff = ((int64_t)b_seq * KV + kvh) * (int64_t)ctx * HD;
  const T* Kp = kc + kvoff;
  const T* Vp = vc + kvoff;

  vec8_t qf[2][4];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int qpos = min(q0 + f * 16 + li, n - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[f][ks] = *reinterpret_cast<const vec8_t*>(q + ((int64_t)(row0 + qpos) * H + h) * HD + ks * 32 + lg * 8);
  }
  f32x4_t o[2][8];
  f32x4_t lsum[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  vec8_t ones8;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones8[e] = (lane_t)1.0f;
  float m_run[2] = {-INFINITY, -INFINITY};
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int d = 0; d < 8; ++d) o[f][d] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = (min(qb0 + 128, n) + 63) >> 6;         // key tiles this workgroup walks
  uint4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
#define WJ_PLOAD1(i, kt)                                                                               \
  {                                                                                                    \
    const int idx = tid + (i) * 256;                                                                   \
    const int krow = min((kt) * 64 + (idx >> 4), n - 1);                                               \
    rk##i = *reinterpret_cast<const uint4*>(Kp + (int64_t)krow * HD + (idx & 15) * 8);                 \
    const int vkey = (kt) * 64 + lane;                                                                 \
    rv##i = *reinterpret_cast<const uint4*>(Vp + (int64_t)min(vkey, n - 1) * HD + ((i) * 4 + wave) * 8); \
    if (vkey >= n) rv##i = uint4{0u, 0u, 0u, 0u};                                                      \
  }
#define WJ_PLOAD(kt) WJ_PLOAD1(0, kt) WJ_PLOAD1(1, kt) WJ_PLOAD1(2, kt) WJ_PLOAD1(3, kt)
#define WJ_PSTORE1(i, buf)                                                                             \
  {                                                                                                    \
    const int idx = tid + (i) * 256;                                                                   \
    *reinterpret_cast<uint4*>(&lds[((buf) * 2 + 0) * 8192 + pk_swz(pk_perm(idx >> 4), idx & 15)]) = rk##i; \
    uint16_t* lvw = &lds[((buf) * 2 + 1) * 8192];                                                      \
    const int d0 = ((i) * 4 + wave) * 8, kc8 = lane >> 3, ko = lane & 7;                               \
    lvw[pv_swz(d0 + 0, kc8) + ko] = (uint16_t)(rv##i.x & 0xffffu);                                     \
    lvw[pv_swz(d0 + 1, kc8) + ko] = (uint16_t)(rv##i.x >> 16);                                         \
    lvw[pv_swz(d0 + 2, kc8) + ko] = (uint16_t)(rv##i.y & 0xffffu);                                     \
    lvw[pv_swz(d0 + 3, kc8) + ko] = (uint16_t)(rv##i.y >> 16);                                         \
    lvw[pv_swz(d0 + 4, kc8) + ko] = (uint16_t)(rv##i.z & 0xffffu);                                     \
    lvw[pv_swz(d0 + 5, kc8) + ko] = (uint16_t)(rv##i.z >> 16);                                         \
    lvw[pv_swz(d0 + 6, kc8) + ko] = (uint16_t)(rv##i.w & 0xffffu);                                     \
    lvw[pv_swz(d0 + 7, kc8) + ko] = (uint16_t)(rv##i.w >> 16);                                         \
  }
#define WJ_PSTORE(buf) WJ_PSTORE1(0, buf) WJ_PSTORE1(1, buf) WJ_PSTORE1(2, buf) WJ_PSTORE1(3, buf)

  WJ_PLOAD(0)
  WJ_PSTORE(0)
  __syncthreads();

  constexpr float c2 = 0.08838834764831845f * 1.44269504088896340736f;    // 1 / sqrt(128) * log2 e
  constexpr float kSlack = 8.0f;
  // MODE 0: every key of the tile is visible to every query of the wave; 1: the tile straddles the diagonal; 2: beyond it
  auto key_tile = [&](const int kt, auto mode_tag) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    const int cur = kt & 1;
    if (kt + 1 < nt) { WJ_PLOAD(kt + 1) }
    if constexpr (MODE < 2) {
      const uint16_t* lk = &lds[(cur * 2 + 0) * 8192];
      const uint16_t* lv = &lds[(cur * 2 + 1) * 8192];
      f32x4_t st[2][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int krow = 16 * kb + li;
        st[0][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        st[1][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const vec8_t kf = *reinterpret_cast<const vec8_t*>(&lk[pk_swz(krow, ks * 4 + lg)]);
          st[0][kb] = mfma16(kf, qf[0][ks], st[0][kb]);
          st[1][kb] = mfma16(kf, qf[1][ks], st[1][kb]);
        }
      }
      // lane (query li of fragment f, lg) holds, for block kb, keys kt*64 + 32*(kb>>1) + 8*lg + 4*(kb&1) + r
      if constexpr (MODE == 1) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int qpos = q0 + f * 16 + li;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 64 + 32 * (kb >> 1) + 8 * lg + 4 * (kb & 1) + r;
              if (key > qpos) st[f][kb][r] = -INFINITY;
            }
        }
      }
      vec8_t pf[2][2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[f][kb][r]);
        mx = pa_quad_max(mx);
        const float mxs = mx * c2;
        if (mxs > m_run[f] + kSlack || m_run[f] == -INFINITY) {
          const float m_new = fmaxf(m_run[f], mxs);
          const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);     // first tile: exp2(-inf) = 0 on zeros
          m_run[f] = m_new;
#pragma unroll
          for (int d = 0; d < 8; ++d)
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
      for (int s2 = 0; s2 < 2; ++s2) {
        lsum[0] = mfma16(ones8, pf[0][s2], lsum[0]);
        lsum[1] = mfma16(ones8, pf[1][s2], lsum[1]);
      }
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const vec8_t vf = *reinterpret_cast<const vec8_t*>(&lv[pv_swz(d * 16 + li, s2 * 4 + lg)]);
          o[0][d] = mfma16(vf, pf[0][s2], o[0][d]);
          o[1][d] = mfma16(vf, pf[1][s2], o[1][d]);
        }
    }
    if (kt + 1 < nt) { WJ_PSTORE(cur ^ 1) }
    __syncthreads();
  };
  for (int kt = 0; kt < nt; ++kt) {
    if (kt * 64 > q0 + 31) key_tile(kt, std::integral_constant<int, 2>{});
    else if (kt * 64 + 63 <= q0) key_tile(kt, std::integral_constant<int, 0>{});
    else key_tile(kt, std::integral_constant<int, 1>{});
  }
#undef WJ_PLOAD
#undef WJ_PLOAD1
#undef WJ_PSTORE
#undef WJ_PSTORE1

  const int64_t ldo = (int64_t)H * HD * (split ? 2 : 1);
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int qpos = q0 + f * 16 + li;
    if (qpos >= n) continue;
    const float inv = 1.0f / lsum[f][0];
    T* dst = out + (int64_t)(row0 + qpos) * ldo + h * HD + lg * 4;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      float v[4] = {o[f][d][0] * inv, o[f][d][1] * inv, o[f][d][2] * inv, o[f][d][3] * inv};
      if (split) st4_split<T>(dst + d * 16, (int64_t)H * HD, v);
      else st4(dst + d * 16, v);
    }
  }
}

}  // namespace

struct wj_qwen {
  wj_ctx* ctx = nullptr;
  wj_qwen_dims d{};
  int dtype = WJ_F16;
  size_t esz = 2;
  const char* blob = nullptr;
  std::vector<int64_t> off;
  int max_seqs = 0, max_ctx = 0, max_rows = 0;
  std::vector<void*> allocs;
  float* x = nullptr;        // f32 [rows][D] residual stream
  void* h = nullptr;         // T   [rows][D]
  void* qkv = nullptr;       // T   [rows][(H + 2 KV) * 128]
  void* q = nullptr;         // T   [rows][H][128]
  void* attn = nullptr;      // T   [rows][H * 128]
  void* gu = nullptr;        // T   [rows][2 F]
  void* act = nullptr;       // T   [rows][F]
  void* kc = nullptr;        // T   [L][seqs][KV][ctx][128]
  void* vc = nullptr;
  float* xl = nullptr;       // f32 [seqs][D] rows that produce logits
  float* logits = nullptr;   // f32 [seqs][ldl]
  int64_t ldl = 0;
  int vocab_pad = 0;         // vocabulary rounded up to 256 when the blob carries the zero rows (qwen.py pads), else the vocabulary
  int32_t *row_seq = nullptr, *row_pos = nullptr, *last_rows = nullptr, *next_tok = nullptr, *finished = nullptr, *n_out = nullptr;
  int32_t *top_id = nullptr, *tokens_out = nullptr, *eos = nullptr;
  float *top_lp = nullptr, *top_lse = nullptr, *lp_out = nullptr;
  int32_t *lim = nullptr, *seen = nullptr, *seen_n = nullptr;   // per-sequence token budget; ids the repetition penalty applies to
  int seen_cap = 0;
  float cur_penalty = 1.f;   // != 1 only while wj_qwen_generate_greedy_ex runs its iterations (run_head applies it)
  int n_seqs = 0;            // sequences of the last prefill
  std::vector<int32_t> prompt_len;   // their prompt lengths (the token budgets are clamped to the room left in the KV cache)
  int32_t* embed_ids = nullptr;      // staging of wj_qwen_embed (its own buffer: row_seq holds decode state between prefill and generate)
  // float16: the GEMMs that write the residual stream (o_proj, down_proj) and the LM head read their activations as
  // [hi | lo] fp16 pairs (x to ~22 bits, W.hi + W.lo in one fp32 accumulator) -- the measure that brought the Whisper
  // decoder inside 1e-3.  wj_tune "qwen_split_act": 0 = off, 1 = decode iterations and the head, 2 (default) = also prompts,
  // 3 = in addition the RMSNorm outputs that feed the q/k/v and gate/up projections (every GEMM of the decoder then sees ~22-bit activations)
  int split_mode = 0;
  // WJ_F8W: the layers' projection matrices as MX-fp8 (e4m3 bytes + E8M0 scales per 32 elements), activations quantised per GEMM
  bool mx8 = false;
  uint8_t* w8 = nullptr;          // all layers: [qkv | o | gate_up | down] bytes
  uint8_t* w8s = nullptr;         // ... their scales
  std::vector<int64_t> w8_off, w8s_off;     // per (layer, matrix 0..3)
  uint8_t* a8 = nullptr;          // activation scratch [max_rows][max K]
  uint8_t* a8s = nullptr;
  int last_used_graph = 0;   // the last generation replayed its iteration from a hipGraph
  int32_t *cmp_src = nullptr, *cmp_seq = nullptr, *cmp_pos = nullptr, *cmp_tok = nullptr;   // batch compaction scratch [max_seqs]
  float2* rope_tab = nullptr; // [max_ctx + 1][64] (cos, sin) of the rotary embedding, see qk_norm_rope_kernel
  float* slab = nullptr;      // f32 [g_qwen_splitk at create][max_seqs][hidden]: split-K slices of o_proj / down_proj (decode batches)
  int slab_ks = 0;
  int4* pwork = nullptr;      // prompt attention work items (sequence slot, first query, first row, length), see prompt_attn_kernel
  int pwork_cap = 0, pwork_n = 0;   // pwork_n > 0 only while a prompt pass (prefill / classify) runs its layers
  std::vector<int32_t> pwork_host;
  int last_compactions = 0;  // times the last generation re-packed its batch
  int64_t last_row_steps = 0;  // sum over its iterations of the live rows (the work actually done)
  int last_steps = 0;        // decode iterations the last generation ran (it leaves the loop when every sequence has ended)
  int last_truncated = 0;    // sequences whose token budget was cut to the room left in the KV cache
  const void* W(int i) const { return blob + off[i]; }
  const float* F(int i) const { return reinterpret_cast<const float*>(blob + off[i]); }
  int layer_base(int l) const { return WJ_Q_N_GLOBAL + l * WJ_QL_N; }
  void* at(void* base, int64_t elems) const { return reinterpret_cast<char*>(base) + elems * (int64_t)esz; }
};

namespace {

#define WJ_TRYQ(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#define TP(T, p) reinterpret_cast<T*>(p)

// Which kernel of gemm.hip carries a decoder GEMM of M rows (scripts/qwen_gemm_sweep.py on the MI355X, 1.7 B geometry,
// profiles/r03_qwen_gemm_sweep.jsonl).  The dispatcher's own choice (0) is the skinny kernel up to 512 rows and the 256-wide
// ping-pong kernel from 1024: right for a few dozen rows and for the wide gate/up projection of a prompt, but between the two
// the 128-tile kernels win by 2-6x (skinny re-reads the activations per 16 columns), and from 1024 rows the 2048-column
// projections (o, down: 32-64 tiles of 256 x 256) leave most of the 256 CUs idle under the 256-tile kernel.
enum QGemm { QG_QKV, QG_O, QG_GATEUP, QG_DOWN };
int gemm_variant(QGemm which, int M, int K, int dt) {
  if (dt == WJ_F32 || K % 64) return 0;             // the exact fp32 kernel has one form; the LDS-DMA tiles take whole 64-column steps
  if (M <= 64) return which == QG_GATEUP && M > 32 ? 73 : 0;
  if (M < 1024) return which == QG_DOWN ? 74 : 3;
  if (M >= 8192) return 0;                          // prompts: every projection fills the chip with 256-wide tiles
  switch (which) {
    case QG_QKV: return M >= 2048 ? 3 : 74;
    case QG_O: return 73;
    case QG_GATEUP: return 0;
    default: return 74;
  }
}

// rows [0, M) through every decoder layer; row_seq / row_pos describe them.  split: the attention output and the SwiGLU
// output are stored as [hi | lo] rows and o_proj / down_proj consume them as split activations (float16 only)
int run_layers(wj_qwen* m, int M, hipStream_t s, bool split) {
  if (m->mx8) split = false;         // MX-fp8 projections: the activations are quantised per GEMM, nothing to split
  const wj_qwen_dims& d = m->d;
  // WJ_F8W: quantise the GEMM's activation rows (float16 [M][K], row stride ld) into the scratch and point the GEMM at them
  auto mx = [&](GemmArgs& g, int l, int which) -> int {
    if (!m->mx8) return WJ_OK;
    WJ_TRYQ(launch_mx8_quantize(WJ_F16, g.A, g.lda, g.M, g.K, m->a8, m->a8s, s));
    g.A = m->a8; g.lda = g.K; g.a_scale = m->a8s;
    g.W = m->w8 + m->w8_off[l * 4 + which]; g.ldw = g.K; g.w_scale = m->w8s + m->w8s_off[l * 4 + which];
    g.mx8 = 1; g.split = 0;
    return WJ_OK;
  };
  const int D = d.hidden, H = d.n_head, KV = d.n_kv_head, F = d.ffn, dt = m->dtype;
  const int W = (H + 2 * KV) * HD;
  const int64_t layer_kv = (int64_t)m->max_seqs * KV * m->max_ctx * HD;
  const bool use_sk = m->slab_ks > 1 && g_qwen_splitk > 1 && dt != WJ_F32 && !m->mx8 && M > 64 && M <= m->max_seqs;
  int pending = 0;       // K slices of a projection waiting in m->slab for the next norm
  for (int l = 0; l < d.n_layer; ++l) {
    const int b0 = m->layer_base(l);
    void* kc = m->at(m->kc, l * layer_kv);
    void* vc = m->at(m->vc, l * layer_kv);
    // the projections' INPUT (h) as [hi | lo] too: mode 3 = both RMSNorm outputs, 4 = the one feeding q / k / v only, 5 = the one feeding
    // gate / up only (per-projection ablations of mode 3, measured in profiles/r05_parity_diag_qwen_split_ablation.jsonl)
    const int sp_qkv = (split && (m->split_mode == 3 || m->split_mode == 4)) ? 1 : 0;
    const int sp_gu = (split && (m->split_mode == 3 || m->split_mode == 5)) ? 1 : 0;
    int sp_in = sp_qkv;
    auto rms = [&](const float* w, void* out) -> int {      // completes the residual stream from pending split-K slices first
      const float* sl = pending ? m->slab : nullptr;
      if (dt == WJ_F32) hipLaunchKernelGGL((rmsnorm_kernel<float>), dim3(M), dim3(256), 0, s, m->x, w, TP(float, out), M, D, d.rms_eps, 0, sl, pending);
      else if (dt == WJ_F16) hipLaunchKernelGGL((rmsnorm_kernel<f16_t>), dim3(M), dim3(256), 0, s, m->x, w, TP(f16_t, out), M, D, d.rms_eps, sp_in, sl, pending);
      else hipLaunchKernelGGL((rmsnorm_kernel<bf16_t>), dim3(M), dim3(256), 0, s, m->x, w, TP(bf16_t, out), M, D, d.rms_eps, sp_in, sl, pending);
      WJ_LAUNCH_CHECK();
      pending = 0;
      return WJ_OK;
    };
    // o_proj / down_proj: into the residual stream, or as K slices for the next norm to add (decode batches of the 16-bit types)
    auto resid_gemm = [&](GemmArgs& g, int variant) -> int {
      int ks = 1;
      if (use_sk) {
        const int keff = g.split ? 2 * g.K : g.K;
        for (ks = m->slab_ks; ks > 1 && (keff % (64 * ks)); ks >>= 1) {}
      }
      if (ks <= 1) return launch_gemm(dt, EPI_RESID_F32, g, s, variant);
      g.out = m->slab; g.ldc = D; g.ksplit = ks;
      WJ_TRYQ(launch_gemm(dt, EPI_PARTIAL_F32, g, s, variant));
      pending = ks;
      return WJ_OK;
    };
    WJ_TRYQ(rms(m->F(b0 + WJ_QL_LN1_W), m->h));
    {
      GemmArgs g;
      g.A = m->h; g.lda = D; g.W = m->W(b0 + WJ_QL_QKV_W); g.ldw = D; g.M = M; g.N = W; g.K = D; g.out = m->qkv; g.ldc = W;
      if (split) { g.split_out = 1; g.ldc = 2 * W; }
      if (sp_in) { g.split = 1; g.lda = 2 * D; }
      WJ_TRYQ(mx(g, l, 0));
      WJ_TRYQ(launch_gemm(dt, EPI_T, g, s, gemm_variant(QG_QKV, M, D, dt)));
    }
    const dim3 rgrid(M, (H + 2 * KV + 3) / 4);
    if (dt == WJ_F32)
      hipLaunchKernelGGL((qk_norm_rope_kernel<float>), rgrid, dim3(256), 0, s, TP(const float, m->qkv), m->F(b0 + WJ_QL_QNORM_W),
                         m->F(b0 + WJ_QL_KNORM_W), m->row_seq, m->row_pos, TP(float, m->q), TP(float, kc), TP(float, vc), H, KV, m->max_ctx, m->rope_tab, d.rms_eps, split ? 1 : 0);
    else if (dt == WJ_F16)
      hipLaunchKernelGGL((qk_norm_rope_kernel<f16_t>), rgrid, dim3(256), 0, s, TP(const f16_t, m->qkv), m->F(b0 + WJ_QL_QNORM_W),
                         m->F(b0 + WJ_QL_KNORM_W), m->row_seq, m->row_pos, TP(f16_t, m->q), TP(f16_t, kc), TP(f16_t, vc), H, KV, m->max_ctx, m->rope_tab, d.rms_eps, split ? 1 : 0);
    else
      hipLaunchKernelGGL((qk_norm_rope_kernel<bf16_t>), rgrid, dim3(256), 0, s, TP(const bf16_t, m->qkv), m->F(b0 + WJ_QL_QNORM_W),
                         m->F(b0 + WJ_QL_KNORM_W), m->row_seq, m->row_pos, TP(bf16_t, m->q), TP(bf16_t, kc), TP(bf16_t, vc), H, KV, m->max_ctx, m->rope_tab, d.rms_eps, split ? 1 : 0);
    WJ_LAUNCH_CHECK();
#define WJ_GQA(T_, G_) hipLaunchKernelGGL((gqa_attn_kernel<T_, G_>), dim3(M, KV), dim3(64), 0, s, TP(const T_, m->q), TP(const T_, kc), \
                                          TP(const T_, vc), m->row_seq, m->row_pos, TP(T_, m->attn), H, KV, m->max_ctx, split ? 1 : 0)
#define WJ_GQA_T(G_) do { if (dt == WJ_F32) WJ_GQA(float, G_); else if (dt == WJ_F16) WJ_GQA(f16_t, G_); else WJ_GQA(bf16_t, G_); } while (0)
    // Decode rows (and fp32 prompts) use the one-row-per-wave kernel.  Round 3 tried giving a wave 4 consecutive prompt rows
    // (8 (row, head) pairs sharing every K / V load): SLOWER (prefill 443 -> 481 ms), the kernel is bound by its per-pair
    // arithmetic and exchanges.  Round 4: 16-bit prompts go to prompt_attn_kernel, the MFMA tile form.
    if (m->pwork_n > 0 && dt != WJ_F32) {      // a prompt pass: lower-triangular tiles on the matrix cores
      if (dt == WJ_F16)
        hipLaunchKernelGGL((prompt_attn_kernel<f16_t>), dim3(m->pwork_n, H), dim3(256), 0, s, TP(const f16_t, m->q), TP(const f16_t, kc),
                           TP(const f16_t, vc), m->pwork, TP(f16_t, m->attn), H, KV, m->max_ctx, split ? 1 : 0);
      else
        hipLaunchKernelGGL((prompt_attn_kernel<bf16_t>), dim3(m->pwork_n, H), dim3(256), 0, s, TP(const bf16_t, m->q), TP(const bf16_t, kc),
                           TP(const bf16_t, vc), m->pwork, TP(bf16_t, m->attn), H, KV, m->max_ctx, split ? 1 : 0);
    } else
    switch (H / KV) {      // query heads per KV head (wj_qwen_create admits 1, 2, 4)
      case 1: WJ_GQA_T(1); break;
      case 2: WJ_GQA_T(2); break;
      default: WJ_GQA_T(4); break;
    }
#undef WJ_GQA_T
#undef WJ_GQA
    WJ_LAUNCH_CHECK();
    {
      GemmArgs g;
      g.A = m->attn; g.lda = (split ? 2 : 1) * H * HD; g.split = split ? 1 : 0;
      g.W = m->W(b0 + WJ_QL_O_W); g.ldw = H * HD; g.M = M; g.N = D; g.K = H * HD; g.out = m->x; g.ldc = D;
      WJ_TRYQ(mx(g, l, 1));
      WJ_TRYQ(resid_gemm(g, gemm_variant(QG_O, M, H * HD, dt)));
    }
    sp_in = sp_gu;
    WJ_TRYQ(rms(m->F(b0 + WJ_QL_LN2_W), m->h));
    const int gu_variant = gemm_variant(QG_GATEUP, M, D, dt);
    // SwiGLU inside the gate-up GEMM's epilogue whenever an MFMA tile kernel serves the pass (16-bit types, more than 64 rows):
    // the [M][2F] gate / up matrix is never written
    const bool fused = g_qwen_fuse_swiglu && is16(dt) && !m->mx8 && M > 64 && (F % 32) == 0 && (D % 64) == 0 &&
                       (gu_variant == 0 || gu_variant == 3 || (gu_variant >= 73 && gu_variant <= 75));
    {
      GemmArgs g;
      g.A = m->h; g.lda = D; g.W = m->W(b0 + WJ_QL_GATEUP_W); g.ldw = D; g.M = M; g.N = 2 * F; g.K = D; g.out = m->gu; g.ldc = 2 * F;
      if (split) { g.split_out = 1; g.ldc = 4 * F; }
      if (sp_in) { g.split = 1; g.lda = 2 * D; }
      if (fused) { g.out = m->act; g.ldc = (split ? 2 : 1) * F; }
      WJ_TRYQ(mx(g, l, 2));
      WJ_TRYQ(launch_gemm(dt, fused ? EPI_SWIGLU_T : EPI_T, g, s, gu_variant));
    }
    if (!fused) {
      const dim3 grid((unsigned)ceil_div64((int64_t)M * F, 256));
      const int sp = split ? 1 : 0;
      if (dt == WJ_F32) hipLaunchKernelGGL((swiglu_kernel<float>), grid, dim3(256), 0, s, TP(const float, m->gu), TP(float, m->act), M, F, 0);
      else if (dt == WJ_F16) hipLaunchKernelGGL((swiglu_kernel<f16_t>), grid, dim3(256), 0, s, TP(const f16_t, m->gu), TP(f16_t, m->act), M, F, sp);
      else hipLaunchKernelGGL((swiglu_kernel<bf16_t>), grid, dim3(256), 0, s, TP(const bf16_t, m->gu), TP(bf16_t, m->act), M, F, sp);
      WJ_LAUNCH_CHECK();
    }
    {
      GemmArgs g;
      g.A = m->act; g.lda = (split ? 2 : 1) * F; g.split = split ? 1 : 0;
      g.W = m->W(b0 + WJ_QL_DOWN_W); g.ldw = F; g.M = M; g.N = D; g.K = F; g.out = m->x; g.ldc = D;
      WJ_TRYQ(mx(g, l, 3));
      WJ_TRYQ(resid_gemm(g, gemm_variant(QG_DOWN, M, F, dt)));
    }
  }
  if (pending) {      // the last down_proj: complete the residual stream without a norm
    if (dt == WJ_F16) hipLaunchKernelGGL((rmsnorm_kernel<f16_t>), dim3(M), dim3(256), 0, s, m->x, (const float*)nullptr, (f16_t*)nullptr, M, D, d.rms_eps, 0, m->slab, pending);
    else hipLaunchKernelGGL((rmsnorm_kernel<bf16_t>), dim3(M), dim3(256), 0, s, m->x, (const float*)nullptr, (bf16_t*)nullptr, M, D, d.rms_eps, 0, m->slab, pending);
    WJ_LAUNCH_CHECK();
  }
  return WJ_OK;
}

// final RMSNorm + tied LM head for `n` rows of `xin` (f32 [n][D]) -> m->logits, then arg-max + log-prob per row
int run_head(wj_qwen* m, const float* xin, int n, hipStream_t s) {
  const wj_qwen_dims& d = m->d;
  const int D = d.hidden, dt = m->dtype;
  const int sp = m->split_mode >= 1 ? 1 : 0;
  if (dt == WJ_F32) hipLaunchKernelGGL((rmsnorm_kernel<float>), dim3(n), dim3(256), 0, s, xin, m->F(WJ_Q_NORM_W), TP(float, m->h), n, D, d.rms_eps, 0);
  else if (dt == WJ_F16) hipLaunchKernelGGL((rmsnorm_kernel<f16_t>), dim3(n), dim3(256), 0, s, xin, m->F(WJ_Q_NORM_W), TP(f16_t, m->h), n, D, d.rms_eps, sp);
  else hipLaunchKernelGGL((rmsnorm_kernel<bf16_t>), dim3(n), dim3(256), 0, s, xin, m->F(WJ_Q_NORM_W), TP(bf16_t, m->h), n, D, d.rms_eps, sp);
  WJ_LAUNCH_CHECK();
  GemmArgs g;
  g.split = sp;
  // >= 1024 rows: N a multiple of 256 admits the 256-tile kernel (the surplus columns are dot products with zero rows; top-1 below
  // reads the first d.vocab columns only)
  g.A = m->h; g.lda = (sp ? 2 : 1) * D; g.W = m->W(WJ_Q_EMBED); g.ldw = D; g.M = n; g.N = n >= 1024 ? m->vocab_pad : d.vocab; g.K = D; g.out = m->logits; g.ldc = m->ldl;
  // 65 .. 1023 rows: the 128-tile kernel.  The dispatcher's own choice up to 512 rows is the skinny kernel, which re-reads the
  // activations for every 16 of the 151 936 columns (2.9 ms per call in the cfg5 trace against 1.2 ms for 1024+ rows)
  WJ_TRYQ(launch_gemm(dt, EPI_F32, g, s, (n > 64 && n < 1024 && dt != WJ_F32 && (D % 64) == 0) ? 3 : 0));
  if (m->cur_penalty != 1.f) {
    hipLaunchKernelGGL(rep_penalty_kernel, dim3(n), dim3(256), 0, s, m->logits, m->ldl, m->seen, m->seen_n, m->seen_cap, m->cur_penalty, m->finished, m->row_seq);
    WJ_LAUNCH_CHECK();
  }
  return launch_topk_logprob(m->logits, m->ldl, n, d.vocab, 1, nullptr, m->top_id, m->top_lp, m->top_lse, s);
}

int qalloc(wj_qwen* m, void** p, size_t bytes) {
  bytes = align_up(bytes ? bytes : 256, 256);
  WJ_HIP(hipMalloc(p, bytes));
  m->allocs.push_back(*p);
  WJ_HIP(hipMemsetAsync(*p, 0, bytes, m->ctx->stream));
  return WJ_OK;
}

// Work items of the prompt attention for a pass that presents n_seqs whole sequences from position 0, rows packed in order
int upload_prompt_work(wj_qwen* m, int n_seqs, const int32_t* n_tokens, hipStream_t s) {
  m->pwork_n = 0;
  if (m->dtype == WJ_F32 || !g_qwen_prompt_mfma) return WJ_OK;
  std::vector<int32_t>& w = m->pwork_host;
  w.clear();
  int row0 = 0;
  for (int b = 0; b < n_seqs; ++b) {
    for (int q0 = 0; q0 < n_tokens[b]; q0 += 128) { w.push_back(b); w.push_back(q0); w.push_back(row0); w.push_back(n_tokens[b]); }
    row0 += n_tokens[b];
  }
  const int n = (int)(w.size() / 4);
  if (n > m->pwork_cap) { set_error("wj_qwen: %d prompt attention work items (room for %d)", n, m->pwork_cap); return WJ_E_INVALID; }
  WJ_HIP(hipMemcpyAsync(m->pwork, w.data(), sizeof(int32_t) * w.size(), hipMemcpyHostToDevice, s));
  m->pwork_n = n;
  return WJ_OK;
}

}  // namespace

extern "C" {

int wj_qwen_free(wj_qwen* m) {
  if (!m) return WJ_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
  return WJ_OK;
}

int wj_qwen_create(wj_ctx* ctx, const wj_qwen_dims* dims, int dtype, const void* blob_dev, size_t blob_bytes,
                   const int64_t* offsets_host, int n_offsets, int max_seqs, int max_ctx, int max_rows, wj_qwen** out) {
  WJ_REQUIRE(ctx && dims && blob_dev && offsets_host && out, "wj_qwen_create: NULL argument");
  const wj_qwen_dims& d = *dims;
  WJ_REQUIRE(d.head_dim == HD, "wj_qwen_create: head_dim %d (the kernels are specialised for 128, every published Qwen3 size)", d.head_dim);
  WJ_REQUIRE(d.hidden > 0 && d.hidden % 8 == 0 && d.ffn > 0 && d.n_head >= 1 && d.n_kv_head >= 1 && d.n_head % d.n_kv_head == 0 &&
             d.n_layer >= 1 && d.vocab >= 2, "wj_qwen_create: bad dimensions");
  // GATEUP rows are interleaved in [16 gate | 16 up] blocks (wjhip.h): the last block of an ffn that is not a multiple of 16
  // would read its up rows past the 2 x ffn rows of the matrix
  WJ_REQUIRE(d.ffn % 16 == 0, "wj_qwen_create: ffn %d is not a multiple of 16 (the gate / up rows are interleaved in blocks of 16)", d.ffn);
  WJ_REQUIRE(d.n_head / d.n_kv_head == 1 || d.n_head / d.n_kv_head == 2 || d.n_head / d.n_kv_head == 4,
             "wj_qwen_create: %d query heads per KV head (the attention kernel is instantiated for 1, 2 and 4)", d.n_head / d.n_kv_head);
  WJ_REQUIRE(dtype == WJ_F32 || dtype == WJ_F16 || dtype == WJ_BF16 || dtype == WJ_F8W, "wj_qwen_create: unknown dtype %d", dtype);
  const bool f8w = dtype == WJ_F8W;
  if (f8w) {
    dtype = WJ_F16;       // storage / activation type of everything that is not an MX-fp8 projection
    WJ_REQUIRE(d.hidden % 128 == 0 && d.ffn % 128 == 0 && (d.n_head * d.head_dim) % 128 == 0,
               "wj_qwen_create: WJ_F8W needs hidden, ffn and heads x head_dim to be multiples of 128 (the k step of the MX matrix-core instruction)");
  }
  WJ_REQUIRE(n_offsets == WJ_Q_N_GLOBAL + d.n_layer * WJ_QL_N, "wj_qwen_create: %d tensor offsets expected, got %d",
             WJ_Q_N_GLOBAL + d.n_layer * WJ_QL_N, n_offsets);
  WJ_REQUIRE(max_seqs >= 1 && max_ctx >= 8 && max_rows >= max_seqs, "wj_qwen_create: need max_rows >= max_seqs >= 1 and max_ctx >= 8");
  for (int i = 0; i < n_offsets; ++i)
    WJ_REQUIRE(offsets_host[i] >= 0 && (size_t)offsets_host[i] < blob_bytes && offsets_host[i] % 16 == 0, "wj_qwen_create: bad offset %d", i);
  WJ_HIP(hipSetDevice(ctx->device));
  wj_qwen* m = new wj_qwen();
  m->ctx = ctx; m->d = d; m->dtype = dtype; m->esz = dtype_size(dtype);
  m->blob = reinterpret_cast<const char*>(blob_dev);
  m->off.assign(offsets_host, offsets_host + n_offsets);
  m->max_seqs = max_seqs; m->max_ctx = max_ctx; m->max_rows = max_rows;
  const size_t e = m->esz, R = max_rows, S = max_seqs;
  const int D = d.hidden, H = d.n_head, KV = d.n_kv_head, F = d.ffn;
  m->split_mode = dtype == WJ_F16 && (D % 64) == 0 && (F % 64) == 0 ? std::max(0, std::min(5, g_qwen_split_act)) : 0;      // 4 / 5: ablations of 3 (qkv input only / gate-up input only)
  m->mx8 = f8w;
  if (f8w && m->split_mode == 0) m->split_mode = 1;      // the LM head keeps fp16 weights and split activations
  const size_t sm = m->split_mode ? 2 : 1;
  {
    const int vp = (d.vocab + 255) / 256 * 256;
    const int64_t room = (n_offsets > 1 ? offsets_host[WJ_Q_EMBED + 1] : (int64_t)blob_bytes) - offsets_host[WJ_Q_EMBED];
    m->vocab_pad = room >= (int64_t)vp * d.hidden * (int64_t)m->esz ? vp : d.vocab;
  }
  m->ldl = (m->vocab_pad + 63) / 64 * 64;
  int rc = 0;
#define QA(field, bytes) do { if (!rc) rc = qalloc(m, reinterpret_cast<void**>(&m->field), (bytes)); } while (0)
  QA(x, R * D * sizeof(float)); QA(h, R * D * e * sm); QA(qkv, R * (size_t)(H + 2 * KV) * HD * e * sm); QA(q, R * (size_t)H * HD * e);
  QA(attn, R * (size_t)H * HD * e * sm); QA(gu, R * 2 * (size_t)F * e * sm); QA(act, R * (size_t)F * e * sm); QA(embed_ids, R * 4);
  QA(kc, (size_t)d.n_layer * S * KV * max_ctx * HD * e); QA(vc, (size_t)d.n_layer * S * KV * max_ctx * HD * e);
  QA(xl, S * D * sizeof(float)); QA(logits, S * m->ldl * sizeof(float));
  QA(row_seq, R * 4); QA(row_pos, R * 4); QA(last_rows, S * 4); QA(next_tok, S * 4); QA(finished, S * 4); QA(n_out, S * 4);
  QA(top_id, S * 4); QA(top_lp, S * 4); QA(top_lse, S * 4); QA(eos, 64);
  m->seen_cap = 2 * max_ctx;      // unique prompt ids (< max_ctx) + generated ids (positions stop at max_ctx)
  QA(lim, S * 4); QA(seen_n, S * 4); QA(seen, S * (size_t)m->seen_cap * 4);
  QA(cmp_src, S * 4); QA(cmp_seq, S * 4); QA(cmp_pos, S * 4); QA(cmp_tok, S * 4);
  m->slab_ks = dtype != WJ_F32 && !m->mx8 ? std::max(1, std::min(8, g_qwen_splitk)) : 1;
  if (m->slab_ks > 1) QA(slab, (int64_t)m->slab_ks * S * D * 4);
  QA(rope_tab, (int64_t)(max_ctx + 1) * 64 * 8);
  m->pwork_cap = m->max_rows / 128 + (int)S + 1;
  QA(pwork, (int64_t)m->pwork_cap * 16);
  if (f8w) {
    const int64_t Wq = (int64_t)(H + 2 * KV) * HD;
    const int64_t rows_of[4] = {Wq, D, 2 * (int64_t)F, D}, cols_of[4] = {D, (int64_t)H * HD, D, F};
    const int which[4] = {WJ_QL_QKV_W, WJ_QL_O_W, WJ_QL_GATEUP_W, WJ_QL_DOWN_W};
    int64_t bytes = 0, sbytes = 0;
    for (int l = 0; l < d.n_layer; ++l)
      for (int k = 0; k < 4; ++k) {
        m->w8_off.push_back(bytes); m->w8s_off.push_back(sbytes);
        bytes += rows_of[k] * cols_of[k]; sbytes += rows_of[k] * cols_of[k] / 32;
      }
    const int64_t maxk = std::max<int64_t>(std::max<int64_t>(D, (int64_t)H * HD), F);
    QA(w8, (size_t)bytes); QA(w8s, (size_t)sbytes); QA(a8, R * (size_t)maxk); QA(a8s, R * (size_t)maxk / 32);
    for (int l = 0; l < d.n_layer && !rc; ++l)
      for (int k = 0; k < 4 && !rc; ++k)
        rc = launch_mx8_quantize(WJ_F16, m->W(m->layer_base(l) + which[k]), cols_of[k], (int)rows_of[k], (int)cols_of[k],
                                 m->w8 + m->w8_off[l * 4 + k], m->w8s + m->w8s_off[l * 4 + k], ctx->stream);
  }
#undef QA
  if (!rc) {
    hipLaunchKernelGGL(rope_table_kernel, dim3(max_ctx + 1), dim3(64), 0, ctx->stream, m->rope_tab, max_ctx + 1, log2f(d.rope_theta));
    if (hipGetLastError() != hipSuccess) { set_error("wj_qwen_create: rope table launch failed"); rc = WJ_E_HIP; }
  }
  if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) { set_error("wj_qwen_create: allocation failed"); rc = WJ_E_HIP; }
  if (rc) { wj_qwen_free(m); return rc; }
  *out = m;
  return WJ_OK;
}

int wj_qwen_embed(wj_qwen* m, const int32_t* tokens_host, int n, float* out_dev, void* stream) {
  WJ_REQUIRE(m && tokens_host && out_dev && n >= 1 && n <= m->max_rows, "wj_qwen_embed: bad argument (n = %d, max_rows %d)", n, m ? m->max_rows : 0);
  for (int i = 0; i < n; ++i) WJ_REQUIRE(tokens_host[i] >= 0 && tokens_host[i] < m->d.vocab, "wj_qwen_embed: token %d out of range", tokens_host[i]);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  // own staging buffer: between wj_qwen_prefill and wj_qwen_generate_greedy row_seq holds the decode rows' sequence ids
  WJ_HIP(hipMemcpyAsync(m->embed_ids, tokens_host, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
  if (m->dtype == WJ_F32) hipLaunchKernelGGL((embed_rows_kernel<float>), dim3(n), dim3(256), 0, s, TP(const float, m->W(WJ_Q_EMBED)), m->embed_ids, out_dev, m->d.hidden);
  else if (m->dtype == WJ_F16) hipLaunchKernelGGL((embed_rows_kernel<f16_t>), dim3(n), dim3(256), 0, s, TP(const f16_t, m->W(WJ_Q_EMBED)), m->embed_ids, out_dev, m->d.hidden);
  else hipLaunchKernelGGL((embed_rows_kernel<bf16_t>), dim3(n), dim3(256), 0, s, TP(const bf16_t, m->W(WJ_Q_EMBED)), m->embed_ids, out_dev, m->d.hidden);
  WJ_LAUNCH_CHECK();
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

int wj_qwen_prefill(wj_qwen* m, const float* embeds_dev, int n_seqs, const int32_t* n_tokens_host, float* logits_out_dev, void* stream) {
  WJ_REQUIRE(m && embeds_dev && n_tokens_host, "wj_qwen_prefill: NULL argument");
  WJ_REQUIRE(n_seqs >= 1 && n_seqs <= m->max_seqs, "wj_qwen_prefill: %d sequences (max %d)", n_seqs, m->max_seqs);
  std::vector<int32_t> seq, pos, last(n_seqs);
  for (int b = 0; b < n_seqs; ++b) {
    WJ_REQUIRE(n_tokens_host[b] >= 1 && n_tokens_host[b] < m->max_ctx, "wj_qwen_prefill: sequence %d has %d tokens (context %d)", b,
               n_tokens_host[b], m->max_ctx);
    for (int t = 0; t < n_tokens_host[b]; ++t) { seq.push_back(b); pos.push_back(t); }
    last[b] = (int32_t)seq.size() - 1;
  }
  const int M = (int)seq.size();
  WJ_REQUIRE(M <= m->max_rows, "wj_qwen_prefill: %d prompt tokens in all (max_rows %d)", M, m->max_rows);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  WJ_HIP(hipMemcpyAsync(m->row_seq, seq.data(), sizeof(int32_t) * M, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->row_pos, pos.data(), sizeof(int32_t) * M, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->last_rows, last.data(), sizeof(int32_t) * n_seqs, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->x, embeds_dev, sizeof(float) * (size_t)M * m->d.hidden, hipMemcpyDeviceToDevice, s));   // packed rows
  m->n_seqs = 0;
  WJ_TRYQ(upload_prompt_work(m, n_seqs, n_tokens_host, s));
  { const int rc_l = run_layers(m, M, s, m->split_mode >= 2); m->pwork_n = 0; if (rc_l) return rc_l; }
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_seqs), dim3(256), 0, s, m->x, m->last_rows, m->xl, m->d.hidden);
  WJ_LAUNCH_CHECK();
  WJ_TRYQ(run_head(m, m->xl, n_seqs, s));
  if (logits_out_dev)
    WJ_HIP(hipMemcpy2DAsync(logits_out_dev, sizeof(float) * m->d.vocab, m->logits, sizeof(float) * m->ldl, sizeof(float) * m->d.vocab, n_seqs,
                            hipMemcpyDeviceToDevice, s));
  // decode state: one row per sequence from here on, at the position after the prompt
  std::vector<int32_t> ids(n_seqs), p1(n_seqs);
  for (int b = 0; b < n_seqs; ++b) { ids[b] = b; p1[b] = n_tokens_host[b]; }
  WJ_HIP(hipStreamSynchronize(s));
  WJ_HIP(hipMemcpyAsync(m->row_seq, ids.data(), sizeof(int32_t) * n_seqs, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->row_pos, p1.data(), sizeof(int32_t) * n_seqs, hipMemcpyHostToDevice, s));
  WJ_HIP(hipStreamSynchronize(s));
  m->n_seqs = n_seqs;
  m->prompt_len.assign(n_tokens_host, n_tokens_host + n_seqs);
  return WJ_OK;
}

int wj_qwen_last_used_graph(const wj_qwen* m) { return m ? m->last_used_graph : 0; }
int wj_qwen_last_steps(const wj_qwen* m) { return m ? m->last_steps : 0; }
int wj_qwen_last_truncated(const wj_qwen* m) { return m ? m->last_truncated : 0; }
int wj_qwen_last_compactions(const wj_qwen* m) { return m ? m->last_compactions : 0; }
int64_t wj_qwen_last_row_steps(const wj_qwen* m) { return m ? m->last_row_steps : 0; }

int wj_qwen_classify(wj_qwen* m, const float* embeds_dev, int n_seqs, const int32_t* n_tokens_host, const int32_t* rows_host, int n_rows,
                     const void* head_w_dev, const float* head_b_dev, int n_labels, int32_t* argmax_out_host, float* logits_out_dev,
                     void* stream) {
  WJ_REQUIRE(m && embeds_dev && n_tokens_host && rows_host && head_w_dev && argmax_out_host, "wj_qwen_classify: NULL argument");
  WJ_REQUIRE(n_seqs >= 1 && n_seqs <= m->max_seqs && n_rows >= 1 && n_rows <= m->max_rows, "wj_qwen_classify: %d sequences / %d rows do not fit", n_seqs, n_rows);
  WJ_REQUIRE(n_labels >= 2 && n_labels <= m->d.vocab, "wj_qwen_classify: %d labels (the logits buffer holds up to the vocabulary's %d)", n_labels, m->d.vocab);
  std::vector<int32_t> seq, pos;
  for (int b = 0; b < n_seqs; ++b) {
    WJ_REQUIRE(n_tokens_host[b] >= 1 && n_tokens_host[b] <= m->max_ctx, "wj_qwen_classify: sequence %d has %d tokens (context %d)", b, n_tokens_host[b], m->max_ctx);
    for (int t = 0; t < n_tokens_host[b]; ++t) { seq.push_back(b); pos.push_back(t); }
  }
  const int M = (int)seq.size();
  WJ_REQUIRE(M <= m->max_rows, "wj_qwen_classify: %d tokens in all (max_rows %d)", M, m->max_rows);
  for (int i = 0; i < n_rows; ++i) WJ_REQUIRE(rows_host[i] >= 0 && rows_host[i] < M, "wj_qwen_classify: row %d out of range", rows_host[i]);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  WJ_HIP(hipMemcpyAsync(m->row_seq, seq.data(), sizeof(int32_t) * M, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->row_pos, pos.data(), sizeof(int32_t) * M, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(m->x, embeds_dev, sizeof(float) * (size_t)M * m->d.hidden, hipMemcpyDeviceToDevice, s));
  m->n_seqs = 0;       // the caches are about to hold a classification pass: generation needs its own prefill
  WJ_TRYQ(upload_prompt_work(m, n_seqs, n_tokens_host, s));
  { const int rc_l = run_layers(m, M, s, m->split_mode >= 2); m->pwork_n = 0; if (rc_l) return rc_l; }
  // the selected rows (e.g. the <timestamp> markers of a forced-alignment prompt): final RMSNorm + the caller's head.  The
  // gather goes through a scratch allocation because x is the residual stream being read.
  int32_t* tmp_rows = nullptr;
  float* gathered = nullptr;
  if (hipMalloc(&gathered, sizeof(float) * (size_t)n_rows * m->d.hidden) != hipSuccess) { set_error("wj_qwen_classify: out of memory"); return WJ_E_HIP; }
  if (hipMalloc(&tmp_rows, sizeof(int32_t) * (size_t)n_rows) != hipSuccess) { (void)hipFree(gathered); set_error("wj_qwen_classify: out of memory"); return WJ_E_HIP; }
  float* d_logits = nullptr;
  int32_t* d_ids = nullptr;
  float* d_lp = nullptr;
  const int64_t ldl = (n_labels + 63) / 64 * 64;
  struct Guard { void *a, *b, *c, *d, *e; ~Guard() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); (void)hipFree(e); } } guard{gathered, tmp_rows, nullptr, nullptr, nullptr};
  if (hipMalloc(&d_logits, sizeof(float) * (size_t)n_rows * ldl) != hipSuccess || hipMalloc(&d_ids, sizeof(int32_t) * n_rows) != hipSuccess ||
      hipMalloc(&d_lp, sizeof(float) * 2 * n_rows) != hipSuccess) {
    guard.c = d_logits; guard.d = d_ids; guard.e = d_lp;
    set_error("wj_qwen_classify: out of memory");
    return WJ_E_HIP;
  }
  guard.c = d_logits; guard.d = d_ids; guard.e = d_lp;
  WJ_HIP(hipMemcpyAsync(tmp_rows, rows_host, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_rows), dim3(256), 0, s, m->x, tmp_rows, gathered, m->d.hidden);
  WJ_LAUNCH_CHECK();
  const int D = m->d.hidden, dt = m->dtype;
  // normed rows go to the head GEMM in the compute type; h holds max_rows x D elements
  if (dt == WJ_F32) hipLaunchKernelGGL((rmsnorm_kernel<float>), dim3(n_rows), dim3(256), 0, s, gathered, m->F(WJ_Q_NORM_W), TP(float, m->h), n_rows, D, m->d.rms_eps);
  else if (dt == WJ_F16) hipLaunchKernelGGL((rmsnorm_kernel<f16_t>), dim3(n_rows), dim3(256), 0, s, gathered, m->F(WJ_Q_NORM_W), TP(f16_t, m->h), n_rows, D, m->d.rms_eps);
  else hipLaunchKernelGGL((rmsnorm_kernel<bf16_t>), dim3(n_rows), dim3(256), 0, s, gathered, m->F(WJ_Q_NORM_W), TP(bf16_t, m->h), n_rows, D, m->d.rms_eps);
  WJ_LAUNCH_CHECK();
  GemmArgs g;
  g.A = m->h; g.lda = D; g.W = head_w_dev; g.ldw = D; g.bias = head_b_dev; g.M = n_rows; g.N = n_labels; g.K = D; g.out = d_logits; g.ldc = ldl;
  WJ_TRYQ(launch_gemm(dt, EPI_F32, g, s, 0));
  WJ_TRYQ(launch_topk_logprob(d_logits, ldl, n_rows, n_labels, 1, nullptr, d_ids, d_lp, d_lp + n_rows, s));
  WJ_HIP(hipMemcpyAsync(argmax_out_host, d_ids, sizeof(int32_t) * n_rows, hipMemcpyDeviceToHost, s));
  if (logits_out_dev)
    WJ_HIP(hipMemcpy2DAsync(logits_out_dev, sizeof(float) * n_labels, d_logits, sizeof(float) * ldl, sizeof(float) * n_labels, n_rows,
                            hipMemcpyDeviceToDevice, s));
  WJ_HIP(hipStreamSynchronize(s));
  m->n_seqs = 0;       // the caches hold a classification pass: generation needs its own prefill
  return WJ_OK;
}

int wj_qwen_generate_greedy(wj_qwen* m, const int32_t* eos_ids_host, int n_eos, int max_new, int32_t* tokens_out, int32_t* n_tokens_out,
                            float* token_logprob_out, void* stream) {
  return wj_qwen_generate_greedy_ex(m, eos_ids_host, n_eos, max_new, nullptr, 1.0f, nullptr, nullptr, tokens_out, n_tokens_out, token_logprob_out, stream);
}

int wj_qwen_generate_greedy_ex(wj_qwen* m, const int32_t* eos_ids_host, int n_eos, int max_new, const int32_t* max_new_per_seq_host,
                               float repetition_penalty, const int32_t* seen_ids_host, const int32_t* seen_offsets_host,
                               int32_t* tokens_out, int32_t* n_tokens_out, float* token_logprob_out, void* stream) {
  WJ_REQUIRE(m && eos_ids_host && tokens_out && n_tokens_out, "wj_qwen_generate_greedy: NULL argument");
  WJ_REQUIRE(m->n_seqs >= 1, "wj_qwen_generate_greedy: call wj_qwen_prefill first");
  WJ_REQUIRE(n_eos >= 1 && n_eos <= 16 && max_new >= 1, "wj_qwen_generate_greedy: 1..16 EOS ids and max_new >= 1");
  WJ_REQUIRE(repetition_penalty > 0.f, "wj_qwen_generate_greedy_ex: repetition_penalty must be positive (1 = off)");
  const bool penalise = repetition_penalty != 1.0f;
  WJ_REQUIRE(!penalise || (seen_ids_host && seen_offsets_host), "wj_qwen_generate_greedy_ex: a repetition penalty needs the prompts' token ids "
             "(transformers penalises every id of input_ids, prompt included)");
  const int S = m->n_seqs;
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  int32_t* d_tok = nullptr;
  float* d_lp = nullptr;
  WJ_HIP(hipMalloc(&d_tok, sizeof(int32_t) * (size_t)S * max_new));
  if (hipMalloc(&d_lp, sizeof(float) * (size_t)S * (max_new + 1)) != hipSuccess) { (void)hipFree(d_tok); set_error("wj_qwen_generate_greedy: out of memory"); return WJ_E_HIP; }
  struct Guard { int32_t* a; float* b; ~Guard() { (void)hipFree(a); (void)hipFree(b); } } guard{d_tok, d_lp};
  WJ_HIP(hipMemsetAsync(d_tok, 0, sizeof(int32_t) * (size_t)S * max_new, s));
  WJ_HIP(hipMemsetAsync(d_lp, 0, sizeof(float) * (size_t)S * (max_new + 1), s));
  WJ_HIP(hipMemsetAsync(m->finished, 0, sizeof(int32_t) * S, s));
  WJ_HIP(hipMemsetAsync(m->n_out, 0, sizeof(int32_t) * S, s));
  WJ_HIP(hipMemcpyAsync(m->eos, eos_ids_host, sizeof(int32_t) * n_eos, hipMemcpyHostToDevice, s));
  std::vector<int32_t> fin(S), lim_h(S, max_new);
  if (max_new_per_seq_host)
    for (int b = 0; b < S; ++b) {
      WJ_REQUIRE(max_new_per_seq_host[b] >= 0, "wj_qwen_generate_greedy_ex: negative token budget for sequence %d", b);
      lim_h[b] = std::min(max_new_per_seq_host[b], max_new);
    }
  // A sequence can never outgrow its KV cache: token g of sequence b is fed at position prompt_len + g - 1, so at most
  // max_ctx - prompt_len tokens fit.  The budget is cut to that room (wj_qwen_last_truncated tells how many sequences were
  // cut): past it the cache position used to be clamped silently and the sequence decoded garbage.
  m->last_truncated = 0;
  for (int b = 0; b < S; ++b) {
    const int room = std::max(0, m->max_ctx - m->prompt_len[b]);
    if (lim_h[b] > room) { lim_h[b] = room; ++m->last_truncated; }
  }
  WJ_HIP(hipMemcpyAsync(m->lim, lim_h.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice, s));
  std::vector<int32_t> seen_h, seen_cnt(S, 0);
  if (penalise) {
    seen_h.assign((size_t)S * m->seen_cap, 0);
    for (int b = 0; b < S; ++b) {
      WJ_REQUIRE(seen_offsets_host[b + 1] >= seen_offsets_host[b], "wj_qwen_generate_greedy_ex: seen_offsets must not decrease");
      std::vector<int32_t> u(seen_ids_host + seen_offsets_host[b], seen_ids_host + seen_offsets_host[b + 1]);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      WJ_REQUIRE(u.empty() || (u.front() >= 0 && u.back() < m->d.vocab), "wj_qwen_generate_greedy_ex: token id out of range in sequence %d", b);
      WJ_REQUIRE((int)u.size() <= m->max_ctx, "wj_qwen_generate_greedy_ex: %d distinct prompt ids for sequence %d (context %d)", (int)u.size(), b, m->max_ctx);
      std::copy(u.begin(), u.end(), seen_h.begin() + (size_t)b * m->seen_cap);
      seen_cnt[b] = (int32_t)u.size();
    }
    WJ_HIP(hipMemcpyAsync(m->seen, seen_h.data(), sizeof(int32_t) * seen_h.size(), hipMemcpyHostToDevice, s));
    WJ_HIP(hipMemcpyAsync(m->seen_n, seen_cnt.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice, s));
    // the prefill left UNPENALISED logits of the last prompt positions in m->logits: penalise them, choose again
    hipLaunchKernelGGL(rep_penalty_kernel, dim3(S), dim3(256), 0, s, m->logits, m->ldl, m->seen, m->seen_n, m->seen_cap, repetition_penalty, m->finished, m->row_seq);
    WJ_LAUNCH_CHECK();
    WJ_TRYQ(launch_topk_logprob(m->logits, m->ldl, S, m->d.vocab, 1, nullptr, m->top_id, m->top_lp, m->top_lse, s));
  }
  struct PenaltyScope { wj_qwen* m; ~PenaltyScope() { m->cur_penalty = 1.f; } } penalty_scope{m};
  m->cur_penalty = repetition_penalty;
  int live = S;          // rows of the decode batch: sequences that have not ended (round 4: the batch is re-packed at the polls)
  auto advance = [&](int first) -> int {
    hipLaunchKernelGGL(advance_kernel, dim3(ceil_div(live, 64)), dim3(64), 0, s, m->top_id, m->top_lp, m->eos, n_eos, m->finished, m->n_out,
                       m->row_pos, m->next_tok, d_tok, d_lp, max_new, live, m->max_ctx, first, m->lim, penalise ? m->seen : nullptr, m->seen_n,
                       m->seen_cap, m->row_seq);
    WJ_LAUNCH_CHECK();
    return WJ_OK;
  };
  // one decode iteration: embed the token chosen last -> layers -> head -> record the next token.  Every step-dependent
  // value (positions, tokens, counters) lives in device memory, so the iteration is captured once and replayed from a
  // hipGraph: ~260 launches of a few microseconds each would otherwise be issued by the host per token.
  auto iteration = [&]() -> int {
    if (m->dtype == WJ_F32) hipLaunchKernelGGL((embed_rows_kernel<float>), dim3(live), dim3(256), 0, s, TP(const float, m->W(WJ_Q_EMBED)), m->next_tok, m->x, m->d.hidden);
    else if (m->dtype == WJ_F16) hipLaunchKernelGGL((embed_rows_kernel<f16_t>), dim3(live), dim3(256), 0, s, TP(const f16_t, m->W(WJ_Q_EMBED)), m->next_tok, m->x, m->d.hidden);
    else hipLaunchKernelGGL((embed_rows_kernel<bf16_t>), dim3(live), dim3(256), 0, s, TP(const bf16_t, m->W(WJ_Q_EMBED)), m->next_tok, m->x, m->d.hidden);
    WJ_LAUNCH_CHECK();
    WJ_TRYQ(run_layers(m, live, s, m->split_mode >= 1));
    WJ_TRYQ(run_head(m, m->x, live, s));
    return advance(0);
  };
  // the prefill left the arg-max of every sequence's last prompt position in top_id / top_lp
  WJ_TRYQ(advance(1));
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  const char* env = getenv("WJ_NO_GRAPH");
  bool use_graph = !(env && env[0] == '1') && max_new > 2;
  int rc_loop = WJ_OK;
  m->last_used_graph = 0;
  m->last_steps = 0;
  m->last_compactions = 0;
  m->last_row_steps = 0;
  m->n_seqs = 0;            // one generation per prefill: the decode state (positions, penalised logits) is consumed below
  std::vector<int32_t> row_of(S);        // sequence id of every row of the current batch (host mirror of row_seq)
  for (int b = 0; b < S; ++b) row_of[b] = b;
  bool capture_next = false;
  for (int k = 1; k <= max_new && rc_loop == WJ_OK; ++k) {
    m->last_steps = k;
    m->last_row_steps += live;
    if ((k == 2 || capture_next) && use_graph) {       // the first iteration ran eagerly (one-time kernel attributes are set); capture the second
      capture_next = false;
      if (exec) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
      if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
      if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = iteration();
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc || e != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
          (void)hipGetLastError();
          if (exec) { (void)hipGraphExecDestroy(exec); exec = nullptr; }
          use_graph = false;
        }
      } else {
        (void)hipGetLastError();
        use_graph = false;
      }
      m->last_used_graph = use_graph ? 1 : 0;
    }
    if (exec) {
      if (hipGraphLaunch(exec, s) != hipSuccess) { set_error("wj_qwen_generate_greedy: graph launch failed"); rc_loop = WJ_E_HIP; }
    } else {
      rc_loop = iteration();
    }
    if ((k & 7) == 0 && k < max_new && rc_loop == WJ_OK) {
      if (hipMemcpyAsync(fin.data(), m->finished, sizeof(int32_t) * S, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        set_error("wj_qwen_generate_greedy: poll failed"); rc_loop = WJ_E_HIP; break;
      }
      if (std::all_of(fin.begin(), fin.end(), [](int32_t f) { return f != 0; })) break;
      // finished sequences leave the batch: an iteration's cost is proportional to its rows (the GEMMs), and the K/V caches are
      // addressed through row_seq, so only three small per-row arrays move.  Done when at least g_qwen_compact_pct % of the rows
      // have ended (each re-pack costs a graph capture).
      std::vector<int32_t> keep;
      for (int r = 0; r < live; ++r)
        if (!fin[row_of[r]]) keep.push_back(r);
      if (g_qwen_compact_pct > 0 && (int)keep.size() < live && (live - (int)keep.size()) * 100 >= live * g_qwen_compact_pct) {
        const int nl = (int)keep.size();
        if (hipMemcpyAsync(m->cmp_src, keep.data(), sizeof(int32_t) * nl, hipMemcpyHostToDevice, s) != hipSuccess) { set_error("wj_qwen_generate_greedy: compaction upload failed"); rc_loop = WJ_E_HIP; break; }
        hipLaunchKernelGGL(compact_decode_rows_kernel, dim3(ceil_div(nl, 64)), dim3(64), 0, s, m->cmp_src, nl, m->row_seq, m->row_pos, m->next_tok,
                           m->cmp_seq, m->cmp_pos, m->cmp_tok);
        (void)hipMemcpyAsync(m->row_seq, m->cmp_seq, sizeof(int32_t) * nl, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(m->row_pos, m->cmp_pos, sizeof(int32_t) * nl, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(m->next_tok, m->cmp_tok, sizeof(int32_t) * nl, hipMemcpyDeviceToDevice, s);
        if (hipStreamSynchronize(s) != hipSuccess) { set_error("wj_qwen_generate_greedy: compaction failed"); rc_loop = WJ_E_HIP; break; }
        std::vector<int32_t> nr(nl);
        for (int r = 0; r < nl; ++r) nr[r] = row_of[keep[r]];
        row_of.swap(nr);
        live = nl;
        ++m->last_compactions;
        capture_next = true;         // the grids depend on the row count: capture the iteration again
      }
    }
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  if (rc_loop) return rc_loop;
  WJ_HIP(hipMemcpyAsync(tokens_out, d_tok, sizeof(int32_t) * (size_t)S * max_new, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(n_tokens_out, m->n_out, sizeof(int32_t) * S, hipMemcpyDeviceToHost, s));
  if (token_logprob_out) WJ_HIP(hipMemcpyAsync(token_logprob_out, d_lp, sizeof(float) * (size_t)S * (max_new + 1), hipMemcpyDeviceToHost, s));
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

}  // extern "C"
