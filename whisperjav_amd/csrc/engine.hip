// engine.hip -- host-side runtime of libwjhip: context, Whisper model object (weights view,
// HBM-resident workspaces, KV caches), encoder / decoder launch sequences, hipGraph-replayed greedy
// decode loop, and the C ABI declared in include/wjhip.h.
#include <stdarg.h>
#include <stdlib.h>

#include <vector>

#include "kernels.hpp"

namespace wj {
static thread_local char g_err[1024] = {0};
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace wj

using namespace wj;

struct wj_profiler {
  struct Pair { int tag; hipEvent_t a, b; int64_t units; };
  std::vector<Pair> pairs;
  int open_tag = -1;
  hipEvent_t open_a = nullptr;
};
static const char* kProfNames[PT_COUNT] = {
    "mel_to_rows", "conv1_gemm", "conv2_gemm", "enc_layernorm", "enc_qk_gemm", "enc_v_gemm", "enc_attention",
    "enc_out_gemm", "enc_fc1_gemm", "enc_fc2_gemm", "cross_kv_gemm", "dec_embed", "dec_layernorm", "dec_qkv_gemm",
    "dec_self_attn", "dec_out_gemm", "dec_cq_gemm", "dec_cross_attn", "dec_cout_gemm", "dec_fc1_gemm", "dec_fc2_gemm",
    "dec_logits_gemm", "dec_sample", "dec_misc"};
namespace wj {
void prof_begin(wj_ctx* ctx, int tag, hipStream_t s) {
  wj_profiler* p = ctx->prof;
  if (!p) return;
  hipEvent_t a = nullptr;
  if (hipEventCreate(&a) != hipSuccess) return;
  (void)hipEventRecord(a, s);
  p->open_tag = tag;
  p->open_a = a;
}
void prof_end(wj_ctx* ctx, hipStream_t s) {
  wj_profiler* p = ctx->prof;
  if (!p || p->open_tag < 0) return;
  hipEvent_t b = nullptr;
  if (hipEventCreate(&b) != hipSuccess) return;
  (void)hipEventRecord(b, s);
  p->pairs.push_back({p->open_tag, p->open_a, b, ctx->prof_units});
  p->open_tag = -1;
}
}  // namespace wj
#define PROF(tag, call)                 \
  do {                                  \
    prof_begin(m->ctx, (tag), s);       \
    int _prc = (call);                  \
    prof_end(m->ctx, s);                \
    if (_prc) return _prc;              \
  } while (0)

int wj_ctx::ensure_scratch(size_t bytes) {
  if (bytes <= scratch_bytes) return WJ_OK;
  if (scratch) {
    WJ_HIP(hipStreamSynchronize(stream));
    WJ_HIP(hipFree(scratch));
    scratch = nullptr;
    scratch_bytes = 0;
  }
  const size_t want = align_up(bytes, (size_t)1 << 20);
  WJ_HIP(hipMalloc(&scratch, want));
  scratch_bytes = want;
  return WJ_OK;
}

// ------------------------------------------------------------------------------------------------
// model object
// ------------------------------------------------------------------------------------------------
static constexpr int kDecKsMax = 16;
static constexpr int kFinCap = 24;   // finished hypotheses kept per window: round(beam * patience) + beam <= 24

// run-time tunables (wj_tune): defaults chosen from the MI355X sweeps recorded in profiles/
struct Tunables {
  int dec_ks_attn = 4;      // split-K factor of the attention out-projection GEMMs of the decode step
  int dec_ks_fc2 = 8;       // split-K factor of the decode fc2 GEMM (K = 4d)
  int dec_tile_min_m = 128; // rows from which the split-K decode GEMMs use the 128x128 tile kernel (0 = never)
  int decode_chains = 1;    // concurrent row chains in the greedy loop
  int dec_ks_proj = 4;      // split-K factor of the K = d projections (QKV, cross-q, fc1); 0 = single-pass kernels
  int dec_proj_min_m = 96;  // rows from which those projections go split-K (tile kernel + reduce kernel)
  int dec_rows = 1;         // decode GEMMs with <= dec_rows_max_m rows use the one-wave-per-row-block kernel
  int dec_rows_max_m = 32;   // measured: faster than the alternatives up to ~32 rows, slower from 64 (vector-L1 line rate)
  int dec_rows_ks_attn = 2; // its split-K factors for the residual projections (K = d) and fc2 (K = 4d)
  int dec_rows_ks_fc2 = 8;
  int dec_tile_reg = 0;     // 1: the decode tile GEMMs stage through registers (global->VGPR->LDS) instead of LDS-DMA
  int dec_ms_stages = 0;    // LDS-DMA stages of ALL decode tile GEMMs (0 = the 2-stage kernel, 2 workgroups per CU: faster for
                            // the wide projections, 746 vs 593 TFLOP/s on 1920 x 3840 x 1280)
  int dec_ms_resid = 0;     // ... of the residual-writing ones only.  Stand-alone (one 150-tile launch, 1920 x 1280 x 1280 / 5120)
                            // 3 stages are 6-10 % faster; in the step these GEMMs run as 2 K-slices = 300 workgroups, which
                            // fit the 256 CUs at once with the 64 KiB kernel (2 per CU) and need a second round with 96 KiB:
                            // measured +18-28 % on the out / cross-out / fc2 GEMMs of the 120-min run.  0 = off
  int dec_fuse_reduce = 1;  // attention kernels consume the q / qkv split-K slices directly (no reduce launch)
  int align_prefill = 1;    // word-timestamp alignment as one full-sequence decoder pass (0: token by token)
  int dec_cross_mfma = 1;   // 16-bit models: cross V kept transposed, cross attention on the matrix cores (read at create)
  int dec_split_act = 1;    // fp16 models: decode-step GEMMs take their activations as hi + lo fp16 pairs (read at create):
                            // 1 = the residual-writing GEMMs (attention out-projections, fc2) and the logits GEMM,
                            // 2 = every decode GEMM, 0 = none
  int dec_adapt_ks = 1;     // halve the split-K factors while the row tiles alone keep >= dec_adapt_wgs workgroups busy
  int dec_adapt_wgs = 256;  // ... 256 = one per CU (the 64 KiB tile kernels fit two per CU); sweeps: profiles/r06_sweep_*
  int self_kv_len = 0;      // positions of the self-attention KV cache per row (read at create; 0 = n_text_ctx).  The cache is
                            // [L][rows][H][positions][64] x 2: 141 GB at 1920 rows x 448 positions, 22 GB at 70 -- a caller that
                            // knows its prompt + max_new_tokens buys batch size with it
  int enc_batch = 0;        // windows per encoder slice (read at create; 0 = max_batch): bounds the encoder workspaces
  int beam_compact = 1;     // device beam search: windows whose search has ended leave the batch at the next poll (the step's cost
                            // is proportional to the live windows: cross attention reads their K/V, the GEMMs their rows)
  int beam_poll = 4;        // decode iterations between two polls of the per-window done flags
  int beam_token_logprobs = 0;  // beam search also carries the cumulative log-prob after every token of every hypothesis, so the
                            // winner's per-token log-probs can be read back (wj_whisper_last_beam_token_logprobs); a diagnostic
                            // mode: the batch is not compacted while it is on
  int beam_compact_pct = 12;  // compact when at least this share of the batch's windows ...
  int beam_compact_min = 8;   // ... and at least this many of them have finished
  int enc_blocked = 1;      // 16-bit models with d_model % 256 == 0 (read at create): the encoder's big GEMMs take BLOCKED operands
                            // ([rows / 256][K / 32][256][32]: every LDS-DMA wave request 1 KiB contiguous) -- weights copied
                            // once into that layout, LayerNorm / attention / fc1 write it; +5..+20 % per GEMM (r04 gemm probe)
  int batch_invariant = 0;  // 16-bit models: ONE kernel family and FIXED split-K factors for every decode / alignment GEMM whatever the row count
                            // (the tile kernels, dec_ks_* as set, no adaptation, no rows / skinny kernels): a window's tokens and log-probs are
                            // then bit-identical whatever shares its batch -- other windows, other shards of a multi-GPU run, compaction.  A
                            // reproducibility mode (small batches run slower); the default picks the fastest kernel per row count, whose fp32
                            // summation orders differ
  int dec_big_min_m = 0;    // rows from which the wide decode projections (qkv, fc1) use the 256x256 kernel; measured at 1920 rows: 14.00 s vs 13.89 s for the 128-tile kernel (160 workgroups do not fill 256 CUs), so off
};
static Tunables g_tune;

struct wj_whisper {
  wj_ctx* ctx = nullptr;
  wj_whisper_dims d{};
  int dtype = WJ_BF16;
  size_t esz = 2;
  const char* blob = nullptr;
  std::vector<int64_t> off;
  int max_batch = 0, max_rows = 0;
  int enc_batch = 0; // windows the encoder workspaces hold: wj_whisper_encode runs larger batches in slices of this size
  int kv_len = 0;    // positions of the self-attention KV cache per row (<= n_text_ctx; wj_tune "self_kv_len" at create)
  int Tpad = 0;      // encoder positions padded to a multiple of 128
  int frames = 0;    // 2 * n_audio_ctx
  std::vector<void*> allocs;
  size_t alloc_bytes = 0;

  // encoder workspaces (all windows of a batch at once; 288 GB of HBM make chunking unnecessary)
  void* mel_rows = nullptr;   // T   [B][frames+2][n_mels]
  void* conv1_out = nullptr;  // T   [B][frames+2][d]
  float* x = nullptr;         // f32 [B][ctx][d]        residual stream
  void* h = nullptr;          // T   [B*ctx][d]         LayerNorm output / encoder output
  void* q = nullptr;          // T   [B][H][Tpad][64]
  void* k = nullptr;          // T   [B][H][Tpad][64]
  void* vt = nullptr;         // T   [B][H][64][Tpad]
  void* attn = nullptr;       // T   [B*ctx][d]
  void* ff = nullptr;         // T   [B*ctx][4d]
  // per-window state kept for decoding
  void* cross_k = nullptr;    // T   [L][max_batch][H][ctx][64]
  void* cross_v = nullptr;    // same, or (cross_tpad > 0) transposed per head: [L][max_batch][H][64][cross_tpad]
  int cross_tpad = 0;
  // blocked GEMM operands of the encoder (wj_tune "enc_blocked"): copies of the encoder / cross-K/V matrices in the layout
  // [N / 256][K / 32][256][32]; wblk_off[tensor index] = element offset of the copy, -1 = none
  bool blk = false;
  void* wblk = nullptr;
  std::vector<int64_t> wblk_off;
  const void* WB(int idx, int64_t row0 = 0, int K = 0) const {   // blocked copy of tensor idx from row row0 (a multiple of 256) on
    return reinterpret_cast<const char*>(wblk) + (wblk_off[idx] + row0 * K) * (int64_t)esz;
  }
  // fp16: the decode-step GEMMs read their activations as [hi | lo] rows (x to ~22 bits; profiles/r02_precision_*):
  // the decode step is HBM / latency bound, the second MFMA per fragment is free, and the per-token log-probs move
  // from ~2e-3 to ~3e-4 of the fp32 evaluation.  dh / dattn / dff rows are then twice as wide.
  bool split_act = false;
  bool split_all = false;     // also the q/k/v, cross-q and fc1 projections (wj_tune dec_split_act = 2)
  // decoder workspaces
  float* dx = nullptr;        // f32 [R][d]
  void* dh = nullptr;         // T   [R][d]
  void* dq = nullptr;         // T   [R][d]
  void* dattn = nullptr;      // T   [R][d]
  void* dff = nullptr;        // T   [R][4d]
  float* partial = nullptr;   // f32 [R][KS_MAX][d] split-K slabs of the decode GEMMs (per row slice)
  float* logits = nullptr;    // f32 [R][ldl]
  int64_t ldl = 0;
  void* self_k = nullptr;     // T   [L][max_rows][H][kv_len][64]
  void* self_v = nullptr;
  int32_t* tokens = nullptr;  // [max_rows][tok_stride]
  int64_t tok_stride = 0;
  int* pos = nullptr;         // device scalar: index of the token being fed
  float* sum_lp = nullptr;    // [R]
  float* tok_lp = nullptr;    // [R][tok_stride]
  int32_t* finished = nullptr;
  float* nsp = nullptr;       // [R] no-speech probability
  int32_t* row_map[2] = {nullptr, nullptr};  // [max_rows][kv_len]
  int cur_map = 0;
  int32_t* parent = nullptr;  // [R]
  int32_t* step_tok = nullptr;  // [R] staging for wj_decode_step
  int32_t* slot_map = nullptr;  // [max_batch] window slot of each decoded window (sub-batch re-decodes)
  bool use_slots = false;
  // device-resident beam search (wj_whisper_decode_beam)
  int32_t* tokens2 = nullptr;     // second history buffer (histories are gathered by parent every step)
  float* beam_score = nullptr;    // [R]
  int32_t* beam_done = nullptr;   // [max_batch] + n_done at [max_batch]
  int32_t* win_ids = nullptr;     // [max_batch] result-array index of each logical window (compaction of finished windows)
  int32_t* src_rows = nullptr;    // [max_rows] staging of a compaction's row gather
  int32_t* fin_count = nullptr;   // [max_batch]
  float* fin_score = nullptr;     // [max_batch][kFinCap]
  int32_t* fin_len = nullptr;
  int32_t* fin_tokens = nullptr;  // [max_batch][kFinCap][tok_stride]
  float* cum2 = nullptr;          // wj_tune beam_token_logprobs: second cumulative-score history (tok_lp is the first), allocated on first use
  float* fin_cum = nullptr;       // ... and the finished hypotheses' [max_batch][kFinCap][tok_stride]
  std::vector<float> last_beam_lp;   // winner's per-token log-probs of the last beam search [batch][max_new + 1], NaN padded
  int last_beam_lp_batch = 0, last_beam_lp_stride = 0;
  // word-timestamp alignment (wj_whisper_align): scratch grown on demand, selection table [L][H]
  void* align_buf = nullptr;
  size_t align_bytes = 0;
  int32_t* align_sel = nullptr;
  float* dump_qk = nullptr;     // non-NULL while the teacher-forced pass of wj_whisper_align runs
  const float* last_align_matrix = nullptr;   // the head-averaged matrix the last wj_whisper_align ran its DTW on (in align_buf)
  int last_align_batch = 0, last_align_T = 0;
  int dump_nsel = 0, dump_tmax = 0;
  int32_t* topk_ids = nullptr;  // [R][16]
  float* topk_lp = nullptr;
  float* topk_lse = nullptr;
  // step-wise decode state
  int open_batch = 0, open_beam = 0, open_rows = 0, host_pos = 0;
  int last_used_graph = 0, last_chains = 1;   // diagnostics of the last decode call
  int last_steps = 0, last_max_new = 0;       // decode iterations it ran / was allowed to run
  int last_compactions = 0;                   // beam search: times the batch was re-packed
  int64_t last_window_steps = 0;              // ... sum over iterations of the live windows (the work actually done)

  const void* W(int idx) const { return blob + off[idx]; }
  const float* F(int idx) const { return reinterpret_cast<const float*>(blob + off[idx]); }
  int enc_base(int l) const { return WJ_T_N_GLOBAL + l * WJ_TE_N; }
  int dec_base(int l) const { return WJ_T_N_GLOBAL + d.n_audio_layer * WJ_TE_N + l * WJ_TD_N; }
  int64_t cross_layer_elems() const { return (int64_t)max_batch * d.n_text_head * d.n_audio_ctx * 64; }
  int64_t cross_v_layer_elems() const {
    return (int64_t)max_batch * d.n_text_head * 64 * (cross_tpad > 0 ? cross_tpad : d.n_audio_ctx);
  }
  int64_t self_layer_elems() const { return (int64_t)max_rows * d.n_text_head * kv_len * 64; }
  void* at(void* base, int64_t elems) const { return reinterpret_cast<char*>(base) + elems * (int64_t)esz; }
};

static int dev_alloc(wj_whisper* m, void** p, size_t bytes, bool zero) {
  bytes = align_up(bytes ? bytes : 256, 256);
  WJ_HIP(hipMalloc(p, bytes));
  m->allocs.push_back(*p);
  m->alloc_bytes += bytes;
  if (zero) WJ_HIP(hipMemsetAsync(*p, 0, bytes, m->ctx->stream));
  return WJ_OK;
}
#define WJ_ALLOC(field, bytes, zero)                                        \
  do {                                                                      \
    int _rc = dev_alloc(m, reinterpret_cast<void**>(&m->field), (bytes), (zero)); \
    if (_rc) { wj_whisper_free(m); return _rc; }                            \
  } while (0)
#define WJ_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc) return _rc;        \
  } while (0)

// ------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------
static int run_encoder(wj_whisper* m, const float* mel, int B, int n_layers, float* enc_out, hipStream_t s, int win0 = 0) {
  const wj_whisper_dims& d = m->d;
  const int D = d.n_audio_state, H = d.n_audio_head, C = d.n_mels, T = d.n_audio_ctx, F = m->frames;
  const int dt = m->dtype;
  m->ctx->prof_units = B;
  PROF(PT_MEL_ROWS, launch_mel_to_rows(dt, mel, m->mel_rows, B, C, F, s));
  {  // conv1 (k=3, pad 1) as a GEMM over the overlapping [t-1, t, t+1] row window + GELU
    GemmArgs g;
    g.A = m->mel_rows; g.lda = C; g.a_batch = (int64_t)(F + 2) * C;
    g.W = m->W(WJ_T_ENC_CONV1_W); g.ldw = 3 * C; g.bias = m->F(WJ_T_ENC_CONV1_B);
    g.M = F; g.N = D; g.K = 3 * C; g.nbatch = B;
    g.out = m->at(m->conv1_out, D); g.ldc = D; g.c_batch = (int64_t)(F + 2) * D;
    PROF(PT_CONV1, launch_gemm(dt, EPI_GELU_T, g, s, 1));
  }
  {  // conv2 (k=3, stride 2, pad 1) + GELU + positional embedding -> fp32 residual stream
    GemmArgs g;
    g.A = m->conv1_out; g.lda = 2 * D; g.a_batch = (int64_t)(F + 2) * D;
    g.W = m->W(WJ_T_ENC_CONV2_W); g.ldw = 3 * D; g.bias = m->F(WJ_T_ENC_CONV2_B);
    g.M = T; g.N = D; g.K = 3 * D; g.nbatch = B;
    g.out = m->x; g.ldc = D; g.c_batch = (int64_t)T * D; g.pos = m->F(WJ_T_ENC_POS);
    PROF(PT_CONV2, launch_gemm(dt, EPI_GELU_POS_F32, g, s, 1));
  }
  const int L = n_layers < 0 ? d.n_audio_layer : n_layers;
  const int M = B * T;
  const int blk = m->blk ? 1 : 0;   // blocked operands: flat rows (the head-split epilogues take the window from the row index)
  for (int l = 0; l < L; ++l) {
    const int b0 = m->enc_base(l);
    PROF(PT_E_LN, launch_layernorm(dt, m->x, m->F(b0 + WJ_TE_LN1_W), m->F(b0 + WJ_TE_LN1_B), m->h, M, D, s, 0, blk));
    {
      GemmArgs g;
      g.A = m->h; g.lda = D; g.a_batch = (int64_t)T * D;
      g.W = m->W(b0 + WJ_TE_QKV_W); g.ldw = D; g.bias = m->F(b0 + WJ_TE_QKV_B);
      g.M = T; g.N = 2 * D; g.K = D; g.nbatch = B;
      g.out = m->q; g.out2 = m->k; g.D = D; g.H = H; g.Tpad = m->Tpad;
      if (blk) { g.blk = 1; g.seq_T = T; g.M = M; g.nbatch = 1; g.a_batch = 0; g.W = m->WB(b0 + WJ_TE_QKV_W); }
      PROF(PT_E_QK, launch_gemm(dt, EPI_QK_HEADS, g, s, 1));
      GemmArgs v = g;
      v.W = blk ? m->WB(b0 + WJ_TE_QKV_W, 2 * D, D)
                : reinterpret_cast<const char*>(m->W(b0 + WJ_TE_QKV_W)) + (int64_t)2 * D * D * m->esz;
      v.bias = m->F(b0 + WJ_TE_QKV_B) + 2 * D;
      v.N = D; v.out = m->vt; v.out2 = nullptr;
      PROF(PT_E_V, launch_gemm(dt, EPI_VT, v, s, 1));
    }
    PROF(PT_E_ATTN, launch_attention_enc(dt, m->q, m->k, m->vt, m->attn, B, T, m->Tpad, H, s, blk));
    {
      GemmArgs g;
      g.A = m->attn; g.lda = D; g.W = m->W(b0 + WJ_TE_OUT_W); g.ldw = D; g.bias = m->F(b0 + WJ_TE_OUT_B);
      g.M = M; g.N = D; g.K = D; g.out = m->x; g.ldc = D;
      if (blk) { g.blk = 1; g.W = m->WB(b0 + WJ_TE_OUT_W); }
      PROF(PT_E_OUT, launch_gemm(dt, EPI_RESID_F32, g, s, 1));
    }
    PROF(PT_E_LN, launch_layernorm(dt, m->x, m->F(b0 + WJ_TE_LN2_W), m->F(b0 + WJ_TE_LN2_B), m->h, M, D, s, 0, blk));
    {
      GemmArgs g;
      g.A = m->h; g.lda = D; g.W = m->W(b0 + WJ_TE_FC1_W); g.ldw = D; g.bias = m->F(b0 + WJ_TE_FC1_B);
      g.M = M; g.N = 4 * D; g.K = D; g.out = m->ff; g.ldc = 4 * D;
      if (blk) { g.blk = 1; g.out_blk = 1; g.W = m->WB(b0 + WJ_TE_FC1_W); }
      PROF(PT_E_FC1, launch_gemm(dt, EPI_GELU_T, g, s, 1));
      GemmArgs g2;
      g2.A = m->ff; g2.lda = 4 * D; g2.W = m->W(b0 + WJ_TE_FC2_W); g2.ldw = 4 * D; g2.bias = m->F(b0 + WJ_TE_FC2_B);
      g2.M = M; g2.N = D; g2.K = 4 * D; g2.out = m->x; g2.ldc = D;
      if (blk) { g2.blk = 1; g2.W = m->WB(b0 + WJ_TE_FC2_W); }
      PROF(PT_E_FC2, launch_gemm(dt, EPI_RESID_F32, g2, s, 1));
    }
  }
  if (n_layers >= 0) {  // bisection mode: raw residual stream, no final norm, no cross K/V
    if (enc_out) WJ_HIP(hipMemcpyAsync(enc_out, m->x, sizeof(float) * (size_t)M * D, hipMemcpyDeviceToDevice, s));
    return WJ_OK;
  }
  PROF(PT_E_LN, launch_layernorm(dt, m->x, m->F(WJ_T_ENC_LNPOST_W), m->F(WJ_T_ENC_LNPOST_B), m->h, M, D, s, 0, blk));
  if (enc_out)
    WJ_TRY(launch_layernorm(WJ_F32, m->x, m->F(WJ_T_ENC_LNPOST_W), m->F(WJ_T_ENC_LNPOST_B), enc_out, M, D, s));
  // cross-attention K/V of every decoder layer, computed once per window and kept in HBM
  for (int l = 0; l < d.n_text_layer; ++l) {
    const int b0 = m->dec_base(l);
    GemmArgs g;
    g.A = m->h; g.lda = D; g.a_batch = (int64_t)T * D;
    g.W = m->W(b0 + WJ_TD_CKV_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_CKV_B);
    g.M = T; g.N = 2 * D; g.K = D; g.nbatch = B;
    if (blk) { g.blk = 1; g.seq_T = T; g.M = M; g.nbatch = 1; g.a_batch = 0; g.W = m->WB(b0 + WJ_TD_CKV_W); }
    // the slice's windows land at window slots win0 .. win0 + B - 1 of the resident cross K/V
    const int64_t kwin = (int64_t)d.n_text_head * T * 64, vwin = (int64_t)d.n_text_head * 64 * (m->cross_tpad > 0 ? m->cross_tpad : T);
    g.out = m->at(m->cross_k, l * m->cross_layer_elems() + win0 * kwin);
    g.out2 = m->at(m->cross_v, l * m->cross_v_layer_elems() + win0 * vwin);
    g.D = D; g.H = d.n_text_head; g.Tpad = T;
    if (m->cross_tpad > 0) {   // K head-split as stored; V transposed per head for the MFMA cross attention
      GemmArgs v = g;
      g.N = D; g.out2 = nullptr;
      PROF(PT_E_CKV, launch_gemm(dt, EPI_QK_HEADS, g, s, 1));
      v.W = blk ? m->WB(b0 + WJ_TD_CKV_W, D, D) : reinterpret_cast<const char*>(m->W(b0 + WJ_TD_CKV_W)) + (int64_t)D * D * m->esz;
      v.bias = m->F(b0 + WJ_TD_CKV_B) + D;
      v.N = D; v.out = m->at(m->cross_v, l * m->cross_v_layer_elems() + win0 * vwin); v.out2 = nullptr;
      v.Tpad = m->cross_tpad;
      PROF(PT_E_CKV, launch_gemm(dt, EPI_VT, v, s, 1));
      continue;
    }
    PROF(PT_E_CKV, launch_gemm(dt, EPI_CKV, g, s, 1));
  }
  return WJ_OK;
}

// ------------------------------------------------------------------------------------------------
// decoder: one token per row
// ------------------------------------------------------------------------------------------------
// Rows [row0, row0 + R) (windows [row0 / beam, ...)) advance by one token.  Rows are independent, so the
// greedy loop runs several disjoint row slices ("chains") concurrently on forked streams: each of the
// ~355 kernels of a step is latency bound and fills at most a third of the chip, two chains overlap.
static int run_decoder_step(wj_whisper* m, int row0, int R, int n_windows, int beam, bool want_logits, hipStream_t s,
                            int* pos = nullptr) {
  if (!pos) pos = m->pos;
  const wj_whisper_dims& d = m->d;
  const int D = d.n_text_state, H = d.n_text_head, dt = m->dtype;
  m->ctx->prof_units = n_windows;
  const int win0 = row0 / beam;
  // activation rows [hi | lo]: `split` for the GEMMs that write the residual stream and the logits (fed by the attention
  // outputs, the GELU output and the final LayerNorm), `psplit` for the projections fed by the per-layer LayerNorms.
  // Measured on the oracle (profiles/r02_precision_ablation_cpu.json + DESIGN.md): splitting only the former leaves
  // 4.4e-4 of per-token log-prob error against 3.5e-4 for all of them, at 43 % instead of 100 % more MFMA work.
  const int split = m->split_act ? 1 : 0, sm = split ? 2 : 1;
  const int psplit = m->split_all ? 1 : 0, psm = psplit ? 2 : 1;
  float* dx = m->dx + (int64_t)row0 * D;
  void* dh = m->at(m->dh, (int64_t)row0 * D * sm);
  void* dq = m->at(m->dq, (int64_t)row0 * D);
  void* dattn = m->at(m->dattn, (int64_t)row0 * D * sm);
  void* dff = m->at(m->dff, (int64_t)row0 * 4 * D * sm);
  const int64_t self_row = (int64_t)H * m->kv_len * 64;         // cache elements per row
  const int64_t cross_win = (int64_t)H * d.n_audio_ctx * 64;    // cross K (or V) elements per window
  // Residual-writing GEMMs (attention out-projections, fc2) can run split-K: each K slice writes a raw fp32
  // slab and the LayerNorm that always follows folds  x += bias + sum(slabs)  in a fixed order (deterministic).
  // This keeps the per-workgroup A traffic at M*K/ksplit and multiplies the number of workgroups streaming W.
  const bool inv = g_tune.batch_invariant && is16(dt);       // see Tunables::batch_invariant
  const bool rows = !inv && is16(dt) && g_tune.dec_rows && R <= g_tune.dec_rows_max_m;
  const int ks_attn = rows ? g_tune.dec_rows_ks_attn : g_tune.dec_ks_attn;
  const int ks_fc2 = rows ? g_tune.dec_rows_ks_fc2 : g_tune.dec_ks_fc2, tile_min_m = inv ? 1 : g_tune.dec_tile_min_m;
  const int proj_min_m = inv ? 1 : g_tune.dec_proj_min_m;
  const int ms = g_tune.dec_ms_stages;
  const int tile_variant = (ms >= 3 && ms <= 5) ? 70 + ms : (g_tune.dec_tile_reg ? 4 : 3);
  const int msr = g_tune.dec_ms_resid;
  const int resid_variant = (msr >= 3 && msr <= 5) ? 70 + msr : tile_variant;
  float* slab = m->partial + (int64_t)row0 * kDecKsMax * D;
  int pend_ks = 0;
  const float* pend_bias = nullptr;
  // Split-K exists to put enough workgroups on the chip when there are few rows; every K slice costs an fp32 slab
  // round trip (and, for the projections, a reduce launch).  With many rows (beam search over a full batch: 1920
  // rows) the row tiles fill the CUs on their own, so the factor is halved while >= 256 workgroups remain -- down to
  // 1, where the GEMM applies its epilogue directly (no slab, no reduce launch).
  const bool tiled = !rows && is16(dt) && tile_min_m > 0 && R >= tile_min_m;
  auto adapt_ks = [&](int ks, int N) -> int {
    if (!tiled || !g_tune.dec_adapt_ks || inv) return ks;
    const int tiles = ceil_div(R, 128) * ceil_div(N, 128);
    while (ks > 1 && tiles * (ks / 2) >= g_tune.dec_adapt_wgs) ks /= 2;
    return ks;
  };
  auto resid_gemm = [&](int tag, const void* A, int K, const void* W, const float* bias, int ks) -> int {
    GemmArgs g;
    g.A = A; g.lda = (int64_t)K * sm; g.split = split; g.W = W; g.ldw = K; g.M = R; g.N = D; g.K = K; g.ldc = D;
    ks = adapt_ks(ks, D);
    if (tiled && ks == 1) {      // enough row tiles: x += acc + bias straight from the tile kernel
      g.bias = bias; g.out = dx;
      PROF(tag, launch_gemm(dt, EPI_RESID_F32, g, s, resid_variant));
      return WJ_OK;
    }
    if (is16(dt) && ks > 1 && ks <= kDecKsMax && K % (64 * ks) == 0) {
      g.out = slab; g.ksplit = ks;
      pend_ks = ks; pend_bias = bias;
      const int variant = rows ? 5 : ((tile_min_m > 0 && R >= tile_min_m) ? resid_variant : 2);
      PROF(tag, launch_gemm(dt, EPI_PARTIAL_F32, g, s, variant));
    } else {
      g.bias = bias; g.out = dx;
      PROF(tag, launch_gemm(dt, EPI_RESID_F32, g, s, rows ? 5 : (inv ? 3 : 0)));      // batch_invariant: never the dispatcher's M-dependent choice
    }
    return WJ_OK;
  };
  // projections with a scatter / activation epilogue: single pass, or (many rows) split-K slabs from the tile
  // kernel + one reduce kernel that applies the very same epilogue
  // `deferred` (may be NULL): the caller's next kernel consumes the raw K-slices itself (attention kernels), so the
  // reduce launch is skipped and *deferred = number of slices left in `slab`
  auto proj_gemm = [&](int tag, Epi epi, GemmArgs& g, int* deferred = nullptr) -> int {
    const int ks = adapt_ks(g_tune.dec_ks_proj, g.N);
    if (deferred) *deferred = 0;
    if (rows) {
      PROF(tag, launch_gemm(dt, epi, g, s, 5));
    } else if (tiled && ks == 1 && R >= proj_min_m) {
      // single pass with the projection's own epilogue.  Wide projections of a big batch (beam search over hundreds of
      // windows: 1920 rows x 3840 / 5120 columns) go to the 256x256 encoder kernel: the 128-tile kernel gives a
      // workgroup only 32 MFMAs per wave per k-step to hide the LDS-DMA round trip behind (measured 440 TFLOP/s on fc1)
      const bool wide = g_tune.dec_big_min_m > 0 && R >= g_tune.dec_big_min_m && !g.split && (g.N % 256) == 0 &&
                        ceil_div(R, 256) * (g.N / 256) >= 96;
      PROF(tag, launch_gemm(dt, epi, g, s, wide ? 6 : tile_variant));
    } else if (is16(dt) && ks > 1 && R >= proj_min_m && (int64_t)ks * g.N <= (int64_t)kDecKsMax * D &&
        g.K % (64 * ks) == 0 && !pend_ks) {
      GemmArgs p = g;
      p.out = slab; p.ldc = g.N; p.ksplit = ks; p.bias = nullptr; p.split_out = 0;
      PROF(tag, launch_gemm(dt, EPI_PARTIAL_F32, p, s, tile_variant));
      if (deferred && g_tune.dec_fuse_reduce) *deferred = ks;
      else PROF(tag, launch_splitk_reduce(dt, epi, g, slab, ks, s));
    } else {
      PROF(tag, launch_gemm(dt, epi, g, s, inv ? 3 : 0));
    }
    return WJ_OK;
  };
  auto norm = [&](const float* w, const float* b, int out_split) -> int {
    if (pend_ks) {
      const int ks = pend_ks;
      pend_ks = 0;
      PROF(PT_D_LN, launch_layernorm_resid(dt, dx, slab, ks, pend_bias, w, b, dh, R, D, s, out_split));
    } else {
      PROF(PT_D_LN, launch_layernorm(dt, dx, w, b, dh, R, D, s, out_split));
    }
    return WJ_OK;
  };
  PROF(PT_D_EMBED, launch_embed(dt, m->W(WJ_T_DEC_TOK_EMB), m->F(WJ_T_DEC_POS), m->tokens + (int64_t)row0 * m->tok_stride,
                                m->tok_stride, pos, dx, R, D, s));
  for (int l = 0; l < d.n_text_layer; ++l) {
    const int b0 = m->dec_base(l);
    int qkv_slices = 0, cq_slices = 0;
    const float *qkv_bias = nullptr, *cq_bias = nullptr;
    void* sk = m->at(m->self_k, l * m->self_layer_elems());
    void* sv = m->at(m->self_v, l * m->self_layer_elems());
    WJ_TRY(norm(m->F(b0 + WJ_TD_LN1_W), m->F(b0 + WJ_TD_LN1_B), psplit));
    {
      GemmArgs g;
      g.A = dh; g.lda = (int64_t)D * psm; g.split = psplit; g.W = m->W(b0 + WJ_TD_QKV_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_QKV_B);
      g.M = R; g.N = 3 * D; g.K = D; g.out = dq;
      g.out2 = m->at(sk, row0 * self_row); g.out3 = m->at(sv, row0 * self_row);
      g.D = D; g.H = H; g.pos_ptr = pos; g.cache_len = m->kv_len;
      WJ_TRY(proj_gemm(PT_D_QKV, EPI_QKV_DEC, g, &qkv_slices));
      qkv_bias = g.bias;
    }
    {
      DecAttnArgs a;   // K/V bases stay absolute: the row map holds absolute physical rows
      a.q = dq; a.K = sk; a.V = sv; a.out = dattn; a.out_split = split; a.G = R; a.nb = 1; a.H = H;
      a.n_keys_ptr = pos; a.kv_stride = m->kv_len;
      a.row_map = m->row_map[m->cur_map] + (int64_t)row0 * m->kv_len;      // the map's row stride IS the cache's (kernel contract)
      if (qkv_slices) {   // the attention kernel sums the K-slices, appends k/v to the cache and attends
        a.slab = slab; a.slab_bias = qkv_bias; a.slab_ks = qkv_slices; a.slab_rows = R; a.slab_ld = 3 * D; a.row_base = row0;
      }
      PROF(PT_D_SELF, launch_attention_dec(dt, a, s));
    }
    WJ_TRY(resid_gemm(PT_D_OUT, dattn, D, m->W(b0 + WJ_TD_OUT_W), m->F(b0 + WJ_TD_OUT_B), ks_attn));
    WJ_TRY(norm(m->F(b0 + WJ_TD_LNX_W), m->F(b0 + WJ_TD_LNX_B), psplit));
    {
      GemmArgs g;
      g.A = dh; g.lda = (int64_t)D * psm; g.split = psplit; g.W = m->W(b0 + WJ_TD_CQ_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_CQ_B);
      g.M = R; g.N = D; g.K = D; g.out = dq; g.ldc = D;
      WJ_TRY(proj_gemm(PT_D_CQ, EPI_T, g, m->cross_tpad > 0 ? &cq_slices : nullptr));
      cq_bias = g.bias;
    }
    {
      DecAttnArgs a;
      a.q = dq;
      if (cq_slices) { a.slab = slab; a.slab_bias = cq_bias; a.slab_ks = cq_slices; a.slab_rows = R; a.slab_ld = D; }
      if (m->dump_qk) {
        a.dump = m->dump_qk; a.dump_sel = m->align_sel + (int64_t)l * H; a.dump_pos_ptr = pos;
        a.dump_nsel = m->dump_nsel; a.dump_tmax = m->dump_tmax; a.dump_row_base = row0;
      }
      const int64_t woff = m->use_slots ? 0 : win0 * cross_win;   // with a slot map the K/V base stays absolute
      a.K = m->at(m->cross_k, l * m->cross_layer_elems() + woff);
      if (m->cross_tpad > 0) {
        const int64_t vwin = (int64_t)H * 64 * m->cross_tpad;
        a.V = m->at(m->cross_v, l * m->cross_v_layer_elems() + (m->use_slots ? 0 : win0 * vwin));
        a.vt_stride = m->cross_tpad;
      } else {
        a.V = m->at(m->cross_v, l * m->cross_layer_elems() + woff);
      }
      a.group_of = m->use_slots ? m->slot_map + win0 : nullptr;
      a.out = dattn; a.out_split = split; a.G = n_windows; a.nb = beam; a.H = H; a.n_keys = d.n_audio_ctx; a.kv_stride = d.n_audio_ctx;
      PROF(PT_D_CROSS, launch_attention_dec(dt, a, s));
    }
    WJ_TRY(resid_gemm(PT_D_COUT, dattn, D, m->W(b0 + WJ_TD_COUT_W), m->F(b0 + WJ_TD_COUT_B), ks_attn));
    WJ_TRY(norm(m->F(b0 + WJ_TD_LN2_W), m->F(b0 + WJ_TD_LN2_B), psplit));
    {
      GemmArgs g;
      g.A = dh; g.lda = (int64_t)D * psm; g.split = psplit; g.W = m->W(b0 + WJ_TD_FC1_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_FC1_B);
      g.M = R; g.N = 4 * D; g.K = D; g.out = dff; g.ldc = (int64_t)4 * D * sm; g.split_out = split;
      WJ_TRY(proj_gemm(PT_D_FC1, EPI_GELU_T, g));
    }
    WJ_TRY(resid_gemm(PT_D_FC2, dff, 4 * D, m->W(b0 + WJ_TD_FC2_W), m->F(b0 + WJ_TD_FC2_B), ks_fc2));
  }
  if (want_logits) {
    WJ_TRY(norm(m->F(WJ_T_DEC_LN_W), m->F(WJ_T_DEC_LN_B), split));
    GemmArgs g;
    g.A = dh; g.lda = (int64_t)D * sm; g.split = split; g.W = m->W(WJ_T_DEC_TOK_EMB); g.ldw = D;
    g.M = R; g.N = d.n_vocab; g.K = D; g.out = m->logits + (int64_t)row0 * m->ldl; g.ldc = m->ldl;
    const bool big_m = is16(dt) && tile_min_m > 0 && R >= tile_min_m;
    PROF(PT_D_LOGITS, launch_gemm(dt, EPI_F32, g, s, rows ? 5 : (big_m ? tile_variant : 0)));
  }
  return WJ_OK;
}

// ------------------------------------------------------------------------------------------------
// decoder: full-sequence teacher-forced pass (word-timestamp alignment).  Rows = windows x Tp positions, run
// through the layers as ONE batch: the decoder weights are streamed once instead of once per token and a
// window's cross K/V is read Tp/16 times instead of Tp times.  Activations live in the encoder workspaces
// (x, h, q, attn, ff: sized for batch x 1500 rows), k/v go to the self-attention cache rows of the windows.
// ------------------------------------------------------------------------------------------------
static int run_decoder_seq(wj_whisper* m, int B, int Tp, int n0, const int32_t* d_ntok, float* d_prob, int eot,
                           const int32_t* d_groups, int nb, hipStream_t s) {
  const wj_whisper_dims& d = m->d;
  const int D = d.n_text_state, H = d.n_text_head, dt = m->dtype;
  const int M = B * Tp;
  const int gv = (g_tune.batch_invariant && is16(dt)) ? 3 : 0;     // batch_invariant: the tile kernel whatever M (the dispatcher's choice depends on it)
  float* x = m->x;
  void *h = m->h, *q = m->q, *attn = m->attn, *ff = m->ff;
  WJ_TRY(launch_embed_seq(dt, m->W(WJ_T_DEC_TOK_EMB), m->F(WJ_T_DEC_POS), m->tokens, m->tok_stride, Tp, x, B, D, s));
  for (int l = 0; l < d.n_text_layer; ++l) {
    const int b0 = m->dec_base(l);
    void* sk = m->at(m->self_k, l * m->self_layer_elems());
    void* sv = m->at(m->self_v, l * m->self_layer_elems());
    WJ_TRY(launch_layernorm(dt, x, m->F(b0 + WJ_TD_LN1_W), m->F(b0 + WJ_TD_LN1_B), h, M, D, s));
    {
      GemmArgs g;
      g.A = h; g.lda = D; g.W = m->W(b0 + WJ_TD_QKV_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_QKV_B);
      g.M = M; g.N = 3 * D; g.K = D; g.out = q; g.out2 = sk; g.out3 = sv;
      g.D = D; g.H = H; g.cache_len = m->kv_len; g.seq_tp = Tp;
      WJ_TRY(launch_gemm(dt, EPI_QKV_DEC, g, s, gv));
    }
    {
      DecAttnArgs a;
      a.q = q; a.K = sk; a.V = sv; a.out = attn; a.G = M; a.nb = 1; a.H = H; a.kv_stride = m->kv_len; a.seq_tp = Tp;
      WJ_TRY(launch_attention_dec(dt, a, s));
    }
    {
      GemmArgs g;
      g.A = attn; g.lda = D; g.W = m->W(b0 + WJ_TD_OUT_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_OUT_B);
      g.M = M; g.N = D; g.K = D; g.out = x; g.ldc = D;
      WJ_TRY(launch_gemm(dt, EPI_RESID_F32, g, s, gv));
    }
    WJ_TRY(launch_layernorm(dt, x, m->F(b0 + WJ_TD_LNX_W), m->F(b0 + WJ_TD_LNX_B), h, M, D, s));
    {
      GemmArgs g;
      g.A = h; g.lda = D; g.W = m->W(b0 + WJ_TD_CQ_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_CQ_B);
      g.M = M; g.N = D; g.K = D; g.out = q; g.ldc = D;
      WJ_TRY(launch_gemm(dt, EPI_T, g, s, gv));
    }
    {
      DecAttnArgs a;
      a.q = q; a.out = attn; a.G = M / nb; a.nb = nb; a.H = H; a.n_keys = d.n_audio_ctx; a.kv_stride = d.n_audio_ctx;
      a.K = m->at(m->cross_k, l * m->cross_layer_elems());
      a.V = m->at(m->cross_v, l * m->cross_v_layer_elems());
      a.vt_stride = m->cross_tpad;
      a.group_of = d_groups;
      a.dump = m->dump_qk; a.dump_sel = m->align_sel + (int64_t)l * H; a.dump_nsel = m->dump_nsel; a.dump_tmax = Tp;
      a.dump_chunks = Tp / nb;
      WJ_TRY(launch_attention_dec(dt, a, s));
    }
    {
      GemmArgs g;
      g.A = attn; g.lda = D; g.W = m->W(b0 + WJ_TD_COUT_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_COUT_B);
      g.M = M; g.N = D; g.K = D; g.out = x; g.ldc = D;
      WJ_TRY(launch_gemm(dt, EPI_RESID_F32, g, s, gv));
    }
    WJ_TRY(launch_layernorm(dt, x, m->F(b0 + WJ_TD_LN2_W), m->F(b0 + WJ_TD_LN2_B), h, M, D, s));
    {
      GemmArgs g;
      g.A = h; g.lda = D; g.W = m->W(b0 + WJ_TD_FC1_W); g.ldw = D; g.bias = m->F(b0 + WJ_TD_FC1_B);
      g.M = M; g.N = 4 * D; g.K = D; g.out = ff; g.ldc = 4 * D;
      WJ_TRY(launch_gemm(dt, EPI_GELU_T, g, s, gv));
      GemmArgs g2;
      g2.A = ff; g2.lda = 4 * D; g2.W = m->W(b0 + WJ_TD_FC2_W); g2.ldw = 4 * D; g2.bias = m->F(b0 + WJ_TD_FC2_B);
      g2.M = M; g2.N = D; g2.K = 4 * D; g2.out = x; g2.ldc = D;
      WJ_TRY(launch_gemm(dt, EPI_RESID_F32, g2, s, gv));
    }
  }
  WJ_TRY(launch_layernorm(dt, x, m->F(WJ_T_DEC_LN_W), m->F(WJ_T_DEC_LN_B), h, M, D, s));
  // text-token probabilities: logits in chunks of max_rows rows (the logits buffer is sized for decode steps)
  for (int r0 = 0; r0 < M; r0 += m->max_rows) {
    const int rows = min(m->max_rows, M - r0);
    GemmArgs g;
    g.A = m->at(h, (int64_t)r0 * D); g.lda = D; g.W = m->W(WJ_T_DEC_TOK_EMB); g.ldw = D;
    g.M = rows; g.N = d.n_vocab; g.K = D; g.out = m->logits; g.ldc = m->ldl;
    WJ_TRY(launch_gemm(dt, EPI_F32, g, s, gv));
    WJ_TRY(launch_align_token_prob_seq(m->logits, m->ldl, eot, m->tokens, m->tok_stride, r0, rows, Tp, n0, d_ntok, d_prob, s));
  }
  return WJ_OK;
}

template <typename T>
__global__ void split_rows_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t M, int K) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M * K; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / K;
    const int k = (int)(i % K);
    st_split<T>(out + m * 2 * K + k, K, in[i]);
  }
}

__global__ void init_rows_kernel(int32_t* map0, int32_t* map1, int R, int stride) {
  const int r = blockIdx.x;
  for (int j = threadIdx.x; j < stride; j += 256) {
    map0[(int64_t)r * stride + j] = r;
    map1[(int64_t)r * stride + j] = r;
  }
}
__global__ void put_tokens_kernel(int32_t* tokens, int64_t stride, const int* pos_ptr, const int32_t* src, int R) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < R) tokens[(int64_t)r * stride + *pos_ptr] = src[r];
}

static int reset_decode_state(wj_whisper* m, int R, hipStream_t s) {
  WJ_HIP(hipMemsetAsync(m->pos, 0, sizeof(int), s));
  WJ_HIP(hipMemsetAsync(m->sum_lp, 0, sizeof(float) * R, s));
  WJ_HIP(hipMemsetAsync(m->finished, 0, sizeof(int32_t) * R, s));
  WJ_HIP(hipMemsetAsync(m->tok_lp, 0, sizeof(float) * (size_t)R * m->tok_stride, s));
  hipLaunchKernelGGL(init_rows_kernel, dim3(R), dim3(256), 0, s, m->row_map[0], m->row_map[1], R, m->kv_len);
  WJ_LAUNCH_CHECK();
  m->cur_map = 0;
  return WJ_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int wj_abi_version(void) { return WJ_ABI_VERSION; }
const char* wj_last_error(void) { return wj::get_error(); }

int wj_init(int device_ordinal, wj_ctx** out) {
  WJ_REQUIRE(out != nullptr, "wj_init: out is NULL");
  int n = 0;
  WJ_HIP(hipGetDeviceCount(&n));
  WJ_REQUIRE(device_ordinal >= 0 && device_ordinal < n, "wj_init: device %d not present (%d visible)", device_ordinal, n);
  WJ_HIP(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  WJ_HIP(hipGetDeviceProperties(&prop, device_ordinal));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("wj_init: libwjhip is built for gfx950 (MI355X) only; device %d is %s", device_ordinal, prop.gcnArchName);
    return WJ_E_UNSUPPORTED;
  }
  wj_ctx* c = new wj_ctx();
  c->device = device_ordinal;
  c->cu_count = prop.multiProcessorCount;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete c;
    return WJ_E_HIP;
  }
  *out = c;
  return WJ_OK;
}

int wj_shutdown(wj_ctx* ctx) {
  if (!ctx) return WJ_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return WJ_OK;
}

int wj_sync(wj_ctx* ctx) {
  WJ_REQUIRE(ctx != nullptr, "wj_sync: ctx is NULL");
  WJ_HIP(hipStreamSynchronize(ctx->stream));
  return WJ_OK;
}

int wj_stream_create(wj_ctx* ctx, int cu_first, int cu_count, void** out) {
  WJ_REQUIRE(ctx && out, "wj_stream_create: NULL argument");
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = nullptr;
  if (cu_count <= 0) {
    WJ_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  } else {
    WJ_REQUIRE(cu_first >= 0 && cu_first + cu_count <= ctx->cu_count, "wj_stream_create: CUs %d..%d outside the device's %d", cu_first,
               cu_first + cu_count - 1, ctx->cu_count);
    const int words = (ctx->cu_count + 31) / 32;
    std::vector<uint32_t> mask(words, 0u);
    for (int c = cu_first; c < cu_first + cu_count; ++c) mask[c >> 5] |= 1u << (c & 31);
    WJ_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()));
  }
  *out = s;
  return WJ_OK;
}

int wj_stream_sync(wj_ctx* ctx, void* stream) {
  WJ_REQUIRE(ctx && stream, "wj_stream_sync: NULL argument");
  WJ_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  return WJ_OK;
}

int wj_stream_destroy(wj_ctx* ctx, void* stream) {
  if (!ctx || !stream) return WJ_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
  return WJ_OK;
}

int wj_device_info(wj_ctx* ctx, int64_t out[4]) {
  WJ_REQUIRE(ctx && out, "wj_device_info: NULL argument");
  hipDeviceProp_t prop;
  WJ_HIP(hipGetDeviceProperties(&prop, ctx->device));
  out[0] = prop.multiProcessorCount;
  out[1] = prop.clockRate;
  out[2] = (int64_t)(prop.totalGlobalMem & 0xffffffffull);
  out[3] = (int64_t)(prop.totalGlobalMem >> 32);
  return WJ_OK;
}

int wj_profile_start(wj_ctx* ctx) {
  WJ_REQUIRE(ctx != nullptr, "wj_profile_start: ctx is NULL");
  if (!ctx->prof) ctx->prof = new wj_profiler();
  return WJ_OK;
}

int wj_profile_tags(void) { return PT_COUNT; }
const char* wj_profile_tag_name(int tag) { return (tag >= 0 && tag < PT_COUNT) ? kProfNames[tag] : ""; }

int wj_profile_stop(wj_ctx* ctx, double* total_ms, int64_t* counts, int n_tags) {
  return wj_profile_stop_ex(ctx, total_ms, counts, nullptr, n_tags);
}

int wj_profile_stop_ex(wj_ctx* ctx, double* total_ms, int64_t* counts, int64_t* units, int n_tags) {
  WJ_REQUIRE(ctx && total_ms && counts && n_tags >= PT_COUNT, "wj_profile_stop: bad arguments");
  wj_profiler* p = ctx->prof;
  WJ_REQUIRE(p != nullptr, "wj_profile_stop: profiler not started");
  WJ_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n_tags; ++i) { total_ms[i] = 0.0; counts[i] = 0; if (units) units[i] = 0; }
  for (auto& pr : p->pairs) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.b) == hipSuccess && hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
      total_ms[pr.tag] += ms;
      counts[pr.tag] += 1;
      if (units) units[pr.tag] += pr.units;
    }
    (void)hipEventDestroy(pr.a);
    (void)hipEventDestroy(pr.b);
  }
  delete p;
  ctx->prof = nullptr;
  return WJ_OK;
}

int wj_tune(const char* key, int value) {
  WJ_REQUIRE(key != nullptr, "wj_tune: NULL key");
  if (!strcmp(key, "dec_ks_attn")) g_tune.dec_ks_attn = value;
  else if (!strcmp(key, "dec_ks_fc2")) g_tune.dec_ks_fc2 = value;
  else if (!strcmp(key, "dec_tile_min_m")) g_tune.dec_tile_min_m = value;
  else if (!strcmp(key, "decode_chains")) g_tune.decode_chains = value;
  else if (!strcmp(key, "attn_enc_variant")) wj::g_attn_enc_variant = value;
  else if (!strcmp(key, "dec_ks_proj")) g_tune.dec_ks_proj = value;
  else if (!strcmp(key, "dec_proj_min_m")) g_tune.dec_proj_min_m = value;
  else if (!strcmp(key, "dec_cross_mfma")) g_tune.dec_cross_mfma = value;
  else if (!strcmp(key, "dec_split_act")) g_tune.dec_split_act = value;
  else if (!strcmp(key, "dec_adapt_ks")) g_tune.dec_adapt_ks = value;
  else if (!strcmp(key, "dec_adapt_wgs")) g_tune.dec_adapt_wgs = value > 0 ? value : 256;
  else if (!strcmp(key, "dec_big_min_m")) g_tune.dec_big_min_m = value;
  else if (!strcmp(key, "batch_invariant")) g_tune.batch_invariant = value;
  else if (!strcmp(key, "self_kv_len")) g_tune.self_kv_len = value;
  else if (!strcmp(key, "enc_batch")) g_tune.enc_batch = value;
  else if (!strcmp(key, "enc_blocked")) g_tune.enc_blocked = value;
  else if (!strcmp(key, "ln_vec")) wj::g_ln_vec = value;
  else if (!strcmp(key, "beam_compact")) g_tune.beam_compact = value;
  else if (!strcmp(key, "beam_token_logprobs")) g_tune.beam_token_logprobs = value;
  else if (!strcmp(key, "beam_poll")) g_tune.beam_poll = value;
  else if (!strcmp(key, "beam_compact_pct")) g_tune.beam_compact_pct = value;
  else if (!strcmp(key, "beam_compact_min")) g_tune.beam_compact_min = value;
  else if (!strcmp(key, "beam_topk_reg")) g_beam_topk_reg = value;
  else if (!strcmp(key, "gemm_big")) g_gemm_big = value;
  else if (!strcmp(key, "tile_l2_kb")) g_tile_l2_kb = value;
  else if (!strcmp(key, "ppb_ns")) g_ppb_ns = value;
  else if (!strcmp(key, "qwen_split_act")) wj::g_qwen_split_act = value;
  else if (!strcmp(key, "qwen_compact_pct")) wj::g_qwen_compact_pct = value;
  else if (!strcmp(key, "qwen_prompt_mfma")) wj::g_qwen_prompt_mfma = value;
  else if (!strcmp(key, "qwen_splitk")) wj::g_qwen_splitk = value;
  else if (!strcmp(key, "qwen_conv_kpad")) wj::g_qwen_conv_kpad = value;
  else if (!strcmp(key, "qwen_tower_split")) wj::g_qwen_tower_split = value;
  else if (!strcmp(key, "qwen_fuse_swiglu")) wj::g_qwen_fuse_swiglu = value;
  else if (!strcmp(key, "ppb_gm")) g_ppb_gm = value;
  else if (!strcmp(key, "epi_wide")) g_epi_wide = value;
  else if (!strcmp(key, "dec_ms_stages")) g_tune.dec_ms_stages = value;
  else if (!strcmp(key, "dec_ms_resid")) g_tune.dec_ms_resid = value;
  else if (!strcmp(key, "dec_tile_reg")) g_tune.dec_tile_reg = value;
  else if (!strcmp(key, "dec_fuse_reduce")) g_tune.dec_fuse_reduce = value;
  else if (!strcmp(key, "align_prefill")) g_tune.align_prefill = value;
  else if (!strcmp(key, "dec_rows")) g_tune.dec_rows = value;
  else if (!strcmp(key, "dec_rows_max_m")) g_tune.dec_rows_max_m = value;
  else if (!strcmp(key, "dec_rows_ks_attn")) g_tune.dec_rows_ks_attn = value;
  else if (!strcmp(key, "dec_rows_ks_fc2")) g_tune.dec_rows_ks_fc2 = value;
  else if (!strcmp(key, "dec_cross_u")) g_dec_cross_u = value;
  else if (!strcmp(key, "dec_cross_nt")) g_dec_cross_nt = value;
  else { set_error("wj_tune: unknown key %s", key); return WJ_E_INVALID; }
  return WJ_OK;
}

int64_t wj_logmel_frames(int64_t n_samples, int mode) {
  if (n_samples <= 0) return 0;
  return (n_samples + (mode == WJ_MEL_FW ? 160 : mode == WJ_MEL_OW ? 480000 : 0)) / 160;
}

int wj_logmel_f32(wj_ctx* ctx, const float* pcm_dev, const int64_t* offsets_host, int n_clips, int n_mels, int mode,
                  int out_frames, float* out_dev, void* stream) {
  WJ_REQUIRE(ctx && pcm_dev && offsets_host && out_dev, "wj_logmel_f32: NULL argument");
  WJ_HIP(hipSetDevice(ctx->device));
  return logmel_run(ctx, pcm_dev, offsets_host, n_clips, n_mels, mode, out_frames, out_dev, ctx->pick(stream));
}

int wj_whisper_free(wj_whisper* m) {
  if (!m) return WJ_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->allocs) (void)hipFree(p);
  if (m->align_buf) (void)hipFree(m->align_buf);
  delete m;
  return WJ_OK;
}

int wj_whisper_create(wj_ctx* ctx, const wj_whisper_dims* dims, int dtype, const void* blob_dev, int64_t blob_bytes,
                      const int64_t* offsets_host, int n_offsets, int max_batch, int max_rows, wj_whisper** out) {
  WJ_REQUIRE(ctx && dims && blob_dev && offsets_host && out, "wj_whisper_create: NULL argument");
  WJ_REQUIRE(dtype == WJ_F32 || dtype == WJ_BF16 || dtype == WJ_F16, "wj_whisper_create: dtype must be WJ_F32, WJ_BF16 or WJ_F16");
  const wj_whisper_dims& d = *dims;
  WJ_REQUIRE(d.n_audio_state == d.n_text_state && d.n_audio_head == d.n_text_head,
             "wj_whisper_create: encoder/decoder width mismatch");
  WJ_REQUIRE(d.n_audio_state == 64 * d.n_audio_head, "wj_whisper_create: head_dim must be 64 (d=%d, heads=%d)",
             d.n_audio_state, d.n_audio_head);
  WJ_REQUIRE(d.n_audio_state <= 1280, "wj_whisper_create: d_model must be <= 1280");
  WJ_REQUIRE(d.n_mels == 80 || d.n_mels == 128, "wj_whisper_create: n_mels must be 80 or 128");
  WJ_REQUIRE(d.n_audio_ctx % 4 == 0 && d.n_audio_ctx > 0 && d.n_text_ctx >= 8, "wj_whisper_create: bad context sizes");
  WJ_REQUIRE(max_batch >= 1 && max_rows >= max_batch, "wj_whisper_create: need max_rows >= max_batch >= 1");
  const int expect = WJ_T_N_GLOBAL + d.n_audio_layer * WJ_TE_N + d.n_text_layer * WJ_TD_N;
  WJ_REQUIRE(n_offsets == expect, "wj_whisper_create: offset table has %d entries, expected %d", n_offsets, expect);
  for (int i = 0; i < n_offsets; ++i)
    WJ_REQUIRE(offsets_host[i] >= 0 && offsets_host[i] < blob_bytes && offsets_host[i] % 256 == 0,
               "wj_whisper_create: offset %d (%lld) outside the blob or not 256-byte aligned", i, (long long)offsets_host[i]);
  WJ_HIP(hipSetDevice(ctx->device));

  wj_whisper* m = new wj_whisper();
  m->ctx = ctx;
  m->d = d;
  m->dtype = dtype;
  m->esz = dtype_size(dtype);
  m->blob = reinterpret_cast<const char*>(blob_dev);
  m->off.assign(offsets_host, offsets_host + n_offsets);
  m->max_batch = max_batch;
  m->max_rows = max_rows;
  m->Tpad = (d.n_audio_ctx + 127) / 128 * 128;
  m->frames = 2 * d.n_audio_ctx;
  m->kv_len = (g_tune.self_kv_len > 0 && g_tune.self_kv_len < d.n_text_ctx) ? (g_tune.self_kv_len + 7) / 8 * 8 : d.n_text_ctx;
  if (m->kv_len > d.n_text_ctx) m->kv_len = d.n_text_ctx;
  m->enc_batch = (g_tune.enc_batch > 0 && g_tune.enc_batch < max_batch) ? g_tune.enc_batch : max_batch;
  const size_t e = m->esz;
  const int D = d.n_audio_state, H = d.n_audio_head, T = d.n_audio_ctx, F = m->frames;
  const size_t B = max_batch, R = max_rows, EB = m->enc_batch;
  WJ_ALLOC(mel_rows, EB * (F + 2) * d.n_mels * e, true);
  WJ_ALLOC(conv1_out, EB * (F + 2) * D * e, true);
  m->blk = is16(dtype) && g_tune.enc_blocked && D % 256 == 0 && D >= 256 && T % 4 == 0 && (int64_t)EB * T < (1 << 24);
  const size_t EBT = m->blk ? (EB * T + 255) / 256 * 256 : EB * T;   // blocked activations: whole 256-row blocks
  WJ_ALLOC(x, EB * T * D * sizeof(float), false);
  WJ_ALLOC(h, EBT * D * e, false);
  WJ_ALLOC(q, EB * H * m->Tpad * 64 * e, true);
  WJ_ALLOC(k, EB * H * m->Tpad * 64 * e, true);
  WJ_ALLOC(vt, EB * H * 64 * m->Tpad * e, true);
  WJ_ALLOC(attn, EBT * D * e, false);
  WJ_ALLOC(ff, EBT * 4 * D * e, false);
  if (m->blk) {
    // blocked copies of the matrices the 256-tile kernel streams: per encoder layer qkv [3D][D], out [D][D], fc1 [4D][D],
    // fc2 [D][4D]; per decoder layer the cross K/V projection [2D][D]
    m->wblk_off.assign(n_offsets, -1);
    int64_t total = 0;
    auto reserve = [&](int idx, int64_t rows, int64_t cols) { m->wblk_off[idx] = total; total += rows * cols; };
    for (int l = 0; l < d.n_audio_layer; ++l) {
      const int b0 = m->enc_base(l);
      reserve(b0 + WJ_TE_QKV_W, 3 * D, D); reserve(b0 + WJ_TE_OUT_W, D, D);
      reserve(b0 + WJ_TE_FC1_W, 4 * D, D); reserve(b0 + WJ_TE_FC2_W, D, 4 * D);
    }
    for (int l = 0; l < d.n_text_layer; ++l) reserve(m->dec_base(l) + WJ_TD_CKV_W, 2 * D, D);
    WJ_ALLOC(wblk, (size_t)total * e, false);
    auto convert = [&](int idx, int rows, int cols) {
      return launch_to_blocked(m->W(idx), cols, rows, cols, reinterpret_cast<char*>(m->wblk) + m->wblk_off[idx] * (int64_t)e, ctx->stream);
    };
    int rc = WJ_OK;
    for (int l = 0; l < d.n_audio_layer && !rc; ++l) {
      const int b0 = m->enc_base(l);
      rc = convert(b0 + WJ_TE_QKV_W, 3 * D, D);
      if (!rc) rc = convert(b0 + WJ_TE_OUT_W, D, D);
      if (!rc) rc = convert(b0 + WJ_TE_FC1_W, 4 * D, D);
      if (!rc) rc = convert(b0 + WJ_TE_FC2_W, D, 4 * D);
    }
    for (int l = 0; l < d.n_text_layer && !rc; ++l) rc = convert(m->dec_base(l) + WJ_TD_CKV_W, 2 * D, D);
    if (rc) { wj_whisper_free(m); return rc; }
  }
  WJ_ALLOC(cross_k, (size_t)d.n_text_layer * m->cross_layer_elems() * e, false);
  m->cross_tpad = (is16(dtype) && g_tune.dec_cross_mfma) ? (d.n_audio_ctx + 31) / 32 * 32 : 0;
  WJ_ALLOC(cross_v, (size_t)d.n_text_layer * m->cross_v_layer_elems() * e, true);   // pad keys stay zero forever
  WJ_ALLOC(dx, R * D * sizeof(float), false);
  m->split_act = dtype == WJ_F16 && g_tune.dec_split_act != 0;
  m->split_all = m->split_act && g_tune.dec_split_act >= 2;
  const size_t sm = m->split_act ? 2 : 1;
  WJ_ALLOC(dh, R * D * e * sm, false);
  WJ_ALLOC(dq, R * D * e, false);
  WJ_ALLOC(dattn, R * D * e * sm, false);
  WJ_ALLOC(dff, R * 4 * D * e * sm, false);
  WJ_ALLOC(partial, R * (size_t)kDecKsMax * D * sizeof(float), false);
  m->ldl = (d.n_vocab + 63) / 64 * 64;
  WJ_ALLOC(logits, R * m->ldl * sizeof(float), false);
  WJ_ALLOC(self_k, (size_t)d.n_text_layer * m->self_layer_elems() * e, true);
  WJ_ALLOC(self_v, (size_t)d.n_text_layer * m->self_layer_elems() * e, true);
  m->tok_stride = d.n_text_ctx + 8;
  WJ_ALLOC(tokens, R * m->tok_stride * sizeof(int32_t), true);
  WJ_ALLOC(pos, 256, true);
  WJ_ALLOC(sum_lp, R * sizeof(float), true);
  WJ_ALLOC(tok_lp, R * m->tok_stride * sizeof(float), true);
  WJ_ALLOC(finished, R * sizeof(int32_t), true);
  WJ_ALLOC(nsp, R * sizeof(float), true);
  WJ_ALLOC(row_map[0], R * m->kv_len * sizeof(int32_t), true);
  WJ_ALLOC(row_map[1], R * m->kv_len * sizeof(int32_t), true);
  WJ_ALLOC(parent, R * sizeof(int32_t), true);
  WJ_ALLOC(step_tok, R * sizeof(int32_t), true);
  WJ_ALLOC(slot_map, B * sizeof(int32_t), true);
  WJ_ALLOC(align_sel, (size_t)d.n_text_layer * d.n_text_head * sizeof(int32_t), true);
  WJ_ALLOC(topk_ids, R * 16 * sizeof(int32_t), true);
  WJ_ALLOC(topk_lp, R * 16 * sizeof(float), true);
  WJ_ALLOC(topk_lse, R * sizeof(float), true);
  WJ_ALLOC(tokens2, R * m->tok_stride * sizeof(int32_t), true);
  WJ_ALLOC(beam_score, R * sizeof(float), true);
  WJ_ALLOC(beam_done, (B + 1) * sizeof(int32_t), true);
  WJ_ALLOC(win_ids, B * sizeof(int32_t), true);
  WJ_ALLOC(src_rows, R * sizeof(int32_t), true);
  WJ_ALLOC(fin_count, B * sizeof(int32_t), true);
  WJ_ALLOC(fin_score, B * kFinCap * sizeof(float), true);
  WJ_ALLOC(fin_len, B * kFinCap * sizeof(int32_t), true);
  WJ_ALLOC(fin_tokens, B * kFinCap * m->tok_stride * sizeof(int32_t), true);
  hipError_t se = hipStreamSynchronize(ctx->stream);
  if (se != hipSuccess) {
    set_error("wj_whisper_create: %s", hipGetErrorString(se));
    wj_whisper_free(m);
    return WJ_E_HIP;
  }
  *out = m;
  return WJ_OK;
}

int64_t wj_whisper_workspace_bytes(const wj_whisper* m) { return m ? (int64_t)m->alloc_bytes : 0; }

int wj_whisper_encode(wj_whisper* m, const float* mel_dev, int batch, int n_layers, float* enc_out_dev, void* stream) {
  WJ_REQUIRE(m && mel_dev, "wj_whisper_encode: NULL argument");
  WJ_REQUIRE(batch >= 1 && batch <= m->max_batch, "wj_whisper_encode: batch %d outside 1..%d", batch, m->max_batch);
  WJ_REQUIRE(n_layers <= m->d.n_audio_layer, "wj_whisper_encode: n_layers too large");
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  const int64_t mel_win = (int64_t)m->d.n_mels * m->frames, out_win = (int64_t)m->d.n_audio_ctx * m->d.n_audio_state;
  for (int w0 = 0; w0 < batch; w0 += m->enc_batch) {      // slices of the encoder workspaces; cross K/V of all windows stays resident
    const int bc = std::min(m->enc_batch, batch - w0);
    WJ_TRY(run_encoder(m, mel_dev + w0 * mel_win, bc, n_layers, enc_out_dev ? enc_out_dev + w0 * out_win : nullptr, s, w0));
  }
  return WJ_OK;
}

int wj_whisper_encode_at(wj_whisper* m, const float* mel_dev, int batch, int slot0, void* stream) {
  WJ_REQUIRE(m && mel_dev, "wj_whisper_encode_at: NULL argument");
  WJ_REQUIRE(batch >= 1 && slot0 >= 0 && slot0 + batch <= m->max_batch, "wj_whisper_encode_at: windows %d..%d outside the %d resident slots",
             slot0, slot0 + batch - 1, m->max_batch);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  const int64_t mel_win = (int64_t)m->d.n_mels * m->frames;
  for (int w0 = 0; w0 < batch; w0 += m->enc_batch) {
    const int bc = std::min(m->enc_batch, batch - w0);
    WJ_TRY(run_encoder(m, mel_dev + w0 * mel_win, bc, -1, nullptr, s, slot0 + w0));
  }
  return WJ_OK;
}

int wj_whisper_decode_greedy(wj_whisper* m, int batch, const int32_t* prompts_host, int prompt_len,
                             const wj_decode_opts* opts, int32_t* tokens_out, int32_t* n_tokens_out,
                             float* sum_logprob_out, float* no_speech_prob_out, float* token_logprob_out,
                             void* stream) {
  return wj_whisper_decode_sample(m, batch, 1, nullptr, prompts_host, prompt_len, opts, 0.0f, 0u, tokens_out, n_tokens_out,
                                  sum_logprob_out, no_speech_prob_out, token_logprob_out, stream);
}

int wj_whisper_decode_sample(wj_whisper* m, int batch, int group, const int32_t* slots_host, const int32_t* prompts_host,
                             int prompt_len, const wj_decode_opts* opts, float temperature, uint32_t seed,
                             int32_t* tokens_out, int32_t* n_tokens_out, float* sum_logprob_out,
                             float* no_speech_prob_out, float* token_logprob_out, void* stream) {
  WJ_REQUIRE(m && prompts_host && opts && tokens_out && n_tokens_out && sum_logprob_out, "wj_whisper_decode_sample: NULL argument");
  WJ_REQUIRE(batch >= 1 && batch <= m->max_batch, "decode: batch %d outside 1..%d", batch, m->max_batch);
  WJ_REQUIRE(group >= 1 && (group <= 6 || group == 8) && batch * group <= m->max_rows,
             "decode: %d samples per window x %d windows does not fit (max_rows %d; group 1..6 or 8)", group, batch, m->max_rows);
  WJ_REQUIRE(temperature >= 0.f, "decode: negative temperature");
  if (slots_host)
    for (int i = 0; i < batch; ++i)
      WJ_REQUIRE(slots_host[i] >= 0 && slots_host[i] < m->max_batch, "decode: window slot %d out of range", slots_host[i]);
  const int max_new = opts->max_new_tokens;
  WJ_REQUIRE(prompt_len >= 1 && max_new >= 1 && prompt_len + max_new <= m->kv_len,
             "decode_greedy: prompt_len %d + max_new_tokens %d exceeds the %d positions of the KV cache (n_text_ctx %d, self_kv_len at create)",
             prompt_len, max_new, m->kv_len, m->d.n_text_ctx);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  const int R = batch * group;   // row r belongs to window r / group
  WJ_TRY(reset_decode_state(m, R, s));
  m->use_slots = slots_host != nullptr;
  if (slots_host) WJ_HIP(hipMemcpyAsync(m->slot_map, slots_host, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
  struct SlotGuard { wj_whisper* m; ~SlotGuard() { m->use_slots = false; } } slot_guard{m};
  // prompt -> device history (row stride tok_stride)
  {
    std::vector<int32_t> hist((size_t)R * m->tok_stride, opts->eot);
    for (int r = 0; r < R; ++r)
      for (int j = 0; j < prompt_len; ++j)
        hist[(size_t)r * m->tok_stride + j] = prompts_host[(size_t)(r / group) * prompt_len + j];
    WJ_HIP(hipMemcpyAsync(m->tokens, hist.data(), hist.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    WJ_HIP(hipStreamSynchronize(s));
  }
  // prompt positions 0 .. prompt_len-2 only fill the KV cache (position 0 also yields no_speech_prob)
  for (int p = 0; p + 1 < prompt_len; ++p) {
    const bool ns = (p == 0) && no_speech_prob_out != nullptr;
    WJ_TRY(run_decoder_step(m, 0, R, batch, group, ns, s));
    if (ns) WJ_TRY(launch_no_speech_prob(m->logits, m->ldl, R, m->d.n_vocab, opts->no_speech, m->nsp, s));
    WJ_TRY(launch_advance_pos(m->pos, s));
  }
  // ---- the decode loop as `chains` INDEPENDENT row slices --------------------------------------------
  // Each chain owns a row range, its own step counter (m->pos + c) and its own stream, and replays its
  // own hipGraph of one decode iteration.  Measured on MI355X / ROCm 7.2 (profiles/): the
  // ~355 short kernels of a step do NOT overlap across streams (2 chains +2 %, 4 chains -26 %; forked
  // branches inside one graph +3 %), i.e. the per-kernel cost is dispatch latency, so the default is one
  // chain and WJ_DECODE_CHAINS keeps the experiment reproducible.
  int chains = g_tune.decode_chains;
  if (const char* ce = getenv("WJ_DECODE_CHAINS")) chains = atoi(ce);
  if (chains < 1) chains = 1;
  if (chains > 4) chains = 4;
  if (prof_on(m->ctx) || group > 1) chains = 1;   // event pairs are recorded on one stream; samples share K/V
  while (chains > 1 && R < 2 * chains) --chains;
  hipStream_t side[4] = {s, nullptr, nullptr, nullptr};
  hipEvent_t ev_start = nullptr;
  for (int c = 1; c < chains; ++c) WJ_HIP(hipStreamCreateWithFlags(&side[c], hipStreamNonBlocking));
  if (chains > 1) {   // the side streams start after the prompt steps; every chain gets a copy of the counter
    for (int c = 1; c < chains; ++c)
      WJ_HIP(hipMemcpyAsync(m->pos + c, m->pos, sizeof(int), hipMemcpyDeviceToDevice, s));
    WJ_HIP(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
    WJ_HIP(hipEventRecord(ev_start, s));
    for (int c = 1; c < chains; ++c) WJ_HIP(hipStreamWaitEvent(side[c], ev_start, 0));
  }
  auto iteration = [&](int c) -> int {
    const int r0 = (int)((int64_t)R * c / chains), r1 = (int)((int64_t)R * (c + 1) / chains);
    hipStream_t s = side[c];   // shadows the outer stream: PROF records on the chain's stream
    int* pos = m->pos + c;
    WJ_TRY(run_decoder_step(m, r0, r1 - r0, (r1 - r0) / group, group, true, s, pos));
    GreedyArgs ga;
    ga.logits = m->logits + (int64_t)r0 * m->ldl; ga.ldl = m->ldl; ga.R = r1 - r0; ga.V = m->d.n_vocab;
    ga.tokens = m->tokens + (int64_t)r0 * m->tok_stride; ga.tok_stride = m->tok_stride; ga.pos_ptr = pos;
    ga.sample_begin = prompt_len; ga.sum_logprob = m->sum_lp + r0; ga.token_logprob = m->tok_lp + (int64_t)r0 * m->tok_stride;
    ga.finished = m->finished + r0; ga.opts = *opts;
    ga.temperature = temperature; ga.seed = seed; ga.row_offset = r0;
    PROF(PT_D_SAMPLE, launch_greedy_sample(ga, s));
    PROF(PT_D_MISC, launch_advance_pos(pos, s));
    return WJ_OK;
  };

  // One decode iteration = ~11 launches per layer; capture it once per chain and replay it from a
  // hipGraph so the loop is not host-launch bound.  Every step-dependent scalar lives in device memory.
  hipGraph_t graph[4] = {nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t exec[4] = {nullptr, nullptr, nullptr, nullptr};
  const char* env = getenv("WJ_NO_GRAPH");
  bool use_graph = !(env && env[0] == '1') && !prof_on(m->ctx);
  for (int c = 0; c < chains && use_graph; ++c) {
    hipError_t e = hipStreamBeginCapture(side[c], hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      int rc = iteration(c);
      e = hipStreamEndCapture(side[c], &graph[c]);
      if (rc || e != hipSuccess || graph[c] == nullptr) use_graph = false;
      else if (hipGraphInstantiate(&exec[c], graph[c], nullptr, nullptr, 0) != hipSuccess) use_graph = false;
    } else {
      use_graph = false;
    }
  }
  if (!use_graph) {
    (void)hipGetLastError();
    for (int c = 0; c < 4; ++c) {
      if (exec[c]) { (void)hipGraphExecDestroy(exec[c]); exec[c] = nullptr; }
      if (graph[c]) { (void)hipGraphDestroy(graph[c]); graph[c] = nullptr; }
    }
  }
  m->last_used_graph = use_graph ? 1 : 0;
  m->last_chains = chains;
  std::vector<int32_t> fin(R);
  m->last_steps = 0; m->last_max_new = max_new;
  m->last_compactions = 0; m->last_window_steps = 0;
  for (int i = 0; i < max_new; ++i) {
    m->last_steps = i + 1;
    m->last_window_steps += batch;
    for (int c = 0; c < chains; ++c) {
      if (use_graph) {
        WJ_HIP(hipGraphLaunch(exec[c], side[c]));
      } else {
        WJ_TRY(iteration(c));
      }
    }
    if ((i & 15) == 15 && i + 1 < max_new) {  // early exit once every row has emitted EOT
      for (int c = 1; c < chains; ++c) WJ_HIP(hipStreamSynchronize(side[c]));
      WJ_HIP(hipMemcpyAsync(fin.data(), m->finished, sizeof(int32_t) * R, hipMemcpyDeviceToHost, s));
      WJ_HIP(hipStreamSynchronize(s));
      bool all = true;
      for (int r = 0; r < R; ++r) all = all && fin[r];
      if (all) break;
    }
  }
  for (int c = 0; c < chains; ++c) WJ_HIP(hipStreamSynchronize(side[c]));
  for (int c = 1; c < chains; ++c) (void)hipStreamDestroy(side[c]);
  if (ev_start) (void)hipEventDestroy(ev_start);
  for (int c = 0; c < 4; ++c) {
    if (exec[c]) (void)hipGraphExecDestroy(exec[c]);
    if (graph[c]) (void)hipGraphDestroy(graph[c]);
  }

  std::vector<int32_t> hist((size_t)R * m->tok_stride);
  std::vector<float> lps((size_t)R * m->tok_stride);
  WJ_HIP(hipMemcpyAsync(hist.data(), m->tokens, hist.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(lps.data(), m->tok_lp, lps.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(sum_logprob_out, m->sum_lp, sizeof(float) * R, hipMemcpyDeviceToHost, s));
  if (no_speech_prob_out) WJ_HIP(hipMemcpyAsync(no_speech_prob_out, m->nsp, sizeof(float) * R, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipStreamSynchronize(s));
  for (int r = 0; r < R; ++r) {
    int n = 0;
    bool done = false;
    for (int j = 0; j < max_new; ++j) {
      const int32_t t = hist[(size_t)r * m->tok_stride + prompt_len + j];
      if (!done && t == opts->eot) done = true;
      tokens_out[(size_t)r * max_new + j] = done ? opts->eot : t;
      if (token_logprob_out) token_logprob_out[(size_t)r * max_new + j] = lps[(size_t)r * m->tok_stride + prompt_len + j];
      if (!done) ++n;
    }
    n_tokens_out[r] = n;
  }
  return WJ_OK;
}

// ------------------------------------------------------------------------------------------------
// beam search, device resident (CTranslate2 semantics; the host-driven restatement in
// whisperjav_amd/search.py stays the cross-check).  A step = decoder step + top-2K per row + per-window
// merge + cache re-binding, captured as two hipGraphs (even / odd steps swap the history and row-map
// buffers).  The host only polls a done counter every 8 steps and ranks the finished lists at the end.
// ------------------------------------------------------------------------------------------------
}  // extern "C"

// flavor 0: CTranslate2's search (faster-whisper); flavor 1: openai-whisper's BeamSearchDecoder + MaximumLikelihoodRanker
static int decode_beam_impl(int flavor, wj_whisper* m, int batch, int beam, const int32_t* slots_host, const int32_t* prompts_host,
                            int prompt_len, const wj_decode_opts* opts, float patience, float length_penalty,
                            int32_t* tokens_out, int32_t* n_tokens_out, float* score_out, float* sum_logprob_out,
                            float* no_speech_prob_out, void* stream) {
  WJ_REQUIRE(m && prompts_host && opts && tokens_out && n_tokens_out && sum_logprob_out, "wj_whisper_decode_beam: NULL argument");
  WJ_REQUIRE(batch >= 1 && batch <= m->max_batch && beam >= 1 && (beam <= 6 || beam == 8) && batch * beam <= m->max_rows,
             "decode_beam: batch %d x beam %d does not fit (max_batch %d, max_rows %d; beam 1..6 or 8)", batch, beam,
             m->max_batch, m->max_rows);
  const int max_new = opts->max_new_tokens, K = beam, R = batch * beam, P = prompt_len;
  WJ_REQUIRE(P >= 1 && max_new >= 1 && P + max_new <= m->kv_len,
             "decode_beam: prompt_len %d + max_new_tokens %d exceeds the %d positions of the KV cache (n_text_ctx %d, self_kv_len at create)",
             P, max_new, m->kv_len, m->d.n_text_ctx);
  WJ_REQUIRE(patience > 0.f, "decode_beam: patience must be positive");
  // CTranslate2: std::round (half away from zero); openai-whisper: Python's round() (half to even)
  const int max_candidates = flavor == 1 ? (int)nearbyintf(K * patience) : (int)lroundf(K * patience);
  WJ_REQUIRE(max_candidates >= 1 && max_candidates + K <= kFinCap, "decode_beam: beam %d x patience %g needs %d finished slots (max %d)",
             K, (double)patience, max_candidates + K, kFinCap);
  // openai-whisper tops a window up to `beam` sequences with the beams alive when the WHOLE batch stops; with
  // round(beam * patience) < beam that depends on the other windows of the batch -- not reproduced on the device
  WJ_REQUIRE(flavor == 0 || max_candidates >= K, "decode_beam (openai flavour): round(beam * patience) = %d < beam %d is not supported",
             max_candidates, K);
  if (slots_host)
    for (int i = 0; i < batch; ++i)
      WJ_REQUIRE(slots_host[i] >= 0 && slots_host[i] < m->max_batch, "decode_beam: window slot %d out of range", slots_host[i]);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  WJ_TRY(reset_decode_state(m, R, s));
  m->use_slots = slots_host != nullptr;
  if (slots_host) WJ_HIP(hipMemcpyAsync(m->slot_map, slots_host, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
  struct SlotGuard { wj_whisper* m; ~SlotGuard() { m->use_slots = false; } } slot_guard{m};
  int32_t* buf[2] = {m->tokens, m->tokens2};
  {
    std::vector<int32_t> hist((size_t)R * m->tok_stride, opts->eot);
    std::vector<float> sc(R, -INFINITY);
    for (int r = 0; r < R; ++r) {
      for (int j = 0; j < P; ++j) hist[(size_t)r * m->tok_stride + j] = prompts_host[(size_t)(r / K) * P + j];
      if (r % K == 0 || flavor == 1) sc[r] = 0.f;  // CTranslate2: one live beam per window at the start; whisper: K copies
    }
    WJ_HIP(hipMemcpyAsync(buf[0], hist.data(), sizeof(int32_t) * hist.size(), hipMemcpyHostToDevice, s));
    WJ_HIP(hipMemcpyAsync(m->beam_score, sc.data(), sizeof(float) * R, hipMemcpyHostToDevice, s));
    WJ_HIP(hipMemsetAsync(m->beam_done, 0, sizeof(int32_t) * (m->max_batch + 1), s));
    WJ_HIP(hipMemsetAsync(m->fin_count, 0, sizeof(int32_t) * m->max_batch, s));
    WJ_HIP(hipStreamSynchronize(s));
  }
  // prompt: all but its last token (its logits are not needed, except for the no-speech probability at step 0)
  for (int p = 0; p + 1 < P; ++p) {
    const bool ns = p == 0 && opts->no_speech >= 0;
    WJ_TRY(run_decoder_step(m, 0, R, batch, K, ns, s));
    if (ns) WJ_TRY(launch_no_speech_prob(m->logits, m->ldl, R, m->d.n_vocab, opts->no_speech, m->nsp, s));
    WJ_TRY(launch_advance_pos(m->pos, s));
  }
  // Windows whose search has ended (round(beam * patience) hypotheses finished) leave the batch: every `beam_poll`
  // iterations the host reads the per-window done flags and, when enough of them are set, re-packs the live windows into
  // the first rows (history, score, row map gathered on the device; KV-cache rows and finished lists do not move: the
  // row map keeps addressing the former, `win_ids` the latter) and re-captures the two step graphs for the smaller batch.
  int n_act = batch;                                  // live logical windows; logical window w = rows [w K, (w + 1) K)
  std::vector<int32_t> act_win(batch), act_slot(batch);
  for (int w = 0; w < batch; ++w) { act_win[w] = w; act_slot[w] = slots_host ? slots_host[w] : w; }
  const bool want_lp = g_tune.beam_token_logprobs != 0;
  float* cbuf[2] = {m->tok_lp, nullptr};
  m->last_beam_lp.clear(); m->last_beam_lp_batch = 0; m->last_beam_lp_stride = 0;
  if (want_lp) {
    if (!m->cum2) WJ_TRY(dev_alloc(m, reinterpret_cast<void**>(&m->cum2), (size_t)m->max_rows * m->tok_stride * sizeof(float), true));
    if (!m->fin_cum) WJ_TRY(dev_alloc(m, reinterpret_cast<void**>(&m->fin_cum), (size_t)m->max_batch * kFinCap * m->tok_stride * sizeof(float), true));
    cbuf[1] = m->cum2;
    WJ_HIP(hipMemsetAsync(cbuf[0], 0, (size_t)R * m->tok_stride * sizeof(float), s));
    WJ_HIP(hipMemsetAsync(cbuf[1], 0, (size_t)R * m->tok_stride * sizeof(float), s));
  }
  const bool compact_ok = g_tune.beam_compact != 0 && !want_lp;
  if (compact_ok) {   // the window -> cross K/V slot and window -> result index maps become explicit
    m->use_slots = true;
    WJ_HIP(hipMemcpyAsync(m->slot_map, act_slot.data(), sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
    WJ_HIP(hipMemcpyAsync(m->win_ids, act_win.data(), sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
    WJ_HIP(hipStreamSynchronize(s));
  }
  auto iteration = [&](int par, bool first) -> int {
    const int Ra = n_act * K;
    int32_t* saved = m->tokens;
    m->tokens = buf[par];                           // the decoder step embeds the token at *pos of this history
    m->cur_map = par;
    int rc = run_decoder_step(m, 0, Ra, n_act, K, true, s);
    m->tokens = saved;
    if (rc) return rc;
    if (first && P == 1 && opts->no_speech >= 0)
      WJ_TRY(launch_no_speech_prob(m->logits, m->ldl, Ra, m->d.n_vocab, opts->no_speech, m->nsp, s));
    BeamArgs a;
    a.logits = m->logits; a.ldl = m->ldl; a.V = m->d.n_vocab; a.K = K;
    a.hist_in = buf[par]; a.hist_out = buf[par ^ 1]; a.tok_stride = m->tok_stride; a.pos_ptr = m->pos;
    a.sample_begin = P; a.max_new = max_new; a.max_candidates = max_candidates; a.opts = *opts;
    a.cand_ids = m->topk_ids; a.cand_lp = m->topk_lp; a.score = m->beam_score; a.parent = m->parent;
    a.win_ids = compact_ok ? m->win_ids : nullptr;
    a.done = m->beam_done; a.n_done = m->beam_done + m->max_batch;
    a.fin_count = m->fin_count; a.fin_score = m->fin_score; a.fin_len = m->fin_len; a.fin_tokens = m->fin_tokens;
    a.fin_cap = kFinCap; a.flavor = flavor;
    if (want_lp) { a.cum_in = cbuf[par]; a.cum_out = cbuf[par ^ 1]; a.fin_cum = m->fin_cum; }
    WJ_TRY(launch_beam_step(a, Ra, n_act, s));
    WJ_TRY(launch_advance_pos(m->pos, s));
    WJ_TRY(launch_rebind_rows(m->row_map[par], m->row_map[par ^ 1], m->parent, m->pos, Ra, m->kv_len, s));
    return WJ_OK;
  };
  // row maps: reset_decode_state initialised both to the identity; parity 0 reads map 0
  hipGraph_t graph[2] = {nullptr, nullptr};
  hipGraphExec_t exec[2] = {nullptr, nullptr};
  const char* env = getenv("WJ_NO_GRAPH");
  bool use_graph = !(env && env[0] == '1') && !prof_on(m->ctx) && !(P == 1 && opts->no_speech >= 0);
  auto drop_graphs = [&]() {
    for (int par = 0; par < 2; ++par) {
      if (exec[par]) { (void)hipGraphExecDestroy(exec[par]); exec[par] = nullptr; }
      if (graph[par]) { (void)hipGraphDestroy(graph[par]); graph[par] = nullptr; }
    }
  };
  auto capture_graphs = [&]() {
    drop_graphs();
    for (int par = 0; par < 2 && use_graph; ++par) {
      hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        int rc = iteration(par, false);
        e = hipStreamEndCapture(s, &graph[par]);
        if (rc || e != hipSuccess || graph[par] == nullptr) use_graph = false;
        else if (hipGraphInstantiate(&exec[par], graph[par], nullptr, nullptr, 0) != hipSuccess) use_graph = false;
      } else {
        use_graph = false;
      }
    }
    if (!use_graph) { (void)hipGetLastError(); drop_graphs(); }
  };
  capture_graphs();
  m->last_used_graph = use_graph ? 1 : 0;
  m->last_chains = 1;
  int rc_loop = WJ_OK;
  m->last_steps = 0; m->last_max_new = max_new;
  m->last_compactions = 0; m->last_window_steps = 0;
  const int poll = std::max(1, g_tune.beam_poll);
  std::vector<int32_t> done_host(m->max_batch + 1);
  int par = 0;
  for (int i = 0; i < max_new && rc_loop == WJ_OK; ++i) {
    m->last_steps = i + 1;
    m->last_window_steps += n_act;
    if (use_graph) {
      if (hipGraphLaunch(exec[par], s) != hipSuccess) { set_error("decode_beam: graph launch failed"); rc_loop = WJ_E_HIP; }
    } else {
      rc_loop = iteration(par, i == 0);
    }
    par ^= 1;                                       // the next iteration reads what this one wrote
    if ((i % poll) == poll - 1 && i + 1 < max_new && rc_loop == WJ_OK) {
      if (hipMemcpyAsync(done_host.data(), m->beam_done, sizeof(int32_t) * (m->max_batch + 1), hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) { set_error("decode_beam: done poll failed"); rc_loop = WJ_E_HIP; break; }
      if (done_host[m->max_batch] >= batch) break;
      if (!compact_ok) continue;
      int live = 0;
      for (int w = 0; w < n_act; ++w) live += done_host[act_win[w]] == 0;
      const int gone = n_act - live;
      if (live == 0 || gone < std::max(1, g_tune.beam_compact_min) || gone * 100 < n_act * g_tune.beam_compact_pct) continue;
      // ---- re-pack the live windows into logical windows 0 .. live-1 ------------------------------------------
      std::vector<int32_t> src((size_t)live * K);
      int nw = 0;
      for (int w = 0; w < n_act; ++w) {
        if (done_host[act_win[w]]) continue;
        for (int b = 0; b < K; ++b) src[(size_t)nw * K + b] = w * K + b;
        act_win[nw] = act_win[w]; act_slot[nw] = act_slot[w];
        ++nw;
      }
      n_act = live;
      // the current state sits in buf[par] / row_map[par] (what the next iteration reads): gather it into the other
      // buffers and read from those instead
      hipError_t e = hipMemcpyAsync(m->src_rows, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipMemcpyAsync(m->slot_map, act_slot.data(), sizeof(int32_t) * n_act, hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipMemcpyAsync(m->win_ids, act_win.data(), sizeof(int32_t) * n_act, hipMemcpyHostToDevice, s);
      if (e != hipSuccess) { set_error("decode_beam: compaction upload failed: %s", hipGetErrorString(e)); rc_loop = WJ_E_HIP; break; }
      rc_loop = launch_compact_rows(m->src_rows, n_act * K, m->row_map[par], m->row_map[par ^ 1], m->kv_len, m->pos, buf[par],
                                    buf[par ^ 1], m->tok_stride, m->beam_score, m->sum_lp, s);
      if (rc_loop) break;
      if (hipMemcpyAsync(m->beam_score, m->sum_lp, sizeof(float) * n_act * K, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) { set_error("decode_beam: compaction failed"); rc_loop = WJ_E_HIP; break; }
      par ^= 1;
      ++m->last_compactions;
      if (use_graph) {
        capture_graphs();
        if (!use_graph) m->last_used_graph = 0;
      }
    }
  }
  drop_graphs();
  m->cur_map = 0;
  if (rc_loop) return rc_loop;
  // finished lists -> best hypothesis per window
  std::vector<int32_t> fcount(batch), flen((size_t)batch * kFinCap), ftok((size_t)batch * kFinCap * m->tok_stride);
  std::vector<float> fscore((size_t)batch * kFinCap), nsp(R);
  WJ_HIP(hipMemcpyAsync(fcount.data(), m->fin_count, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(flen.data(), m->fin_len, sizeof(int32_t) * flen.size(), hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(fscore.data(), m->fin_score, sizeof(float) * fscore.size(), hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(ftok.data(), m->fin_tokens, sizeof(int32_t) * ftok.size(), hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(nsp.data(), m->nsp, sizeof(float) * R, hipMemcpyDeviceToHost, s));
  std::vector<float> fcum, live_cum;
  if (want_lp) {
    fcum.resize(ftok.size());
    WJ_HIP(hipMemcpyAsync(fcum.data(), m->fin_cum, sizeof(float) * fcum.size(), hipMemcpyDeviceToHost, s));
    m->last_beam_lp.assign((size_t)batch * (max_new + 1), NAN);
    m->last_beam_lp_batch = batch; m->last_beam_lp_stride = max_new + 1;
  }
  // openai flavour: a window that holds fewer than `beam` finished sequences when the loop ends is topped up with its
  // live beams (BeamSearchDecoder.finalize): their histories and cumulative log-probs, in the buffers the next
  // iteration would have read
  std::vector<int32_t> live_hist;
  std::vector<float> live_score;
  std::vector<int> where(batch, -1);              // logical window of each window still in the batch
  if (flavor == 1) {
    live_hist.resize((size_t)n_act * K * m->tok_stride);
    live_score.resize((size_t)n_act * K);
    WJ_HIP(hipMemcpyAsync(live_hist.data(), buf[par], sizeof(int32_t) * live_hist.size(), hipMemcpyDeviceToHost, s));
    WJ_HIP(hipMemcpyAsync(live_score.data(), m->beam_score, sizeof(float) * live_score.size(), hipMemcpyDeviceToHost, s));
    if (want_lp) {
      live_cum.resize(live_hist.size());
      WJ_HIP(hipMemcpyAsync(live_cum.data(), cbuf[par], sizeof(float) * live_cum.size(), hipMemcpyDeviceToHost, s));
    }
    for (int w = 0; w < n_act; ++w) where[act_win[w]] = w;
  }
  WJ_HIP(hipStreamSynchronize(s));
  const int n_gen = m->last_steps;                 // tokens every live beam has generated
  for (int w = 0; w < batch; ++w) {
    const int n = std::min(fcount[w], kFinCap);
    struct Hyp { double score; int len; const int32_t* tok; const float* cum; };
    std::vector<Hyp> hyps;
    for (int i = 0; i < n; ++i)
      hyps.push_back({(double)fscore[(size_t)w * kFinCap + i], flen[(size_t)w * kFinCap + i],
                      &ftok[((size_t)w * kFinCap + i) * m->tok_stride], want_lp ? &fcum[((size_t)w * kFinCap + i) * m->tok_stride] : nullptr});
    if (flavor == 1 && n < K) {
      WJ_REQUIRE(where[w] >= 0, "decode_beam: window %d left the batch with %d < %d finished sequences", w, n, K);
      std::vector<int> order(K);
      for (int b = 0; b < K; ++b) order[b] = b;
      const float* sc = &live_score[(size_t)where[w] * K];
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sc[x] > sc[y]; });
      for (int b : order) {
        if ((int)hyps.size() >= K) break;
        const int32_t* t = &live_hist[((size_t)where[w] * K + b) * m->tok_stride + P];
        int len = 0;                                // tokens before the first EOT (whisper slices at the first EOT)
        while (len < n_gen && t[len] != opts->eot) ++len;
        bool dup = false;                           // sequences are dict keys upstream: an equal one only overwrites
        for (const Hyp& h : hyps)
          if (h.len == len && std::equal(t, t + len, h.tok)) { dup = true; break; }
        if (!dup) hyps.push_back({(double)sc[b], len, t, want_lp ? &live_cum[((size_t)where[w] * K + b) * m->tok_stride + P] : nullptr});
      }
    }
    WJ_REQUIRE(!hyps.empty(), "decode_beam: window %d finished no hypothesis", w);
    int best = 0;
    double best_norm = -INFINITY;
    for (int i = 0; i < (int)hyps.size(); ++i) {
      const int len = hyps[i].len;
      double norm;
      if (flavor == 1) {      // MaximumLikelihoodRanker: sum / length, or the GNMT penalty ((5 + length) / 6) ** alpha
        const double pen = length_penalty < 0.f ? (double)len : pow((5.0 + len) / 6.0, (double)length_penalty);
        norm = pen != 0.0 ? hyps[i].score / pen : -INFINITY;
      } else {                // CTranslate2: score / len ** length_penalty, first one wins ties
        norm = length_penalty != 0.f ? hyps[i].score / pow((double)std::max(len, 1), (double)length_penalty) : hyps[i].score;
      }
      if (norm > best_norm || i == 0) { best_norm = norm; best = i; }
    }
    const int len = hyps[best].len;
    for (int j = 0; j < max_new; ++j) tokens_out[(size_t)w * max_new + j] = j < len ? hyps[best].tok[j] : opts->eot;
    n_tokens_out[w] = len;
    sum_logprob_out[w] = (float)hyps[best].score;
    if (score_out) score_out[w] = (float)best_norm;
    if (want_lp && hyps[best].cum) {     // differences of the cumulative history; entry len = what the end of the sequence added (EOT)
      float* lp = &m->last_beam_lp[(size_t)w * (max_new + 1)];
      const float* cum = hyps[best].cum;
      for (int j = 0; j < len && j < max_new; ++j) lp[j] = cum[j] - (j ? cum[j - 1] : 0.f);
      if (len <= max_new) lp[len] = (float)hyps[best].score - (len ? cum[len - 1] : 0.f);
    }
    if (no_speech_prob_out) no_speech_prob_out[w] = nsp[(size_t)w * K];
  }
  return WJ_OK;
}

extern "C" {

int wj_whisper_decode_beam(wj_whisper* m, int batch, int beam, const int32_t* slots_host, const int32_t* prompts_host,
                           int prompt_len, const wj_decode_opts* opts, float patience, float length_penalty,
                           int32_t* tokens_out, int32_t* n_tokens_out, float* score_out, float* sum_logprob_out,
                           float* no_speech_prob_out, void* stream) {
  return decode_beam_impl(0, m, batch, beam, slots_host, prompts_host, prompt_len, opts, patience, length_penalty, tokens_out,
                          n_tokens_out, score_out, sum_logprob_out, no_speech_prob_out, stream);
}

int wj_whisper_decode_beam_openai(wj_whisper* m, int batch, int beam, const int32_t* slots_host, const int32_t* prompts_host,
                                  int prompt_len, const wj_decode_opts* opts, float patience, float length_penalty,
                                  int32_t* tokens_out, int32_t* n_tokens_out, float* score_out, float* sum_logprob_out,
                                  float* no_speech_prob_out, void* stream) {
  return decode_beam_impl(1, m, batch, beam, slots_host, prompts_host, prompt_len, opts, patience, length_penalty, tokens_out,
                          n_tokens_out, score_out, sum_logprob_out, no_speech_prob_out, stream);
}

int wj_decode_open(wj_whisper* m, int batch, int beam, void* stream) {
  WJ_REQUIRE(m != nullptr, "wj_decode_open: NULL model");
  WJ_REQUIRE(batch >= 1 && batch <= m->max_batch && beam >= 1 && batch * beam <= m->max_rows,
             "wj_decode_open: batch %d x beam %d does not fit (max_batch %d, max_rows %d)", batch, beam, m->max_batch, m->max_rows);
  WJ_REQUIRE(beam <= 6 || beam == 8, "wj_decode_open: beam must be 1..6 or 8");
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  m->open_batch = batch;
  m->open_beam = beam;
  m->open_rows = batch * beam;
  m->host_pos = 0;
  return reset_decode_state(m, m->open_rows, s);
}

int wj_decode_step(wj_whisper* m, const int32_t* tokens_host, const int32_t* parent_host, int want_logits, void* stream) {
  WJ_REQUIRE(m && tokens_host, "wj_decode_step: NULL argument");
  WJ_REQUIRE(m->open_rows > 0, "wj_decode_step: call wj_decode_open first");
  WJ_REQUIRE(m->host_pos < m->kv_len, "wj_decode_step: context of %d tokens exhausted", m->kv_len);
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  const int R = m->open_rows;
  WJ_HIP(hipMemcpyAsync(m->step_tok, tokens_host, sizeof(int32_t) * R, hipMemcpyHostToDevice, s));
  if (parent_host) {
    WJ_HIP(hipMemcpyAsync(m->parent, parent_host, sizeof(int32_t) * R, hipMemcpyHostToDevice, s));
    WJ_TRY(launch_rebind_rows(m->row_map[m->cur_map], m->row_map[m->cur_map ^ 1], m->parent, m->pos, R, m->kv_len, s));
    m->cur_map ^= 1;
  }
  hipLaunchKernelGGL(put_tokens_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, s, m->tokens, m->tok_stride, m->pos, m->step_tok, R);
  WJ_LAUNCH_CHECK();
  WJ_TRY(run_decoder_step(m, 0, R, m->open_batch, m->open_beam, want_logits != 0, s));
  WJ_TRY(launch_advance_pos(m->pos, s));
  WJ_HIP(hipStreamSynchronize(s));  // host buffers may be reused by the caller
  m->host_pos += 1;
  return WJ_OK;
}

float* wj_decode_logits_dev(wj_whisper* m) { return m ? m->logits : nullptr; }

// ------------------------------------------------------------------------------------------------
// word-timestamp alignment
// ------------------------------------------------------------------------------------------------
int wj_whisper_align(wj_whisper* m, int batch, const int32_t* slots_host, const int32_t* tokens_host, int n_tokens_max,
                     const int32_t* n_tokens_host, int n_prefix, const int32_t* heads_host, int n_heads,
                     const int32_t* num_frames_host, int medfilt_width, int eot, int32_t* path_text_out,
                     int32_t* path_time_out, int32_t* path_len_out, float* token_prob_out, void* stream) {
  WJ_REQUIRE(m && tokens_host && n_tokens_host && heads_host && num_frames_host && path_text_out && path_time_out &&
             path_len_out && token_prob_out, "wj_whisper_align: NULL argument");
  const wj_whisper_dims& d = m->d;
  const int L = d.n_text_layer, H = d.n_text_head, nctx = d.n_audio_ctx, n0 = n_prefix - 1;
  WJ_REQUIRE(batch >= 1 && batch <= m->max_batch && batch <= m->max_rows, "align: batch %d outside 1..%d", batch, m->max_batch);
  WJ_REQUIRE(n_tokens_max >= n_prefix + 1 && n_tokens_max <= m->kv_len, "align: %d tokens per window outside %d..%d",
             n_tokens_max, n_prefix + 1, m->kv_len);
  // full-sequence pass: positions padded to a multiple of 16 (the cross-attention kernels take 16 / 8 query rows
  // of a window per workgroup); needs the sequence to fit the encoder workspaces
  const bool prefill = g_tune.align_prefill && (n_tokens_max + 15) / 16 * 16 <= 512 && (n_tokens_max + 15) / 16 * 16 <= d.n_audio_ctx &&
                       (int64_t)batch * ((n_tokens_max + 15) / 16 * 16) <= (int64_t)m->enc_batch * d.n_audio_ctx;
  const int T = prefill ? (n_tokens_max + 15) / 16 * 16 : n_tokens_max;     // rows of the score / matrix buffers per window
  WJ_REQUIRE(n_prefix >= 2 && n_heads >= 1 && n_heads <= L * H, "align: bad prefix length / head count");
  WJ_REQUIRE(eot > 0 && eot <= d.n_vocab, "align: eot id out of range");
  std::vector<int32_t> sel((size_t)L * H, -1), nf2(batch);
  for (int i = 0; i < n_heads; ++i) {
    const int l = heads_host[2 * i], h = heads_host[2 * i + 1];
    WJ_REQUIRE(l >= 0 && l < L && h >= 0 && h < H, "align: head (%d, %d) outside the %d x %d decoder", l, h, L, H);
    WJ_REQUIRE(sel[(size_t)l * H + h] < 0, "align: head (%d, %d) listed twice", l, h);
    sel[(size_t)l * H + h] = i;
  }
  for (int b = 0; b < batch; ++b) {
    WJ_REQUIRE(n_tokens_host[b] >= n_prefix + 1 && n_tokens_host[b] <= n_tokens_max, "align: window %d has %d tokens (prefix %d + eot .. %d)", b,
               n_tokens_host[b], n_prefix, n_tokens_max);
    WJ_REQUIRE(num_frames_host[b] >= 2 && num_frames_host[b] / 2 <= nctx, "align: window %d: %d feature frames", b, num_frames_host[b]);
    nf2[b] = num_frames_host[b] / 2;
    if (slots_host) WJ_REQUIRE(slots_host[b] >= 0 && slots_host[b] < m->max_batch, "align: window slot %d out of range", slots_host[b]);
  }
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  // scratch: qk [B][n_heads][T][nctx] f32 | matrix [B][T][nctx] f32 | trace [B][T+1][nctx+1] i8 | paths | meta | probs
  const size_t plen = (size_t)T + nctx;
  const size_t b_qk = align_up(sizeof(float) * (size_t)batch * n_heads * T * nctx, 256);
  const size_t b_mx = align_up(sizeof(float) * (size_t)batch * T * nctx, 256);
  const size_t b_tr = align_up((size_t)batch * (T + 1) * (nctx + 1), 256);
  const size_t b_path = align_up(sizeof(int32_t) * (size_t)batch * plen, 256);
  const size_t b_meta = align_up(sizeof(int32_t) * (size_t)batch, 256);
  const size_t b_prob = align_up(sizeof(float) * (size_t)batch * T, 256);
  const size_t need = b_qk + b_mx + b_tr + 2 * b_path + 3 * b_meta + b_prob;
  if (need > m->align_bytes) {
    if (m->align_buf) { WJ_HIP(hipStreamSynchronize(s)); (void)hipFree(m->align_buf); m->align_buf = nullptr; m->align_bytes = 0; m->last_align_matrix = nullptr; }
    WJ_HIP(hipMalloc(&m->align_buf, need));
    m->align_bytes = need;
  }
  char* base = reinterpret_cast<char*>(m->align_buf);
  float* qk = reinterpret_cast<float*>(base);
  float* matrix = reinterpret_cast<float*>(base + b_qk);
  int8_t* trace = reinterpret_cast<int8_t*>(base + b_qk + b_mx);
  int32_t* p_text = reinterpret_cast<int32_t*>(base + b_qk + b_mx + b_tr);
  int32_t* p_time = reinterpret_cast<int32_t*>(base + b_qk + b_mx + b_tr + b_path);
  int32_t* d_ntok = reinterpret_cast<int32_t*>(base + b_qk + b_mx + b_tr + 2 * b_path);
  int32_t* d_nf2 = d_ntok + b_meta / sizeof(int32_t);
  int32_t* d_plen = d_nf2 + b_meta / sizeof(int32_t);
  float* d_prob = reinterpret_cast<float*>(base + b_qk + b_mx + b_tr + 2 * b_path + 3 * b_meta);

  const int R = batch;
  WJ_TRY(reset_decode_state(m, R, s));
  WJ_HIP(hipMemcpyAsync(m->align_sel, sel.data(), sizeof(int32_t) * sel.size(), hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(d_ntok, n_tokens_host, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemcpyAsync(d_nf2, nf2.data(), sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
  WJ_HIP(hipMemsetAsync(d_prob, 0, b_prob, s));
  m->use_slots = slots_host != nullptr;
  if (slots_host) WJ_HIP(hipMemcpyAsync(m->slot_map, slots_host, sizeof(int32_t) * batch, hipMemcpyHostToDevice, s));
  {
    std::vector<int32_t> hist((size_t)R * m->tok_stride, eot);
    for (int r = 0; r < R; ++r)
      for (int j = 0; j < n_tokens_max; ++j) hist[(size_t)r * m->tok_stride + j] = tokens_host[(size_t)r * n_tokens_max + j];
    WJ_HIP(hipMemcpyAsync(m->tokens, hist.data(), sizeof(int32_t) * hist.size(), hipMemcpyHostToDevice, s));
    WJ_HIP(hipStreamSynchronize(s));   // `hist`, `sel`, `nf2` go out of scope / are reused
  }
  m->dump_qk = qk; m->dump_nsel = n_heads; m->dump_tmax = T;
  struct Guard { wj_whisper* m; ~Guard() { m->dump_qk = nullptr; m->use_slots = false; } } guard{m};
  if (prefill) {
    const int nb = is16(m->dtype) && m->cross_tpad > 0 ? 16 : 8;     // query rows of a window per cross-attention workgroup
    const int chunks = T / nb;
    std::vector<int32_t> groups((size_t)batch * chunks);
    for (int b = 0; b < batch; ++b)
      for (int c = 0; c < chunks; ++c) groups[(size_t)b * chunks + c] = slots_host ? slots_host[b] : b;
    int32_t* d_groups = reinterpret_cast<int32_t*>(trace);                // the trace is written after the pass: borrow its head
    WJ_REQUIRE(sizeof(int32_t) * groups.size() <= b_tr, "align: group table does not fit its scratch");
    WJ_HIP(hipMemcpyAsync(d_groups, groups.data(), sizeof(int32_t) * groups.size(), hipMemcpyHostToDevice, s));
    WJ_HIP(hipStreamSynchronize(s));
    WJ_TRY(run_decoder_seq(m, batch, T, n0, d_ntok, d_prob, eot, d_groups, nb, s));
  } else
  // teacher-forced pass: position t reads token t of the history; logits only where a text token is predicted
  for (int t = 0; t < T; ++t) {
    const bool want = t >= n0 && t + 1 < T;
    WJ_TRY(run_decoder_step(m, 0, R, batch, 1, want, s));
    if (want)
      WJ_TRY(launch_align_token_prob(m->logits, m->ldl, eot, m->tokens, m->tok_stride, m->pos, n0, d_prob, T, R, s));
    WJ_TRY(launch_advance_pos(m->pos, s));
  }
  WJ_TRY(launch_align_post(qk, matrix, trace, d_ntok, d_nf2, R, n_heads, T, nctx, n0, medfilt_width, p_text, p_time, d_plen, s));
  m->last_align_matrix = matrix; m->last_align_batch = batch; m->last_align_T = T;
  // device rows are T wide; the ABI's are n_tokens_max wide
  const size_t plen_out = (size_t)n_tokens_max + nctx;
  WJ_HIP(hipMemcpy2DAsync(path_text_out, sizeof(int32_t) * plen_out, p_text, sizeof(int32_t) * plen, sizeof(int32_t) * plen_out,
                          batch, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpy2DAsync(path_time_out, sizeof(int32_t) * plen_out, p_time, sizeof(int32_t) * plen, sizeof(int32_t) * plen_out,
                          batch, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(path_len_out, d_plen, sizeof(int32_t) * batch, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpy2DAsync(token_prob_out, sizeof(float) * n_tokens_max, d_prob, sizeof(float) * T, sizeof(float) * n_tokens_max,
                          batch, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

int wj_whisper_last_align_matrix(const wj_whisper* m, int batch, int n_rows, int n_cols, float* out_host) {
  WJ_REQUIRE(m && out_host, "wj_whisper_last_align_matrix: NULL argument");
  WJ_REQUIRE(m->last_align_matrix && m->align_buf, "wj_whisper_last_align_matrix: no wj_whisper_align call on this model yet");
  WJ_REQUIRE(batch >= 1 && batch <= m->last_align_batch && n_rows >= 1 && n_rows <= m->last_align_T && n_cols >= 1 &&
             n_cols <= m->d.n_audio_ctx, "wj_whisper_last_align_matrix: [%d][%d][%d] outside the last call's [%d][%d][%d]", batch, n_rows,
             n_cols, m->last_align_batch, m->last_align_T, m->d.n_audio_ctx);
  WJ_HIP(hipSetDevice(m->ctx->device));
  const int nctx = m->d.n_audio_ctx, T = m->last_align_T;
  for (int b = 0; b < batch; ++b)
    WJ_HIP(hipMemcpy2D(out_host + (size_t)b * n_rows * n_cols, sizeof(float) * n_cols, m->last_align_matrix + (size_t)b * T * nctx,
                       sizeof(float) * nctx, sizeof(float) * n_cols, n_rows, hipMemcpyDeviceToHost));
  return WJ_OK;
}

int wj_whisper_last_decode_info(const wj_whisper* m, int32_t out[6]) {
  WJ_REQUIRE(m && out, "wj_whisper_last_decode_info: NULL argument");
  out[0] = m->last_used_graph;
  out[1] = m->last_chains;
  out[2] = m->last_steps;
  out[3] = m->last_max_new;
  out[4] = m->last_compactions;
  out[5] = (int32_t)std::min<int64_t>(m->last_window_steps, INT32_MAX);
  return WJ_OK;
}

int wj_whisper_last_beam_token_logprobs(const wj_whisper* m, int batch, int stride, float* out_host) {
  WJ_REQUIRE(m && out_host, "wj_whisper_last_beam_token_logprobs: NULL argument");
  WJ_REQUIRE(m->last_beam_lp_batch > 0, "wj_whisper_last_beam_token_logprobs: the last beam search did not carry them (wj_tune beam_token_logprobs 1 first)");
  WJ_REQUIRE(batch == m->last_beam_lp_batch && stride == m->last_beam_lp_stride,
             "wj_whisper_last_beam_token_logprobs: the last beam search had %d windows x (%d + 1) entries", m->last_beam_lp_batch, m->last_beam_lp_stride - 1);
  memcpy(out_host, m->last_beam_lp.data(), sizeof(float) * m->last_beam_lp.size());
  return WJ_OK;
}

int wj_decode_logits_copy(wj_whisper* m, int rows, float* dst_dev, void* stream) {
  WJ_REQUIRE(m && dst_dev && rows >= 1 && rows <= m->max_rows, "wj_decode_logits_copy: bad arguments");
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  WJ_HIP(hipMemcpy2DAsync(dst_dev, sizeof(float) * m->d.n_vocab, m->logits, sizeof(float) * m->ldl,
                          sizeof(float) * m->d.n_vocab, rows, hipMemcpyDeviceToDevice, s));
  return WJ_OK;
}

int wj_decode_topk(wj_whisper* m, int rows, int k, const uint8_t* ban_dev, int32_t* ids_out_host,
                   float* logprob_out_host, float* lse_out_host, void* stream) {
  WJ_REQUIRE(m && ids_out_host && logprob_out_host, "wj_decode_topk: NULL argument");
  WJ_REQUIRE(rows >= 1 && rows <= m->max_rows && k >= 1 && k <= 16, "wj_decode_topk: rows/k out of range");
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  WJ_TRY(launch_topk_logprob(m->logits, m->ldl, rows, m->d.n_vocab, k, ban_dev, m->topk_ids, m->topk_lp, m->topk_lse, s));
  WJ_HIP(hipMemcpyAsync(ids_out_host, m->topk_ids, sizeof(int32_t) * rows * k, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(logprob_out_host, m->topk_lp, sizeof(float) * rows * k, hipMemcpyDeviceToHost, s));
  if (lse_out_host) WJ_HIP(hipMemcpyAsync(lse_out_host, m->topk_lse, sizeof(float) * rows, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

int wj_decode_topk_rules(wj_whisper* m, int rows, int k, const wj_decode_opts* opts, const int32_t* row_rules_host,
                         const int32_t* ban_host, int maxb, const int32_t* pen_host, int maxp, float penalty,
                         int32_t* ids_out_host, float* logprob_out_host, void* stream) {
  WJ_REQUIRE(m && opts && row_rules_host && ids_out_host && logprob_out_host, "wj_decode_topk_rules: NULL argument");
  WJ_REQUIRE(rows >= 1 && rows <= m->max_rows && k >= 1 && k <= 16, "wj_decode_topk_rules: rows/k out of range");
  WJ_REQUIRE(maxb >= 0 && maxp >= 0 && (maxb == 0 || ban_host) && (maxp == 0 || pen_host),
             "wj_decode_topk_rules: list pointers do not match their lengths");
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  const size_t b_rules = align_up(sizeof(int32_t) * 4 * rows, 256);
  const size_t b_ban = align_up(sizeof(int32_t) * (size_t)maxb * rows, 256);
  const size_t b_pen = align_up(sizeof(int32_t) * (size_t)maxp * rows, 256);
  WJ_TRY(m->ctx->ensure_scratch(b_rules + b_ban + b_pen));
  char* base = reinterpret_cast<char*>(m->ctx->scratch);
  int32_t* d_rules = reinterpret_cast<int32_t*>(base);
  int32_t* d_ban = reinterpret_cast<int32_t*>(base + b_rules);
  int32_t* d_pen = reinterpret_cast<int32_t*>(base + b_rules + b_ban);
  WJ_HIP(hipMemcpyAsync(d_rules, row_rules_host, sizeof(int32_t) * 4 * rows, hipMemcpyHostToDevice, s));
  if (maxb) WJ_HIP(hipMemcpyAsync(d_ban, ban_host, sizeof(int32_t) * (size_t)maxb * rows, hipMemcpyHostToDevice, s));
  if (maxp) WJ_HIP(hipMemcpyAsync(d_pen, pen_host, sizeof(int32_t) * (size_t)maxp * rows, hipMemcpyHostToDevice, s));
  WJ_TRY(launch_topk_rules(m->logits, m->ldl, rows, m->d.n_vocab, k, *opts, d_rules, d_ban, maxb, d_pen, maxp, penalty,
                           m->topk_ids, m->topk_lp, s));
  WJ_HIP(hipMemcpyAsync(ids_out_host, m->topk_ids, sizeof(int32_t) * rows * k, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipMemcpyAsync(logprob_out_host, m->topk_lp, sizeof(float) * rows * k, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

int wj_decode_no_speech(wj_whisper* m, int rows, int no_speech_id, float* out_host, void* stream) {
  WJ_REQUIRE(m && out_host && rows >= 1 && rows <= m->max_rows, "wj_decode_no_speech: bad arguments");
  WJ_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->pick(stream);
  WJ_TRY(launch_no_speech_prob(m->logits, m->ldl, rows, m->d.n_vocab, no_speech_id, m->nsp, s));
  WJ_HIP(hipMemcpyAsync(out_host, m->nsp, sizeof(float) * rows, hipMemcpyDeviceToHost, s));
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

}  // extern "C"

// wj_k_gemm / wj_k_gemm_timed, variant 88 (blocked operands, row-major output) and 89 (blocked output as well, converted
// back for the caller): the entry builds the blocked copies itself, so a test hands over ordinary row-major matrices and can
// compare the result bit for bit with the other kernels.  reps > 0: only the GEMM launches are timed.
static int k_gemm_blocked(wj_ctx* ctx, int dtype, const void* a_dev, const void* w_dev, const float* bias_dev, void* c_dev, int M,
                          int N, int K, int act_gelu, int out_f32, int variant, hipStream_t s, int reps, float* ms_per_launch) {
  WJ_REQUIRE(is16(dtype), "wj_k_gemm: blocked operands are a 16-bit feature");
  WJ_REQUIRE(!(variant == 89 && out_f32), "wj_k_gemm: variant 89 (blocked output) needs a 16-bit output");
  const int64_t mp = ((int64_t)M + 255) / 256 * 256;
  void *ab = nullptr, *wb = nullptr, *cb = nullptr;
  int rc = WJ_OK;
  auto cleanup = [&]() { if (ab) (void)hipFree(ab); if (wb) (void)hipFree(wb); if (cb) (void)hipFree(cb); };
  if (hipMalloc(&ab, (size_t)mp * K * 2) != hipSuccess || hipMalloc(&wb, (size_t)N * K * 2) != hipSuccess ||
      (variant == 89 && hipMalloc(&cb, (size_t)mp * N * 2) != hipSuccess)) {
    cleanup();
    set_error("wj_k_gemm: out of device memory for the blocked copies");
    return WJ_E_HIP;
  }
  rc = launch_to_blocked(a_dev, K, M, K, ab, s);
  if (!rc) rc = launch_to_blocked(w_dev, K, N, K, wb, s);
  GemmArgs g;
  g.A = ab; g.W = wb; g.bias = bias_dev; g.M = M; g.N = N; g.K = K; g.out = variant == 89 ? cb : c_dev; g.ldc = N;
  g.blk = 1; g.out_blk = variant == 89;
  const Epi e = out_f32 ? EPI_F32 : (act_gelu ? EPI_GELU_T : EPI_T);
  if (!rc) rc = launch_gemm(dtype, e, g, s, 0);
  if (!rc && reps > 0) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps && !rc; ++i) rc = launch_gemm(dtype, e, g, s, 0);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_launch = ms / reps;
  }
  if (!rc && variant == 89) rc = launch_from_blocked(cb, M, N, c_dev, N, s);
  (void)hipStreamSynchronize(s);
  cleanup();
  return rc;
}

extern "C" {

// ---- kernel-level entry points (tests / micro-benchmarks) ---------------------------------------
int wj_k_gemm(wj_ctx* ctx, int dtype, const void* a_dev, const void* w_dev, const float* bias_dev, void* c_dev, int M,
              int N, int K, int act_gelu, int out_f32, int variant, void* stream) {
  WJ_REQUIRE(ctx && a_dev && w_dev && c_dev, "wj_k_gemm: NULL argument");
  WJ_HIP(hipSetDevice(ctx->device));
  GemmArgs g;
  g.A = a_dev; g.lda = K; g.W = w_dev; g.ldw = K; g.bias = bias_dev; g.M = M; g.N = N; g.K = K; g.out = c_dev; g.ldc = N;
  Epi e = out_f32 ? EPI_F32 : (act_gelu ? EPI_GELU_T : EPI_T);
  WJ_REQUIRE(!(out_f32 && act_gelu), "wj_k_gemm: gelu with f32 output is not a fused variant");
  if (variant == 88 || variant == 89)
    return k_gemm_blocked(ctx, dtype, a_dev, w_dev, bias_dev, c_dev, M, N, K, act_gelu, out_f32, variant, ctx->pick(stream), 0, nullptr);
  return launch_gemm(dtype, e, g, ctx->pick(stream), variant);
}

int wj_k_gemm_mx8(wj_ctx* ctx, int dtype, const float* a_f32_dev, const float* w_f32_dev, const float* bias_dev, void* c_dev, int M, int N,
                  int K, int out_f32, uint8_t* a8_out_dev, uint8_t* a_scale_out_dev, uint8_t* w8_out_dev, uint8_t* w_scale_out_dev, int reps,
                  float* ms_per_launch, void* stream) {
  WJ_REQUIRE(ctx && a_f32_dev && w_f32_dev && c_dev, "wj_k_gemm_mx8: NULL argument");
  WJ_REQUIRE(is16(dtype) && M >= 1 && N >= 1 && K >= 128 && K % 128 == 0, "wj_k_gemm_mx8: 16-bit output types, K a multiple of 128");
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->pick(stream);
  uint8_t *a8 = nullptr, *sa = nullptr, *w8 = nullptr, *sw = nullptr;
  auto cleanup = [&]() { (void)hipFree(a8); (void)hipFree(sa); (void)hipFree(w8); (void)hipFree(sw); };
  if (hipMalloc(&a8, (size_t)M * K) != hipSuccess || hipMalloc(&sa, (size_t)M * K / 32) != hipSuccess ||
      hipMalloc(&w8, (size_t)N * K) != hipSuccess || hipMalloc(&sw, (size_t)N * K / 32) != hipSuccess) {
    cleanup();
    set_error("wj_k_gemm_mx8: out of device memory");
    return WJ_E_HIP;
  }
  int rc = launch_mx8_quantize(WJ_F32, a_f32_dev, K, M, K, a8, sa, s);
  if (!rc) rc = launch_mx8_quantize(WJ_F32, w_f32_dev, K, N, K, w8, sw, s);
  GemmArgs g;
  g.A = a8; g.lda = K; g.W = w8; g.ldw = K; g.a_scale = sa; g.w_scale = sw; g.mx8 = 1; g.bias = bias_dev;
  g.M = M; g.N = N; g.K = K; g.out = c_dev; g.ldc = N;
  const Epi e = out_f32 ? EPI_F32 : EPI_T;
  if (!rc) rc = launch_gemm(dtype, e, g, s, 0);
  if (!rc && reps > 0 && ms_per_launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps && !rc; ++i) rc = launch_gemm(dtype, e, g, s, 0);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_launch = ms / reps;
  }
  if (!rc && a8_out_dev) rc = hipMemcpyAsync(a8_out_dev, a8, (size_t)M * K, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : WJ_E_HIP;
  if (!rc && a_scale_out_dev) rc = hipMemcpyAsync(a_scale_out_dev, sa, (size_t)M * K / 32, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : WJ_E_HIP;
  if (!rc && w8_out_dev) rc = hipMemcpyAsync(w8_out_dev, w8, (size_t)N * K, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : WJ_E_HIP;
  if (!rc && w_scale_out_dev) rc = hipMemcpyAsync(w_scale_out_dev, sw, (size_t)N * K / 32, hipMemcpyDeviceToDevice, s) == hipSuccess ? 0 : WJ_E_HIP;
  (void)hipStreamSynchronize(s);
  cleanup();
  return rc;
}

int wj_k_gemm_split(wj_ctx* ctx, int dtype, const float* a_f32_dev, const void* w_dev, const float* bias_dev, float* c_dev,
                    int M, int N, int K, int variant, void* stream) {
  WJ_REQUIRE(ctx && a_f32_dev && w_dev && c_dev, "wj_k_gemm_split: NULL argument");
  WJ_REQUIRE(is16(dtype) && M >= 1 && N >= 1 && K >= 64, "wj_k_gemm_split: 16-bit dtypes only");
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->pick(stream);
  WJ_TRY(ctx->ensure_scratch((size_t)M * 2 * K * 2));
  const unsigned blocks = (unsigned)std::min<int64_t>(4096, ceil_div64((int64_t)M * K, 256));
  if (dtype == WJ_F16)
    hipLaunchKernelGGL(split_rows_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, a_f32_dev, (f16_t*)ctx->scratch, (int64_t)M, K);
  else
    hipLaunchKernelGGL(split_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, a_f32_dev, (bf16_t*)ctx->scratch, (int64_t)M, K);
  WJ_LAUNCH_CHECK();
  GemmArgs g;
  g.A = ctx->scratch; g.lda = 2 * (int64_t)K; g.split = 1; g.W = w_dev; g.ldw = K; g.bias = bias_dev;
  g.M = M; g.N = N; g.K = K; g.out = c_dev; g.ldc = N;
  return launch_gemm(dtype, EPI_F32, g, s, variant);
}

int wj_k_gemm_timed(wj_ctx* ctx, int dtype, const void* a_dev, const void* w_dev, const float* bias_dev, void* c_dev,
                    int M, int N, int K, int act_gelu, int out_f32, int variant, int reps, float* ms_per_launch) {
  WJ_REQUIRE(ctx && a_dev && w_dev && c_dev && ms_per_launch && reps >= 1, "wj_k_gemm_timed: bad arguments");
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  GemmArgs g;
  g.A = a_dev; g.lda = K; g.W = w_dev; g.ldw = K; g.bias = bias_dev; g.M = M; g.N = N; g.K = K; g.out = c_dev; g.ldc = N;
  Epi e = out_f32 ? EPI_F32 : (act_gelu ? EPI_GELU_T : EPI_T);
  if (variant == 88 || variant == 89)
    return k_gemm_blocked(ctx, dtype, a_dev, w_dev, bias_dev, c_dev, M, N, K, act_gelu, out_f32, variant, s, reps, ms_per_launch);
  int rc = launch_gemm(dtype, e, g, s, variant);  // warm-up + argument validation
  if (rc) return rc;
  hipEvent_t e0, e1;
  WJ_HIP(hipEventCreate(&e0));
  WJ_HIP(hipEventCreate(&e1));
  WJ_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) {
    rc = launch_gemm(dtype, e, g, s, variant);
    if (rc) return rc;
  }
  WJ_HIP(hipEventRecord(e1, s));
  WJ_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  WJ_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = ms / reps;
  return WJ_OK;
}

int wj_k_layernorm(wj_ctx* ctx, int dtype, const float* x_dev, const float* w_dev, const float* b_dev, void* out_dev, int M,
                   int D, void* stream) {
  WJ_REQUIRE(ctx && x_dev && w_dev && b_dev && out_dev, "wj_k_layernorm: NULL argument");
  WJ_HIP(hipSetDevice(ctx->device));
  return launch_layernorm(dtype, x_dev, w_dev, b_dev, out_dev, M, D, ctx->pick(stream));
}

}  // extern "C"

// helpers for the test entry points ------------------------------------------------------------------
template <typename T>
__global__ void qkv_split_kernel(const float* __restrict__ qkv, T* __restrict__ Q, T* __restrict__ K, T* __restrict__ Vt,
                                 int Tn, int Tpad, int H) {
  const int b = blockIdx.y, t = blockIdx.x, D = H * 64;
  const float* row = qkv + ((int64_t)b * Tn + t) * 3 * D;
  for (int c = threadIdx.x; c < D; c += 256) {
    const int h = c >> 6, dd = c & 63;
    const int64_t bh = (int64_t)b * H + h;
    Elem<T>::st(Q + (bh * Tpad + t) * 64 + dd, row[c]);
    Elem<T>::st(K + (bh * Tpad + t) * 64 + dd, row[D + c]);
    Elem<T>::st(Vt + (bh * 64 + dd) * Tpad + t, row[2 * D + c]);
  }
}
// float32 [G][H][n_keys][64] -> bf16 [G][H][64][kp] (pad columns are left as they are: the caller zeroes them)
template <typename T>
__global__ void v_transpose_h_kernel(const float* in, T* out, int64_t n, int n_keys, int kp) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int dd = (int)(i & 63);
    const int64_t t = i >> 6;
    const int key = (int)(t % n_keys);
    const int64_t gh = t / n_keys;
    Elem<T>::st(out + (gh * 64 + dd) * kp + key, in[i]);
  }
}

template <typename T>
__global__ void T_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = Elem<T>::ld(in + i);
}

extern "C" {

int wj_k_attention_enc(wj_ctx* ctx, int dtype, const float* qkv_f32_dev, void* out_dev, int B, int T, int H, void* stream) {
  WJ_REQUIRE(ctx && qkv_f32_dev && out_dev, "wj_k_attention_enc: NULL argument");
  WJ_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->pick(stream);
  const int Tpad = (T + 127) / 128 * 128;
  const size_t esz = dtype_size(dtype);
  const size_t one = align_up((size_t)B * H * Tpad * 64 * esz, 256);
  WJ_TRY(ctx->ensure_scratch(3 * one));
  char* base = reinterpret_cast<char*>(ctx->scratch);
  WJ_HIP(hipMemsetAsync(base, 0, 3 * one, s));
  if (dtype == WJ_F32)
    hipLaunchKernelGGL(qkv_split_kernel<float>, dim3(T, B), dim3(256), 0, s, qkv_f32_dev, (float*)base, (float*)(base + one),
                       (float*)(base + 2 * one), T, Tpad, H);
  else if (dtype == WJ_F16)
    hipLaunchKernelGGL(qkv_split_kernel<f16_t>, dim3(T, B), dim3(256), 0, s, qkv_f32_dev, (f16_t*)base,
                       (f16_t*)(base + one), (f16_t*)(base + 2 * one), T, Tpad, H);
  else
    hipLaunchKernelGGL(qkv_split_kernel<bf16_t>, dim3(T, B), dim3(256), 0, s, qkv_f32_dev, (bf16_t*)base,
                       (bf16_t*)(base + one), (bf16_t*)(base + 2 * one), T, Tpad, H);
  WJ_LAUNCH_CHECK();
  return launch_attention_enc(dtype, base, base + one, base + 2 * one, out_dev, B, T, Tpad, H, s);
}

int wj_k_attention_enc_timed(wj_ctx* ctx, int dtype, const float* qkv_f32_dev, void* out_dev, int B, int T, int H,
                             int reps, float* ms_per_launch) {
  WJ_REQUIRE(ctx && qkv_f32_dev && out_dev && ms_per_launch && reps >= 1, "wj_k_attention_enc_timed: bad arguments");
  int rc = wj_k_attention_enc(ctx, dtype, qkv_f32_dev, out_dev, B, T, H, nullptr);   // builds Q/K/Vt in the scratch + warm-up
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  const int Tpad = (T + 127) / 128 * 128;
  const size_t esz = dtype_size(dtype);
  const size_t one = align_up((size_t)B * H * Tpad * 64 * esz, 256);
  char* base = reinterpret_cast<char*>(ctx->scratch);
  hipEvent_t e0, e1;
  WJ_HIP(hipEventCreate(&e0));
  WJ_HIP(hipEventCreate(&e1));
  WJ_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) {
    rc = launch_attention_enc(dtype, base, base + one, base + 2 * one, out_dev, B, T, Tpad, H, s);
    if (rc) return rc;
  }
  WJ_HIP(hipEventRecord(e1, s));
  WJ_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  WJ_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_per_launch = ms / reps;
  return WJ_OK;
}

static int attention_dec_common(wj_ctx* ctx, int dtype, const float* q_dev, const float* k_dev, const float* v_dev,
                                float* out_dev, int G, int nb, int H, int n_keys, int layout, int reps, float* ms,
                                hipStream_t s) {
  const size_t esz = dtype_size(dtype);
  const bool vt = is16(dtype) && layout != 1;   // the engine's 16-bit layout: V transposed, MFMA kernel
  const int kp = (n_keys + 31) / 32 * 32;
  const int64_t nq = (int64_t)G * nb * H * 64, nkv = (int64_t)G * H * n_keys * 64, nvt = (int64_t)G * H * 64 * kp;
  const size_t bq = align_up(nq * esz, 256), bk = align_up(nkv * esz, 256), bv = align_up((vt ? nvt : nkv) * esz, 256);
  WJ_TRY(ctx->ensure_scratch(2 * bq + bk + bv));
  char* base = reinterpret_cast<char*>(ctx->scratch);
  void *tq = base, *tk = base + bq, *tv = base + bq + bk, *to = base + bq + bk + bv;
  WJ_TRY(launch_f32_to_T(dtype, q_dev, tq, nq, s));
  WJ_TRY(launch_f32_to_T(dtype, k_dev, tk, nkv, s));
  if (vt) {
    WJ_HIP(hipMemsetAsync(tv, 0, bv, s));
    if (dtype == WJ_F16)
      hipLaunchKernelGGL(v_transpose_h_kernel<f16_t>, dim3((unsigned)min((int64_t)4096, ceil_div64(nkv, 256))), dim3(256), 0, s,
                         v_dev, reinterpret_cast<f16_t*>(tv), nkv, n_keys, kp);
    else
      hipLaunchKernelGGL(v_transpose_h_kernel<bf16_t>, dim3((unsigned)min((int64_t)4096, ceil_div64(nkv, 256))), dim3(256), 0, s,
                         v_dev, reinterpret_cast<bf16_t*>(tv), nkv, n_keys, kp);
    WJ_LAUNCH_CHECK();
  } else {
    WJ_TRY(launch_f32_to_T(dtype, v_dev, tv, nkv, s));
  }
  DecAttnArgs a;
  a.q = tq; a.K = tk; a.V = tv; a.out = to; a.G = G; a.nb = nb; a.H = H; a.n_keys = n_keys; a.kv_stride = n_keys;
  a.vt_stride = vt ? kp : 0;
  WJ_TRY(launch_attention_dec(dtype, a, s));
  if (reps > 0) {
    hipEvent_t e0, e1;
    WJ_HIP(hipEventCreate(&e0)); WJ_HIP(hipEventCreate(&e1));
    WJ_HIP(hipEventRecord(e0, s));
    int rc = WJ_OK;
    for (int i = 0; i < reps && rc == WJ_OK; ++i) rc = launch_attention_dec(dtype, a, s);
    WJ_HIP(hipEventRecord(e1, s));
    WJ_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    WJ_HIP(hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc != WJ_OK) return rc;
    if (ms) *ms = t / reps;
  }
  const int blocks = (int)min((int64_t)1024, ceil_div64(nq, 256));
  if (dtype == WJ_F32)
    hipLaunchKernelGGL(T_to_f32_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)to, out_dev, nq);
  else if (dtype == WJ_F16)
    hipLaunchKernelGGL(T_to_f32_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, (const f16_t*)to, out_dev, nq);
  else
    hipLaunchKernelGGL(T_to_f32_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)to, out_dev, nq);
  WJ_LAUNCH_CHECK();
  return WJ_OK;
}

int wj_k_attention_dec(wj_ctx* ctx, int dtype, const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev,
                       int G, int nb, int H, int n_keys, void* stream) {
  WJ_REQUIRE(ctx && q_dev && k_dev && v_dev && out_dev, "wj_k_attention_dec: NULL argument");
  WJ_HIP(hipSetDevice(ctx->device));
  return attention_dec_common(ctx, dtype, q_dev, k_dev, v_dev, out_dev, G, nb, H, n_keys, 0, 0, nullptr, ctx->pick(stream));
}

int wj_k_attention_dec_timed(wj_ctx* ctx, int dtype, const float* q_dev, const float* k_dev, const float* v_dev,
                             float* out_dev, int G, int nb, int H, int n_keys, int layout, int reps, float* ms_per_launch) {
  WJ_REQUIRE(ctx && q_dev && k_dev && v_dev && out_dev && ms_per_launch && reps > 0, "wj_k_attention_dec_timed: bad argument");
  WJ_HIP(hipSetDevice(ctx->device));
  return attention_dec_common(ctx, dtype, q_dev, k_dev, v_dev, out_dev, G, nb, H, n_keys, layout, reps, ms_per_launch,
                              ctx->pick(nullptr));
}

}  // extern "C"
