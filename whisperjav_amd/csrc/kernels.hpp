// kernels.hpp -- internal launcher interface between the engine and the kernel translation units.
#pragma once

#include "common.hpp"

namespace wj {

// ---------------- GEMM: C = A[M,K] . W[N,K]^T (+bias) with fused epilogues ------------------
enum Epi : int {
  EPI_T = 0,         // out T  [z][M][ldc]  = acc + bias
  EPI_GELU_T,        // out T               = gelu(acc + bias)
  EPI_F32,           // out f32 [z][M][ldc] = acc + bias            (N tail safe)
  EPI_RESID_F32,     // out f32 [z][M][ldc] += acc + bias           (residual stream)
  EPI_GELU_POS_F32,  // out f32 [z][M][ldc] = gelu(acc + bias) + pos[m][n]   (conv2 + positional)
  EPI_QK_HEADS,      // n <  D : out [z][h][m][64] (Q) ; n >= D : out2 [z][h][m][64] (K); rows/head = Tpad
  EPI_VT,            // out [z][h][dd][Tpad]  (V transposed per head; "MN" accumulator orientation)
  EPI_QKV_DEC,       // decode step: n<D: out T [M][D]; then K,V -> caches out2/out3 [m][h][cache_len][64] at *pos_ptr
  EPI_CKV,           // cross K/V: n<D: out [z][h][m][64] ; n>=D: out2 [z][h][m][64]; rows/head = Tpad
  EPI_PARTIAL_F32,   // split-K slab: out f32 [ksplit][M][ldc] = raw partial sums (no bias); reduced by the consumer
  EPI_SWIGLU_T,      // out T [M][ldc] (N / 2 columns) = silu(gate) * up; W's rows are [16 gate | 16 up] blocks (column 32 b + c = gate
                     // 16 b + c, 32 b + 16 + c = up 16 b + c), so a lane holds a gate / up pair in neighbouring fragments; MFMA tile
                     // kernels of the 16-bit types only (round 5: the Qwen decoder's gate-up projection), no bias; split_out = lo at + N / 2
  EPI_COUNT
};

struct GemmArgs {
  const void* A = nullptr;   // T, row stride lda (elements), batch stride a_batch
  int64_t lda = 0, a_batch = 0;
  const void* W = nullptr;   // T [N][ldw]
  int64_t ldw = 0;
  const float* bias = nullptr;
  int M = 0, N = 0, K = 0, nbatch = 1;
  void* out = nullptr;
  int64_t ldc = 0, c_batch = 0;
  void* out2 = nullptr;
  void* out3 = nullptr;
  const float* pos = nullptr;   // [M][N] fp32
  int D = 0, H = 0, Tpad = 0;
  const int* pos_ptr = nullptr;
  int cache_len = 0;
  int ksplit = 1;               // EPI_PARTIAL_F32: K is cut into ksplit slices (grid.z resp. grid.y)
  int seq_tp = 0;               // EPI_QKV_DEC, > 0: row m is position m % seq_tp of cache row m / seq_tp (full-sequence pass)
  // Split activations (16-bit decode GEMMs, wj_tune "dec_split_act"): row m of A holds [hi(K) | lo(K)] with
  // hi = T(x), lo = T(x - hi), i.e. x to ~22 bits; C = W.hi + W.lo in the same fp32 accumulator.  lda >= 2 K.
  int split = 0;
  // EPI_T / EPI_GELU_T: also store the rounding residual of the output at column offset N of the same row
  // (the consumer GEMM reads it as a split activation; ldc >= 2 N)
  int split_out = 0;
  // 2-byte row-major / head-split epilogues of the MFMA tile kernels: 16-byte stores after a v_permlane16_swap of
  // neighbouring fragments (set by launch_gemm from wj_tune "epi_wide"; 0 = the 8-byte stores, for A/B and cross-checks)
  int epi_wide = 1;
  // Blocked operands (round 4, encoder GEMMs): A is [ceil(M / 256)][K / 32][256][32], W is [N / 256][K / 32][256][32]
  // (lda / ldw / a_batch unused); out_blk: the EPI_T / EPI_GELU_T output is written in the same layout with N columns
  // (it is the next GEMM's A).  seq_T > 0: flat rows -- the head-split epilogues (EPI_QK_HEADS / EPI_VT / EPI_CKV) take
  // window z = m / seq_T and position m % seq_T from the row index instead of from the batch dimension.
  int blk = 0, out_blk = 0, seq_T = 0;
  float inv_seq_T = 0.f;          // set by launch_gemm
  int gm = 8;                     // row tiles per group of the blocked kernel's tile order (set by its launcher)
  // MX-fp8 operands (round 4, the Qwen decoder's WJ_F8W type): A and W are OCP e4m3 bytes [rows][K] (lda / ldw in BYTES) with one
  // E8M0 scale byte per 32-element block, a_scale [M][K / 32] and w_scale [N][K / 32]; the product runs on
  // v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulation, the epilogue is the 16-bit one of the dtype passed to launch_gemm
  int mx8 = 0;
  const uint8_t* a_scale = nullptr;
  const uint8_t* w_scale = nullptr;
};

// variant: 0 = auto; 1 = tiled MFMA kernel (default staging); 2 = skinny (decode) kernel;
//          3 = tiled kernel with LDS-DMA staging; 4 = tiled kernel with register staging
int launch_gemm(int dtype, Epi epi, const GemmArgs& a, hipStream_t s, int variant = 0);

// row-major 2-byte [rows][ld] -> blocked [ceil(rows / 256)][cols / 32][256][32] (pad rows zero-filled) and back
int launch_to_blocked(const void* src, int64_t ld, int rows, int cols, void* dst, hipStream_t s);
int launch_from_blocked(const void* src, int rows, int cols, void* dst, int64_t ld, hipStream_t s);

// MX quantisation of 2-byte / fp32 rows: out8 [rows][K] e4m3 bytes, scales [rows][K / 32] E8M0 bytes (OCP MX: the block's scale is
// 2^(floor(log2 amax) - 8), elements round to nearest even and saturate at +-448).  src_dtype WJ_F32 / WJ_F16 / WJ_BF16, ld in elements.
int launch_mx8_quantize(int src_dtype, const void* src, int64_t ld, int rows, int K, uint8_t* out8, uint8_t* scales, hipStream_t s);

// second half of a split-K GEMM whose epilogue has no consumer kernel to fold the reduction into:
// out = EPI(bias + sum_s slab[s][M][N]); `a` carries the epilogue operands exactly as for launch_gemm
int launch_splitk_reduce(int dtype, Epi epi, const GemmArgs& a, const float* slab, int ks, hipStream_t s);

// ---------------- normalisation / elementwise ------------------------------------------------
// split != 0 (16-bit types): out rows are [hi(D) | lo(D)] (row stride 2 D), see GemmArgs::split
// blk != 0 (16-bit types, D % 256 == 0): out is written in the blocked GEMM-operand layout (GemmArgs::blk)
extern int g_ln_vec;
int launch_layernorm(int dtype, const float* x, const float* w, const float* b, void* out, int M, int D,
                     hipStream_t s, int split = 0, int blk = 0);
// residual update fused with LayerNorm: x[m][:] += bias + sum_s partial[s][m][:]  (fixed order -> deterministic),
// then out = LayerNorm(x).  Consumer side of the split-K decode GEMMs.
int launch_layernorm_resid(int dtype, float* x, const float* partial, int ksplit, const float* bias, const float* w,
                           const float* b, void* out, int M, int D, hipStream_t s, int split = 0);
// mel f32 [B][n_mels][frames] -> engine layout T [B][frames+2][n_mels] (row 0 and frames+1 stay zero)
int launch_mel_to_rows(int dtype, const float* mel, void* out, int B, int n_mels, int frames, hipStream_t s);
// decoder embedding: x[r][:] = tok_emb[token[r]][:] + pos_emb[*pos][:]
int launch_embed_seq(int dtype, const void* tok_emb, const float* pos_emb, const int32_t* tokens, int64_t tok_stride,
                     int Tp, float* x, int B, int D, hipStream_t s);
int launch_embed(int dtype, const void* tok_emb, const float* pos_emb, const int32_t* tokens, int64_t tok_stride,
                 const int* pos_ptr, float* x, int R, int D, hipStream_t s);
int launch_f32_to_T(int dtype, const float* in, void* out, int64_t n, hipStream_t s);

// ---------------- attention --------------------------------------------------------------------
// encoder: Q,K [B][H][Tpad][64], Vt [B][H][64][Tpad] (T dtype) -> out T [B][T][H*64]
extern int g_attn_enc_variant;
// out_blk != 0 (16-bit): out is written in the blocked GEMM-operand layout (GemmArgs::blk) over B*T rows, H*64 columns
int launch_attention_enc(int dtype, const void* Q, const void* K, const void* Vt, void* out, int B, int T,
                         int Tpad, int H, hipStream_t s, int out_blk = 0);
// decode: q T [G*nb][H*64]; K,V laid out [group][H][kv_stride][64]. n_keys from n_keys_ptr (device,
// +1 applied when SELF) or the constant n_keys.  row_map (device, may be NULL): for SELF attention,
// src_row[r][j] = physical cache row holding position j of logical row r (beam indirection).
struct DecAttnArgs {
  const void* q = nullptr;
  const void* K = nullptr;
  const void* V = nullptr;
  void* out = nullptr;
  int G = 0, nb = 1, H = 0;
  int n_keys = 0;
  const int* n_keys_ptr = nullptr;   // if set: n_keys = *ptr + 1
  int kv_stride = 0;                 // positions allocated per (group, head)
  const int32_t* row_map = nullptr;  // [G][kv_stride]
  const int32_t* group_of = nullptr; // [G] window slot per group (cross attention), NULL = identity
  int vt_stride = 0;                 // > 0: bf16 cross attention on the matrix cores; V is [group][H][64][vt_stride]
  // Fused split-K consumer (bf16): the projection that feeds this attention left `slab_ks` raw fp32 K-slices
  // [slab_ks][slab_rows][slab_ld] instead of q (cross: slab_ld = D) or q|k|v (self: slab_ld = 3 D); the kernel
  // sums them in slice order, adds slab_bias, rounds to bf16 exactly like the projection's own epilogue would,
  // and (self) appends k, v to the cache at position *n_keys_ptr of physical row row_base + group.
  const float* slab = nullptr;
  const float* slab_bias = nullptr;
  int slab_ks = 0, slab_rows = 0, slab_ld = 0, row_base = 0;
  // Word-timestamp alignment: cross-attention heads with dump_sel[h] >= 0 copy their scaled scores to
  // dump[query row][dump_sel[h]][*dump_pos_ptr][n_keys] (query row counted from dump_row_base)
  float* dump = nullptr;
  const int32_t* dump_sel = nullptr;   // [H] device
  const int* dump_pos_ptr = nullptr;
  int dump_nsel = 0, dump_tmax = 0, dump_row_base = 0;
  // Full-sequence (teacher-forced) pass.  Self attention: seq_tp > 0 -> query row g is position g % seq_tp of cache
  // row g / seq_tp and sees keys 0..position.  Cross attention dump: dump_chunks > 0 -> group g is chunk
  // g % dump_chunks of window g / dump_chunks, query b of it is position chunk * nb + b.
  int seq_tp = 0, dump_chunks = 0;
  // out_split != 0 (16-bit types): output rows are [hi(D) | lo(D)] (row stride 2 D) for a split-activation GEMM
  int out_split = 0;
};
extern int g_dec_cross_u;
extern int g_dec_cross_nt;
extern int g_gemm_big;
extern int g_tile_l2_kb;   // wj_tune("tile_l2_kb"), gemm.hip
extern int g_ppb_ns, g_ppb_gm;
extern int g_qwen_split_act, g_qwen_compact_pct, g_qwen_prompt_mfma, g_qwen_splitk, g_qwen_fuse_swiglu;
extern int g_qwen_conv_kpad, g_qwen_tower_split;                          // qwen_audio.hip      // qwen.hip
extern int g_epi_wide;
int launch_attention_dec(int dtype, const DecAttnArgs& a, hipStream_t s);

// ---------------- word-timestamp alignment (align.hip) ------------------------------------------------
int launch_align_token_prob(const float* logits, int64_t ldl, int limit, const int32_t* tokens, int64_t tok_stride,
                            const int* pos_ptr, int n0, float* prob_out, int tmax, int R, hipStream_t s);
int launch_align_token_prob_seq(const float* logits, int64_t ldl, int limit, const int32_t* tokens, int64_t tok_stride,
                                int row0, int rows, int Tp, int n0, const int32_t* n_tok, float* prob_out, hipStream_t s);
int launch_align_post(float* qk, float* matrix, int8_t* trace, const int32_t* n_tok, const int32_t* nf2, int R, int nsel,
                      int tmax, int nctx, int n0, int width, int32_t* path_text, int32_t* path_time, int32_t* path_len,
                      hipStream_t s);

// ---------------- sampling -----------------------------------------------------------------------
struct GreedyArgs {
  float* logits = nullptr;        // [R][ldl] (processors edit them in place)
  int64_t ldl = 0;
  int R = 0, V = 0;
  int32_t* tokens = nullptr;      // [R][tok_stride] full history (prompt + sampled)
  int64_t tok_stride = 0;
  const int* pos_ptr = nullptr;   // device: index of the LAST token fed (history length - 1)
  int sample_begin = 0;           // prompt length
  float* sum_logprob = nullptr;   // [R]
  float* token_logprob = nullptr; // [R][tok_stride] or NULL
  int32_t* finished = nullptr;    // [R]
  wj_decode_opts opts;
  float temperature = 0.f;        // > 0: sample from softmax(filtered logits / T) (Gumbel-max), log-probs stay unscaled
  uint32_t seed = 0;
  int row_offset = 0;             // absolute index of row 0 (keeps the random stream independent of chains)
};
int launch_greedy_sample(const GreedyArgs& a, hipStream_t s);
int launch_no_speech_prob(const float* logits, int64_t ldl, int R, int V, int no_speech_id, float* out,
                          hipStream_t s);
// masked log-softmax + top-k (k <= 16) per row
int launch_topk_logprob(const float* logits, int64_t ldl, int R, int V, int k, const uint8_t* ban,
                        int32_t* ids, float* logprobs, float* lse, hipStream_t s);
// device-resident beam search step (see sampler.hip)
struct BeamArgs {
  float* logits = nullptr;            // [R][ldl], edited in place by the processors
  int64_t ldl = 0;
  int V = 0, K = 0;                   // vocabulary, beam size (rows per window)
  const int32_t* hist_in = nullptr;   // [R][tok_stride] token history read this step (prompt + generated)
  int32_t* hist_out = nullptr;        // ... written for the next step (gathered by parent + new token)
  int64_t tok_stride = 0;
  const int* pos_ptr = nullptr;       // device: index of the token fed this step
  int sample_begin = 0, max_new = 0, max_candidates = 0;
  wj_decode_opts opts;
  int32_t* cand_ids = nullptr;        // [R][16]
  float* cand_lp = nullptr;           // [R][16]
  float* score = nullptr;             // [R] cumulative log-prob of the live beams (-inf = dead)
  int32_t* parent = nullptr;          // [R] absolute parent row of next step's row
  const int32_t* win_ids = nullptr;   // [B] or NULL: window id of logical window w in the per-window arrays below (the
                                      // search compacts finished windows out of the batch; their lists stay where they were)
  int32_t* done = nullptr;            // [B]
  int32_t* n_done = nullptr;          // [1]
  int32_t* fin_count = nullptr;       // [B]
  float* fin_score = nullptr;         // [B][fin_cap]
  int32_t* fin_len = nullptr;         // [B][fin_cap]
  int32_t* fin_tokens = nullptr;      // [B][fin_cap][tok_stride]
  // optional (NULL = off): cumulative log-prob after every history position, gathered like the tokens -- the per-token
  // log-probs of the winning hypothesis are its differences (wj_whisper_last_beam_token_logprobs)
  const float* cum_in = nullptr;      // [R][tok_stride]
  float* cum_out = nullptr;
  float* fin_cum = nullptr;           // [B][fin_cap][tok_stride]: entries 0..len-1 after each generated token, entry len = the total
  int fin_cap = 0;
  int flavor = 0;                     // 0 = CTranslate2 (faster-whisper), 1 = openai-whisper BeamSearchDecoder
};
extern int g_beam_topk_reg;
int launch_beam_step(const BeamArgs& a, int R, int B, hipStream_t s);

// beam search: logits processors + timestamp rules + masked log-softmax + top-k (see sampler.hip)
int launch_topk_rules(float* logits, int64_t ldl, int R, int V, int k, const wj_decode_opts& o, const int32_t* row_rules,
                      const int32_t* ban, int maxb, const int32_t* pen, int maxp, float penalty, int32_t* ids,
                      float* logprobs, hipStream_t s);
int launch_advance_pos(int* pos_ptr, hipStream_t s);
// row_map update for beam search: new_map[r][0..pos-1] = old_map[parent[r]][..]; new_map[r][pos] = r
// beam-search compaction: logical row r of the shrunk batch continues old logical row src_rows[r] (history, score, row map)
int launch_compact_rows(const int32_t* src_rows, int R_new, const int32_t* old_map, int32_t* new_map, int map_stride,
                        const int* pos_ptr, const int32_t* old_hist, int32_t* new_hist, int64_t tok_stride,
                        const float* old_score, float* new_score, hipStream_t s);
int launch_rebind_rows(const int32_t* old_map, int32_t* new_map, const int32_t* parent, const int* pos_ptr,
                       int R, int stride, hipStream_t s);

// ---------------- profiler (engine.hip) -------------------------------------------------------
enum ProfTag : int {
  PT_MEL_ROWS = 0, PT_CONV1, PT_CONV2, PT_E_LN, PT_E_QK, PT_E_V, PT_E_ATTN, PT_E_OUT, PT_E_FC1, PT_E_FC2, PT_E_CKV,
  PT_D_EMBED, PT_D_LN, PT_D_QKV, PT_D_SELF, PT_D_OUT, PT_D_CQ, PT_D_CROSS, PT_D_COUT, PT_D_FC1, PT_D_FC2, PT_D_LOGITS,
  PT_D_SAMPLE, PT_D_MISC, PT_COUNT
};
void prof_begin(wj_ctx* ctx, int tag, hipStream_t s);
void prof_end(wj_ctx* ctx, hipStream_t s);
inline bool prof_on(const wj_ctx* ctx) { return ctx->prof != nullptr; }

// ---------------- log-mel ----------------------------------------------------------------------
int logmel_run(wj_ctx* ctx, const float* pcm, const int64_t* offsets_host, int n_clips, int n_mels, int mode,
               int out_frames, float* out, hipStream_t s);

}  // namespace wj
