/*
 * wjhip.h -- C ABI of libwjhip.so, the MI355X (gfx950) implementation of WhisperJAV's
 * balanced/fidelity hot path: log-mel -> VAD scorer -> Whisper encoder -> decode.
 *
 * The reference (meizhong986/WhisperJAV v1.8.14) has NO native boundary: every FLOP of this
 * path runs inside third-party wheels that the reference calls through Python objects
 * (SURVEY.md section 8b).  Each entry point below therefore cites the *Python call site* it
 * replaces; whisperjav_amd/ (ctypes) re-creates those Python seams on top of this header.
 *
 * Conventions: plain pointers and sizes only (no torch / HIP types); every `*_dev` pointer is
 * a device (HBM) address on the context's GPU; `stream` is a hipStream_t passed as void*
 * (NULL = the context's own stream); every function returns 0 on success or a negative
 * WJ_E_* code, with a thread-local message available from wj_last_error().  No exceptions
 * cross the boundary; nothing runs at exit (the reference leaves via os._exit,
 * whisperjav/main.py:2490-2494).  One context per process per GPU, spawn-safe.
 */
#ifndef WJHIP_H
#define WJHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WJ_ABI_VERSION 6

enum {
  WJ_OK = 0,
  WJ_E_INVALID = -1,   /* bad argument */
  WJ_E_HIP = -2,       /* HIP runtime error (message has the hipError string) */
  WJ_E_NOMEM = -3,
  WJ_E_STATE = -4,     /* call sequence error */
  WJ_E_UNSUPPORTED = -5
};

/* compute/storage type of matrices & activations.  WJ_F32: exact fp32 kernels.  WJ_BF16 / WJ_F16: 16-bit matrices and
 * GEMM / attention operands on the matrix cores with fp32 accumulation and an fp32 residual stream, LayerNorm,
 * softmax and logits.  WJ_F16 is the arithmetic the reference runs on a GPU (ctranslate2 compute_type="float16",
 * whisper fp16=True: whisperjav/modules/whisper_pro_asr.py:201-218). */
enum { WJ_F32 = 0, WJ_BF16 = 1, WJ_F16 = 2,
       /* ABI 4, wj_qwen_create only (BASELINE cfg5: "fp8 MFMA"): the blob holds float16 matrices; the four projection matrices of
        * every decoder layer are re-quantised at create to MX-fp8 (OCP e4m3, E8M0 scale per 32 elements) and multiplied with
        * activations quantised the same way on the fly, on v_mfma_scale_f32_16x16x128_f8f6f4; embeddings, norms, attention, KV
        * caches and the LM head stay float16 (the head with split activations).  A throughput type: 3 mantissa bits on both GEMM
        * operands put the log-probs ~1e-1 from the fp32 evaluation (tests/test_gpu_qwen.py states the measured bound). */
       WJ_F8W = 3 };
enum { WJ_MEL_FW = 0, WJ_MEL_OW = 1, WJ_MEL_RAW = 2 };   /* faster-whisper / openai-whisper mel semantics; RAW = no zero padding
                                                         * (Qwen3-ASR's feature extractor: the same formula on the clip as it is) */

typedef struct wj_ctx wj_ctx;
typedef struct wj_whisper wj_whisper;
typedef struct wj_vad wj_vad;

/* ---- context ------------------------------------------------------------------------- */
int wj_abi_version(void);
const char* wj_last_error(void);
/* Replaces: device selection at whisperjav/utils/device_detector.py:84-157 (returns "cpu" on
 * AMD today).  Creates a HIP context + stream on `device_ordinal`. */
int wj_init(int device_ordinal, wj_ctx** out);
int wj_shutdown(wj_ctx* ctx);
int wj_sync(wj_ctx* ctx);
/* Extra streams for the entry points that take a `stream` argument.  cu_count > 0 restricts the stream to the compute units
 * cu_first .. cu_first + cu_count - 1 (hipExtStreamCreateWithCUMask): two such streams over disjoint CU sets let the encoder
 * of the next chunk of windows (matrix-core bound) and the decode loop of the current one (HBM / latency bound) run SIDE BY
 * SIDE instead of time-slicing the chip (the 256-tile GEMM workgroups take a CU's whole register file and LDS, so unmasked
 * streams only alternate).  cu_count <= 0: a plain non-blocking stream. */
int wj_stream_create(wj_ctx* ctx, int cu_first, int cu_count, void** out);
int wj_stream_sync(wj_ctx* ctx, void* stream);
int wj_stream_destroy(wj_ctx* ctx, void* stream);
/* device properties for roofline reporting: out[0]=CU count, out[1]=clock kHz, out[2]=HBM bytes (lo32), out[3]=(hi32) */
int wj_device_info(wj_ctx* ctx, int64_t out[4]);

/* Run-time tunables (A/B switches behind the sweeps under profiles/; the defaults are the measured optimum).
 * Decode step: "dec_ks_attn", "dec_ks_fc2", "dec_ks_proj", "dec_proj_min_m", "dec_tile_min_m" (split-K factors and the
 * row counts that select them), "dec_adapt_ks" (halve them while the row tiles alone fill the chip), "dec_big_min_m" (rows from which wide projections use
 * the 256-tile kernel), "dec_rows", "dec_rows_max_m", "dec_rows_ks_attn", "dec_rows_ks_fc2" (one-wave-per-row-
 * block GEMM for small batches), "dec_ms_stages", "dec_tile_reg" (decode tile GEMM staging), "dec_fuse_reduce",
 * "decode_chains", "dec_cross_mfma" and "dec_split_act" (fp16 models: decode GEMM activations as hi + lo pairs; both read
 * at wj_whisper_create), "dec_cross_u", "dec_cross_nt".
 * Encoder: "attn_enc_variant" (bit0 XCD remap, bit1 base-2 softmax, bit2 lazy rescale, bit3 lean softmax),
 * "gemm_big" (256-tile GEMM kernel: 1 lockstep, 3-5 ping-pong with that many 32-wide ring stages, 6 ping-pong over
 * 64-wide pairs = default), "epi_wide" (16-byte epilogue stores of the MFMA tile kernels), "dec_ms_resid" (ring stages of
 * the residual-writing decode tile GEMMs).  Alignment: "align_prefill".  Search: "beam_topk_reg" (register-resident top-2K of the
 * device beam search; 0 = the multi-pass sweep), "beam_compact" (1 = windows whose search has ended leave the batch),
 * "beam_poll" (iterations between polls of the done flags), "beam_compact_pct" / "beam_compact_min" (share / number of
 * finished windows that triggers a re-pack).  Unknown keys are an error. */
int wj_tune(const char* key, int value);

/* ---- profiler --------------------------------------------------------------------------
 * Kernel time per launch class, measured with hipEvent pairs recorded on the stream the kernels
 * are launched on (used by bench.py for the live roofline figures).  While a profile is open the
 * decode loop runs eagerly (events cannot be timed inside a captured graph). */
int wj_profile_start(wj_ctx* ctx);
int wj_profile_tags(void);
const char* wj_profile_tag_name(int tag);
int wj_profile_stop(wj_ctx* ctx, double* total_ms, int64_t* counts, int n_tags);
/* as wj_profile_stop, plus units[tag] = sum over the class's launches of the windows in the launch's batch (the
 * algorithmic bytes of a decode cross-attention launch are windows x 2 x H x 1500 x 64 x element size) */
int wj_profile_stop_ex(wj_ctx* ctx, double* total_ms, int64_t* counts, int64_t* units, int n_tags);

/* ---- log-mel ---------------------------------------------------------------------------
 * Replaces: faster_whisper.feature_extractor.FeatureExtractor.__call__ (entered from
 * whisperjav/modules/faster_whisper_pro_asr.py:819) [WJ_MEL_FW] and
 * whisper.audio.log_mel_spectrogram (entered from whisperjav/modules/whisper_pro_asr.py:433)
 * [WJ_MEL_OW].
 *
 * Batched over `n_clips` clips resident in HBM: clip i is pcm_dev[offsets[i] .. offsets[i+1])
 * (float32 mono 16 kHz; offsets is a HOST array of n_clips+1 sample offsets).
 * Frames per clip: FW: (n+160)/160, OW: (n+480000)/160, RAW: n/160  (wj_logmel_frames()).
 * out_dev receives float32 [n_clips][n_mels][out_frames]; frames past a clip's own count are
 * filled with 0.0 (FW pad_or_trim semantics) / are real zero-audio frames (OW), frames beyond
 * out_frames are dropped.  scratch is managed by the context. */
int64_t wj_logmel_frames(int64_t n_samples, int mode);
int wj_logmel_f32(wj_ctx* ctx, const float* pcm_dev, const int64_t* offsets_host, int n_clips,
                  int n_mels, int mode, int out_frames, float* out_dev, void* stream);

/* Scene detection front end.  Replaces the per-frame energy computation inside auditok.split as
 * driven by AuditokSceneDetector._detect_pass1/_detect_pass2 (scene_detection_backends/auditok_backend.py:385,
 * 561-567): for every analysis frame (offset, length in samples of the resident float32 clip) the exact integer
 * sum of squares of the PCM16-quantised samples ((x * 32767) truncated to int16).  The host turns the sums into
 * auditok's 20 log10(sqrt(mean)) energies and runs its tokenizer (whisperjav_amd/scenes.py). */
int wj_frame_sumsq(wj_ctx* ctx, const float* pcm_dev, int64_t n_samples, const int64_t* frame_off_host,
                   const int32_t* frame_len_host, int64_t n_frames, int64_t* sums_out_host, void* stream);

/* ---- Whisper model -------------------------------------------------------------------- */
typedef struct {
  int32_t n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int32_t n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wj_whisper_dims;

/* Canonical order of the offset table (byte offsets into the weight blob).  Matrices are
 * [out,in] row-major in the model dtype; vectors are always float32.  Per-layer groups
 * follow the globals: all encoder layers, then all decoder layers. */
enum {
  WJ_T_ENC_CONV1_W = 0, /* [d, 3*n_mels]  K index = tap*n_mels + c */
  WJ_T_ENC_CONV1_B, WJ_T_ENC_CONV2_W /* [d, 3*d] */, WJ_T_ENC_CONV2_B,
  WJ_T_ENC_POS /* f32 [n_audio_ctx, d] */, WJ_T_ENC_LNPOST_W, WJ_T_ENC_LNPOST_B,
  WJ_T_DEC_TOK_EMB /* [n_vocab, d] */, WJ_T_DEC_POS /* f32 [n_text_ctx, d] */,
  WJ_T_DEC_LN_W, WJ_T_DEC_LN_B,
  WJ_T_N_GLOBAL
};
enum { /* per encoder layer */
  WJ_TE_LN1_W = 0, WJ_TE_LN1_B, WJ_TE_QKV_W /* [3d,d] q;k;v */, WJ_TE_QKV_B, WJ_TE_OUT_W, WJ_TE_OUT_B,
  WJ_TE_LN2_W, WJ_TE_LN2_B, WJ_TE_FC1_W, WJ_TE_FC1_B, WJ_TE_FC2_W, WJ_TE_FC2_B,
  WJ_TE_N
};
enum { /* per decoder layer */
  WJ_TD_LN1_W = 0, WJ_TD_LN1_B, WJ_TD_QKV_W, WJ_TD_QKV_B, WJ_TD_OUT_W, WJ_TD_OUT_B,
  WJ_TD_LNX_W, WJ_TD_LNX_B, WJ_TD_CQ_W, WJ_TD_CQ_B, WJ_TD_CKV_W /* [2d,d] k;v */, WJ_TD_CKV_B,
  WJ_TD_COUT_W, WJ_TD_COUT_B, WJ_TD_LN2_W, WJ_TD_LN2_B, WJ_TD_FC1_W, WJ_TD_FC1_B,
  WJ_TD_FC2_W, WJ_TD_FC2_B,
  WJ_TD_N
};

/* Replaces: faster_whisper.WhisperModel(...) (whisperjav/modules/faster_whisper_pro_asr.py:247-253)
 * and whisper.load_model (whisperjav/modules/whisper_pro_asr.py:182).
 * The blob stays owned by the caller (it is typically a torch uint8 tensor that was
 * RCCL-broadcast to every rank); it must outlive the model.
 * max_batch = windows resident at once (encoder outputs + cross K/V are kept for all of them);
 * max_rows  = decode rows (windows x beams) the KV cache is sized for. */
int wj_whisper_create(wj_ctx* ctx, const wj_whisper_dims* dims, int dtype, const void* blob_dev,
                      int64_t blob_bytes, const int64_t* offsets_host, int n_offsets,
                      int max_batch, int max_rows, wj_whisper** out);
int wj_whisper_free(wj_whisper* m);
int64_t wj_whisper_workspace_bytes(const wj_whisper* m);

/* Replaces: ctranslate2 Whisper.encode / whisper.model.AudioEncoder.forward (+ the per-window
 * cross-attention K/V projection the decoders cache).  mel_dev: float32 [batch][n_mels][3000]
 * (the layout wj_logmel_f32 writes).  Results stay resident in the model (slots 0..batch-1).
 * n_layers < 0 runs the full stack; 0..L-1 stops early WITHOUT the final LayerNorm (used by the
 * layer-bisection parity tests).  enc_out_dev (optional, may be NULL): float32 [batch][1500][d]. */
int wj_whisper_encode(wj_whisper* m, const float* mel_dev, int batch, int n_layers,
                      float* enc_out_dev, void* stream);
/* The same, leaving the windows' cross K/V in the resident slots slot0 .. slot0 + batch - 1, asynchronously on `stream`
 * (NULL = the context's stream): the encoder's workspaces are its own, so a caller may encode the NEXT batch of windows on a
 * second stream while a decode call runs on the slots of the current one -- the encoder is matrix-core bound, the decode step
 * HBM / latency bound, and the two overlap (whisper_model.transcribe_many does this).  Not to be overlapped with
 * wj_whisper_align, which borrows the encoder workspaces. */
int wj_whisper_encode_at(wj_whisper* m, const float* mel_dev, int batch, int slot0, void* stream);

typedef struct {
  int32_t max_new_tokens;         /* <= n_text_ctx/2 */
  int32_t suppress_blank;         /* faster_whisper.py:292 / whisper DecodingOptions */
  int32_t without_timestamps;
  int32_t max_initial_timestamp_index; /* round(max_initial_timestamp / 0.02); <0 = none */
  int32_t eot, no_timestamps, timestamp_begin, blank, no_speech;
  const uint8_t* suppress_mask_dev;   /* [n_vocab] 1 = always suppressed (may be NULL) */
  /* device-sampler logits processors (ctranslate2 RepetitionPenalty / NoRepeatNgram, applied before the
   * suppress / timestamp rules over [last prompt token] + generated); <= 0 or 1.0 / 0 = off.  Honoured by
   * wj_whisper_decode_greedy / _sample; the step API takes them per call (wj_decode_topk_rules). */
  float repetition_penalty;
  int32_t no_repeat_ngram_size;
} wj_decode_opts;

/* Greedy decode, fully device resident (one host sync at the end; per-step launches are
 * replayed from a hipGraph).  Replaces: ctranslate2 Whisper.generate(beam_size=1) /
 * whisper.decoding.GreedyDecoder (temperature 0) for `batch` windows previously encoded.
 * prompts_host: int32 [batch][prompt_len] (sot, language, task[, notimestamps]); all rows use
 * the same prompt_len.  Outputs (HOST pointers): tokens_out [batch][max_new_tokens] (eot padded),
 * n_tokens_out [batch] (count before eot), sum_logprob_out [batch], no_speech_prob_out [batch],
 * token_logprob_out [batch][max_new_tokens] (optional, may be NULL). */
int wj_whisper_decode_greedy(wj_whisper* m, int batch, const int32_t* prompts_host, int prompt_len,
                             const wj_decode_opts* opts, int32_t* tokens_out, int32_t* n_tokens_out,
                             float* sum_logprob_out, float* no_speech_prob_out,
                             float* token_logprob_out, void* stream);

/* Generalisation used by the temperature-fallback ladder (faster_whisper generate_with_fallback /
 * whisper decode_with_fallback): `group` rows per window (best_of samples) share the window's cross K/V,
 * `slots_host` (may be NULL = identity) names the resident window slot of each decoded window so a subset of
 * an encoded batch can be re-decoded, temperature > 0 draws from softmax(filtered logits / T) with a
 * counter-based generator (seed, row, step, token) -- reproducible; reported log-probs are unscaled.
 * prompts_host is [batch][prompt_len]; outputs are per ROW ([batch * group]...). */
int wj_whisper_decode_sample(wj_whisper* m, int batch, int group, const int32_t* slots_host, const int32_t* prompts_host,
                             int prompt_len, const wj_decode_opts* opts, float temperature, uint32_t seed,
                             int32_t* tokens_out, int32_t* n_tokens_out, float* sum_logprob_out,
                             float* no_speech_prob_out, float* token_logprob_out, void* stream);

/* Beam search, fully device resident.  Replaces: ctranslate2 Whisper.generate(beam_size = beam, patience,
 * length_penalty, repetition_penalty, no_repeat_ngram_size [the last two in opts]) as called by
 * faster_whisper.WhisperModel.generate_with_fallback (faster_whisper_pro_asr.py:819).  Semantics of
 * CTranslate2's search: per step the best 2*beam of beam x vocabulary, EOT candidates among the first `beam`
 * retire to the window's finished list and are replaced by the next non-EOT candidates, a window stops once
 * round(beam * patience) hypotheses are finished or max_new_tokens is reached; the hypothesis with the best
 * score / len^length_penalty wins.  Outputs per WINDOW: tokens_out [batch][max_new_tokens] (eot padded),
 * n_tokens_out, score_out (normalised, may be NULL), sum_logprob_out (cumulative log-prob incl. EOT),
 * no_speech_prob_out (may be NULL).  slots_host (may be NULL = 0..batch-1) names the resident window of each row
 * group, as for wj_whisper_decode_sample. */
int wj_whisper_decode_beam(wj_whisper* m, int batch, int beam, const int32_t* slots_host, const int32_t* prompts_host,
                           int prompt_len, const wj_decode_opts* opts, float patience, float length_penalty,
                           int32_t* tokens_out, int32_t* n_tokens_out, float* score_out, float* sum_logprob_out,
                           float* no_speech_prob_out, void* stream);

/* The same loop with openai-whisper's search rules.  Replaces: whisper.decoding.DecodingTask with beam_size set
 * (BeamSearchDecoder.update / finalize + MaximumLikelihoodRanker), reached through whisper.transcribe from
 * whisperjav/modules/whisper_pro_asr.py:433 -- the fidelity pipeline's search (defaults beam_size = 2, patience = 1.2:
 * whisperjav/config/components/asr/openai_whisper.py:229-255).  Per step every beam proposes its top (beam + 1) tokens,
 * candidates with equal token sequences count once, the best `beam` non-EOT ones continue (all beams stay alive, they
 * start as `beam` copies), EOT-terminated ones are admitted best first while the window holds fewer than
 * round(beam * patience) finished sequences, which is also when it stops; at the length limit a window holding fewer
 * than `beam` sequences is topped up with its live beams; the winner maximises sum_logprob / length (length_penalty < 0,
 * upstream's None) or sum_logprob / ((5 + length) / 6) ** length_penalty.  opts->repetition_penalty and
 * no_repeat_ngram_size must be neutral (1.0 / 0: upstream has no such processors).  round(beam * patience) >= beam is
 * required.  Arguments and outputs as wj_whisper_decode_beam. */
int wj_whisper_decode_beam_openai(wj_whisper* m, int batch, int beam, const int32_t* slots_host, const int32_t* prompts_host,
                                  int prompt_len, const wj_decode_opts* opts, float patience, float length_penalty,
                                  int32_t* tokens_out, int32_t* n_tokens_out, float* score_out, float* sum_logprob_out,
                                  float* no_speech_prob_out, void* stream);

/* ---- Qwen3 text decoder (the LLM of Qwen3-ASR; BASELINE cfg5, SURVEY.md 8f-3: first correct path) ---------------------
 * Replaces (un-vendored upstream): the `qwen_asr` package's generation loop behind QwenASR.transcribe
 * (whisperjav/modules/qwen_asr.py:638-757; checkpoints Qwen/Qwen3-ASR-1.7B and Qwen3-ForcedAligner-0.6B named at :192-193),
 * i.e. what a TextGenerator (whisperjav/modules/subtitle_pipeline/protocols.py:60-110) runs per VAD group.  RMSNorm, fused
 * QKV projection, per-head RMSNorm on q / k, rotary embedding (rotate-half), grouped-query causal attention with head_dim
 * 128, SwiGLU MLP, tied LM head.  The prompt enters as EMBEDDINGS (fp32, device, rows of all sequences packed back to
 * back), so audio embeddings -- the projected output of the audio tower, not part of this slice -- simply replace the rows
 * of the <audio> placeholder tokens.
 * Blob: matrices [out][in] row-major in the compute type, vectors fp32, every tensor 256-byte aligned, offsets in the order
 * of the enumerators below (globals, then WJ_QL_* per layer).  QKV = q rows, then k rows, then v rows.  GATEUP (ABI 5 onwards,
 * every dtype): the 2 x ffn rows INTERLEAVED in blocks of 16 -- rows [32 b, 32 b + 16) = gate rows [16 b, 16 b + 16), rows
 * [32 b + 16, 32 b + 32) = up rows [16 b, 16 b + 16) -- so that neighbouring matrix-core fragments of a lane hold a (gate, up)
 * pair and silu(g) * u is computed in the GEMM's epilogue; ffn must be a multiple of 16 (wj_qwen_create refuses otherwise).
 * Rounds 3-4 (ABI <= 4) stored gate rows, then up rows: a blob packed that way decodes garbage here. */
typedef struct wj_qwen wj_qwen;
typedef struct {
  int32_t hidden, n_layer, n_head, n_kv_head, head_dim, ffn, vocab;
  float rope_theta, rms_eps;
} wj_qwen_dims;
enum { WJ_Q_EMBED = 0, WJ_Q_NORM_W, WJ_Q_N_GLOBAL };
enum { WJ_QL_LN1_W = 0, WJ_QL_QKV_W, WJ_QL_QNORM_W, WJ_QL_KNORM_W, WJ_QL_O_W, WJ_QL_LN2_W, WJ_QL_GATEUP_W, WJ_QL_DOWN_W, WJ_QL_N };
/* max_seqs sequences of up to max_ctx positions (KV cache [layer][seq][kv_head][max_ctx][128] x 2), max_rows = packed
 * prompt tokens of one prefill call */
int wj_qwen_create(wj_ctx* ctx, const wj_qwen_dims* dims, int dtype, const void* blob_dev, size_t blob_bytes,
                   const int64_t* offsets_host, int n_offsets, int max_seqs, int max_ctx, int max_rows, wj_qwen** out);
int wj_qwen_free(wj_qwen* m);
/* token embeddings (fp32 [n][hidden], device) of n host token ids: the caller scatters its audio rows over the result */
int wj_qwen_embed(wj_qwen* m, const int32_t* tokens_host, int n, float* out_dev, void* stream);
/* Runs the prompts of n_seqs sequences (embeds_dev: fp32 [sum(n_tokens)][hidden], sequence after sequence) through the
 * decoder, fills the KV caches and leaves every sequence at the position after its prompt.  logits_out_dev (may be NULL):
 * fp32 [n_seqs][vocab] logits of every sequence's last prompt position. */
int wj_qwen_prefill(wj_qwen* m, const float* embeds_dev, int n_seqs, const int32_t* n_tokens_host, float* logits_out_dev, void* stream);
/* Greedy continuation of the prefilled sequences until one of the EOS ids (not stored) or max_new tokens.  Host outputs:
 * tokens_out [n_seqs][max_new], n_tokens_out [n_seqs], token_logprob_out [n_seqs][max_new + 1] (may be NULL; entry n of a
 * sequence that stopped at EOS after n tokens is the EOS token's log-prob). */
int wj_qwen_generate_greedy(wj_qwen* m, const int32_t* eos_ids_host, int n_eos, int max_new, int32_t* tokens_out, int32_t* n_tokens_out,
                            float* token_logprob_out, void* stream);

/* The same with the two generation controls the reference's pipeline sets (pipelines/qwen_pipeline.py:157-158 ->
 * modules/qwen_asr.py:382-437):
 *  - max_new_per_seq_host (may be NULL): a token budget per sequence, clamped to max_new -- the reference scales
 *    max_new_tokens with each clip's duration (max_tokens_per_audio_second, floor 256);
 *  - repetition_penalty (1 = off): transformers' RepetitionPenaltyLogitsProcessor -- before every choice the logit of each id
 *    already in the sequence (PROMPT ids included: seen_ids_host holds them, sequence b's at [seen_offsets_host[b],
 *    seen_offsets_host[b + 1])) is divided by the penalty when positive and multiplied when negative.  Reported log-probs are
 *    those of the penalised distribution (transformers' `scores`). */
int wj_qwen_generate_greedy_ex(wj_qwen* m, const int32_t* eos_ids_host, int n_eos, int max_new, const int32_t* max_new_per_seq_host,
                               float repetition_penalty, const int32_t* seen_ids_host, const int32_t* seen_offsets_host,
                               int32_t* tokens_out, int32_t* n_tokens_out, float* token_logprob_out, void* stream);

/* 1 if the last wj_qwen_generate_greedy replayed its decode iteration from a hipGraph */
int wj_qwen_last_used_graph(const wj_qwen* m);
/* ABI 4.  Decode iterations the last generation ran (the loop polls the per-sequence finished flags every 8 iterations and
 * leaves when every sequence has ended on EOS or on its budget), and the number of sequences whose token budget was cut to the
 * room left in the KV cache (max_ctx - prompt length; the reference budgets up to 4096 new tokens, qwen_asr.py:414-437 --
 * size max_ctx for prompt + budget, or read this count). */
int wj_qwen_last_steps(const wj_qwen* m);
int wj_qwen_last_truncated(const wj_qwen* m);
/* batch compaction of the greedy loop (finished sequences leave the batch at the polls; wj_tune "qwen_compact_pct"): how often the last
 * generation re-packed, and the sum over its iterations of the live rows (= the row-iterations of work it did) */
int wj_qwen_last_compactions(const wj_qwen* m);
int64_t wj_qwen_last_row_steps(const wj_qwen* m);

/* Token classification over a full (non-generative) pass: the forced aligner (Qwen3-ForcedAligner-0.6B, reference
 * whisperjav/modules/qwen_asr.py:1198-1320 -> TextAligner, protocols.py:128-179) is this decoder + audio tower with a
 * linear head over time bins, read at the <timestamp> marker positions of a prompt that interleaves the transcript's words
 * with markers.  rows_host: indices into the packed token axis; head_w_dev: [n_labels][hidden] in the compute type (device),
 * head_b_dev: fp32 [n_labels] or NULL.  argmax_out_host [n_rows]; logits_out_dev (may be NULL): fp32 [n_rows][n_labels]. */
int wj_qwen_classify(wj_qwen* m, const float* embeds_dev, int n_seqs, const int32_t* n_tokens_host, const int32_t* rows_host, int n_rows,
                     const void* head_w_dev, const float* head_b_dev, int n_labels, int32_t* argmax_out_host, float* logits_out_dev,
                     void* stream);

/* ---- Qwen3-ASR audio tower (same slice): log-mel -> audio embeddings for the decoder's <audio> placeholders ----------
 * Replaces (un-vendored upstream): the audio encoder of the `qwen_asr` model (whisperjav/modules/qwen_asr.py:545-636,
 * run per VAD group from :638-757).  mel: wj_logmel_f32(..., WJ_MEL_RAW, ...) of every clip, frame axis zero-padded to a
 * multiple of 100.  Chunks of 100 frames -> three 3x3 stride-2 convolutions (GELU) as GEMMs over gathered patches ->
 * linear to d_model -> + 13-position sinusoid table per chunk -> padding tokens dropped -> pre-LN transformer layers with
 * self-attention inside windows of n_window_infer / 100 chunks -> ln_post -> projector (linear, GELU, linear).
 * Blob: matrices in the compute type, vectors fp32, 256-byte aligned, order of the enumerators below; the patch-matrix
 * orders the convolution weights need are documented in whisperjav_amd/qwen.py::pack_audio_blob. */
typedef struct wj_qwen_audio wj_qwen_audio;
typedef struct {
  int32_t n_mels, n_layer, n_head, ffn, d_model, n_window, n_window_infer, conv_hidden, out_dim;
} wj_qwen_audio_dims;
enum { WJ_QA_CONV1_W = 0, WJ_QA_CONV1_B, WJ_QA_CONV2_W, WJ_QA_CONV2_B, WJ_QA_CONV3_W, WJ_QA_CONV3_B, WJ_QA_CONVOUT_W, WJ_QA_POS,
       WJ_QA_LNPOST_W, WJ_QA_LNPOST_B, WJ_QA_PROJ1_W, WJ_QA_PROJ1_B, WJ_QA_PROJ2_W, WJ_QA_PROJ2_B, WJ_QA_N_GLOBAL };
enum { WJ_QAL_LN1_W = 0, WJ_QAL_LN1_B, WJ_QAL_QKV_W, WJ_QAL_QKV_B, WJ_QAL_OUT_W, WJ_QAL_OUT_B, WJ_QAL_LN2_W, WJ_QAL_LN2_B,
       WJ_QAL_FC1_W, WJ_QAL_FC1_B, WJ_QAL_FC2_W, WJ_QAL_FC2_B, WJ_QAL_N };
/* max_chunks = 1 s chunks of one encode call (all clips together) */
int wj_qwen_audio_create(wj_ctx* ctx, const wj_qwen_audio_dims* dims, int dtype, const void* blob_dev, size_t blob_bytes,
                         const int64_t* offsets_host, int n_offsets, int max_chunks, wj_qwen_audio** out);
int wj_qwen_audio_free(wj_qwen_audio* m);
/* audio tokens a clip of n_frames mel frames yields (13 per full chunk + the tail chunk's share) */
int wj_qwen_audio_tokens(int n_frames);
/* mel_dev: fp32 [n_clips][n_mels][frames_max] (device); n_frames_host[c] valid frames.  out_dev: fp32 [sum tokens][out_dim]
 * (device, clip after clip); n_tokens_out_host[c] = tokens of clip c. */
int wj_qwen_audio_encode(wj_qwen_audio* m, const float* mel_dev, int n_clips, int frames_max, const int32_t* n_frames_host,
                         float* out_dev, int32_t* n_tokens_out_host, void* stream);

/* Word-timestamp alignment.  Replaces: ctranslate2 Whisper.align (faster_whisper.transcribe.WhisperModel
 * .find_alignment, reached with word_timestamps=True from faster_whisper_pro_asr.py:819) and whisper/timing.py
 * find_alignment (whisper_pro_asr.py:433).  Teacher-forced decoder pass over tokens_host [batch][n_tokens_max]
 * (= sot sequence, <|notimestamps|>, text tokens, eot; rows padded with eot; n_tokens_host[b] counts them,
 * n_prefix = len(sot sequence) + 1) for the windows resident in slots_host (NULL = 0..batch-1); the scaled
 * cross-attention scores of the n_heads (layer, head) pairs in heads_host go through softmax over the first
 * num_frames/2 encoder positions, (w - mean) / std over the tokens, a median filter of width medfilt_width
 * (odd, <= 15, reflect padding), the mean over heads and a DTW of the negated matrix.
 * Outputs (host): the DTW path of window b in path_text_out / path_time_out [b][n_tokens_max + n_audio_ctx]
 * (path_len_out[b] entries; text index 0 = the row of <|notimestamps|>), token_prob_out [b][n_tokens_max]
 * = softmax(logits[:eot]) probability of text token i of window b at index i. */
int wj_whisper_align(wj_whisper* m, int batch, const int32_t* slots_host, const int32_t* tokens_host, int n_tokens_max,
                     const int32_t* n_tokens_host, int n_prefix, const int32_t* heads_host, int n_heads,
                     const int32_t* num_frames_host, int medfilt_width, int eot, int32_t* path_text_out,
                     int32_t* path_time_out, int32_t* path_len_out, float* token_prob_out, void* stream);

/* ABI 5, diagnostic: the matrix the last wj_whisper_align ran its DTW on -- (w - mean) / std, median filtered, averaged over the
 * alignment heads (whisper/timing.py find_alignment's `matrix`; the DTW minimises the sum of its NEGATED entries).  out_host
 * [batch][n_rows][n_cols] float32: row i of window b = text position i (0 = <|notimestamps|>), column j = encoder frame j; entries
 * past a window's own token / frame counts are scratch.  Lets a test price one pass's path on the other pass's matrix (16-bit
 * compute types: two summation orders give different paths exactly where the cost surface is flat). */
int wj_whisper_last_align_matrix(const wj_whisper* m, int batch, int n_rows, int n_cols, float* out_host);

/* diagnostics of the last wj_whisper_decode_{greedy,sample,beam} call: out[0] = 1 if the step was replayed from a
 * hipGraph, out[1] = number of concurrent row chains, out[2] = decode iterations actually run (the loops leave early
 * once every row has emitted EOT resp. every window has round(beam * patience) finished hypotheses), out[3] = the
 * iterations it was allowed (max_new_tokens), out[4] = times the beam search re-packed its batch (windows whose search
 * has ended leave it), out[5] = sum over the iterations of the live windows (window-steps of work actually done) */
int wj_whisper_last_decode_info(const wj_whisper* m, int32_t out[6]);

/* Per-token log-probs of the WINNING hypothesis of every window of the last wj_whisper_decode_beam{,_openai} call, when that
 * call ran with wj_tune("beam_token_logprobs", 1) (a diagnostic mode: the search then carries the cumulative log-prob after
 * every token of every hypothesis through its history gathers, and does not compact its batch).  out_host [batch][stride],
 * stride = that call's max_new_tokens + 1: entries 0 .. n_tokens-1 = log p of each generated token under the masked
 * distribution the search ranked it in, entry n_tokens = what ending the sequence added (log p(EOT); 0 at the length limit),
 * NaN beyond.  They sum to sum_logprob_out.  The reference's libraries expose only the sum (ctranslate2 WhisperGenerationResult
 * .scores, whisper DecodingResult.avg_logprob); this is what the parity tests bound per token.  WJ_E_INVALID when the last
 * search did not carry them or batch / stride differ from that call's. */
int wj_whisper_last_beam_token_logprobs(const wj_whisper* m, int batch, int stride, float* out_host);

/* Step-wise decoder for host-driven search (beam search with CTranslate2's patience /
 * repetition-penalty / no-repeat-ngram processors lives in whisperjav_amd/search.py).
 * rows = batch*beam, row r belongs to window r / beam.
 *   wj_decode_open : resets the KV cache for `rows` rows.
 *   wj_decode_step : feeds one token per row (tokens_host[rows]) after re-binding each row's
 *                    history to parent_host[r] (NULL = identity), and leaves float32 logits
 *                    [rows][n_vocab] in the model's logits buffer (wj_decode_logits_dev).
 *   wj_decode_topk : masked log-softmax + top-k per row on device.  ban_dev: uint8 [rows][n_vocab]
 *                    or NULL; penal_*: optional repetition penalty lists.  Outputs on host. */
int wj_decode_open(wj_whisper* m, int batch, int beam, void* stream);
int wj_decode_step(wj_whisper* m, const int32_t* tokens_host, const int32_t* parent_host,
                   int want_logits, void* stream);
float* wj_decode_logits_dev(wj_whisper* m);
/* compact copy of the last step's logits: dst_dev float32 [rows][n_vocab] */
int wj_decode_logits_copy(wj_whisper* m, int rows, float* dst_dev, void* stream);
int wj_decode_topk(wj_whisper* m, int rows, int k, const uint8_t* ban_dev,
                   int32_t* ids_out_host, float* logprob_out_host, float* lse_out_host, void* stream);

/* Beam-search scoring of the last step's logits (modified in place):
 *   CTranslate2's RepetitionPenalty on pen_host[r][0..maxp) (distinct token ids, -1 padded),
 *   NoRepeatNgram bans ban_host[r][0..maxb) (-1 padded), then suppress mask / suppress_blank /
 *   Whisper timestamp rules driven by row_rules_host[r] = {first_step, last_was_timestamp,
 *   penultimate_was_timestamp, timestamp_floor (-1 = none)}, log-softmax over what is left, top-k.
 * ids / log-probs go to host arrays [rows][k] (id -1, -inf when fewer than k tokens are allowed). */
int wj_decode_topk_rules(wj_whisper* m, int rows, int k, const wj_decode_opts* opts, const int32_t* row_rules_host,
                         const int32_t* ban_host, int maxb, const int32_t* pen_host, int maxp, float penalty,
                         int32_t* ids_out_host, float* logprob_out_host, void* stream);
/* softmax probability of the no-speech token in the last step's (unfiltered) logits */
int wj_decode_no_speech(wj_whisper* m, int rows, int no_speech_id, float* out_host, void* stream);

/* ---- VAD scorer for TorchScript archives (ABI 5; ABI 6: stages, exchange area, fused execution) -------------------------
 * Replaces: the per-window forward of the reference's DEFAULT segmenter network -- the silero-v3.1 / v4.0 torch.hub archive
 * (whisperjav/main.py:1867-1876; loader whisperjav/modules/speech_segmentation/backends/silero.py:197-206; called as
 * model(chunk, 16000) on consecutive 1536-sample windows from the archive's get_speech_timestamps, :258-273).  The archive's
 * graph is walked on the host (whisperjav_amd/vad_graph.py: abstract interpretation at the window shape) and handed over as an
 * instruction stream of float32 tensor ops: strided element-wise ops, conv1d, padding, mean over an axis, linear, multi-layer
 * LSTM whose (h, c) live in a per-stream state row.  The LSTM instructions cut the program into stages; operand spaces:
 * 0 = the per-window ARENA (tensors that live inside one stage, laid out by liveness: arena_floats per window -- LDS of the
 * workgroup that runs the window's stage when it fits 160 KiB, HBM otherwise), 1 = constants, 2 = state, 3 = the per-window
 * EXCHANGE area in HBM (xchg_floats per window: what crosses a stage or what an LSTM touches).  words: the program (layout
 * in vad_graph.py / csrc/vadgraph.hip; every offset is validated here); consts_host: the archive's parameters and folded
 * constants; state_init_host [state_floats]: the state a stream starts from (what the archive's reset_states() leaves);
 * input_space / input_offset, output_space / output_offset: where the window's samples go and where its probability appears;
 * max_windows: windows per launch group (an upper bound; the library lowers it to keep the exchange areas under 2 GiB);
 * mode: 0 = one launch per stage with the arena in LDS when it fits (else as 1), 1 = one launch per instruction over an HBM
 * arena and the general LSTM kernel (round 5's executor: the fall-back and the cross-check), 2 = as 0 but fail when the arena
 * does not fit.  Per-window memory is allocated by wj_vadg_scores for the windows of the call (grow-only). */
typedef struct wj_vadg wj_vadg;
int wj_vadg_create(wj_ctx* ctx, const int32_t* words, int n_words, int n_instr, const float* consts_host, int64_t n_consts,
                   const float* state_init_host, int state_floats, int64_t arena_floats, int64_t xchg_floats, int input_space,
                   int input_offset, int output_space, int output_offset, int window, int max_windows, int mode, wj_vadg** out);
int wj_vadg_free(wj_vadg* h);
/* ABI 6.  out8 = {fused (0 / 1), LDS bytes per workgroup, stages, LSTM weights in registers (0 / 1), windows per launch group,
 * LSTM instructions, 0, 0}: how wj_vadg_scores will run the program. */
int wj_vadg_info(wj_vadg* h, int32_t* out8);
/* As wj_vad_scores, on `window`-sample windows: stream i = pcm_dev[offsets[i]..offsets[i+1]) (the last window zero padded, as
 * utils_vad.get_speech_timestamps pads it), its ceil(n_i / window) probabilities at probs_dev[prob_offsets[i] ...]
 * (prob_offsets contiguous).  Every stage is one launch over all windows of all streams (one workgroup per window), the LSTM
 * one workgroup per stream in window order; every stream starts from state_init. */
int wj_vadg_scores(wj_vadg* h, const float* pcm_dev, const int64_t* offsets_host, const int64_t* prob_offsets_host,
                   int n_streams, float* probs_dev, void* stream);

/* ---- VAD scorer -------------------------------------------------------------------------
 * Replaces: the per-window forward of the silero-vad JIT model inside
 * silero_vad.get_speech_timestamps (called at
 * whisperjav/modules/speech_segmentation/backends/silero_v6.py:205-210 and silero.py:269-273).
 * Architecture (silero-vad v5/v6, 16 kHz): 64-sample context + 512-sample chunk -> reflect pad
 * 64 -> conv-STFT(256, hop 128) magnitude [129 x 4] -> 4 x (Conv1d k3 + ReLU) with strides
 * 1,2,2,1 -> LSTM cell(128) -> ReLU -> Conv1d(128->1) -> sigmoid.  State resets per stream.
 * weights_host: float32 blob in the order documented in whisperjav_amd/vad_weights.py. */
int wj_vad_create(wj_ctx* ctx, const float* weights_host, int64_t n_floats, wj_vad** out);
int wj_vad_free(wj_vad* v);
/* Scores `n_streams` independent audio streams in lock-step (one wavefront per stream).
 * Stream i = pcm_dev[offsets[i]..offsets[i+1]); probs_dev receives float32, stream i's
 * ceil(n_i/512) window probabilities at probs_dev[prob_offsets[i] ...]. */
int wj_vad_scores(wj_vad* v, const float* pcm_dev, const int64_t* offsets_host,
                  const int64_t* prob_offsets_host, int n_streams, float* probs_dev, void* stream);

/* ---- multi-GPU: the one collective of the path ---------------------------------------------
 * Scenes shard over the GPUs of a node with no exchange step; the only traffic is ONE broadcast of the
 * packed weight blob at start-up, over RCCL (xGMI).  One process per GPU: rank `root` calls wj_comm_unique_id and hands
 * the 128 bytes to the other ranks out of band (environment, file, MPI, torch.distributed ...), every rank calls
 * wj_comm_init and wj_bcast_weights on its own device copy of the blob (same size everywhere), then wj_comm_destroy.
 * RCCL is resolved at run time (dlopen): a single-GPU process never loads it; WJ_E_UNSUPPORTED when it is absent.
 * (whisperjav_amd/sharding.py does the same broadcast through torch.distributed, whose "nccl" backend IS RCCL.) */
typedef struct wj_comm wj_comm;
int wj_comm_unique_id(char out[128]);
int wj_comm_init(wj_ctx* ctx, int nranks, int rank, const char id[128], wj_comm** out);
int wj_bcast_weights(wj_comm* comm, void* blob_dev, int64_t bytes, int root, void* stream);
/* ABI 4: the rank count RCCL reports for the communicator (ncclCommCount), for multi-GPU assertions */
int wj_comm_count(wj_comm* comm, int* n_ranks_out);
int wj_comm_destroy(wj_comm* comm);

/* ---- kernel-level entry points (parity tests and micro-benchmarks only) ----------------- */
/* C[M,N] = A[M,K] . W[N,K]^T + bias, plain row-major output in `dtype` (out_f32=0) or float32. */
int wj_k_gemm(wj_ctx* ctx, int dtype, const void* a_dev, const void* w_dev, const float* bias_dev,
              void* c_dev, int M, int N, int K, int act_gelu, int out_f32, int variant, void* stream);
/* same launch repeated `reps` times between two hipEvents on the context stream (micro-benchmarks) */
int wj_k_gemm_timed(wj_ctx* ctx, int dtype, const void* a_dev, const void* w_dev, const float* bias_dev,
                    void* c_dev, int M, int N, int K, int act_gelu, int out_f32, int variant, int reps,
                    float* ms_per_launch);

/* ABI 4, MX-fp8 GEMM (the arithmetic of the Qwen decoder's WJ_F8W compute type, BASELINE cfg5 "fp8 MFMA"): a and w arrive as fp32
 * [M][K] / [N][K]; both are quantised on the device to OCP e4m3 with one E8M0 scale per 32-element block (OCP MX v1.0: scale
 * 2^(floor(log2 amax) - 8), round to nearest even, saturation at +-448) and multiplied on v_mfma_scale_f32_16x16x128_f8f6f4 with
 * fp32 accumulation.  dtype = type of the 16-bit output (out_f32 != 0: fp32 [M][N]).  The *_out_dev pointers (may be NULL) receive
 * the quantised bytes and scale bytes, so a test can rebuild the exact product on the host; reps > 0 times the GEMM launches. */
int wj_k_gemm_mx8(wj_ctx* ctx, int dtype, const float* a_f32_dev, const float* w_f32_dev, const float* bias_dev, void* c_dev, int M, int N,
                  int K, int out_f32, uint8_t* a8_out_dev, uint8_t* a_scale_out_dev, uint8_t* w8_out_dev, uint8_t* w_scale_out_dev, int reps,
                  float* ms_per_launch, void* stream);
/* Split-activation GEMM of the fp16 / bf16 decode step: a_f32_dev float32 [M][K] is stored as [hi | lo] 16-bit rows
 * (hi = T(a), lo = T(a - hi)) and C = W.hi + W.lo is accumulated in one fp32 accumulator; c_dev float32 [M][N].
 * variant as for wj_k_gemm (rows / skinny / LDS-DMA tile kernels). */
int wj_k_gemm_split(wj_ctx* ctx, int dtype, const float* a_f32_dev, const void* w_dev, const float* bias_dev,
                    float* c_dev, int M, int N, int K, int variant, void* stream);
int wj_k_layernorm(wj_ctx* ctx, int dtype, const float* x_dev, const float* w_dev, const float* b_dev,
                   void* out_dev, int M, int D, void* stream);
/* encoder self-attention: qkv in the engine's head-split layouts built from a
 * row-major float32 [B][T][3*D] tensor by the call itself. out: dtype [B][T][D]. */
int wj_k_attention_enc(wj_ctx* ctx, int dtype, const float* qkv_f32_dev, void* out_dev, int B, int T,
                       int H, void* stream);
int wj_k_attention_enc_timed(wj_ctx* ctx, int dtype, const float* qkv_f32_dev, void* out_dev, int B, int T,
                             int H, int reps, float* ms_per_launch);
/* decode attention over a K/V set shared by `nb` query rows: q float32 [G][nb][H*64],
 * k,v float32 [G][H][n_keys][64]; out float32 [G][nb][H*64]. */
int wj_k_attention_dec(wj_ctx* ctx, int dtype, const float* q_dev, const float* k_dev, const float* v_dev,
                       float* out_dev, int G, int nb, int H, int n_keys, void* stream);
/* same kernel, `reps` back-to-back launches timed with HIP events; layout 0 = the engine's layout for the dtype
 * (bf16: V transposed per head, matrix-core kernel), 1 = row-major V with the vector kernel (A/B comparisons) */
int wj_k_attention_dec_timed(wj_ctx* ctx, int dtype, const float* q_dev, const float* k_dev, const float* v_dev,
                             float* out_dev, int G, int nb, int H, int n_keys, int layout, int reps,
                             float* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* WJHIP_H */
