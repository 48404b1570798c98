"""Generate the committed golden vectors (run in the build container, CPU only):

    python tests/golden/make_golden.py [--large]

The upstream wheels that implement this path (faster-whisper / ctranslate2 / openai-whisper) are not
installable offline and the reference's tests pin no numeric value at this boundary,
so the vectors are produced by the CPU oracle -- itself pinned against transformers' independent
Whisper / feature-extractor implementations (tests/test_oracle_*.py) -- on seeded synthetic weights
and audio that both sides can regenerate bit-identically:

  * golden_small.npz    : 2-layer d=128 model, fp32 oracle: mel, encoder probes, greedy tokens/log-probs
  * golden_large_v3.npz : the BASELINE cfg2 geometry (large-v3: 128 mels, d=1280, 32+32 layers,
                          vocab 51866), one 30 s window of synthetic speech: encoder probes and
                          the first greedy tokens with their log-probs (timestamps on)
  * golden_logmel.npz   : log-mel probes of the synthetic clip under both upstream semantics
  * golden_large_v3_r2_<weights>.npz (--large-r2): large-v3 geometry with weights that are NOT pre-rounded to the
                          engine's storage type (``none``: raw fp32 draws) resp. rounded to fp16 (``float16``: the
                          storage type of the published checkpoints): 32 greedy tokens with per-token log-probs AND
                          the BASELINE cfg3 search (beam 5, patience 1.2, repetition penalty 1.5, no-repeat-3-gram,
                          32 new tokens) -- every finished hypothesis with its cumulative log-prob.  These carry the
                          north-star's 1e-3 log-prob bar for the 16-bit compute types (tests/test_gpu_pipeline.py).
  * golden_large_v3_r3_eot.npz (--large-r3): large-v3 geometry, fp16-representable weights WITH the end-of-text ramp,
                          duration cue and peaked cross-attention (``synth_weights(**weights.SPEECHLIKE)``), two windows
                          (a 6 s and a 2.5 s clip of synthetic speech): greedy decode until EOT and the BASELINE cfg3 beam
                          search until ``round(beam * patience)`` hypotheses have finished -- every finished hypothesis,
                          different lengths, with the oracle's trace of refills / stop reason (tests/test_gpu_search_eot.py).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import decoding, logmel, whisper_ref  # noqa: E402
from tests import helpers  # noqa: E402
from whisperjav_amd import dims as pdims, synth, weights as pweights  # noqa: E402

PROBE_T = np.array([0, 1, 7, 311, 749, 1200, 1499])
PROBE_D = np.array([0, 1, 5, 63, 64, 100, 127])


def probes(enc: torch.Tensor) -> np.ndarray:
    d = enc.shape[-1]
    cols = np.unique(np.minimum(np.concatenate([PROBE_D, [d // 2, d - 1]]), d - 1))
    return enc[0][PROBE_T][:, cols].numpy(), cols


def run(dims: pdims.WhisperDims, seed: int, n_new: int, out_name: str, mel: np.ndarray):
    t0 = time.time()
    w = pweights.synth_weights(dims, seed=seed)
    oracle = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), w)
    print(f"[{out_name}] weights {time.time() - t0:.1f}s", flush=True)
    lay = decoding.TokenLayout.for_vocab(dims.n_vocab)
    toks = pdims.special_tokens(dims.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev,
                toks.no_speech)
    cfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    with torch.no_grad():
        t0 = time.time()
        enc = oracle.encode(torch.from_numpy(mel))
        print(f"[{out_name}] encode {time.time() - t0:.1f}s", flush=True)
        t0 = time.time()
        res = decoding.greedy_decode(oracle, enc, prompt, n_new, cfg)
        print(f"[{out_name}] decode {time.time() - t0:.1f}s", flush=True)
    pv, cols = probes(enc)
    np.savez_compressed(
        os.path.join(HERE, out_name), seed=seed, dims=np.array(list(dims.as_dict().values())),
        prompt=np.array(prompt), suppress=np.array(suppress), tokens=np.array(res.tokens[0]),
        token_logprob=np.array(res.token_logprob[0], dtype=np.float32), sum_logprob=res.sum_logprob,
        no_speech_prob=res.no_speech_prob, enc_probe=pv.astype(np.float32), probe_t=PROBE_T, probe_d=cols,
        enc_mean=np.float32(enc.mean()), enc_abs_mean=np.float32(enc.abs().mean()), eot=lay.eot)
    print(f"[{out_name}] tokens {res.tokens[0]} lp {np.round(res.token_logprob[0], 4)}")


def run_r2(exact: str, n_new: int, mel: np.ndarray):
    """Round-2 golden: greedy + cfg3 beam search at the large-v3 geometry on weights of rounding mode ``exact``."""
    dims = pdims.dims_for("large-v3")
    t0 = time.time()
    w = pweights.synth_weights(dims, seed=1234, exact=exact)
    oracle = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), w)
    lay = decoding.TokenLayout.for_vocab(dims.n_vocab)
    toks = pdims.special_tokens(dims.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev,
                toks.no_speech)
    gcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    bfil = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(5, 1.2, 1.0, 1.5, 3, n_new)
    name = f"golden_large_v3_r2_{exact}.npz"
    with torch.no_grad():
        enc = oracle.encode(torch.from_numpy(mel))
        print(f"[{name}] weights + encode {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        res = decoding.greedy_decode(oracle, enc, prompt, n_new, gcfg)
        print(f"[{name}] greedy {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        hyps, nsp = decoding.beam_search(oracle, enc, prompt, bcfg, bfil)
        print(f"[{name}] beam {time.time() - t0:.0f}s, {len(hyps)} finished hypotheses", flush=True)
    pv, cols = probes(enc)
    width = max(len(t) for t, _, _ in hyps)
    beam_tokens = np.full((len(hyps), width), lay.eot, dtype=np.int64)
    for i, (t, _, _) in enumerate(hyps):
        beam_tokens[i, : len(t)] = t
    np.savez_compressed(
        os.path.join(HERE, name), seed=1234, exact=exact, dims=np.array(list(dims.as_dict().values())),
        prompt=np.array(prompt), suppress=np.array(suppress), tokens=np.array(res.tokens[0]),
        token_logprob=np.array(res.token_logprob[0], dtype=np.float32), sum_logprob=res.sum_logprob,
        no_speech_prob=res.no_speech_prob, enc_probe=pv.astype(np.float32), probe_t=PROBE_T, probe_d=cols,
        enc_abs_mean=np.float32(enc.abs().mean()), eot=lay.eot,
        beam=np.array([5, 1.2, 1.0, 1.5, 3, n_new]), beam_tokens=beam_tokens,
        beam_len=np.array([len(t) for t, _, _ in hyps]), beam_norm=np.array([n for _, n, _ in hyps], dtype=np.float64),
        beam_cum=np.array([c for _, _, c in hyps], dtype=np.float64), beam_no_speech=np.float64(nsp))
    print(f"[{name}] greedy {res.tokens[0][:10]}... beam best {hyps[0][0][:10]}... cum {hyps[0][2]:.4f} "
          f"(runner-up {hyps[1][2]:.4f})" if len(hyps) > 1 else "")


def run_r3(max_new: int = 64):
    """Round-3 golden: searches that END (greedy + cfg3 beam) at the large-v3 geometry, two windows."""
    dims = pdims.dims_for("large-v3")
    t0 = time.time()
    w = pweights.synth_weights(dims, seed=1234, exact="float16", **pweights.SPEECHLIKE)
    oracle = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), w)
    lay = decoding.TokenLayout.for_vocab(dims.n_vocab)
    toks = pdims.special_tokens(dims.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev,
                toks.no_speech)
    gcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    bfil = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(5, 1.2, 1.0, 1.5, 3, max_new)
    clips = ((6.0, 1234), (2.5, 99))
    mel = np.stack([logmel.window_features(synth.speech_like(s, seed=k), 128, "fw") for s, k in clips])
    name = "golden_large_v3_r3_eot.npz"
    out = dict(seed=1234, exact="float16", weights="SPEECHLIKE", dims=np.array(list(dims.as_dict().values())), prompt=np.array(prompt),
               suppress=np.array(suppress), eot=lay.eot, beam=np.array([5, 1.2, 1.0, 1.5, 3, max_new]),
               clip_seconds=np.array([s for s, _ in clips]), clip_seeds=np.array([k for _, k in clips]))
    with torch.no_grad():
        enc = oracle.encode(torch.from_numpy(mel))
        print(f"[{name}] weights + encode {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        res = decoding.greedy_decode(oracle, enc, prompt, max_new, gcfg)
        print(f"[{name}] greedy {time.time() - t0:.0f}s lens {[len(t) for t in res.tokens]}", flush=True)
        for b in range(2):
            tr = {}
            t0 = time.time()
            hyps, nsp = decoding.beam_search(oracle, enc[b:b + 1], prompt, bcfg, bfil, trace=tr)
            print(f"[{name}] beam w{b} {time.time() - t0:.0f}s lens {[len(t) for t, _, _ in hyps]} trace {tr}", flush=True)
            width = max(len(t) for t, _, _ in hyps)
            bt = np.full((len(hyps), width), lay.eot, dtype=np.int64)
            for i, (t, _, _) in enumerate(hyps):
                bt[i, : len(t)] = t
            out.update({f"beam{b}_tokens": bt, f"beam{b}_len": np.array([len(t) for t, _, _ in hyps]),
                        f"beam{b}_norm": np.array([n for _, n, _ in hyps], dtype=np.float64),
                        f"beam{b}_cum": np.array([c for _, _, c in hyps], dtype=np.float64), f"beam{b}_no_speech": np.float64(nsp),
                        f"beam{b}_refills": tr["refills"], f"beam{b}_steps": tr["steps"], f"beam{b}_stop": tr["stop"],
                        f"beam{b}_finish_steps": np.array(tr["finish_steps"]),
                        # the winner's per-token log-probs (+ the EOT's): what wj_whisper_last_beam_token_logprobs is held to
                        f"beam{b}_token_logprobs": np.array(tr["token_logprobs"][0], dtype=np.float64)})
            gl = len(res.tokens[b])
            out.update({f"greedy{b}_tokens": np.array(res.tokens[b]), f"greedy{b}_logprob": np.array(res.token_logprob[b], dtype=np.float32)})
            assert len(res.token_logprob[b]) == gl + 1 or gl == max_new
    pv, cols = probes(enc)
    out.update(greedy_sum=res.sum_logprob, greedy_no_speech=res.no_speech_prob, enc_probe=pv.astype(np.float32), probe_t=PROBE_T,
               probe_d=cols, enc_abs_mean=np.float32(enc.abs().mean()))
    np.savez_compressed(os.path.join(HERE, name), **out)


def main():
    if "--only-r3" in sys.argv:
        run_r3()
        return
    audio = synth.speech_like(30.0, seed=1234)
    fw128 = logmel.window_features(audio, 128, "fw")
    if "--only-r2" in sys.argv:
        for exact in ("none", "float16"):
            run_r2(exact, 32, fw128[None])
        return
    fw80 = logmel.window_features(audio, 80, "fw")
    ow128 = logmel.window_features(audio[: 16000 * 11], 128, "ow")
    cols = np.array([0, 1, 2, 100, 1000, 1099, 1100, 1101, 1500, 2998, 2999])
    np.savez_compressed(os.path.join(HERE, "golden_logmel.npz"), cols=cols, fw128=fw128[:, cols], fw80=fw80[:, cols],
                        ow128=ow128[:, cols], fw128_sum=np.float64(fw128.astype(np.float64).sum()),
                        frames_fw=np.int64(logmel.logmel_fw(audio, 128).shape[1]))
    small = helpers.small_dims()
    run(small, 21, 24, "golden_small.npz", fw80[None])
    if "--large" in sys.argv:
        run(pdims.dims_for("large-v3"), 1234, 16, "golden_large_v3.npz", fw128[None])
    if "--large-r2" in sys.argv:
        for exact in ("none", "float16"):
            run_r2(exact, 32, fw128[None])
    if "--large-r3" in sys.argv:
        run_r3()


if __name__ == "__main__":
    main()
