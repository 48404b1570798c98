"""What the REAL upstream packages return for the oracle's pinning inputs -- computed live where the wheel is installed, read from a
committed fixture where it is not.  TEST INFRASTRUCTURE.

The hot path's arithmetic lives in wheels the reference does not vendor (faster-whisper, ctranslate2, openai-whisper, silero-vad and
the torch.hub archives, auditok, soundfile -- /root/reference/uv.lock) and this build environment cannot install.  Every case
below is a function that runs ONE of those packages on inputs both sides regenerate from seeds (``synth.speech_like``,
``weights.synth_weights``) and returns plain arrays.  ``scripts/make_upstream_fixtures.py`` calls them on any machine that has the
wheels and writes ``tests/golden/upstream_<case>.npz``; ``reference(case)`` hands a test the live result when the wheel imports, the
fixture when the file exists, and skips otherwise -- so ONE outside run pins the oracle permanently, and the tests in
tests/test_upstream_wheels.py (and the two W-tests that live beside their code) are the same code in all three situations.

A fixture holds inputs' seeds implicitly (they are in this file) and the package's OUTPUTS only -- never package source.  The
silero cases also keep the network's parameters / the TorchScript archive when the generator is asked to
(``--include-archives``): MIT-licensed model files, data in the sense of the task's fixture rule.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, Dict, Tuple

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


class Unavailable(Exception):
    """The case cannot be computed live here (a model directory or hub cache is missing) although the wheel imports."""


# ---- inputs shared by a case and its test -----------------------------------------------------------------------------------
FW_MEL_CLIPS = ((3.7, 5), (11.0, 6), (30.0, 7))
OW_MEL_CLIP = (7.3, 9)
OW_DECODE_CONFIGS = ((None, None), (2, 1.2), (5, 2.0))
OW_DECODE_CLIPS = ((1.5, 30), (4.0, 31), (6.0, 32))
CT2_CLIPS = ((2.0, 1), (5.5, 2))
AUDITOK_PARAMS = ((0.3, 2700.0, 1.8, 32), (0.3, 28.0, 0.94, 38), (0.2, 10.0, 0.3, 50))
SILERO_SETTINGS = ((0.35, 350), (0.5, 30))
EDGE = 64          # frames kept from both ends of a feature matrix that is too large to commit whole


def pcm16_input() -> np.ndarray:
    rng = np.random.default_rng(11)
    return np.concatenate([rng.uniform(-1.0, 1.0, 5000), (np.arange(-40, 41) + 0.5) / 32767.0, [1.0, -1.0, 0.0]]).astype(np.float32)


def _pad_ragged(seqs, fill=-1) -> np.ndarray:
    n = max((len(s) for s in seqs), default=0)
    out = np.full((len(seqs), max(1, n)), fill, dtype=np.int64)
    for i, s in enumerate(seqs):
        out[i, : len(s)] = list(s)
    return out


def unpad(row) -> list:
    return [int(t) for t in row if t >= 0]


# ---- cases ------------------------------------------------------------------------------------------------------------------
def fw_mel() -> Dict[str, np.ndarray]:
    """faster_whisper.feature_extractor.FeatureExtractor.__call__(audio, padding=160) and faster_whisper.audio.pad_or_trim
    (reference call site faster_whisper_pro_asr.py:819)."""
    import torch
    from faster_whisper.audio import pad_or_trim
    from faster_whisper.feature_extractor import FeatureExtractor
    from whisperjav_amd import synth
    out: Dict[str, np.ndarray] = {}
    for n_mels in (80, 128):
        fe = FeatureExtractor(feature_size=n_mels)
        for i, (seconds, seed) in enumerate(FW_MEL_CLIPS):
            audio = synth.speech_like(seconds, seed=seed)
            ref = np.asarray(fe(audio, padding=160), dtype=np.float32)
            win = np.asarray(pad_or_trim(torch.from_numpy(ref), 3000), dtype=np.float32)
            key = f"{n_mels}_{i}"
            out[f"shape_{key}"] = np.asarray(ref.shape, dtype=np.int64)
            if ref.shape[1] <= 1200:
                out[f"ref_{key}"] = ref
            else:        # a 30 s clip: the ends and a strided sample pin it without 1.5 MB per matrix
                out[f"head_{key}"], out[f"tail_{key}"], out[f"stride_{key}"] = ref[:, :EDGE], ref[:, -EDGE:], ref[:, ::37]
            out[f"win_pad_absmax_{key}"] = np.asarray(np.abs(win[:, ref.shape[1]:]).max() if ref.shape[1] < 3000 else 0.0, dtype=np.float32)
            out[f"win_equals_ref_{key}"] = np.asarray(np.array_equal(win[:, : min(3000, ref.shape[1])], ref[:, :3000]))
    return out


def ow_mel() -> Dict[str, np.ndarray]:
    """whisper.audio.log_mel_spectrogram(audio, n_mels, padding=N_SAMPLES) (whisper_pro_asr.py:433, inside whisper.transcribe)."""
    import torch
    import whisper.audio as wa
    from whisperjav_amd import synth
    out: Dict[str, np.ndarray] = {}
    audio = synth.speech_like(*OW_MEL_CLIP[:1], seed=OW_MEL_CLIP[1])
    for n_mels in (80, 128):
        ref = wa.log_mel_spectrogram(torch.from_numpy(audio), n_mels, padding=wa.N_SAMPLES).numpy().astype(np.float32)
        out[f"shape_{n_mels}"] = np.asarray(ref.shape, dtype=np.int64)
        out[f"head_{n_mels}"] = ref[:, :800]
        out[f"stride_{n_mels}"] = ref[:, ::41]
    return out


def ow_decoding() -> Dict[str, np.ndarray]:
    """whisper.decoding.DecodingTask -- greedy, BeamSearchDecoder + MaximumLikelihoodRanker -- on the synthetic SPEECHLIKE weights
    loaded into the real whisper.model.Whisper (the search of fidelity mode, whisper_pro_asr.py:433)."""
    import torch
    import whisper
    from oracle import logmel
    from tests import helpers
    from whisperjav_amd import synth, weights as pweights
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=21, exact="none", **pweights.SPEECHLIKE)
    model = whisper.model.Whisper(whisper.model.ModelDimensions(**d.as_dict()))
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "alignment_heads" not in k] and not unexpected, (missing, unexpected)
    model.eval()
    clips = [synth.speech_like(s, seed=seed) for s, seed in OW_DECODE_CLIPS]
    mel = torch.from_numpy(np.stack([logmel.window_features(c, d.n_mels, "ow") for c in clips]))
    out: Dict[str, np.ndarray] = {}
    for ci, (beam, patience) in enumerate(OW_DECODE_CONFIGS):
        opts = whisper.DecodingOptions(language="ja", task="transcribe", beam_size=beam, patience=patience, fp16=False, sample_len=48,
                                       suppress_tokens="", temperature=0.0)
        with torch.no_grad():
            results = whisper.decode(model, mel, opts)
        out[f"tokens_{ci}"] = _pad_ragged([r.tokens for r in results])
        out[f"avg_logprob_{ci}"] = np.asarray([r.avg_logprob for r in results], dtype=np.float64)
        out[f"no_speech_prob_{ci}"] = np.asarray([r.no_speech_prob for r in results], dtype=np.float64)
    return out


def _ct2_seeded_dir(tmp: str):
    from tests import helpers
    from whisperjav_amd import ct2_format, weights as pweights
    d = helpers.small_dims()
    w = pweights.synth_weights(d, seed=21, exact="float16", **pweights.SPEECHLIKE)
    ct2_format.write_ct2_whisper(tmp, d, w, dtype="float32", alignment_heads=[(1, 0)])
    return d, w


def ct2_generate_seeded() -> Dict[str, np.ndarray]:
    """ctranslate2.models.Whisper.generate (beam 5, patience 1.2, repetition penalty 1.5, no-repeat 3-gram, 5 hypotheses: what
    faster-whisper calls from faster_whisper_pro_asr.py:819-822) on a SEEDED toy Whisper written as a CTranslate2 directory by
    whisperjav_amd.ct2_format (both sides regenerate the weights from the seed: no download); also the encoder output of the
    same directory (a strided sample)."""
    import tempfile
    import ctranslate2 as ct2
    from oracle import logmel
    from whisperjav_amd import dims as pdims, synth
    out: Dict[str, np.ndarray] = {}
    with tempfile.TemporaryDirectory(prefix="wj_ct2_") as tmp:
        d, _ = _ct2_seeded_dir(tmp)
        model = ct2.models.Whisper(tmp, device="cpu", compute_type="float32")
        toks = pdims.special_tokens(d.n_vocab)
        prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
        suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
        for i, (seconds, seed) in enumerate(CT2_CLIPS):
            mel = logmel.window_features(synth.speech_like(seconds, seed=seed), d.n_mels, "fw")[None]
            res = model.generate(ct2.StorageView.from_array(mel), [prompt], beam_size=5, patience=1.2, length_penalty=1.0, repetition_penalty=1.5,
                                 no_repeat_ngram_size=3, max_length=224, return_scores=True, return_no_speech_prob=True, suppress_blank=True,
                                 suppress_tokens=list(suppress), max_initial_timestamp_index=0, num_hypotheses=5)[0]
            out[f"sequences_{i}"] = _pad_ragged(res.sequences_ids)
            out[f"scores_{i}"] = np.asarray(res.scores, dtype=np.float64)
            out[f"no_speech_prob_{i}"] = np.asarray(res.no_speech_prob, dtype=np.float64)
            if i == 0:
                out["encoder_stride"] = np.asarray(model.encode(ct2.StorageView.from_array(mel)), dtype=np.float32)[0, ::25]
    out["ctranslate2_version"] = np.asarray(ct2.__version__)
    return out


def ct2_generate_checkpoint() -> Dict[str, np.ndarray]:
    """As ct2_generate_seeded on a PUBLISHED checkpoint held in both formats (WJ_CT2_MODEL_DIR: a converted faster-whisper directory,
    WJ_HF_MODEL_DIR: the same checkpoint as Hugging Face safetensors).  Live only: the comparison needs the checkpoint on both sides."""
    import ctranslate2 as ct2           # noqa: F401
    if not (os.environ.get("WJ_CT2_MODEL_DIR") and os.environ.get("WJ_HF_MODEL_DIR")):
        raise Unavailable("set WJ_CT2_MODEL_DIR / WJ_HF_MODEL_DIR to the same Whisper checkpoint in both formats")
    return {"ct2_dir": np.asarray(os.environ["WJ_CT2_MODEL_DIR"]), "hf_dir": np.asarray(os.environ["WJ_HF_MODEL_DIR"])}


def auditok_split() -> Dict[str, np.ndarray]:
    """auditok.split(bytes, sampling_rate, channels=1, sample_width=2, min_dur, max_dur, max_silence, energy_threshold,
    drop_trailing_silence=True) as called at auditok_backend.py:396,567."""
    import auditok
    from oracle import auditok_ref
    from whisperjav_amd import synth
    pcm = auditok_ref.to_pcm16(synth.speech_like(95.0, seed=4, noisy=True))
    out: Dict[str, np.ndarray] = {}
    for i, (min_dur, max_dur, max_sil, thr) in enumerate(AUDITOK_PARAMS):
        regs = auditok.split(pcm.tobytes(), sampling_rate=16000, channels=1, sample_width=2, min_dur=min_dur, max_dur=max_dur, max_silence=max_sil,
                             energy_threshold=thr, drop_trailing_silence=True)
        out[f"regions_{i}"] = np.asarray([(r.start if hasattr(r, "start") else r.meta.start, r.end if hasattr(r, "end") else r.meta.end) for r in regs],
                                         dtype=np.float64).reshape(-1, 2)
    return out


def soundfile_pcm16() -> Dict[str, np.ndarray]:
    """sf.write(path, x, sr, subtype="PCM_16") (scene_detection_backends/utils.py:140) then sf.read(path, dtype="float32")
    (faster_whisper_pro_asr.py:477), and the raw int16 samples."""
    import tempfile
    import soundfile as sf
    x = pcm16_input()
    with tempfile.TemporaryDirectory(prefix="wj_sf_") as tmp:
        path = os.path.join(tmp, "rt.wav")
        sf.write(path, x, 16000, subtype="PCM_16")
        back, sr = sf.read(path, dtype="float32")
        raw, _ = sf.read(path, dtype="int16")
    return {"back": np.asarray(back, dtype=np.float32), "raw": np.asarray(raw, dtype=np.int16), "sr": np.asarray(sr)}


def silero_v5() -> Dict[str, np.ndarray]:
    """silero_vad.load_silero_vad(): window probabilities of the bundled v5/v6 archive on a seeded clip, get_speech_timestamps at two
    settings (reference call backends/silero_v6.py:205-210), and the archive's parameters (what the HIP blob is packed from)."""
    import silero_vad
    import torch
    from whisperjav_amd import synth
    jit = silero_vad.load_silero_vad()
    audio = synth.speech_like(20.0, seed=3)
    jit.reset_states()
    out: Dict[str, np.ndarray] = {"probs": np.asarray([float(jit(torch.from_numpy(audio[i: i + 512]), 16000)) for i in range(0, len(audio) - 511, 512)],
                                                     dtype=np.float32)}
    for i, (thr, pad) in enumerate(SILERO_SETTINGS):
        stamps = silero_vad.get_speech_timestamps(torch.from_numpy(audio), jit, threshold=thr, sampling_rate=16000, min_speech_duration_ms=100,
                                                  min_silence_duration_ms=100, speech_pad_ms=pad)
        out[f"stamps_{i}"] = np.asarray([(s["start"], s["end"]) for s in stamps], dtype=np.int64).reshape(-1, 2)
    for k, v in jit.state_dict().items():
        out["sd." + k] = np.asarray(v.detach().cpu().numpy())
    return out


def _silero_hub(version: str) -> Dict[str, np.ndarray]:
    import torch
    from whisperjav_amd import synth
    hub = os.path.join(torch.hub.get_dir(), f"snakers4_silero-vad_{version}")
    if not os.path.isdir(hub):
        raise Unavailable(f"torch.hub cache has no snakers4/silero-vad:{version} (the reference's loader, backends/silero.py:199-206)")
    model, utils = torch.hub.load(repo_or_dir=f"snakers4/silero-vad:{version}", model="silero_vad", onnx=False, trust_repo=True)
    audio = synth.speech_like(12.0, seed=3)
    model.reset_states()
    probs = [float(model(torch.nn.functional.pad(torch.from_numpy(audio[i: i + 1536]), (0, max(0, 1536 - len(audio[i: i + 1536])))), 16000))
             for i in range(0, len(audio), 1536)]
    g = model.forward.graph.copy()
    torch._C._jit_pass_inline(g)
    kinds = sorted({n.kind() for n in _all_nodes(g)})
    out: Dict[str, np.ndarray] = {"probs": np.asarray(probs, dtype=np.float32), "op_kinds": np.asarray(kinds)}
    defaults = {"v3.1": (0.125, 90, 300, 700), "v4.0": (0.25, 150, 300, 700)}[version]       # segmenters.HipSileroSpeechSegmenter.VERSION_DEFAULTS
    for i, (thr, ms, sil, pad) in enumerate((defaults, (0.5, 100, 300, 400))):
        stamps = utils[0](torch.from_numpy(audio), model, sampling_rate=16000, threshold=thr, min_speech_duration_ms=ms, min_silence_duration_ms=sil,
                          speech_pad_ms=pad)
        out[f"stamps_{i}"] = np.asarray([(s["start"], s["end"]) for s in stamps], dtype=np.int64).reshape(-1, 2)
        out[f"stamps_params_{i}"] = np.asarray([thr, ms, sil, pad], dtype=np.float64)
    files = [os.path.join(r, f) for r, _, fs in os.walk(hub) for f in fs if f.endswith(".jit") and "16k" not in f and "8k" not in f]
    out["archive_path"] = np.asarray(files[0] if files else "")
    return out


def _all_nodes(block):
    for n in block.nodes():
        yield n
        for b in n.blocks():
            yield from _all_nodes(b)


def silero_hub_v31() -> Dict[str, np.ndarray]:
    """torch.hub.load("snakers4/silero-vad:v3.1", "silero_vad", onnx=False) -- the reference's DEFAULT segmenter network
    (main.py:1867-1876; loader backends/silero.py:197-206): window probabilities on 1536-sample windows, the archive's own
    get_speech_timestamps at the reference's defaults and at the balanced preset, and the op inventory of its inlined graph."""
    return _silero_hub("v3.1")


def silero_hub_v40() -> Dict[str, np.ndarray]:
    """As silero_hub_v31 for the v4.0 tag (backends/silero.py:68-72)."""
    return _silero_hub("v4.0")


CASES: Dict[str, Callable[[], Dict[str, np.ndarray]]] = {
    "fw_mel": fw_mel, "ow_mel": ow_mel, "ow_decoding": ow_decoding, "ct2_generate_seeded": ct2_generate_seeded,
    "ct2_generate_checkpoint": ct2_generate_checkpoint, "auditok_split": auditok_split, "soundfile_pcm16": soundfile_pcm16,
    "silero_v5": silero_v5, "silero_hub_v31": silero_hub_v31, "silero_hub_v40": silero_hub_v40,
}
LIVE_ONLY = {"ct2_generate_checkpoint"}          # nothing to commit: the comparison needs the checkpoint itself


def fixture_path(case: str) -> Path:
    return GOLDEN / f"upstream_{case}.npz"


def archive_path(case: str) -> Path:
    return GOLDEN / f"upstream_{case}.jit"


def status(case: str) -> str:
    """"live" (the wheel imports and the case computes), "fixture" (a committed file stands in for it) or "unpinned"."""
    try:
        CASES[case]()
        return "live"
    except (ImportError, Unavailable, OSError):
        pass
    return "fixture" if fixture_path(case).exists() else "unpinned"


def reference(case: str) -> Tuple[Dict[str, np.ndarray], str]:
    """The upstream package's outputs for ``case``: (arrays, "live" | "fixture"); ``pytest.skip`` when there is neither."""
    import pytest
    try:
        return CASES[case](), "live"
    except (ImportError, Unavailable, OSError) as e:
        why = f"{type(e).__name__}: {e}"
    path = fixture_path(case)
    if case not in LIVE_ONLY and path.exists():
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}, "fixture"
    pytest.skip(f"upstream case {case!r}: wheel absent offline and no tests/golden/upstream_{case}.npz yet (parity unpinned, PARITY.md; "
                f"scripts/make_upstream_fixtures.py writes it wherever the wheel exists) [{why}]")
