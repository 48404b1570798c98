"""The registration recipe of INTEGRATION.md executed against the REFERENCE's own factories (imported from source).
Needs /root/reference (present in the build container, absent on the GPU box: the tests skip there) -- it proves the
plugin seams accept the classes of this package with the parameters the reference's resolver produces."""
import importlib
import os
import sys
import types

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "whisperjav")), reason="reference tree not present")


@pytest.fixture()
def ref_modules(monkeypatch):
    """Load reference sub-packages by path without running whisperjav/__init__.py's application imports."""
    saved = {k: v for k, v in sys.modules.items() if k == "whisperjav" or k.startswith("whisperjav.")}
    for k in saved:
        del sys.modules[k]
    for name, path in (("whisperjav", f"{REF}/whisperjav"), ("whisperjav.modules", f"{REF}/whisperjav/modules")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    yield
    for k in [k for k in sys.modules if k == "whisperjav" or k.startswith("whisperjav.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_speech_segmenter_factory_creates_the_hip_backends(ref_modules, monkeypatch):
    from whisperjav_amd import hipbind, segmenters
    monkeypatch.setattr(hipbind, "context", lambda device=0: types.SimpleNamespace(handle=None, device=device))
    factory = importlib.import_module("whisperjav.modules.speech_segmentation.factory")
    for name, target in segmenters.REGISTRY_ENTRIES.items():
        factory._BACKEND_REGISTRY[name] = target
        factory._BACKEND_DEPENDENCIES[name] = {"packages": ["whisperjav_amd"], "install_hint": "build libwjhip.so",
                                                "always_available": False}
    factory._PARAM_SCHEMAS["silero-hip"] = factory._PARAM_SCHEMAS["silero-v6.2"]
    factory._PARAM_SCHEMAS["silero-v6.2-hip"] = factory._PARAM_SCHEMAS["silero-v6.2"]
    ok, _ = factory.SpeechSegmenterFactory.is_backend_available("silero-hip")
    assert ok
    # what FasterWhisperProASR hands over: resolver VAD presets merged with the speech_segmenter section (:104-122);
    # GUI-style string values must be coerced by the reference's sanitiser, unknown keys stripped
    cfg = {"threshold": "0.35", "min_speech_duration_ms": 100, "min_silence_duration_ms": 300, "speech_pad_ms": 400,
           "chunk_threshold_s": 2.5, "max_group_duration_s": 6.0, "backend": "silero-hip", "not_a_parameter": 1}
    seg = factory.SpeechSegmenterFactory.create("silero-hip", config=cfg)
    assert type(seg).__name__ == "HipSileroV6SpeechSegmenter"
    assert seg.threshold == pytest.approx(0.35) and seg.max_group_duration_s == 6.0 and seg.speech_pad_ms == 400
    assert seg.name.startswith("silero") and 16000 in seg.get_supported_sample_rates()
    legacy = factory.SpeechSegmenterFactory.create("silero-v4.0-hip", config={"threshold": 0.4})
    assert type(legacy).__name__ == "HipSileroSpeechSegmenter"
    seg.cleanup(); legacy.cleanup()


def test_scene_detector_factory_creates_the_hip_backend(ref_modules, monkeypatch):
    from whisperjav_amd import hipbind
    monkeypatch.setattr(hipbind, "context", lambda device=0: types.SimpleNamespace(handle=None, device=device))
    monkeypatch.setattr(hipbind, "lib", lambda: None)
    for stub in ("soundfile", "librosa"):
        monkeypatch.setitem(sys.modules, stub, types.ModuleType(stub))
    factory = importlib.import_module("whisperjav.modules.scene_detection_backends.factory")
    factory._BACKEND_REGISTRY["auditok-hip"] = "whisperjav_amd.scenes.HipAuditokSceneDetector"
    if hasattr(factory, "_BACKEND_DEPENDENCIES"):
        factory._BACKEND_DEPENDENCIES["auditok-hip"] = {"packages": ["whisperjav_amd"], "install_hint": "build libwjhip.so",
                                                        "always_available": False}
    det = factory.SceneDetectorFactory.create("auditok-hip", max_duration=29.0, min_duration=0.3)
    assert type(det).__name__ == "HipAuditokSceneDetector" and det.name == "auditok-hip"
    assert det._config.max_duration == 29.0 and det._config.pass2_max_duration == 28.0
    legacy = factory.SceneDetectorFactory.create_from_legacy_kwargs(method="auditok-hip", max_duration=20.0, pass1_max_silence_s=2.5)
    assert legacy._config.max_duration == 20.0 and legacy._config.pass1_max_silence == 2.5
    det.cleanup()


def test_reference_asr_module_drives_the_model_shim_with_its_balanced_preset(ref_modules, monkeypatch, tmp_path):
    """INTEGRATION.md "smallest possible diff": the reference's FasterWhisperProASR stays, only
    ``faster_whisper.WhisperModel`` is swapped for ``HipWhisperModel``.  Run here with the reference class imported from
    source, its ``balanced`` preset values (config/components/asr/faster_whisper.py) and the shim over an engine double:
    the shim must accept every keyword the reference sends and its segments must flow through the reference's
    post-processing."""
    import numpy as np
    from tests import test_asr_adapter as doubles
    from whisperjav_amd import dims as pdims, whisper_model as wm
    tb = pdims.special_tokens(51865).timestamp_begin
    shim = doubles._model([[tb, 11, 12, tb + 100, tb + 100, 13, tb + 150]])
    shim.model.align = lambda rows, n_prefix, heads, num_frames, slots=None, medfilt_width=7: [
        (np.repeat(np.arange(len(r) - 4), 10), np.arange(10 * (len(r) - 4)), np.full(len(r) - 5, 0.8, np.float32)) for r in rows]
    shim.dims = pdims.custom_dims(80, 128, 2, 2, 51865)
    seen = []
    real = shim.transcribe

    class SwappedWhisperModel:
        def __init__(self, *a, **kw):
            self.ctor = (a, kw)

        def transcribe(self, audio, **params):
            seen.append(params)
            return real(audio, **params)
    fw = types.ModuleType("faster_whisper"); fw.WhisperModel = SwappedWhisperModel
    sf = types.ModuleType("soundfile"); sf.SoundFileError = Exception
    audio = (np.sin(np.arange(16000 * 9) * 0.05) * 0.2).astype(np.float32)
    sf.read = lambda path, dtype="float32", **kw: (audio.copy(), 16000)
    for name, mod in (("faster_whisper", fw), ("soundfile", sf), ("srt", types.ModuleType("srt"))):
        monkeypatch.setitem(sys.modules, name, mod)
    ref = importlib.import_module("whisperjav.modules.faster_whisper_pro_asr")
    base = importlib.import_module("whisperjav.modules.speech_segmentation.base")

    class Seg:
        name = "silero-v6.2"

        def segment(self, a, sample_rate=16000, **kw):
            s = [base.SpeechSegment(start_sec=1.0, end_sec=4.0, start_sample=16000, end_sample=64000)]
            return base.SegmentationResult(segments=s, groups=[s], method=self.name, audio_duration_sec=len(a) / sample_rate, parameters={})

        def cleanup(self):
            pass
    monkeypatch.setattr(ref.SpeechSegmenterFactory, "create", staticmethod(lambda name, config=None, **kw: Seg()))
    balanced = dict(task="transcribe", language="ja", beam_size=1, best_of=2, patience=1.2, length_penalty=None, prefix=None,
                    suppress_tokens=None, suppress_blank=True, without_timestamps=False, max_initial_timestamp=0.0,
                    temperature=[0.0], compression_ratio_threshold=2.4, logprob_threshold=-1.0, logprob_margin=0.0,
                    no_speech_threshold=0.65, drop_nonverbal_vocals=False, condition_on_previous_text=False, initial_prompt=None,
                    word_timestamps=True, prepend_punctuations=None, append_punctuations=None, clip_timestamps=None)
    engine_opts = dict(chunk_length=None, repetition_penalty=1.5, no_repeat_ngram_size=3, prompt_reset_on_temperature=None,
                       hotwords=None, multilingual=False, max_new_tokens=None, language_detection_threshold=None,
                       language_detection_segments=None, log_progress=False, hallucination_silence_threshold=None)
    params = {"decoder": balanced, "provider": engine_opts, "vad": {"threshold": 0.28, "min_speech_duration_ms": 100},
              "speech_segmenter": {"backend": "silero-v6.2"}}
    a = ref.FasterWhisperProASR({"model_name": "large-v3", "device": "cuda", "compute_type": "float16"}, params, "transcribe")
    out = a.transcribe(tmp_path / "scene_0001.wav")
    assert len(seen) == 1 and seen[0]["word_timestamps"] is True and seen[0]["vad_filter"] is False
    assert "hallucination_silence_threshold" not in seen[0] and seen[0]["temperature"] == 0.0
    # two sub-segments decoded by the shim, shifted by the group start (1.0 s) by the reference's own code
    assert [s["text"] for s in out["segments"]] == ["<11><12>", "<13>"]
    assert out["segments"][0]["start"] >= 1.0 and out["segments"][-1]["end"] <= 4.0 + 1e-6


def test_reference_fidelity_module_drives_the_openai_shim_with_its_balanced_preset(ref_modules, monkeypatch, tmp_path):
    """The fidelity seam: the reference's WhisperProASR (from source) with ``whisper.load_model`` returning
    ``HipOpenAIWhisperModel`` (over an engine double) and the reference's own ``balanced`` preset
    (config/components/asr/openai_whisper.py:225-258), None values included."""
    import numpy as np
    from tests import test_asr_adapter as doubles
    from whisperjav_amd import dims as pdims
    tb = pdims.special_tokens(51865).timestamp_begin
    shim = doubles._ow_model([[tb, 11, 12, tb + 100, tb + 100, 13, tb + 150]])
    shim.model.align = lambda rows, n_prefix, heads, num_frames, slots=None, medfilt_width=7: [
        (np.repeat(np.arange(len(r) - 4), 10), np.arange(10 * (len(r) - 4)), np.full(len(r) - 5, 0.8, np.float32)) for r in rows]
    shim.dims = pdims.custom_dims(80, 128, 2, 2, 51865)
    seen = []
    real = shim.transcribe

    def transcribe(audio, **params):
        seen.append(params)
        return real(audio, **params)
    shim.transcribe = transcribe
    wh = types.ModuleType("whisper"); wh.load_model = lambda name, device=None, **kw: shim
    sf = types.ModuleType("soundfile"); sf.SoundFileError = Exception
    audio = (np.sin(np.arange(16000 * 9) * 0.05) * 0.2).astype(np.float32)
    sf.read = lambda path, dtype="float32", **kw: (audio.copy(), 16000)
    for name, mod in (("whisper", wh), ("soundfile", sf), ("srt", types.ModuleType("srt"))):
        monkeypatch.setitem(sys.modules, name, mod)
    ref = importlib.import_module("whisperjav.modules.whisper_pro_asr")
    base = importlib.import_module("whisperjav.modules.speech_segmentation.base")

    class Seg:
        name = "silero-v4.0"

        def segment(self, a, sample_rate=16000, **kw):
            s = [base.SpeechSegment(start_sec=1.0, end_sec=4.0, start_sample=16000, end_sample=64000)]
            return base.SegmentationResult(segments=s, groups=[s], method=self.name, audio_duration_sec=len(a) / sample_rate, parameters={})

        def cleanup(self):
            pass
    monkeypatch.setattr(ref.SpeechSegmenterFactory, "create", staticmethod(lambda name, config=None, **kw: Seg()))
    decoder = dict(task="transcribe", language="ja", beam_size=1, best_of=2, patience=1.2, length_penalty=None, prefix=None,
                   suppress_tokens=None, suppress_blank=True, without_timestamps=False, max_initial_timestamp=0.0,
                   temperature=[0.0], compression_ratio_threshold=2.4, logprob_threshold=-1.0, logprob_margin=0.0,
                   no_speech_threshold=0.71, drop_nonverbal_vocals=False, condition_on_previous_text=False, initial_prompt=None,
                   word_timestamps=True, prepend_punctuations=None, append_punctuations=None, clip_timestamps=None)
    provider = dict(verbose=None, carry_initial_prompt=None, prompt=None, fp16=True, hallucination_silence_threshold=None)
    params = {"decoder": decoder, "provider": provider, "vad": {"threshold": 0.3}, "speech_segmenter": {"backend": "silero-v4.0"}}
    a = ref.WhisperProASR({"model_name": "large-v2", "device": "cuda"}, params, "transcribe")
    out = a.transcribe(tmp_path / "scene_0001.wav")
    assert len(seen) >= 1 and seen[0]["fp16"] is True and seen[0]["word_timestamps"] is True
    assert [s["text"] for s in out["segments"]] == ["<11><12>", "<13>"]
    assert out["segments"][0]["start"] >= 1.0 and out["segments"][-1]["end"] <= 4.0 + 1e-6


def test_device_gate(ref_modules, monkeypatch):
    """``whisperjav_amd.device.install()`` swaps the AMD-aware ``get_best_device`` into the
    reference's ``utils/device_detector`` (which answers "cpu" on ROCm, :146-157).  No GPU here: the HIP branch is
    exercised with a doubled probe, the fall-through must be the reference's own answer."""
    import torch
    from whisperjav_amd import device
    sys.modules["whisperjav.utils"] = types.ModuleType("whisperjav.utils")
    sys.modules["whisperjav.utils"].__path__ = [f"{REF}/whisperjav/utils"]
    dd = importlib.import_module("whisperjav.utils.device_detector")
    ref_answer = dd.get_best_device()
    assert device.install() and dd.get_best_device is device.get_best_device
    assert device.install()                                              # idempotent
    assert dd._reference_get_best_device() == ref_answer
    assert dd.get_best_device(prefer_cpu=True) == "cpu"
    if not torch.cuda.is_available():
        assert dd.get_best_device() == ref_answer == "cpu"               # nothing to enable without a device
    monkeypatch.setattr(device, "hip_path_available", lambda: (True, "AMD Instinct MI355X"))
    assert dd.get_best_device() == "cuda"
    monkeypatch.setattr(device, "hip_path_available", lambda: (False, "AMD Instinct MI355X"))
    if not torch.cuda.is_available():
        assert dd.get_best_device() == "cpu"                             # device seen, library unusable: the reference's answer


def test_qwen_adapters_satisfy_the_reference_protocols(ref_modules):
    """``HipQwenTextGenerator`` / ``HipQwenForcedAligner`` are instances of the reference's runtime-checkable
    ``TextGenerator`` / ``TextAligner`` protocols (modules/subtitle_pipeline/protocols.py:60-179), and their result objects
    carry the fields of the reference's ``TranscriptionResult`` / ``AlignmentResult`` / ``WordTimestamp`` (types.py)."""
    import dataclasses
    from whisperjav_amd import qwen
    sys.modules["whisperjav.modules.subtitle_pipeline"] = types.ModuleType("whisperjav.modules.subtitle_pipeline")
    sys.modules["whisperjav.modules.subtitle_pipeline"].__path__ = [f"{REF}/whisperjav/modules/subtitle_pipeline"]
    protocols = importlib.import_module("whisperjav.modules.subtitle_pipeline.protocols")
    ref_types = importlib.import_module("whisperjav.modules.subtitle_pipeline.types")
    d = qwen.Qwen3Dims(hidden=64, n_layer=1, n_head=1, n_kv_head=1, head_dim=128, ffn=64, vocab=32)
    ad = qwen.Qwen3AudioDims(n_layer=1, n_head=1, ffn=64, d_model=64, conv_hidden=8, out_dim=64)
    w = {**qwen.synth_weights(d), **qwen.synth_audio_weights(ad)}
    gen = qwen.HipQwenTextGenerator(d, w, audio_dims=ad)
    al = qwen.HipQwenForcedAligner(d, ad, w)
    assert isinstance(gen, protocols.TextGenerator) and isinstance(al, protocols.TextAligner)
    for mine, ref in ((qwen.TranscriptionResult, ref_types.TranscriptionResult), (qwen.AlignmentResult, ref_types.AlignmentResult),
                      (qwen.WordTimestamp, ref_types.WordTimestamp)):
        assert [f.name for f in dataclasses.fields(mine)] == [f.name for f in dataclasses.fields(ref)]


def test_reference_vad_grouped_framer_frames_scenes_through_the_hip_segmenter(ref_modules, monkeypatch):
    """INTEGRATION.md section f: qwen mode needs no framer of ours -- the reference's ``VadGroupedFramer``
    (framers/vad_grouped.py:67-76, loaded from source) creates its segmenter through ``SpeechSegmenterFactory``, so
    registering the HIP segmenter there makes the device VAD frame the scenes.  Run here with the speech-timestamp seam of
    the segmenter scripted (no GPU): frames, per-frame speech regions, the short-group filter and the step-down ``reframe``
    (a NEW segmenter with the tighter group limit, then the original one restored) must come out of the reference's code."""
    import numpy as np
    from unittest.mock import MagicMock
    from whisperjav_amd import hipbind, segmenters
    monkeypatch.setattr(hipbind, "context", lambda device=0: types.SimpleNamespace(handle=None, device=device))
    factory = importlib.import_module("whisperjav.modules.speech_segmentation.factory")
    for name, target in segmenters.REGISTRY_ENTRIES.items():
        factory._BACKEND_REGISTRY[name] = target
        factory._BACKEND_DEPENDENCIES[name] = {"packages": ["whisperjav_amd"], "install_hint": "build libwjhip.so",
                                                "always_available": False}
    factory._PARAM_SCHEMAS["silero-hip"] = factory._PARAM_SCHEMAS["silero-v6.2"]
    stamps = [{"start": 16000, "end": 40000}, {"start": 48000, "end": 80000},          # 1.0-2.5 s, 3.0-5.0 s: one group
              {"start": 160000, "end": 160800},                                         # 10.0-10.05 s: its own group, too short
              {"start": 240000, "end": 400000}, {"start": 408000, "end": 560000}]      # 15-25 s, 25.5-35 s: split by the group limit
    created = []
    real_create = factory.SpeechSegmenterFactory.create

    def create(name, *a, **kw):
        seg = real_create(name, *a, **kw)
        seg._model = MagicMock()
        seg._get_speech_timestamps = lambda audio, model, **k: [dict(t) for t in stamps]
        created.append(seg)
        return seg
    monkeypatch.setattr(factory.SpeechSegmenterFactory, "create", staticmethod(create))
    framers = importlib.import_module("whisperjav.modules.subtitle_pipeline.framers.vad_grouped")
    protocols = importlib.import_module("whisperjav.modules.subtitle_pipeline.protocols")
    framer = framers.VadGroupedFramer(segmenter_backend="silero-hip", max_group_duration_s=29.0, chunk_threshold_s=1.0,
                                      segmenter_config={"speech_pad_ms": 0, "threshold": 0.3})
    assert isinstance(framer, protocols.TemporalFramer)
    audio = np.zeros(16000 * 40, dtype=np.float32)
    res = framer.frame(audio, 16000)
    assert type(created[0]).__name__ == "HipSileroV6SpeechSegmenter" and created[0].max_group_duration_s == 29.0
    got = [(round(f.start, 3), round(f.end, 3), f.source) for f in res.frames]
    assert got == [(1.0, 5.0, "vad-grouped"), (15.0, 35.0, "vad-grouped")], got
    assert res.metadata["groups_skipped"] == 1 and res.metadata["total_segments"] == 5
    assert res.metadata["segmenter_backend"].endswith("-hip")
    assert [[(round(a, 2), round(b, 2)) for a, b in regs] for regs in res.metadata["speech_regions"]] == \
        [[(1.0, 2.5), (3.0, 5.0)], [(15.0, 25.0), (25.5, 35.0)]]
    # step-down retry (orchestrator._run_stepdown_pass): tighter groups from a fresh segmenter, the original one kept
    tight = framer.reframe(audio, 16000, max_group_duration_s=12.0)
    assert len(created) == 2 and created[1].max_group_duration_s == 12.0
    assert [(round(f.start, 2), round(f.end, 2)) for f in tight.frames] == [(1.0, 5.0), (15.0, 25.0), (25.5, 35.0)]
    assert framer._segmenter is created[0] and framer._max_group == 29.0
    framer.cleanup()


def test_dynamic_token_budget_equals_the_reference_method(ref_modules, monkeypatch):
    """``QwenASR._compute_dynamic_token_limit`` (modules/qwen_asr.py:414-437) run from source on an instance built without
    its constructor, against the product helper and the oracle's restatement, over a grid of durations / rates / limits."""
    import itertools
    from oracle import qwen3_ref
    from whisperjav_amd import qwen
    sw = types.ModuleType("stable_whisper")
    sw.WhisperResult = object                 # only named in annotations of the module
    monkeypatch.setitem(sys.modules, "stable_whisper", sw)
    jp = types.ModuleType("whisperjav.modules.japanese_postprocessor")
    jp.JapanesePostProcessor = object
    monkeypatch.setitem(sys.modules, "whisperjav.modules.japanese_postprocessor", jp)
    ref = importlib.import_module("whisperjav.modules.qwen_asr")
    asr = ref.QwenASR.__new__(ref.QwenASR)
    for rate, floor, max_new, dur in itertools.product((0.0, 5.0, 20.0, 33.3), (0, 64, 256), (100, 512, 4096),
                                                       (0.0, 0.4, 2.49, 12.8, 29.99, 31.0, 250.0)):
        asr.max_tokens_per_audio_second, asr.min_tokens_floor, asr.max_new_tokens = rate, floor, max_new
        want = asr._compute_dynamic_token_limit(dur)
        assert qwen.dynamic_token_limit(dur, max_new, rate, floor) == want == qwen3_ref.dynamic_token_limit(dur, max_new, rate, floor)
    # the defaults the HIP generator carries are the reference generator's (generators/qwen3.py:32-42)
    gen = importlib.import_module("whisperjav.modules.subtitle_pipeline.generators.qwen3")
    import inspect
    sig = inspect.signature(gen.Qwen3TextGenerator.__init__).parameters
    mine = inspect.signature(qwen.HipQwenTextGenerator.__init__).parameters
    for name in ("repetition_penalty", "max_tokens_per_audio_second"):
        assert mine[name].default == sig[name].default


def test_punctuation_merge_equals_the_reference_function(ref_modules, monkeypatch):
    """``merge_master_with_timestamps`` (modules/qwen_asr.py:33-151, run from source) against the product's restatement on
    the docstring's example, the edge cases the reference names (empty transcript, no timestamps, leading / trailing
    punctuation, a word the transcript lacks, empty words, repeated words) and 400 random transcripts."""
    import random
    from whisperjav_amd import qwen
    sw = types.ModuleType("stable_whisper"); sw.WhisperResult = object
    monkeypatch.setitem(sys.modules, "stable_whisper", sw)
    jp = types.ModuleType("whisperjav.modules.japanese_postprocessor"); jp.JapanesePostProcessor = object
    monkeypatch.setitem(sys.modules, "whisperjav.modules.japanese_postprocessor", jp)
    ref = importlib.import_module("whisperjav.modules.qwen_asr").merge_master_with_timestamps

    def both(master, items):
        want = ref(master, [{"text": w, "start_time": a, "end_time": b} for w, a, b in items])
        got = qwen.merge_words_with_master(master, items)
        assert [(d["word"], d["start"], d["end"]) for d in want] == got, (master, items, want, got)
        return got
    assert both("けども、お前が。", [("けども", 1.0, 1.5), ("お前", 2.0, 2.3), ("が", 2.3, 2.5)])[0][0] == "けども、"
    both("", [("a", 0.0, 1.0)]); both("   ", []); both("そう。", []); both("「はい」 そう", [("はい", 0, 1), ("xx", 1, 2), ("そう", 2, 3)])
    both("。。。", [("", 0, 1)]); both("ああああ", [("あ", 0, 1), ("ああ", 1, 2), ("あ", None, 3)]); both("abc", [("zzz", None, None)])
    rng = random.Random(7)
    alphabet = "あいうえおかきくけこ"
    punct = "、。！？「」… "
    for _ in range(400):
        words = ["".join(rng.choice(alphabet) for _ in range(rng.randint(1, 4))) for _ in range(rng.randint(0, 7))]
        master = "".join(rng.choice(punct) for _ in range(rng.randint(0, 2)))
        for w in words:
            if rng.random() < 0.85:
                master += w
            master += "".join(rng.choice(punct) for _ in range(rng.choice((0, 0, 1, 2))))
        t, items = 0.0, []
        for w in words:
            if rng.random() < 0.1:
                w = rng.choice(("", "ん", w[::-1]))
            items.append((w, round(t, 2), round(t + 0.3, 2)))
            t += 0.4
        both(master, items)


def _wav_sf_stub():
    """A two-function stand-in for soundfile over the stdlib ``wave`` module (PCM16), for the orchestrator's audio I/O."""
    import wave
    import numpy as np
    sf = types.ModuleType("soundfile")

    def read(path, dtype="float32", **kw):
        with wave.open(str(path), "rb") as wf:
            pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
            return pcm, wf.getframerate()

    def write(path, audio, sr, **kw):
        with wave.open(str(path), "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(sr)
            wf.writeframes(np.clip(np.rint(np.asarray(audio) * 32767), -32768, 32767).astype("<i2").tobytes())
    sf.read, sf.write = read, write
    return sf


def test_reference_orchestrator_steps_run_pooled_over_the_hip_adapters(ref_modules, monkeypatch, tmp_path):
    """``qwen_pipeline.hip_decoupled_pipeline_class()``: the reference's ``DecoupledSubtitlePipeline`` (from source) frames the
    scenes, then its generation and alignment steps run over the HIP adapters (device engines replaced by doubles: no GPU
    here).  Pooled, the decoder sees ONE batch with every scene's clip and the aligner one batch with every (clip, text);
    texts, word timings and the inherited per-scene bookkeeping equal the un-pooled reference class's, which makes one engine
    call per scene."""
    import numpy as np
    import torch
    import wave
    from whisperjav_amd import qwen, qwen_pipeline
    monkeypatch.setitem(sys.modules, "soundfile", _wav_sf_stub())
    orch = importlib.import_module("whisperjav.modules.subtitle_pipeline.orchestrator")
    monkeypatch.setattr(orch, "sf", sys.modules["soundfile"])
    stypes = importlib.import_module("whisperjav.modules.subtitle_pipeline.types")
    FullScene = importlib.import_module("whisperjav.modules.subtitle_pipeline.framers.full_scene").FullSceneFramer
    Passthrough = importlib.import_module("whisperjav.modules.subtitle_pipeline.cleaners.passthrough").PassthroughCleaner
    d = qwen.Qwen3Dims(hidden=64, n_layer=1, n_head=1, n_kv_head=1, head_dim=128, ffn=64, vocab=400, audio_token_id=9)
    ad = qwen.Qwen3AudioDims(n_layer=1, n_head=1, ffn=64, d_model=64, conv_hidden=8, out_dim=64)
    calls = {"generate": [], "classify": []}

    class Tower:
        def encode(self, clips):
            return [torch.full((max(1, len(c) // 4000), d.hidden), float(len(c) % 97)) for c in clips]

        def close(self):
            pass

    class Decoder:
        def prompt_embeddings_many(self, prompts, audios):
            self.prompts = [list(p) for p in prompts]
            n = np.array([len(p) for p in prompts], dtype=np.int32)
            return torch.zeros((int(n.sum()), d.hidden)), n

        def prefill_packed(self, packed, n, want_logits=False):
            self.n = n

        def generate(self, max_new, repetition_penalty=1.0, prompt_ids=None, max_new_per_seq=None):
            calls["generate"].append(len(self.n))
            return qwen.GenerateResult([[100 + int(k) % 200, 7, 8][: (max_new_per_seq[b] if max_new_per_seq else 3)] for b, k in enumerate(self.n)],
                                       [[0.0] * 3 for _ in self.n])

        def classify(self, embeds, rows, head_w, head_b):
            calls["classify"].append(len(embeds))
            return [np.arange(len(r), dtype=np.int32) * 3 + int(e.shape[0]) % 5 for e, r in zip(embeds, rows)]

        def close(self):
            pass

    class Gen(qwen.HipQwenTextGenerator):
        def load(self):
            self._model, self._tower = self._model or Decoder(), self._tower or Tower()

    class Al(qwen.HipQwenForcedAligner):
        def load(self):
            self._model, self._tower = self._model or Decoder(), self._tower or Tower()
            self._head_w = self._head_b = None

    def word_prompt(n_audio, words, language):
        ids, marks = [11] + [d.audio_token_id] * n_audio, []
        for wd in words:
            ids += [20 + len(wd)]
            marks += [len(ids), len(ids) + 1]
            ids += [5, 5]
        return ids, marks

    def build(pooled):
        gen = Gen(d, {}, audio_dims=ad, prompt_builder=lambda n, lang, ctx: [11] + [d.audio_token_id] * n + [12],
                  detokenize=lambda t: "t" + "、".join(map(str, t)) + "。", batch_size=64, max_new_tokens=3)
        al = Al(d, ad, {}, word_prompt=word_prompt, split_words=lambda text, lang: text.replace("。", "").split("、"), batch_size=64)
        cls = qwen_pipeline.hip_decoupled_pipeline_class() if pooled else orch.DecoupledSubtitlePipeline
        return cls(FullScene(), gen, Passthrough(), al, stypes.HardeningConfig(), language="ja", context="cast: A")
    paths, durations = [], []
    for i, sec in enumerate((1.5, 2.25, 0.75)):
        path = tmp_path / f"scene_{i}.wav"
        sys.modules["soundfile"].write(path, 0.1 * np.sin(np.arange(int(sec * 16000)) * 0.01 * (i + 1)), 16000)
        paths.append(path); durations.append(sec)
    out = {}
    for pooled in (True, False):
        calls["generate"].clear(); calls["classify"].clear()
        pipe = build(pooled)
        assert isinstance(pipe.generator, importlib.import_module("whisperjav.modules.subtitle_pipeline.protocols").TextGenerator)
        frames, fpaths, _ = pipe._step1_frame_and_slice(paths, durations)
        texts = pipe._step2_4_generate_and_clean(frames, fpaths, durations)
        words = pipe._step5_7_align(frames, fpaths, texts, durations)
        out[pooled] = (texts, words, list(calls["generate"]), list(calls["classify"]))
        assert pipe.generator._model is None and pipe.aligner._model is None          # unloaded by the inherited finally blocks
    assert out[True][2] == [3] and out[True][3] == [3]              # one decoder batch, one aligner batch for the three scenes
    assert out[False][2] == [1, 1, 1] and out[False][3] == [1, 1, 1]
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]
    assert all(len(t) == 1 and t[0].startswith("t1") for t in out[True][0])
    assert [len(w[0]) for w in out[True][1]] == [3, 3, 3] and out[True][1][0][0][0]["word"].endswith("、")       # punctuation merged back


def test_reference_stable_ts_module_drives_the_hip_stable_shim(ref_modules, monkeypatch, tmp_path):
    """BASELINE cfg1 plumbing: the reference's ``StableTSASR`` (modules/stable_ts_asr.py, turbo mode = the
    ``faster`` / ``fast`` pipelines) imported from source, with ONE symbol swapped -- ``stable_whisper.load_faster_whisper`` ->
    ``whisperjav_amd.stable_shim.load_faster_whisper`` -- and its ``balanced`` preset as the legacy resolver packs it
    (config/components/asr/stable_ts.py:374-436, config/legacy.py:289-328).  The shim (over an engine double here) must accept
    every keyword ``_prepare_transcribe_parameters`` sends (stable_ts_asr.py:363-420), its result must flow through
    ``transcribe`` / ``_postprocess`` / ``_save_to_srt`` (:477-509, :623-655) and ``transcribe_to_srt`` must write the SRT and
    the diagnostic JSON (:570-595).  Both result flavours: the built-in one (stable-ts absent) and a ``WhisperResult`` class
    supplied by the ``stable_whisper`` module (stable-ts present: regrouping and silence suppression are ITS methods)."""
    import json
    import numpy as np
    import torch
    from tests import test_asr_adapter as doubles
    from whisperjav_amd import dims as pdims, stable_shim
    tb = pdims.special_tokens(51865).timestamp_begin
    engine = doubles._model([[tb, 11, 12, tb + 100, tb + 100, 13, tb + 150]])
    engine.model.align = lambda rows, n_prefix, heads, num_frames, slots=None, medfilt_width=7: [
        (np.repeat(np.arange(len(r) - 4), 10), np.arange(10 * (len(r) - 4)), np.full(len(r) - 5, 0.8, np.float32)) for r in rows]
    engine.dims = pdims.custom_dims(80, 128, 2, 2, 51865)
    # the preset searches with beam 2: the scripted engine answers the device beam call with its script
    beams = []
    engine.device_beam = True
    engine.model.decode_beam = lambda prompts, options, beam_size=5, patience=1.0, length_penalty=1.0, **kw: (
        beams.append((beam_size, patience, length_penalty)), engine.model.decode_greedy(prompts, options))[1]
    seen, loads = [], []
    real = engine.transcribe

    def spy(audio, **params):
        seen.append(params)
        return real(audio, **params)
    engine.transcribe = spy

    def load_faster_whisper(name, **kw):
        loads.append((name, kw))
        return stable_shim.HipStableWhisperModel(name, model=engine, **kw)
    sw = types.ModuleType("stable_whisper")
    sw.WhisperResult = object                     # "stable-ts absent" for the shim; only annotations use it in the module
    sw.load_faster_whisper = load_faster_whisper
    sf = types.ModuleType("soundfile")
    audio = (np.sin(np.arange(16000 * 9) * 0.05) * 0.2).astype(np.float32)
    sf.read = lambda path, dtype="float32", always_2d=False, **kw: (audio.copy(), 16000)
    for name, mod in (("stable_whisper", sw), ("soundfile", sf), ("librosa", types.ModuleType("librosa"))):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(torch.hub, "load", lambda *a, **kw: None)          # _precache_silero_vad: no network here
    ref = importlib.import_module("whisperjav.modules.stable_ts_asr")
    monkeypatch.setattr(ref, "snapshot_download", None)                    # _prefetch_faster_whisper_weights returns early
    decoder = dict(task="transcribe", language="ja", beam_size=2, best_of=1, patience=2.0, suppress_blank=True, without_timestamps=False)
    provider = dict(temperature=[0.0, 0.1], compression_ratio_threshold=2.4, logprob_threshold=-1.2, logprob_margin=0.2,
                    no_speech_threshold=0.5, drop_nonverbal_vocals=False, condition_on_previous_text=False, word_timestamps=True,
                    suppress_ts_tokens=False, gap_padding=" ...", only_ffmpeg=False, max_instant_words=0.5, ignore_compatibility=True,
                    nonspeech_error=0.1, only_voice_freq=False, regroup=True, ts_num=0, suppress_silence=True, suppress_word_ts=True,
                    suppress_attention=False, use_word_position=True, q_levels=20, k_size=5, demucs=False, vad=True, vad_threshold=0.25,
                    vad_repo="snakers4/silero-vad")
    asr = ref.StableTSASR({"model_name": "large-v2", "device": "cuda", "compute_type": "int8"},
                          {"decoder": decoder, "provider": provider}, "transcribe", turbo_mode=True)
    assert loads == [("large-v2", {"device": "cuda", "compute_type": "int8"})]
    out = asr.transcribe_to_srt(tmp_path / "scene_0001.wav", tmp_path / "out" / "scene_0001.srt")
    # what reached the HIP model: faster-whisper keywords only, word timings forced, stable-ts's own keywords consumed
    assert len(seen) == 1
    p = seen[0]
    assert p["word_timestamps"] is True and p["vad_filter"] is False and p["beam_size"] == 2 and p["patience"] == 2.0
    assert p["temperature"] == (0.0, 0.1) and p["language"] == "ja" and p["task"] == "transcribe"
    assert not ({"vad", "vad_threshold", "regroup", "batch_size", "verbose", "logprob_margin", "q_levels"} & set(p))
    assert beams and beams[0] == (2, 2.0, 1.0)             # beam 2, patience 2.0, CTranslate2's default length penalty
    text = out.read_text(encoding="utf-8")
    assert out == tmp_path / "out" / "scene_0001.srt"
    # segment edges come from the word timings (faster-whisper moves them there when word_timestamps is on)
    assert text == "1\n00:00:00,000 --> 00:00:00,400\n<11><12>\n\n2\n00:00:00,400 --> 00:00:00,600\n<13>\n", text
    dumped = json.loads((tmp_path / "out" / "scene_0001.transcribe.json").read_text(encoding="utf-8"))
    assert [s["text"] for s in dumped["segments"]] == ["<11><12>", "<13>"] and dumped["language"] == "ja"
    assert all(len(s["words"]) >= 1 for s in dumped["segments"])
    # a keyword neither faster-whisper nor stable-ts knows must surface as TypeError: the reference's retry keys on it (:517-548)
    with pytest.raises(TypeError):
        asr.model.transcribe(audio, not_an_option=1)

    # ---- stable-ts present: the result is ITS WhisperResult, silence suppression and regrouping are ITS methods -----------
    calls = []

    class WhisperResult:
        def __init__(self, result):
            calls.append(("init", [s["text"] for s in result["segments"]], result["language"]))
            self.segments = [types.SimpleNamespace(**s) for s in result["segments"]]

        def adjust_by_silence(self, samples, vad=False, vad_threshold=0.35, **kw):
            calls.append(("silence", len(samples), vad, vad_threshold))

        def regroup(self, algo=True):
            calls.append(("regroup", algo))
            return self
    sw.WhisperResult = WhisperResult
    res = asr.model.transcribe(audio, task="transcribe", language="ja", temperature=(0.0,), beam_size=2, vad=True, vad_threshold=0.25,
                               regroup=True, batch_size=8)
    assert isinstance(res, WhisperResult)
    assert calls == [("init", ["<11><12>", "<13>"], "ja"), ("silence", len(audio), True, 0.25), ("regroup", True)]
    asr.cleanup()


def test_reference_faster_pipeline_produces_its_srt_through_the_hip_stable_shim(ref_modules, monkeypatch, tmp_path):
    """BASELINE cfg1 as SURVEY 8 scopes it ("plumbing only: adapters import, SRT produced"): the reference's ``FasterPipeline``
    (pipelines/faster_pipeline.py:19-270) imported from source, its ``process()`` run unchanged -- audio extraction and SRT
    post-processing (outside the seam) replaced by file-level doubles -- over ``StableTSASR`` whose
    ``stable_whisper.load_faster_whisper`` is ``whisperjav_amd.stable_shim``'s.  The final ``<basename>.ja.whisperjav.srt`` must
    hold what the engine (a double here) decoded, through the shim's result object and the reference's own ``_save_to_srt``."""
    import wave
    import numpy as np
    import torch
    from tests import test_asr_adapter as doubles
    from whisperjav_amd import dims as pdims, stable_shim
    tb = pdims.special_tokens(51865).timestamp_begin
    engine = doubles._model([[tb, 21, 22, tb + 150, tb + 150, 23, tb + 300]])
    engine.model.align = lambda rows, n_prefix, heads, num_frames, slots=None, medfilt_width=7: [
        (np.repeat(np.arange(len(r) - 4), 10), np.arange(10 * (len(r) - 4)), np.full(len(r) - 5, 0.8, np.float32)) for r in rows]
    engine.dims = pdims.custom_dims(80, 128, 2, 2, 51865)
    engine.device_beam = True
    engine.model.decode_beam = lambda prompts, options, **kw: engine.model.decode_greedy(prompts, options)

    def wav(path, seconds):
        path.parent.mkdir(parents=True, exist_ok=True)
        pcm = (np.sin(np.arange(int(16000 * seconds)) * 0.05) * 8000).astype(np.int16)
        with wave.open(str(path), "wb") as wf:
            wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(pcm.tobytes())
        return path

    def sf_read(path, dtype="float32", always_2d=False, **kw):
        with wave.open(str(path), "rb") as wf:
            data = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
            return data, wf.getframerate()
    stubs = {name: types.ModuleType(name) for name in ("stable_whisper", "soundfile", "librosa", "pysrt", "srt", "tqdm", "ffmpeg", "jsonschema")}
    stubs["stable_whisper"].WhisperResult = object
    stubs["stable_whisper"].load_faster_whisper = lambda name, **kw: stable_shim.HipStableWhisperModel(name, model=engine, **kw)
    stubs["soundfile"].read = sf_read
    stubs["soundfile"].SoundFileError = Exception
    for attr in ("SubRipItem", "SubRipFile", "SubRipTime"):
        setattr(stubs["pysrt"], attr, type(attr, (), {}))
    stubs["tqdm"].tqdm = lambda it=None, **kw: it
    for name, mod in stubs.items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(torch.hub, "load", lambda *a, **kw: None)
    for _ in range(40):                 # anything else the application imports at module level and this container lacks
        try:
            fp = importlib.import_module("whisperjav.pipelines.faster_pipeline")
            break
        except ModuleNotFoundError as e:
            if e.name.startswith("whisperjav"):
                raise
            monkeypatch.setitem(sys.modules, e.name, types.ModuleType(e.name))
    else:
        pytest.skip("reference faster pipeline not importable here")
    sta = importlib.import_module("whisperjav.modules.stable_ts_asr")
    monkeypatch.setattr(sta, "snapshot_download", None)
    monkeypatch.setattr(fp, "AudioExtractor", lambda *a, **kw: types.SimpleNamespace(extract=lambda path, out: (wav(out, 12.0), 12.0)))

    class Post:
        def __init__(self, language="ja", **kw):
            self.language = language

        def process(self, srt_path, out_path, **kw):
            tmp = srt_path.with_suffix(".sanitized.srt")
            tmp.write_text(srt_path.read_text(encoding="utf-8"), encoding="utf-8")
            return tmp, {"total_subtitles": 2, "empty_removed": 0}
    monkeypatch.setattr(fp, "SRTPostProcessor", Post)
    resolved = {"model": {"model_name": "large-v2", "device": "cuda", "compute_type": "int8"},
                "params": {"decoder": dict(task="transcribe", language="ja", beam_size=2, best_of=1, patience=2.0),
                           "provider": dict(temperature=[0.0, 0.1], no_speech_threshold=0.5, condition_on_previous_text=False,
                                            word_timestamps=True, regroup=True, vad=True, vad_threshold=0.25, vad_repo="snakers4/silero-vad"),
                           "vad": {}},
                "features": {"post_processing": {}}, "task": "transcribe"}
    pipe = fp.FasterPipeline(output_dir=str(tmp_path / "out"), temp_dir=str(tmp_path / "temp"), keep_temp_files=True,
                             subs_language="native", resolved_config=resolved)
    assert isinstance(pipe.asr.model, stable_shim.HipStableWhisperModel)
    media = wav(tmp_path / "movie.wav", 12.0)
    pipe.process({"path": str(media), "basename": "movie", "type": "audio", "duration": 12.0})
    final = tmp_path / "out" / "movie.ja.whisperjav.srt"
    assert final.exists()
    assert final.read_text(encoding="utf-8") == "1\n00:00:00,000 --> 00:00:00,400\n<21><22>\n\n2\n00:00:00,400 --> 00:00:00,600\n<23>\n"


def test_reference_factories_create_the_hip_qwen_back_ends_from_checkpoint_directories(ref_modules, monkeypatch, tmp_path):
    """cfg5 as a runnable mode (VERDICT r3 missing #3): the reference's ``TextGeneratorFactory`` / ``TextAlignerFactory``
    (modules/subtitle_pipeline/{generators,aligners}/factory.py, imported from source) create the HIP back ends from ONE
    registry line each, with exactly the keyword sets ``QwenPipeline._build_subtitle_pipeline`` passes
    (pipelines/qwen_pipeline.py:455-466, 499-506).  The instances satisfy the reference's runtime-checkable protocols, load
    nothing at construction, and resolve a checkpoint DIRECTORY (written here by ``save_pretrained`` of transformers' qwen3_asr
    port, with its processor files) into engine geometry, tensors and tokenizer-side callables -- no hand-written plug-in."""
    import json
    pytest.importorskip("transformers")
    from tests import test_qwen_host as host
    from whisperjav_amd import qwen
    sys.modules["whisperjav.modules.subtitle_pipeline"] = types.ModuleType("whisperjav.modules.subtitle_pipeline")
    sys.modules["whisperjav.modules.subtitle_pipeline"].__path__ = [f"{REF}/whisperjav/modules/subtitle_pipeline"]
    protocols = importlib.import_module("whisperjav.modules.subtitle_pipeline.protocols")
    gfac = importlib.import_module("whisperjav.modules.subtitle_pipeline.generators.factory")
    afac = importlib.import_module("whisperjav.modules.subtitle_pipeline.aligners.factory")
    monkeypatch.setitem(gfac._REGISTRY, "qwen3-hip", "whisperjav_amd.qwen.HipQwen3TextGeneratorBackend")
    monkeypatch.setitem(afac._REGISTRY, "qwen3-hip", "whisperjav_amd.qwen.HipQwen3ForcedAlignerBackend")
    # ---- generator: a checkpoint directory with processor files ---------------------------------------------------------
    asr_dir = tmp_path / "asr"
    asr_dir.mkdir()
    host._tiny_transformers_checkpoint(asr_dir)
    proc, vocab = host._tiny_processor()
    proc.save_pretrained(str(asr_dir))
    gen = gfac.TextGeneratorFactory.create("qwen3-hip", model_id=str(asr_dir), device="cuda", dtype="auto", batch_size=4, max_new_tokens=4096,
                                           language="ja", repetition_penalty=1.1, max_tokens_per_audio_second=20.0,
                                           attn_implementation="auto")
    assert isinstance(gen, protocols.TextGenerator) and isinstance(gen, qwen.HipQwen3TextGeneratorBackend)
    assert gen.is_loaded is False and gen.dtype == "float16" and gen.batch_size == 4 and gen.max_new_tokens == 4096
    assert gen.repetition_penalty == 1.1 and gen.max_tokens_per_audio_second == 20.0
    gen._resolve()                                             # what load() does before touching the device
    assert gen.dims.hidden == 256 and gen.dims.n_layer == 2 and gen.audio_dims.d_model == 64
    assert "model.language_model.embed_tokens.weight" in gen._weights
    ids = gen.prompt_builder(4, "ja", None)
    assert ids.count(vocab["<|audio_pad|>"]) == 4 and gen.detokenize([vocab["hello"], vocab["world"]]) == "hello world"
    gen.unload()
    assert gen._weights == {} and gen.is_loaded is False
    # ---- aligner: the token-classification flavour of the family ---------------------------------------------------------
    import torch
    from transformers import Qwen3ASRConfig, Qwen3ASRForTokenClassification
    cfg = Qwen3ASRConfig.from_pretrained(str(asr_dir))
    cfg.num_labels = 12
    torch.manual_seed(1)
    al_dir = tmp_path / "aligner"
    Qwen3ASRForTokenClassification(cfg).save_pretrained(str(al_dir), safe_serialization=True)
    proc.save_pretrained(str(al_dir))
    raw = json.loads((al_dir / "config.json").read_text())
    raw["timestamp_token_id"] = vocab["<timestamp>"]
    (al_dir / "config.json").write_text(json.dumps(raw))
    al = afac.TextAlignerFactory.create("qwen3-hip", aligner_id=str(al_dir), device="cuda", dtype="bfloat16", language="Japanese")
    assert isinstance(al, protocols.TextAligner) and al.is_loaded is False and al.dtype == "bfloat16"
    al._resolve()
    assert al._weights["score.weight"].shape == (12, 256)
    words = al.split_words("hello world", "en")
    pids, marks = al.word_prompt(3, words, "en")
    assert words == ["hello", "world"] and len(marks) == 4 and all(pids[i] == vocab["<timestamp>"] for i in marks)
    # a generator checkpoint is not an aligner checkpoint; an unknown repository is not fetched
    bad = afac.TextAlignerFactory.create("qwen3-hip", aligner_id=str(asr_dir), device="cuda", dtype="auto", language="Japanese")
    with pytest.raises(KeyError, match="forced-aligner"):
        bad._resolve()
    with pytest.raises(FileNotFoundError, match="never fetches"):
        gfac.TextGeneratorFactory.create("qwen3-hip", model_id="Qwen/Qwen3-ASR-1.7B", device="cuda", dtype="auto", batch_size=1,
                                         max_new_tokens=64, language="ja", repetition_penalty=1.1, max_tokens_per_audio_second=20.0,
                                         attn_implementation="auto")._resolve()
    with pytest.raises(ValueError, match="no CPU path"):
        gfac.TextGeneratorFactory.create("qwen3-hip", model_id=str(asr_dir), device="cpu", dtype="auto", batch_size=1, max_new_tokens=64,
                                         language="ja", repetition_penalty=1.1, max_tokens_per_audio_second=20.0, attn_implementation="auto")
