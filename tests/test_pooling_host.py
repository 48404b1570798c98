"""Host logic of round 2's seams, with fakes (no GPU): cross-scene pooling in the ASR adapter, the priming path that
serves the reference's per-scene loop, the retry ladder, the explicit VAD parameter sources, the ``none`` / TEN
segmenters and ``HipBalancedPipeline`` driven through the REFERENCE's own ``BalancedPipeline.process`` (from source)."""
import importlib
import os
import sys
import types
import wave

import numpy as np
import pytest

from tests.test_asr_adapter import CONFIG, FakeSegmenter, FakeWhisper, _seg
from whisperjav_amd import asr, pipeline, segmenters, vad_weights

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "whisperjav")), reason="reference tree not present")


def _scene_wav(path, seconds, freq=0.05):
    pcm = (np.sin(np.arange(int(16000 * seconds)) * freq) * 8000).astype("<i2")
    with wave.open(str(path), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(pcm.tobytes())
    return path


class LengthSegmenter(FakeSegmenter):
    """Groups depend on the clip length, so every scene gets its own plan (and an 11 s scene gets none)."""

    def __init__(self):
        super().__init__([])
        self.many_calls = 0

    def segment(self, audio, sample_rate=16000, **kw):
        dur = len(audio) / sample_rate
        self.groups = [] if abs(dur - 11.0) < 0.01 else [[(0.5, min(2.0, dur / 2))], [(dur / 2 + 0.25, dur - 0.5)]]
        return super().segment(audio, sample_rate, **kw)

    def segment_many(self, audios, sample_rates):
        self.many_calls += 1
        return [self.segment(a, sample_rate=sr) for a, sr in zip(audios, sample_rates)]


def _script(i, clip):
    n = len(clip)
    return [_seg(1, 0.1, 0.6, f" clip{n} "), _seg(2, 0.7, 0.9, "Thank you", lp=-0.95 + 0.2 * (n % 3))]


def test_pooled_scenes_equal_per_scene_calls(tmp_path):
    paths = [_scene_wav(tmp_path / f"m_scene_{i:04d}.wav", s) for i, s in enumerate((6.0, 11.0, 9.0, 4.0))]
    fake, seg = FakeWhisper(_script), LengthSegmenter()
    a = asr.HipFasterWhisperProASR({"model_name": "large-v3"}, CONFIG, "transcribe", whisper_model=fake, segmenter=seg)
    per_scene = [a.transcribe(p) for p in paths]
    stats_loop, calls_loop = a.get_filter_statistics(), len(fake.calls)
    a.reset_statistics()
    pooled = a.transcribe_scenes(paths)
    assert pooled == per_scene
    assert a.get_filter_statistics() == stats_loop
    assert calls_loop == 3 and len(fake.calls) == 4 and fake.calls[-1][0] == 6      # 3 scenes x 2 groups in ONE call
    assert seg.many_calls == 1                                                       # one segmentation launch
    vad = a.get_vad_segments_per_scene()
    assert len(vad) == 4 and vad[1] == [] and len(vad[0]) == 2
    # in-memory scenes (audio, rate) are accepted as well
    clips = [asr.read_audio(p) for p in paths]
    assert a.transcribe_scenes(clips) == per_scene


def test_priming_serves_the_reference_loop_from_one_pooled_pass(tmp_path):
    paths = [_scene_wav(tmp_path / f"m_scene_{i:04d}.wav", s) for i, s in enumerate((6.0, 9.0, 4.0))]
    fake = FakeWhisper(_script)
    a = asr.HipFasterWhisperProASR({"model_name": "large-v3"}, CONFIG, "transcribe", whisper_model=fake, segmenter=LengthSegmenter())
    loop = []
    for p in paths:                                           # the unprimed loop: one engine call per scene
        srt = a.transcribe_to_srt(p, tmp_path / "loop" / (p.stem + ".srt"), task="transcribe")
        loop.append((srt.read_text(encoding="utf-8"), a.get_last_vad_segments()))
    assert len(fake.calls) == 3
    a.prime_scenes(paths)
    primed = []
    for p in paths:                                           # BalancedPipeline.process' loop shape (:429-500)
        srt = a.transcribe_to_srt(p, tmp_path / "primed" / (p.stem + ".srt"), task="transcribe")
        primed.append((srt.read_text(encoding="utf-8"), a.get_last_vad_segments()))
    assert primed == loop
    assert len(fake.calls) == 4 and fake.calls[-1][0] == 6 and a.pooled_calls == 1
    # a path that was never announced still works, and the cache does not serve stale results
    extra = _scene_wav(tmp_path / "other.wav", 5.0)
    a.transcribe(extra)
    a.transcribe(paths[0])
    assert len(fake.calls) == 6


def test_engine_failure_falls_back_to_minimal_parameters_then_drops_the_clip(tmp_path):
    class Flaky(FakeWhisper):
        def transcribe_many(self, clips, **params):
            if "patience" in params:
                raise RuntimeError("engine rejected the full parameter set")
            if len(clips[0]) < 40000:
                raise RuntimeError("still broken for the short clip")
            return super().transcribe_many(clips, **params)
    fake = Flaky(_script)
    a = asr.HipFasterWhisperProASR({}, CONFIG, "transcribe", whisper_model=fake, segmenter=LengthSegmenter())
    res = a.transcribe(_scene_wav(tmp_path / "s.wav", 9.0))
    assert len(fake.calls) == 1 and set(fake.calls[0][2]) == {"task", "language", "temperature", "beam_size", "log_progress"}
    assert fake.calls[0][2]["beam_size"] == 3 and len(res["segments"]) >= 1          # the long group survived, the short one was dropped


def test_one_bad_clip_does_not_demote_the_whole_pooled_call(tmp_path):
    """a pooled engine error is bisected with the FULL parameters; only the clip that still fails alone goes
    down the reference's per-group ladder (minimal parameters, then dropped)."""
    paths = [_scene_wav(tmp_path / f"m_scene_{i:04d}.wav", s) for i, s in enumerate((6.0, 9.0, 4.0))]
    clean = asr.HipFasterWhisperProASR({"model_name": "large-v3"}, CONFIG, "transcribe", whisper_model=FakeWhisper(_script),
                                       segmenter=LengthSegmenter())
    good = clean.transcribe_scenes(paths)
    # poison only clips of exactly the 4 s scene's second group length
    second = int((4.0 - 0.5) * 16000) - int((2.0 + 0.25) * 16000)

    class Poisoned(FakeWhisper):
        def transcribe_many(self, clips, **params):
            if "patience" in params and second in [len(c) for c in clips]:
                raise RuntimeError("device fault on one clip")
            return super().transcribe_many(clips, **params)
    fake2 = Poisoned(_script)
    b = asr.HipFasterWhisperProASR({"model_name": "large-v3"}, CONFIG, "transcribe", whisper_model=fake2, segmenter=LengthSegmenter())
    got = b.transcribe_scenes(paths)
    full = [c for c in fake2.calls if "patience" in c[2]]
    minimal = [c for c in fake2.calls if "patience" not in c[2]]
    assert sum(c[0] for c in full) == 5 and len(minimal) == 1 and minimal[0][0] == 1      # 5 clips kept the full parameters
    assert got[0] == good[0] and got[1] == good[1]                                       # untouched scenes: identical transcripts


def test_unreadable_announced_scene_fails_on_its_own_call(tmp_path):
    """the pooled pass over a primed scene list skips files it cannot read (they fail when the loop reaches
    them) and a failing pooled pass falls through to the per-scene path."""
    paths = [_scene_wav(tmp_path / f"m_scene_{i:04d}.wav", s) for i, s in enumerate((6.0, 9.0))]
    broken = tmp_path / "m_scene_0002.wav"
    broken.write_bytes(b"not a wav file")
    a = asr.HipFasterWhisperProASR({"model_name": "large-v3"}, CONFIG, "transcribe", whisper_model=FakeWhisper(_script),
                                   segmenter=LengthSegmenter())
    a.prime_scenes(paths + [broken])
    first = a.transcribe(paths[0])                               # must not raise because of scene 2
    assert first["segments"] and a.pooled_calls == 1
    assert a.transcribe(paths[1])["segments"]
    with pytest.raises(Exception):
        a.transcribe(broken)

    class BoomSegmenter(LengthSegmenter):
        def segment_many(self, audios, sample_rates):
            raise RuntimeError("segmenter fault in the pooled pass")
    b = asr.HipFasterWhisperProASR({"model_name": "large-v3"}, CONFIG, "transcribe", whisper_model=FakeWhisper(_script),
                                   segmenter=BoomSegmenter())
    b.prime_scenes(paths)
    assert b.transcribe(paths[0]) == first and b.pooled_calls == 0          # served by the per-scene path


def test_tracer_receives_one_record_per_group(tmp_path):
    seen = []

    class Tracer:
        def emit_transcribe_params(self, params, audio_info, context):
            seen.append((context, audio_info["sample_rate"], sorted(audio_info)))
    a = asr.HipFasterWhisperProASR({}, CONFIG, "transcribe", tracer=Tracer(), whisper_model=FakeWhisper(_script),
                                   segmenter=LengthSegmenter())
    a.transcribe(_scene_wav(tmp_path / "s.wav", 6.0))
    assert [c for c, _, _ in seen] == ["vad_group_0.50s-2.00s", "vad_group_3.25s-5.50s"] and seen[0][1] == 16000


def test_none_backend_and_unknown_backend_without_the_reference_package(monkeypatch):
    monkeypatch.setitem(sys.modules, "whisperjav", None)      # stand-alone: the reference factory is not importable
    seg = asr.HipFasterWhisperProASR._create_segmenter("none", {"threshold": 0.3})
    assert seg.name == "none"
    res = seg.segment(np.zeros(32000, dtype=np.float32), sample_rate=16000)
    assert len(res.groups) == 1 and res.segments[0].end_sample == 32000 and res.segments[0].metadata == {"bypass": True}
    with pytest.raises(ValueError, match="Unknown speech segmenter backend"):
        asr.HipFasterWhisperProASR._create_segmenter("whisper-vad", {})
    with pytest.raises(ValueError, match="Unknown speech segmenter backend"):
        asr.HipFasterWhisperProASR._create_segmenter("nemo", {})


def test_vad_parameters_are_never_silently_synthetic(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "silero_vad", None)
    with pytest.raises(FileNotFoundError, match="trained Silero VAD parameters"):
        vad_weights.resolve(None)
    w = vad_weights.resolve("synthetic")
    assert vad_weights.pack(w).shape[0] == vad_weights.BLOB_FLOATS
    np.savez(tmp_path / "vad.npz", **w)
    back = vad_weights.resolve(None, str(tmp_path / "vad.npz"))
    assert all(np.array_equal(back[k], w[k]) for k in w)
    # a file in the TorchScript naming of silero_vad.load_silero_vad().state_dict()
    jit = {"_model.stft.forward_basis_buffer": w["stft.forward_basis_buffer"], "_model.decoder.decoder.2.weight": w["out.weight"],
           "_model.decoder.decoder.2.bias": w["out.bias"]}
    for i in range(4):
        jit[f"_model.encoder.{i}.reparam_conv.weight"] = w[f"encoder.{i}.weight"]
        jit[f"_model.encoder.{i}.reparam_conv.bias"] = w[f"encoder.{i}.bias"]
    for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        jit[f"_model.decoder.rnn.{n}"] = w[f"rnn.{n}"]
    np.savez(tmp_path / "jit.npz", **jit)
    assert np.array_equal(vad_weights.pack(vad_weights.load_file(str(tmp_path / "jit.npz"))), vad_weights.pack(w))


def test_real_silero_state_dict_loads_when_the_package_is_present():
    """Consumes the trained model when ``silero_vad`` is installed (it is not, offline): the TorchScript state dict must
    map onto the blob layout and the oracle restatement must reproduce the package's own probabilities."""
    silero_vad = pytest.importorskip("silero_vad")
    import torch
    from oracle import silero_ref
    from whisperjav_amd import synth
    jit = silero_vad.load_silero_vad()
    w = vad_weights.from_jit_state_dict(jit.state_dict())
    assert vad_weights.pack(w).shape[0] == vad_weights.BLOB_FLOATS
    audio = synth.speech_like(6.0, seed=3)
    jit.reset_states()
    ref = [float(jit(torch.from_numpy(audio[i: i + 512]), 16000)) for i in range(0, len(audio) - 511, 512)]
    got = silero_ref.SileroOracle(w).probs(audio)[: len(ref)]
    assert np.abs(np.array(ref) - got).max() < 1e-5


def test_pcm16_round_trip_and_resampling():
    # libsndfile: write = lrint(x * 0x7FFF) stored into a short (no clipping), read = int16 / 0x8000
    x = np.array([0.0, 0.5, -1.0, 0.999999, 1.0 / 32768 * 0.4, 1.0, 1.0001, 3.0 / 32767 * 0.5], dtype=np.float32)
    assert pipeline.pcm16_encode(x).tolist() == [0, 16384, -32767, 32767, 0, 32767, -32766, 2]      # half to even; wrap
    q = pipeline.pcm16_roundtrip(x)
    assert q.dtype == np.float32 and q.tolist()[:6] == [0.0, 0.5, -32767 / 32768, 32767 / 32768, 0.0, 32767 / 32768]
    assert pipeline.pcm16_encode(x.astype(np.float64)).tolist()[:6] == [0, 16384, -32767, 32767, 0, 32767]
    t = np.arange(48000) / 48000.0
    y = pipeline.to_16k(np.sin(2 * np.pi * 440 * t).astype(np.float32), 48000)
    assert y.shape == (16000,) and y.dtype == np.float32
    assert abs(np.abs(np.fft.rfft(y)).argmax() - 440) <= 1


def test_pcm16_round_trip_matches_soundfile():
    """The scene round trip of the reference -- ``sf.write(path, x, sr, subtype="PCM_16")`` (scene_detection_backends/utils.py:140)
    then ``sf.read(path, dtype="float32")`` (faster_whisper_pro_asr.py:477) -- must equal ``pcm16_roundtrip`` bit for bit, and the raw
    int16 samples ``pcm16_encode``.  soundfile (libsndfile) live where it is installed, ``tests/golden/upstream_soundfile_pcm16.npz``
    (scripts/make_upstream_fixtures.py) where it is not, skipped while there is neither (parity unpinned, PARITY.md)."""
    from tests import upstream_cases as U
    ref, _ = U.reference("soundfile_pcm16")
    x = U.pcm16_input()
    assert int(ref["sr"]) == 16000 and np.array_equal(ref["back"], pipeline.pcm16_roundtrip(x))
    assert np.array_equal(ref["raw"], pipeline.pcm16_encode(x))


def test_recording_transcriber_stitches_in_scene_order():
    class Det:
        def split_clip(self, audio, sr):
            return [(8.0, 14.0, 1, {}), (0.0, 6.0, 1, {}), (20.0, 29.0, 2, {})], []
    fake = FakeWhisper(_script)
    a = asr.HipFasterWhisperProASR({}, CONFIG, "transcribe", whisper_model=fake, segmenter=LengthSegmenter())
    audio = (np.sin(np.arange(16000 * 30) * 0.05) * 0.25).astype(np.float32)
    out = pipeline.RecordingTranscriber(a, Det()).transcribe(audio, 16000)
    starts = [s["start"] for s in out["segments"]]
    assert starts == sorted(starts) and starts[0] == pytest.approx(0.6) and len(fake.calls) == 1
    assert out["vad_segments"][0][0]["start_sec"] == 8.5 and len(out["per_scene"]) == 3
    single = pipeline.RecordingTranscriber(a, Det()).transcribe(audio, 16000, pooled=False)
    assert single["segments"] == out["segments"] and len(fake.calls) == 4


# ---- against the reference's own classes (from source) -------------------------------------------------------------
@pytest.fixture()
def ref_modules(monkeypatch):
    saved = {k: v for k, v in sys.modules.items() if k == "whisperjav" or k.startswith("whisperjav.")}
    for k in saved:
        del sys.modules[k]
    for name, path in (("whisperjav", f"{REF}/whisperjav"), ("whisperjav.modules", f"{REF}/whisperjav/modules"),
                       ("whisperjav.pipelines", f"{REF}/whisperjav/pipelines")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    yield
    for k in [k for k in sys.modules if k == "whisperjav" or k.startswith("whisperjav.")]:
        del sys.modules[k]
    sys.modules.update(saved)


@needs_ref
def test_ten_segmenter_reuses_the_reference_post_ops_with_a_pluggable_scorer(ref_modules):
    """``HipTenSpeechSegmenter`` inherits every post-op of the reference's TenSpeechSegmenter; with a scorer that returns
    the probabilities a fake ``TenVad`` model would, the two classes give identical segments and groups."""
    ten = importlib.import_module("whisperjav.modules.speech_segmentation.backends.ten")
    rng = np.random.default_rng(5)
    audio = (rng.standard_normal(16000 * 12) * 0.1).astype(np.float32)
    hop = 256
    n_frames = (len(audio) + hop - 1) // hop
    t = np.arange(n_frames) * hop / 16000.0
    probs = np.clip(0.5 + 0.45 * np.sin(2 * np.pi * t / 3.1) + 0.05 * rng.standard_normal(n_frames), 0, 1)

    class FakeTenVad:           # the frame API of ten_vad.TenVad (ten.py:232-239)
        def __init__(self, threshold):
            self.i, self.thr = 0, threshold
            self.out_flags, self.out_probability = types.SimpleNamespace(value=0), types.SimpleNamespace(value=0.0)

        def process(self, frame):
            p = float(probs[self.i]); self.i += 1
            self.out_probability.value, self.out_flags.value = p, int(p >= self.thr)
    kw = dict(threshold=0.3, hop_size=hop, min_silence_duration_ms=120, max_speech_duration_s=2.5, chunk_threshold_s=0.8,
              max_group_duration_s=6.0)
    ref = ten.TenSpeechSegmenter(**kw)
    ref._model = FakeTenVad(0.3)
    want = ref.segment(audio, sample_rate=16000)
    seen = []
    cls = segmenters.hip_ten_segmenter_class()
    assert issubclass(cls, ten.TenSpeechSegmenter)
    mine = cls(scorer=lambda pcm16, h: (seen.append((pcm16.dtype, len(pcm16), h)) or probs), **kw)
    got = mine.segment(audio, sample_rate=16000)
    assert seen == [(np.dtype("int16"), len(audio), hop)]                         # ONE scorer call per clip
    assert mine.name == "ten-hip" and len(got.segments) >= 3
    assert [(s.start_sample, s.end_sample, s.start_sec, s.end_sec) for s in got.segments] == \
           [(s.start_sample, s.end_sample, s.start_sec, s.end_sec) for s in want.segments]
    assert [[(s.start_sample, s.end_sample) for s in g] for g in got.groups] == [[(s.start_sample, s.end_sample) for s in g] for g in want.groups]
    assert got.to_legacy_format() == want.to_legacy_format()


def _reference_pipeline_with_doubles(monkeypatch, tmp_path, module_name):
    """Import ``whisperjav.pipelines.<module_name>`` from source (third-party wheels stubbed) and replace what lies
    outside the seam -- ffmpeg extraction, scene detection, stitching, post-processing -- by file-level doubles."""
    stubs = {}
    for name in ("faster_whisper", "whisper", "soundfile", "srt", "pysrt", "jsonschema", "tqdm", "ffmpeg", "librosa", "auditok"):
        stubs[name] = types.ModuleType(name)
    stubs["faster_whisper"].WhisperModel = object
    stubs["whisper"].load_model = lambda *a, **k: None
    stubs["soundfile"].SoundFileError = Exception
    stubs["soundfile"].read = lambda path, dtype="float32", **kw: _wave_read(path)
    for attr in ("SubRipItem", "SubRipFile", "SubRipTime"):
        setattr(stubs["pysrt"], attr, type(attr, (), {}))
    stubs["tqdm"].tqdm = lambda it=None, **kw: it
    for name, mod in stubs.items():
        monkeypatch.setitem(sys.modules, name, mod)
    missing = []
    for _ in range(40):             # anything else the application imports at module level and the container lacks
        try:
            bp = importlib.import_module(f"whisperjav.pipelines.{module_name}")
            break
        except ModuleNotFoundError as e:
            if e.name.startswith("whisperjav"):
                raise
            missing.append(e.name)
            monkeypatch.setitem(sys.modules, e.name, types.ModuleType(e.name))
    else:
        pytest.skip(f"reference pipeline not importable here (stubbed {missing})")
    scenes = [(0.0, 6.0), (8.0, 17.0), (20.0, 24.0)]

    class Detector:
        name = "auditok-hip"

        def detect_scenes(self, audio_path, output_dir, media_basename, **kw):
            output_dir.mkdir(parents=True, exist_ok=True)
            tuples = []
            for i, (a, b) in enumerate(scenes):
                p = _scene_wav(output_dir / f"{media_basename}_scene_{i:04d}.wav", b - a)
                tuples.append((p, a, b, b - a))
            return types.SimpleNamespace(
                to_legacy_tuples=lambda: tuples,
                to_metadata_dict=lambda: {"scenes_detected": [{"scene_index": i} for i in range(len(tuples))]})

        def cleanup(self):
            pass
    monkeypatch.setattr(bp.SceneDetectorFactory, "safe_create_from_legacy_kwargs", staticmethod(lambda **kw: Detector()))
    monkeypatch.setattr(bp, "AudioExtractor", lambda sample_rate=16000: types.SimpleNamespace(
        extract=lambda path, out: (_scene_wav(out, 25.0), 25.0)))
    stitched = {}

    class Stitcher:
        def stitch(self, scene_srt_info, out_path):
            stitched["info"] = [(str(p), s) for p, s in scene_srt_info]
            out_path.write_text("".join(p.read_text(encoding="utf-8") for p, _ in scene_srt_info), encoding="utf-8")
            return sum(p.read_text(encoding="utf-8").count("-->") for p, _ in scene_srt_info)
    monkeypatch.setattr(bp, "SRTStitcher", Stitcher)

    class Post:
        def __init__(self, language="ja", **kw):
            pass

        def process(self, srt_path, out_path, **kw):
            out_path.write_text(srt_path.read_text(encoding="utf-8"), encoding="utf-8")
            return out_path, {"total_subtitles": 0, "empty_removed": 0, "removed_hallucinations": 0, "removed_repetitions": 0,
                              "duration_adjustments": 0, "cps_filtered": 0, "logprob_filtered": 0, "nonverbal_filtered": 0}
    monkeypatch.setattr(bp, "StandardPostProcessor", Post)
    return bp, stitched


def _run_pipeline(cls, resolved, tmp_path, stitched):
    pipe = cls(output_dir=str(tmp_path / "out"), temp_dir=str(tmp_path / "temp"), keep_temp_files=True, subs_language="native",
               resolved_config=resolved)
    media = tmp_path / "movie.wav"
    _scene_wav(media, 25.0)
    try:
        pipe.process({"path": str(media), "basename": "movie", "type": "audio", "duration": 25.0})
    except Exception:            # anything past the ASR phase that the doubles do not model is outside this seam
        if not stitched:
            raise


RESOLVED = {"model": {"model_name": "large-v3", "device": "cuda", "compute_type": "float16"},
            "params": {"decoder": dict(CONFIG["decoder"]), "provider": dict(CONFIG["provider"]), "vad": {"threshold": 0.28},
                       "speech_segmenter": {"backend": "silero-v6.2-hip"}},
            "features": {"scene_detection": {"method": "auditok-hip"}, "post_processing": {}}, "task": "transcribe"}


@needs_ref
def test_hip_balanced_pipeline_pools_inside_the_reference_process_loop(ref_modules, monkeypatch, tmp_path):
    """The pipeline-level seam: ``HipBalancedPipeline`` IS the reference's ``BalancedPipeline`` (imported from source;
    ffmpeg extraction, stitching and post-processing replaced by file-level doubles) -- its inherited ``process`` loop
    calls ``transcribe_to_srt`` per scene, and the engine is entered ONCE with the groups of all scenes; the per-scene
    SRT files equal those of the unpooled module."""
    bp, stitched = _reference_pipeline_with_doubles(monkeypatch, tmp_path, "balanced_pipeline")
    fake = FakeWhisper(_script)
    real_asr = asr.HipFasterWhisperProASR

    def make_asr(model_config, params, task, tracer=None):
        return real_asr(model_config, params, task, tracer, whisper_model=fake, segmenter=LengthSegmenter())
    monkeypatch.setattr(asr, "HipFasterWhisperProASR", make_asr)
    cls = pipeline.hip_balanced_pipeline_class()
    assert issubclass(cls, bp.BalancedPipeline) and issubclass(cls, importlib.import_module("whisperjav.pipelines.base_pipeline").BasePipeline)
    _run_pipeline(cls, RESOLVED, tmp_path, stitched)
    assert len(fake.calls) == 1 and fake.calls[0][0] == 6, fake.calls           # ONE engine call: 3 scenes x 2 groups
    assert [s for _, s in stitched["info"]] == [0.0, 8.0, 20.0]                  # one SRT per scene reached the stitcher
    # the same scenes through the unpooled module give byte-identical per-scene SRTs
    loop = real_asr(RESOLVED["model"], RESOLVED["params"], "transcribe", whisper_model=FakeWhisper(_script), segmenter=LengthSegmenter())
    for i in range(3):
        wavp = tmp_path / "temp" / "scenes" / f"movie_scene_{i:04d}.wav"
        want = loop.transcribe_to_srt(wavp, tmp_path / "loop" / f"{i}.srt").read_text(encoding="utf-8")
        assert (tmp_path / "temp" / "scene_srts" / f"movie_scene_{i:04d}.srt").read_text(encoding="utf-8") == want


@needs_ref
def test_hip_fidelity_pipeline_pools_inside_the_reference_process_loop(ref_modules, monkeypatch, tmp_path):
    """The same for fidelity mode: ``HipFidelityPipeline`` IS the reference's ``FidelityPipeline``; the ASR module its
    ``process()`` creates as a local (``WhisperProASR(**self._asr_config)``, fidelity_pipeline.py:270) becomes
    ``HipWhisperProASR``, primed with the scene list, and the model is entered once for all scenes."""
    fp, stitched = _reference_pipeline_with_doubles(monkeypatch, tmp_path, "fidelity_pipeline")
    calls = []

    class DictModel:             # whisper.load_model(...) object: transcribe(audio, **kw) -> dict
        def transcribe_many(self, clips, **params):
            calls.append(len(clips))
            return [_script(i, c) for i, c in enumerate(clips)], [None] * len(clips)

        def close(self):
            pass
    real_asr = asr.HipWhisperProASR
    made = []

    def make_asr(model_config, params, task, tracer=None):
        made.append(real_asr(model_config, params, task, tracer, whisper_model=DictModel(), segmenter=LengthSegmenter()))
        return made[-1]
    monkeypatch.setattr(asr, "HipWhisperProASR", make_asr)
    cls = pipeline.hip_fidelity_pipeline_class()
    assert issubclass(cls, fp.FidelityPipeline)
    saved = fp.WhisperProASR
    _run_pipeline(cls, RESOLVED, tmp_path, stitched)
    assert fp.WhisperProASR is saved                                             # the module-level name is restored
    assert calls == [6] and len(made) == 1 and made[0].pooled_calls == 1
    assert [s for _, s in stitched["info"]] == [0.0, 8.0, 20.0]
    assert made[0].post_model_filter_enabled is True                            # fidelity's post-model gate default (whisper_pro_asr.py:123-127)


def _wave_read(path):
    with wave.open(str(path), "rb") as wf:
        sr = wf.getframerate()
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
    return pcm, sr
