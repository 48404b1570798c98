"""Generates tests/golden/reference_asr_adapter.json by driving the REFERENCE's own ``FasterWhisperProASR``
(/root/reference/whisperjav/modules/faster_whisper_pro_asr.py, imported from source) with scripted doubles for the
three things that are not installable offline: ``faster_whisper.WhisperModel`` (returns scripted segments that depend
only on the clip length), ``soundfile.read`` (serves in-memory clips) and the speech segmenter (scripted groups built
from the reference's own ``SpeechSegment`` / ``SegmentationResult``).  What it pins, bit for bit, is the reference's
ASR-module logic around the model call: parameter normalisation, group slicing, timestamp shifting, the logprob /
nonverbal / suppress-phrase filters, the VAD fail-over, statistics and the returned dict -- the behaviour
``whisperjav_amd.asr.HipFasterWhisperProASR`` has to reproduce (tests/test_asr_adapter.py).

Run from the repo root inside the build container:  python tests/golden/make_asr_adapter_fixtures.py
"""
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.append("/root/reference")      # after the repo: `tests` must resolve to ours, `whisperjav` to the reference

from tests.helpers import scripted_segments  # noqa: E402  (shared with the test so both sides see the same "model")

CALLS = []
AUDIO = {}


class FakeInfo:
    language, language_probability = "ja", 1.0


class FakeSegment:
    def __init__(self, d):
        self.__dict__.update(d)
        self.words = None


class FakeWhisperModel:
    def __init__(self, *a, **kw):
        self.ctor = (a, kw)

    def transcribe(self, audio, **params):
        CALLS.append({"n": int(len(audio)), "params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in params.items()}})
        return iter([FakeSegment(d) for d in scripted_segments(len(audio))]), FakeInfo()


def install():
    fw = types.ModuleType("faster_whisper")
    fw.WhisperModel = FakeWhisperModel
    sys.modules["faster_whisper"] = fw
    sf = types.ModuleType("soundfile")
    sf.SoundFileError = type("SoundFileError", (Exception,), {})
    sf.read = lambda path, dtype="float32", **kw: (AUDIO[str(path)].copy(), 16000)
    sys.modules["soundfile"] = sf
    sys.modules["srt"] = types.ModuleType("srt")
    return importlib.import_module("whisperjav.modules.faster_whisper_pro_asr")


SCENARIOS = [
    dict(name="two_groups", seconds=10.0, groups=[[(1.0, 2.0), (2.2, 3.0)], [(5.0, 6.5)]], post_filter=True, threshold=-1.0,
         margin=0.0, drop_nonverbal=False),
    dict(name="filter_off", seconds=12.0, groups=[[(0.5, 4.0)], [(6.0, 6.4)], [(8.0, 11.5)]], post_filter=False, threshold=-1.0,
         margin=0.0, drop_nonverbal=False),
    dict(name="strict_threshold", seconds=20.0, groups=[[(0.0, 5.9)], [(7.0, 12.0), (12.5, 13.0)]], post_filter=True,
         threshold=-0.5, margin=0.1, drop_nonverbal=True),
    dict(name="no_speech", seconds=8.0, groups=[], post_filter=True, threshold=-1.0, margin=0.0, drop_nonverbal=False),
    dict(name="tiny_coverage_failover", seconds=60.0, groups=[[(10.0, 10.2)]], post_filter=True, threshold=-1.0, margin=0.0,
         drop_nonverbal=False),
]


def main():
    mod = install()
    base = importlib.import_module("whisperjav.modules.speech_segmentation.base")
    out = []
    for sc in SCENARIOS:
        groups = sc["groups"]

        class Seg:
            name = "silero-v6.2"
            display_name = "fake"

            def segment(self, audio, sample_rate=16000, **kw):
                segs = [[base.SpeechSegment(start_sec=a, end_sec=b, start_sample=int(a * sample_rate), end_sample=int(b * sample_rate))
                         for a, b in g] for g in groups]
                return base.SegmentationResult(segments=[s for g in segs for s in g], groups=segs, method=self.name,
                                               audio_duration_sec=len(audio) / sample_rate, parameters={})

            def cleanup(self):
                pass

        mod.SpeechSegmenterFactory.create = staticmethod(lambda name, config=None, **kw: Seg())
        params = {"decoder": {"task": "transcribe", "language": "ja", "beam_size": 2, "patience": 1.2, "suppress_tokens": None,
                              "logprob_threshold": sc["threshold"], "no_repeat_ngram_size": 3.0, "temperature": [0.0, 0.2],
                              "fp16": True, "post_model_filter_enabled": sc["post_filter"], "logprob_margin": sc["margin"],
                              "drop_nonverbal_vocals": sc["drop_nonverbal"]},
                  "provider": {"repetition_penalty": 1.5, "hallucination_silence_threshold": None, "word_timestamps": True},
                  "vad": {"threshold": 0.28}, "speech_segmenter": {"backend": "silero-v6.2"}}
        asr = mod.FasterWhisperProASR({"model_name": "large-v3", "device": "cuda", "compute_type": "float16"}, params, "transcribe")
        path = f"/virtual/{sc['name']}.wav"
        AUDIO[path] = (np.sin(np.arange(int(16000 * sc["seconds"])) * 0.05) * 0.25).astype(np.float32)
        CALLS.clear()
        res = asr.transcribe(path)
        out.append({**sc, "params": params, "result": res, "calls": list(CALLS), "filter_stats": asr.get_filter_statistics(),
                    "vad_segments": asr.get_last_vad_segments()})
        print(sc["name"], len(res["segments"]), asr.get_filter_statistics(), [c["n"] for c in CALLS])
    filters = importlib.import_module("whisperjav.modules.segment_filters")
    texts = ["♪♪", "♪ ～", "[音楽]", "(music)", "ああっ", "あっ…", "んんっ！", "ah", "mmm~", "はぁはぁ", "こんにちは", "Thank you",
             "[laughs]", "喘ぎ声", "うん", "えっと", "ahh nice", "ふふふ", "……", "", "   ", "オーッ", "ohhhhhhhh", "{sigh}", "m", "no",
             "<<breathing>>", "あああああああ", "はっ、はっ", "Moaning softly", "うめき", "アッ！", "noon", "ほほう", "ambient noise",
             "今日はいい天気", "ん", "ふぅ〜", "ha ha ha", "a.h.m", "ｱｯ"]
    nonverbal = [[t, bool(filters.SegmentFilterHelper._looks_nonverbal(t))] for t in texts]
    with open(os.path.join(HERE, "reference_asr_adapter.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False)
    with open(os.path.join(HERE, "reference_nonverbal.json"), "w") as f:
        json.dump(nonverbal, f, ensure_ascii=False)
    print("nonverbal", sum(v for _, v in nonverbal), "of", len(nonverbal))


if __name__ == "__main__":
    main()
