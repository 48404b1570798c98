"""Generates tests/golden/reference_fidelity_adapter.json by driving the REFERENCE's own ``WhisperProASR``
(/root/reference/whisperjav/modules/whisper_pro_asr.py, imported from source; the fidelity pipeline's ASR module) with
scripted doubles for ``whisper.load_model`` (a model whose ``transcribe`` returns openai-style dicts that depend only on
the clip length), ``soundfile.read`` and the speech segmenter.  Pins the module logic around the model call: parameter
preparation, group slicing, the fallback call, ``_process_segments`` (timestamp shift, suppress phrases, post-model
gate ON by default), statistics, the returned dict.  Companion of make_asr_adapter_fixtures.py.

Run from the repo root inside the build container:  python tests/golden/make_fidelity_adapter_fixtures.py
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
sys.path.append("/root/reference")

from tests.helpers import scripted_segments  # noqa: E402

CALLS = []
AUDIO = {}


class FakeOpenAIWhisper:
    def transcribe(self, audio, **params):
        CALLS.append({"n": int(len(audio)), "params": {k: (list(v) if isinstance(v, tuple) else v) for k, v in params.items()}})
        segs = scripted_segments(len(audio))
        return {"text": "".join(s["text"] for s in segs), "segments": segs, "language": "ja"}


def install():
    wh = types.ModuleType("whisper")
    wh.load_model = lambda name, device=None, **kw: FakeOpenAIWhisper()
    sys.modules["whisper"] = wh
    sf = types.ModuleType("soundfile")
    sf.SoundFileError = type("SoundFileError", (Exception,), {})
    sf.read = lambda path, dtype="float32", **kw: (AUDIO[str(path)].copy(), 16000)
    sys.modules["soundfile"] = sf
    sys.modules["srt"] = types.ModuleType("srt")
    return importlib.import_module("whisperjav.modules.whisper_pro_asr")


SCENARIOS = [
    dict(name="two_groups", seconds=10.0, groups=[[(1.0, 2.0), (2.2, 3.0)], [(5.0, 6.5)]], decoder_extra={}),
    dict(name="gate_default_on", seconds=14.0, groups=[[(0.5, 4.0)], [(6.0, 6.4)], [(8.0, 13.5)]], decoder_extra={}),
    dict(name="gate_off_margin", seconds=20.0, groups=[[(0.0, 5.9)], [(7.0, 12.0), (12.5, 13.0)]],
         decoder_extra={"post_model_filter_enabled": False, "logprob_margin": 0.2}),
    dict(name="strict_nonverbal", seconds=9.0, groups=[[(0.2, 8.8)]],
         decoder_extra={"logprob_threshold": -0.6, "drop_nonverbal_vocals": True, "logprob_margin": 0.1}),
    dict(name="no_speech", seconds=8.0, groups=[], decoder_extra={}),
]


def main():
    mod = install()
    base = importlib.import_module("whisperjav.modules.speech_segmentation.base")
    out = []
    for sc in SCENARIOS:
        groups = sc["groups"]

        class Seg:
            name = "silero-v4.0"
            display_name = "fake"

            def segment(self, audio, sample_rate=16000, **kw):
                segs = [[base.SpeechSegment(start_sec=a, end_sec=b, start_sample=int(a * sample_rate), end_sample=int(b * sample_rate))
                         for a, b in g] for g in groups]
                return base.SegmentationResult(segments=[s for g in segs for s in g], groups=segs, method=self.name,
                                               audio_duration_sec=len(audio) / sample_rate, parameters={})

            def cleanup(self):
                pass

        mod.SpeechSegmenterFactory.create = staticmethod(lambda name, config=None, **kw: Seg())
        params = {"decoder": {"task": "transcribe", "language": "ja", "beam_size": 5, "best_of": 5, "patience": 2.0,
                              "suppress_tokens": "-1", "logprob_threshold": -1.0, "temperature": [0.0, 0.2, 0.4],
                              "condition_on_previous_text": False, "word_timestamps": True, **sc["decoder_extra"]},
                  "provider": {"fp16": True, "hallucination_silence_threshold": None, "carry_initial_prompt": False},
                  "vad": {"threshold": 0.3}, "speech_segmenter": {"backend": "silero-v4.0"}}
        asr = mod.WhisperProASR({"model_name": "large-v2", "device": "cuda"}, params, "transcribe")
        path = f"/virtual/{sc['name']}.wav"
        AUDIO[path] = (np.sin(np.arange(int(16000 * sc["seconds"])) * 0.05) * 0.25).astype(np.float32)
        CALLS.clear()
        res = asr.transcribe(path)
        out.append({**sc, "params": params, "result": res, "calls": list(CALLS), "filter_stats": asr.get_filter_statistics()})
        print(sc["name"], len(res["segments"]), asr.get_filter_statistics(), [c["n"] for c in CALLS])
    with open(os.path.join(HERE, "reference_fidelity_adapter.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False)


if __name__ == "__main__":
    main()
