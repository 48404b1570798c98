"""stable_shim (cfg1 plumbing, /root/reference/whisperjav/modules/stable_ts_asr.py:296-305,477-509): stable-ts options the shim
does not implement are announced once each instead of being dropped silently; their stable-ts default values and the
options that act here stay quiet."""
import logging
import types

import numpy as np

from whisperjav_amd import stable_shim


class _Engine:
    def __init__(self):
        self.calls = []

    def transcribe(self, audio, **params):
        self.calls.append(params)
        return iter(()), types.SimpleNamespace(language="ja")


def test_unimplemented_stable_ts_options_warn_once(caplog):
    eng = _Engine()
    model = stable_shim.HipStableWhisperModel("tiny", model=eng)
    audio = np.zeros(16000, dtype=np.float32)
    with caplog.at_level(logging.WARNING, logger="whisperjav_amd"):
        model.transcribe(audio, language="ja", denoiser="demucs", only_voice_freq=True, min_word_dur=0.2, regroup=False,
                         suppress_word_ts=True, nonspeech_error=0.1, q_levels=20, k_size=5, vad=False, verbose=None)
        model.transcribe(audio, language="ja", denoiser="demucs", only_voice_freq=True, regroup=False)
    msgs = [r.getMessage() for r in caplog.records if "does not implement" in r.getMessage()]
    assert sorted(m.split("(")[1].split("=")[0] for m in msgs) == ["denoiser", "min_word_dur", "only_voice_freq"]     # once per key, defaults quiet
    assert all(set(c) & {"denoiser", "only_voice_freq", "min_word_dur", "regroup", "vad"} == set() for c in eng.calls)   # consumed, not forwarded
    assert all(c["word_timestamps"] is True for c in eng.calls)
