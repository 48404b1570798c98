"""Segmenter plugins vs vectors produced by the REFERENCE'S OWN CODE (tests/golden/make_reference_fixtures.py),
plus the reference's fake-scorer test pattern (tests/test_vad_threshold_padding_e2e.py:400-575) replayed on
the HIP-backed classes: ``seg._model`` / ``seg._get_speech_timestamps`` are replaced, no GPU needed."""
import json
import os
from unittest.mock import MagicMock

import numpy as np
import pytest

from whisperjav_amd import asr, segmenters

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_grouping.json")))


def _mock(seg, stamps):
    seg._model = MagicMock()
    seg._get_speech_timestamps = lambda audio, model, **kw: [dict(t) for t in stamps]
    return seg


def test_group_segments_matches_reference():
    for case in FIX["group_cases"]:
        segs = [segmenters.SpeechSegment(start_sec=a, end_sec=b, start_sample=int(a * 16000), end_sample=int(b * 16000))
                for a, b in case["segments"]]
        groups = segmenters.group_segments(segs, case["max_group_duration_s"], case["chunk_threshold_s"])
        assert [[(s.start_sec, s.end_sec) for s in g] for g in groups] == [[tuple(x) for x in g] for g in case["groups"]]
        res = segmenters.SegmentationResult(segments=segs, groups=groups, method="x",
                                            audio_duration_sec=1.0, parameters={})
        assert res.to_legacy_format() == case["legacy"]
        assert res.speech_coverage_sec == pytest.approx(case["coverage"], abs=1e-9)


def test_silero_v31_v40_arithmetic_matches_reference():
    """start/end pad, clamp to len-16, overlap fix, grouping: bit-exact sample indices."""
    for case in FIX["silero_cases"]:
        seg = _mock(segmenters.HipSileroSpeechSegmenter(**case["kw"]), case["stamps"])
        res = seg.segment(np.zeros(case["n_samples"], dtype=np.float32), sample_rate=16000)
        assert [(s.start_sample, s.end_sample, s.start_sec, s.end_sec) for s in res.segments] == \
            [tuple(x) for x in case["segments"]]
        assert [[(s.start_sample, s.end_sample) for s in g] for g in res.groups] == \
            [[tuple(x) for x in g] for g in case["groups"]]
        assert res.to_legacy_format() == case["legacy"]
        assert seg.name == case["name"] + "-hip"


def test_silero_v6_matches_reference():
    for case in FIX["silero_v6_cases"]:
        seg = _mock(segmenters.HipSileroV6SpeechSegmenter(**case["kw"]), case["stamps"])
        res = seg.segment(np.zeros(case["n_samples"], dtype=np.float32), sample_rate=16000)
        assert [(s.start_sample, s.end_sample, s.start_sec, s.end_sec) for s in res.segments] == \
            [tuple(x) for x in case["segments"]]
        assert [[(s.start_sample, s.end_sample) for s in g] for g in res.groups] == \
            [[tuple(x) for x in g] for g in case["groups"]]
        assert seg._get_parameters() == case["params"]


def test_v6_swallows_scorer_errors_like_reference():
    seg = segmenters.HipSileroV6SpeechSegmenter()
    seg._model = MagicMock()

    def boom(*a, **k):
        raise RuntimeError("scorer died")
    seg._get_speech_timestamps = boom
    res = seg.segment(np.zeros(16000, dtype=np.float32))
    assert res.segments == [] and res.groups == [] and res.method == "silero-v6.2-hip"
    legacy = segmenters.HipSileroSpeechSegmenter()
    legacy._model = MagicMock()
    legacy._get_speech_timestamps = boom
    with pytest.raises(RuntimeError):       # the v3.1/v4.0 backend propagates (silero.py has no try/except)
        legacy.segment(np.zeros(16000, dtype=np.float32))


def test_protocol_surface():
    for cls in (segmenters.HipSileroV6SpeechSegmenter, segmenters.HipSileroSpeechSegmenter):
        seg = cls()
        assert isinstance(seg.name, str) and seg.name.startswith("silero") and isinstance(seg.display_name, str)
        assert seg.get_supported_sample_rates() == [16000]
        seg.cleanup()
        seg.cleanup()   # idempotent
    assert set(segmenters.REGISTRY_ENTRIES) >= {"silero-hip", "silero-v6.2-hip", "silero-v4.0-hip", "silero-v3.1-hip"}


def test_failover_and_logprob_gate_match_reference():
    for case in FIX["failover_cases"]:
        assert asr.should_force_full_transcribe(case["groups"], case["duration"]) == case["force"]
    for case in FIX["filter_cases"]:
        helper = asr.SegmentFilterHelper(asr.SegmentFilterConfig(**case["cfg"]))
        drop, reason, thr = helper.should_filter(avg_logprob=case["avg_logprob"], duration=case["duration"], text="こんにちは")
        assert (drop, reason) == (case["drop"], case["reason"])
        assert thr == case["threshold"] or thr == pytest.approx(case["threshold"])
