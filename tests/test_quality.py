"""whisperjav_amd.quality pinned against the reference's own ``whisperjav/bench`` code run from source."""
import importlib.util
import os
import random
import sys
import types

import pytest

from whisperjav_amd import quality

REF = "/root/reference/whisperjav/bench"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _load_ref():
    """metrics.py / matcher.py by path (``matcher`` imports ``whisperjav.bench.metrics``)."""
    saved = {k: v for k, v in sys.modules.items() if k == "whisperjav" or k.startswith("whisperjav.")}
    for name, path in (("whisperjav", "/root/reference/whisperjav"), ("whisperjav.bench", REF)):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    try:
        mods = []
        for name in ("metrics", "matcher"):
            spec = importlib.util.spec_from_file_location(f"whisperjav.bench.{name}", os.path.join(REF, f"{name}.py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[spec.name] = mod
            spec.loader.exec_module(mod)
            mods.append(mod)
        return mods
    finally:
        for k in [k for k in sys.modules if k == "whisperjav" or k.startswith("whisperjav.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def _subs(rng, n, jitter, drop=0.0, swap=0.0):
    pool = "あいうえおかきくけこさしすせそたちつてとなにぬねのはひふへほ。、！？（佐藤）ＡＢＣabc 　\n"
    t, out = 0.0, []
    for _ in range(n):
        t += rng.uniform(0.1, 3.0)
        d = rng.uniform(0.4, 4.0)
        out.append({"start": round(t, 3), "end": round(t + d, 3), "text": "".join(rng.choice(pool) for _ in range(rng.randint(1, 18)))})
        t += d * rng.uniform(0.6, 1.1)
    test = []
    for s in out:
        if rng.random() < drop:
            continue
        txt = "".join(c if rng.random() > swap else rng.choice(pool) for c in s["text"])
        test.append({"start": round(s["start"] + rng.uniform(-jitter, jitter), 3), "end": round(s["end"] + rng.uniform(-jitter, jitter), 3),
                     "text": txt})
    if test and rng.random() < 0.5:
        test.insert(rng.randrange(len(test)), dict(test[-1]))       # an ordering fault
    for i, s in enumerate(test):
        s["index"] = i + 1
    return out, test


@needs_ref
def test_mirror_equals_the_reference_metric_code():
    metrics, matcher = _load_ref()
    assert quality._ref_metrics is None          # the mirror is what runs here (no whisperjav package installed)
    rng = random.Random(11)
    for case in range(25):
        gt, test = _subs(rng, rng.randint(0, 30), jitter=rng.choice([0.0, 0.3, 1.5]), drop=rng.choice([0.0, 0.2]), swap=rng.choice([0.0, 0.1, 0.5]))
        for s in gt + test:
            assert quality.normalize_text(s["text"]) == metrics.normalize_text(s["text"])
        hyp, ref = "".join(s["text"] for s in test), "".join(s["text"] for s in gt)
        assert quality.cer(hyp, ref) == metrics.compute_cer(hyp, ref)
        assert quality.cer(hyp, ref) == metrics.compute_cer_from_segments([s["text"] for s in test], [s["text"] for s in gt])
        want = matcher.match_subtitles(gt, test)
        got = quality.match(gt, test)
        assert got == want
        rep = quality.compare(gt, test)
        assert rep["timing_iou"] == pytest.approx(metrics.compute_timing_score(want["matched"]), abs=1e-12)
        off = metrics.compute_timing_offsets(want["matched"])
        assert rep["start_offset_abs_mean_ms"] == pytest.approx(off["start_offset_abs_mean_ms"], abs=1e-9)
        assert rep["end_offset_abs_mean_ms"] == pytest.approx(off["end_offset_abs_mean_ms"], abs=1e-9)
        order = metrics.analyze_temporal_order(test)
        assert rep["temporal_order"] == {k: order[k] for k in rep["temporal_order"]}
        assert (rep["matched"], rep["missed"], rep["hallucinated"]) == (len(want["matched"]), len(want["missed"]), len(want["hallucinated"]))
    assert quality.edit_distance("kitten", "sitting") == metrics._levenshtein_distance("kitten", "sitting") == 3
    assert quality.iou(0, 2, 1, 3) == metrics.compute_iou(0, 2, 1, 3) == pytest.approx(1 / 3)


def test_srt_round_trip_and_cli(tmp_path, capsys):
    from whisperjav_amd import asr
    segs = [{"start": 1.25, "end": 2.5, "text": "こんにちは"}, {"start": 3.0, "end": 4.004, "text": "二行\nの字幕"}]
    text = asr.compose_srt(segs)
    back = quality.parse_srt(text)
    assert [(s["start"], s["end"], s["text"]) for s in back] == [(1.25, 2.5, "こんにちは"), (3.0, 4.004, "二行\nの字幕")]
    (tmp_path / "a.srt").write_text(text, encoding="utf-8")
    (tmp_path / "b.srt").write_text(asr.compose_srt([dict(segs[0], text="こんばんは"), segs[1]]), encoding="utf-8")
    assert quality.main([str(tmp_path / "a.srt"), str(tmp_path / "b.srt"), "--json"]) == 0
    import json
    rep = json.loads(capsys.readouterr().out)
    assert rep["matched"] == 2 and rep["timing_iou"] == pytest.approx(1.0) and 0 < rep["cer"] < 0.5
    same = quality.compare(segs, segs)
    assert same["cer"] == 0.0 and same["timing_iou"] == 1.0 and same["missed"] == same["hallucinated"] == 0
