"""Quality-parity harness: CER and timing IoU between two transcripts of the same recording.

Numeric parity (tests/) pins the arithmetic on seeded weights; once trained weights are at hand the question becomes
"do the subtitles of the MI355X path score like the reference's on real media".  The reference measures that with
``whisperjav/bench`` (CER ``bench/metrics.py:77-111``, matching ``bench/matcher.py:18-96``, timing IoU ``metrics.py:114-160``,
temporal order ``:167-247``).  This module drives exactly those functions when the ``whisperjav`` package is importable
and carries field-for-field equivalents otherwise (``tests/test_quality.py`` pins them against the reference's own code run
from source), so the same report can be produced on the GPU box:

    python -m whisperjav_amd.quality hip_output.srt reference_or_ground_truth.srt [--json]

``compare(gt, test)`` takes lists of ``{"start", "end", "text"}`` dicts (what ``asr`` / ``pipeline`` return) and
reports ``cer``, ``timing_iou``, matched / missed / hallucinated counts, offsets and ordering faults.
"""
from __future__ import annotations

import difflib
import json
import re
import sys
import unicodedata
from typing import Any, Dict, List, Sequence, Tuple

try:  # inside WhisperJAV: the reference's own metric code
    from whisperjav.bench import matcher as _ref_matcher, metrics as _ref_metrics  # type: ignore
except Exception:  # noqa: BLE001
    _ref_matcher = _ref_metrics = None

_SPEAKER = re.compile(r"[（(][^）)]*[）)]")
_DROP = set("。、！？「」『』（）()…・〜～.,!?\"' ")


def normalize_text(text: str) -> str:
    """NFKC, speaker labels in parentheses dropped, whitespace and system-dependent punctuation removed."""
    if _ref_metrics is not None:
        return _ref_metrics.normalize_text(text)
    text = _SPEAKER.sub("", unicodedata.normalize("NFKC", text))
    return "".join(c for c in "".join(text.split()) if c not in _DROP)


def edit_distance(a: str, b: str) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a):
        cur = [i + 1]
        for j, cb in enumerate(b):
            cur.append(min(prev[j + 1] + 1, cur[j] + 1, prev[j] + (ca != cb)))
        prev = cur
    return prev[-1]


def cer(hypothesis: str, reference: str) -> float:
    """Character error rate on the normalised strings; an empty reference scores 0 (nothing said) or 1."""
    if _ref_metrics is not None:
        return _ref_metrics.compute_cer(hypothesis, reference)
    hyp, ref = normalize_text(hypothesis), normalize_text(reference)
    if not ref:
        return 0.0 if not hyp else 1.0
    return edit_distance(hyp, ref) / len(ref)


def iou(s1: float, e1: float, s2: float, e2: float) -> float:
    if _ref_metrics is not None:
        return _ref_metrics.compute_iou(s1, e1, s2, e2)
    inter = max(0.0, min(e1, e2) - max(s1, s2))
    union = max(e1, e2) - min(s1, s2)
    return inter / union if union > 0 else 0.0


def match(gt: Sequence[dict], test: Sequence[dict], min_overlap_sec: float = 0.1, min_text_similarity: float = 0.2):
    """Greedy GT-order matching by temporal overlap, best text similarity wins (``bench/matcher.py:18-96``)."""
    if _ref_matcher is not None:
        return _ref_matcher.match_subtitles(list(gt), list(test), min_overlap_sec, min_text_similarity)
    used, matched, missed = set(), [], []
    for g in gt:
        g_text = normalize_text(g["text"])
        best, best_score, best_idx = None, -1.0, -1
        for i, t in enumerate(test):
            if i in used or min(g["end"], t["end"]) - max(g["start"], t["start"]) < min_overlap_sec:
                continue
            score = difflib.SequenceMatcher(None, g_text, normalize_text(t["text"])).ratio()
            if score > best_score:
                best, best_score, best_idx = t, score, i
        if best is not None and best_score >= min_text_similarity:
            matched.append((g, best))
            used.add(best_idx)
        else:
            missed.append(g)
    return {"matched": matched, "missed": missed, "hallucinated": [t for i, t in enumerate(test) if i not in used]}


def temporal_order(subs: Sequence[dict]) -> Dict[str, Any]:
    """Regressions (a start before the previous start) and overlaps of consecutive subtitles (``metrics.py:167-247``)."""
    if _ref_metrics is not None:
        r = _ref_metrics.analyze_temporal_order(list(subs))
        return {k: r[k] for k in ("is_monotonic", "regression_count", "max_regression_sec", "overlap_count", "total_overlap_sec")}
    reg = ovl = 0
    max_reg = tot = 0.0
    for a, b in zip(subs, subs[1:]):
        if b["start"] < a["start"]:
            reg += 1
            max_reg = max(max_reg, a["start"] - b["start"])
        elif b["start"] < a["end"]:
            ovl += 1
            tot += a["end"] - b["start"]
    return {"is_monotonic": reg == 0, "regression_count": reg, "max_regression_sec": round(max_reg, 3),
            "overlap_count": ovl, "total_overlap_sec": round(tot, 3)}


def compare(gt: Sequence[dict], test: Sequence[dict]) -> Dict[str, Any]:
    """The report of ``whisperjav/bench`` for one (ground truth | reference run, test run) pair."""
    m = match(gt, test)
    pairs: List[Tuple[dict, dict]] = m["matched"]
    n = len(pairs)
    starts = [(t["start"] - g["start"]) * 1000 for g, t in pairs]
    ends = [(t["end"] - g["end"]) * 1000 for g, t in pairs]
    return {
        "cer": cer("".join(t["text"] for t in test), "".join(g["text"] for g in gt)),      # global: texts concatenated in order
        "timing_iou": sum(iou(g["start"], g["end"], t["start"], t["end"]) for g, t in pairs) / n if n else 0.0,
        "matched": n, "missed": len(m["missed"]), "hallucinated": len(m["hallucinated"]),
        "gt_subtitles": len(gt), "test_subtitles": len(test),
        "start_offset_abs_mean_ms": sum(abs(o) for o in starts) / n if n else 0.0,
        "end_offset_abs_mean_ms": sum(abs(o) for o in ends) / n if n else 0.0,
        "temporal_order": temporal_order(list(test)),
        "metric_code": "whisperjav.bench" if _ref_metrics is not None else "whisperjav_amd.quality (mirror)",
    }


_TS = re.compile(r"(\d+):(\d\d):(\d\d)[,.](\d{1,3})\s*-->\s*(\d+):(\d\d):(\d\d)[,.](\d{1,3})")


def parse_srt(text: str) -> List[dict]:
    """Minimal SRT reader: [{index, start, end, text}] (seconds)."""
    out: List[dict] = []
    for block in re.split(r"\n\s*\n", text.replace("\r\n", "\n").strip()):
        lines = [ln for ln in block.split("\n") if ln.strip()]
        for k, ln in enumerate(lines):
            mt = _TS.search(ln)
            if mt:
                g = [int(x) for x in mt.groups()]
                to_s = lambda h, m_, s, ms: h * 3600 + m_ * 60 + s + ms / 1000.0   # noqa: E731
                out.append({"index": len(out) + 1, "start": to_s(*g[:4]), "end": to_s(*g[4:]), "text": "\n".join(lines[k + 1:])})
                break
    return out


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    as_json = "--json" in argv
    paths = [a for a in argv if not a.startswith("--")]
    if len(paths) != 2:
        print(__doc__)
        return 2
    test = parse_srt(open(paths[0], encoding="utf-8").read())
    gt = parse_srt(open(paths[1], encoding="utf-8").read())
    rep = compare(gt, test)
    if as_json:
        print(json.dumps(rep, ensure_ascii=False))
    else:
        for k, v in rep.items():
            print(f"{k:28s} {v}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
