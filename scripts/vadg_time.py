"""Time the reference's DEFAULT segmenter (silero-v3.1 contract over a lowered TorchScript archive) on a synthetic recording:
scene detection -> HipSileroSpeechSegmenter.segment_many over all scenes (device probabilities + regions + padding + grouping).

    python scripts/vadg_time.py [--minutes 120] [--reps 3] [--out gpurun_out/vadg_time.json]

Prints one JSON object: per executor (fused / per-instruction) the scorer's time alone (``scores_ms``: HBM-resident scenes ->
host probabilities) and the whole ``segment_many`` (``segment_many_ms``), and the agreement of the two executors; per region
route (certified / archive) the host time of the regions.  The archive is tests/silero_standin.py's (the real hub archive is
not obtainable offline): an input generator, not a checker.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=120.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--unfused", action="store_true", help="also time round 5's per-instruction executor")
    args = ap.parse_args()
    from whisperjav_amd import pipeline, scenes, segmenters, synth, vad_graph
    from whisperjav_amd.standin_vad import build, get_speech_timestamps
    audio = synth.speech_like_long(60.0 * args.minutes, seed=1234, noisy=True)
    det = scenes.HipAuditokSceneDetector(pass1_energy_threshold=52, pass2_energy_threshold=56)

    class _Asr:       # RecordingTranscriber only needs an object for the scene helpers used here
        pass

    runner = pipeline.RecordingTranscriber(_Asr(), det)
    scn = runner.detect(audio, 16000)
    clips = [c for c, _ in runner._device_clips(audio, 16000, scn)]          # HBM-resident scene clips, as the bench's step hands them over
    archive = build("v4", seed=7)
    out = {"minutes": args.minutes, "scenes": len(clips), "windows": int(sum((int(c.numel()) + 1535) // 1536 for c in clips))}
    probs = {}
    for name, fused in (("fused", None),) + ((("per_instruction", False),) if args.unfused else ()):
        sc = vad_graph.HipGraphVadScorer(archive, fused=fused)
        sc.scores(clips[:4])
        best = 1e9
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            p = sc.scores(clips)
            best = min(best, time.perf_counter() - t)
        probs[name] = np.concatenate(p)
        out[name] = {"scores_ms": round(1e3 * best, 2), "fused": sc.fused, "lds_bytes": sc.lds_bytes, "lstm_in_registers": sc.lstm_in_registers,
                     "arena_floats": sc.program.arena_floats, "xchg_floats": sc.program.xchg_floats, "instructions": sc.program.n_instr}
        sc.close()
    if args.unfused:
        out["max_abs_diff_between_executors"] = float(np.abs(probs["fused"] - probs["per_instruction"]).max())
    vad31 = dict(threshold=0.5, min_speech_duration_ms=100, min_silence_duration_ms=300, speech_pad_ms=400, chunk_threshold_s=2.5, max_group_duration_s=6.0)
    utils = (get_speech_timestamps, None, None, None, None)
    for route in ("certified", "archive"):
        seg = segmenters.HipSileroSpeechSegmenter(version="v3.1", scorer=(archive, utils), region_route=route, **vad31)
        res = seg.segment_many(clips, 16000)
        best = 1e9
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            res = seg.segment_many(clips, 16000)
            best = min(best, time.perf_counter() - t)
        out[f"segment_many_{route}"] = {"segment_many_ms": round(1e3 * best, 2), "segments": int(sum(len(r.segments) for r in res)),
                                        "groups": int(sum(len(r.groups) for r in res)), "region_stats": dict(seg.region_stats)}
        seg.cleanup()
    line = json.dumps(out)
    print(line)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
