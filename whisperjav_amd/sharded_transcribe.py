"""Scene-parallel transcription of one long recording on the GPUs of a node (BASELINE cfg4).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m whisperjav_amd.sharded_transcribe recording.wav out.srt --model /models/whisper-large-v3

One process per GPU.  Rank 0 packs the weights once (one RCCL broadcast), every rank loads the clip, detects the scenes
with the same deterministic device kernel (``scenes.HipAuditokSceneDetector``; nothing to exchange), takes its
longest-processing-time-first share, runs VAD -> groups -> batched transcription on its scenes and rank 0 writes the
SRT in scene order.  Mirrors what the reference does serially in ``BalancedPipeline.process`` steps 2-4
(/root/reference/whisperjav/pipelines/balanced_pipeline.py:281-514) + ``SRTStitcher.stitch``
(modules/srt_stitching.py:18-84), minus the files on disk between the steps.
"""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import sharding


def transcribe_scenes(audio: np.ndarray, sr: int, scenes: Sequence[Tuple[float, float]],
                      transcribe_scene: Callable[[np.ndarray, float], List[Dict[str, Any]]]) -> Optional[List[Dict[str, Any]]]:
    """``transcribe_scene(scene_audio, scene_start_s)`` -> segments with ABSOLUTE times; returns on rank 0 the
    concatenation over all scenes sorted by start (None elsewhere)."""
    def work(i: int) -> List[Dict[str, Any]]:
        a, b = scenes[i]
        return transcribe_scene(audio[int(a * sr): int(b * sr)], float(a))
    per_scene = sharding.scene_parallel(scenes, work)
    if per_scene is None:
        return None
    merged = [seg for scene in per_scene for seg in scene]
    merged.sort(key=lambda s: (s["start"], s["end"]))
    return merged


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("audio")
    ap.add_argument("srt")
    ap.add_argument("--model", required=True, help="directory with model.pt or config.json + model.safetensors (+ tokenizer.json)")
    ap.add_argument("--compute-type", default="bfloat16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--beam-size", type=int, default=2)
    ap.add_argument("--language", default="ja")
    ap.add_argument("--max-new-tokens", type=int, default=None)
    ap.add_argument("--scene-energy-db", type=int, default=32, help="auditok pass-1 energy threshold (pass 2 = +6 dB)")
    args = ap.parse_args(argv)
    import torch
    from . import asr, scenes as scn, segmenters, weights as W, whisper_model as wm
    info = sharding.init_distributed()
    torch.cuda.set_device(info.local_rank)
    dev = torch.device("cuda", info.local_rank)
    blob = offsets = meta = None
    if info.rank == 0:
        loader = wm.HipWhisperModel.__new__(wm.HipWhisperModel)
        loader.tokenizer = wm.IdTokenizer()
        dims, sd = loader._load_checkpoint(args.model)
        blob, offsets = W.pack_blob(dims, sd, "bfloat16" if args.compute_type != "float32" else "float32")
        meta = {"dims": dims, "alignment_heads": getattr(loader, "_alignment_heads", None)}
    meta = sharding.broadcast_object(meta)
    dims = meta["dims"]
    dev_blob, offsets = sharding.broadcast_blob(blob, offsets, dev)
    model = wm.HipWhisperModel(args.model, compute_type=args.compute_type, dims=dims, blob=dev_blob, offsets=offsets,
                               device_index=info.local_rank, max_batch=args.batch, max_beam=max(2, args.beam_size))
    if meta["alignment_heads"]:
        model._alignment_heads = list(meta["alignment_heads"])
    tok = Path(args.model) / "tokenizer.json"
    if tok.exists():
        model.tokenizer = wm.HfTokenizer(str(tok))
    audio, sr = asr.read_audio(Path(args.audio))
    detector = scn.HipAuditokSceneDetector(device=info.local_rank, pass1_energy_threshold=args.scene_energy_db,
                                           pass2_energy_threshold=args.scene_energy_db + 6)
    found, _ = detector.split_clip(audio, sr)
    scene_list = [(a, b) for a, b, _, _ in found]
    segmenter = segmenters.HipSileroV6SpeechSegmenter(device=info.local_rank)
    kw = dict(language=args.language, beam_size=args.beam_size, patience=1.2, temperature=[0.0], repetition_penalty=1.5,
              no_repeat_ngram_size=3, condition_on_previous_text=False, max_initial_timestamp=0.0, word_timestamps=True,
              max_new_tokens=args.max_new_tokens)

    def transcribe_scene(clip: np.ndarray, start_s: float) -> List[Dict[str, Any]]:
        res = segmenter.segment(clip, sample_rate=sr)
        spans = [(g[0].start_sample, g[-1].end_sample) for g in res.groups if g]
        clips = [clip[a:b] for a, b in spans if b - a > 400]
        out: List[Dict[str, Any]] = []
        if not clips:
            return out
        segs, _ = model.transcribe_many(clips, **kw)
        for (a, _), group in zip([s for s in spans if s[1] - s[0] > 400], segs):
            for s in group:
                out.append({"start": start_s + a / sr + s.start, "end": start_s + a / sr + s.end, "text": s.text.strip(),
                            "avg_logprob": s.avg_logprob})
        return out
    merged = transcribe_scenes(audio, sr, scene_list, transcribe_scene)
    if merged is not None:
        Path(args.srt).parent.mkdir(parents=True, exist_ok=True)
        Path(args.srt).write_text(asr.compose_srt(merged), encoding="utf-8")
    sharding.barrier()
    model.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
