"""Scene-parallel transcription of one long recording on the GPUs of a node (BASELINE cfg4).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m whisperjav_amd.sharded_transcribe recording.wav out.srt --model /models/whisper-large-v3

One process per GPU.  Rank 0 packs the weights once (one RCCL broadcast), every rank loads the clip, detects the scenes
with the same deterministic device kernel (``scenes.HipAuditokSceneDetector``; nothing to exchange), takes its
longest-processing-time-first share and transcribes it POOLED (one VAD launch and one batched engine call for all
of the rank's scenes, with the ASR adapter's per-scene semantics: ``asr.HipFasterWhisperProASR.transcribe_scenes``),
and rank 0 writes the SRT in scene order.  Mirrors what the reference does serially in ``BalancedPipeline.process`` steps 2-4
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


def balanced_params(language: str = "ja", beam_size: int = 2, max_new_tokens: Optional[int] = None,
                    word_timestamps: bool = True, vad_threshold: float = 0.28) -> Dict[str, Any]:
    """The ``params`` dict the reference's resolver hands ``FasterWhisperProASR`` in balanced mode
    (config/components/asr/faster_whisper.py:223-315 decoder + provider defaults, config/components/vad/silero.py:105-114)."""
    decoder = dict(task="transcribe", language=language, beam_size=beam_size, best_of=2, patience=1.2, length_penalty=None,
                   suppress_blank=True, without_timestamps=False, max_initial_timestamp=0.0, temperature=[0.0],
                   compression_ratio_threshold=2.4, logprob_threshold=-1.0, no_speech_threshold=0.65,
                   condition_on_previous_text=False, word_timestamps=word_timestamps)
    provider = dict(repetition_penalty=1.5, no_repeat_ngram_size=3, max_new_tokens=max_new_tokens, log_progress=False)
    vad = dict(threshold=vad_threshold, min_speech_duration_ms=100, min_silence_duration_ms=300, speech_pad_ms=400,
               chunk_threshold_s=2.5, max_group_duration_s=6.0)
    return {"decoder": decoder, "provider": provider, "vad": vad, "speech_segmenter": {"backend": "silero-v6.2-hip"}}


def fidelity_params(language: str = "ja", beam_size: int = 2, sample_len: Optional[int] = None,
                    word_timestamps: bool = True, vad_threshold: float = 0.28) -> Dict[str, Any]:
    """The ``params`` dict the reference's resolver hands ``WhisperProASR`` in fidelity mode
    (config/components/asr/openai_whisper.py:229-255, "balanced" sensitivity; BASELINE cfg4)."""
    decoder = dict(task="transcribe", language=language, beam_size=beam_size, best_of=2, patience=1.2, length_penalty=None,
                   prefix=None, suppress_tokens=None, suppress_blank=True, without_timestamps=False, max_initial_timestamp=0.0,
                   temperature=[0.0], compression_ratio_threshold=2.4, logprob_threshold=-1.0, logprob_margin=0.0,
                   no_speech_threshold=0.71, drop_nonverbal_vocals=False, condition_on_previous_text=False, initial_prompt=None,
                   word_timestamps=word_timestamps, verbose=None, fp16=True)
    if sample_len is not None:
        decoder["sample_len"] = sample_len
    vad = dict(threshold=vad_threshold, min_speech_duration_ms=100, min_silence_duration_ms=300, speech_pad_ms=400,
               chunk_threshold_s=2.5, max_group_duration_s=6.0)
    return {"decoder": decoder, "provider": {}, "vad": vad, "speech_segmenter": {"backend": "silero-v6.2-hip"}}


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("audio")
    ap.add_argument("srt")
    ap.add_argument("--model", required=True, help="directory with model.pt or config.json + model.safetensors (+ tokenizer.json)")
    ap.add_argument("--mode", default="balanced", choices=["balanced", "fidelity"],
                    help="balanced: faster-whisper contract (FasterWhisperProASR, CTranslate2 search); fidelity: openai-whisper "
                         "contract (WhisperProASR: its mel padding, its beam search, post-model gate) -- BASELINE cfg4")
    ap.add_argument("--compute-type", default="float16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--beam-size", type=int, default=2)
    ap.add_argument("--language", default="ja")
    ap.add_argument("--max-new-tokens", type=int, default=None)
    ap.add_argument("--scene-energy-db", type=int, default=32, help="auditok pass-1 energy threshold (pass 2 = +6 dB)")
    ap.add_argument("--vad-weights", default=None, help="trained Silero VAD parameters (silero_vad.jit / .npz / .safetensors); "
                    "default: the silero_vad package's bundled model; 'synthetic' = seeded random parameters (tests only)")
    ap.add_argument("--logprob-threshold", type=float, default=None,
                    help="override the preset's logprob_threshold (-1.0): the fallback trigger and, in fidelity mode, the post-model gate")
    ap.add_argument("--per-scene", action="store_true", help="the reference's call pattern (one engine call per scene) instead of pooling")
    args = ap.parse_args(argv)
    import torch
    from . import asr, pipeline, scenes as scn, segmenters, weights as W, whisper_model as wm
    info = sharding.init_distributed()
    torch.cuda.set_device(info.local_rank)
    dev = torch.device("cuda", info.local_rank)
    blob = offsets = meta = None
    ct = {"auto": "float16", "default": "float16"}.get(args.compute_type, args.compute_type)
    if info.rank == 0:
        loader = wm.HipWhisperModel.__new__(wm.HipWhisperModel)
        loader.tokenizer = wm.IdTokenizer()
        dims, sd = loader._load_checkpoint(args.model)
        blob, offsets = W.pack_blob(dims, sd, ct)
        meta = {"dims": dims, "alignment_heads": getattr(loader, "_alignment_heads", None)}
    meta = sharding.broadcast_object(meta)
    dims = meta["dims"]
    dev_blob, offsets = sharding.broadcast_blob(blob, offsets, dev)
    model_cls = wm.HipOpenAIWhisperModel if args.mode == "fidelity" else wm.HipWhisperModel
    model = model_cls(args.model, compute_type=ct, dims=dims, blob=dev_blob, offsets=offsets,
                      device_index=info.local_rank, max_batch=args.batch, max_beam=max(2, args.beam_size))
    if meta["alignment_heads"]:
        model._alignment_heads = list(meta["alignment_heads"])
    tok = Path(args.model) / "tokenizer.json"
    if tok.exists():
        model.tokenizer = wm.HfTokenizer(str(tok))
    audio, sr = asr.read_audio(Path(args.audio))
    audio = pipeline.to_16k(audio, sr)          # every sample index below is a 16 kHz index
    detector = scn.HipAuditokSceneDetector(device=info.local_rank, pass1_energy_threshold=args.scene_energy_db,
                                           pass2_energy_threshold=args.scene_energy_db + 6)
    params = (fidelity_params(args.language, args.beam_size, args.max_new_tokens) if args.mode == "fidelity"
              else balanced_params(args.language, args.beam_size, args.max_new_tokens))
    if args.logprob_threshold is not None:
        params["decoder"]["logprob_threshold"] = args.logprob_threshold
    vad_kw = dict(params["vad"])
    if args.vad_weights == "synthetic":
        vad_kw["weights"] = "synthetic"
    elif args.vad_weights:
        vad_kw["weights_path"] = args.vad_weights
    segmenter = segmenters.HipSileroV6SpeechSegmenter(device=info.local_rank, **vad_kw)
    # the per-scene semantics (VAD fail-over, suppress lists, post-model gate, timestamp shifts) are the ASR adapter's
    asr_cls = asr.HipWhisperProASR if args.mode == "fidelity" else asr.HipFasterWhisperProASR
    module = asr_cls({"model_name": args.model, "device": "cuda", "compute_type": ct}, params, "transcribe",
                     whisper_model=model, segmenter=segmenter)
    runner = pipeline.RecordingTranscriber(module, detector)
    scene_list = runner.detect(audio, pipeline.SR)          # deterministic: every rank computes the same list
    plan = sharding.assign_lpt([b - a for a, b in scene_list], info.world)
    mine = plan[info.rank]
    results = runner.transcribe_scenes(audio, pipeline.SR, [scene_list[i] for i in mine], pooled=not args.per_scene) if mine else []
    gathered = sharding.gather_objects(dict(zip(mine, results)), dst=0)
    if info.rank == 0:
        per_scene = sharding.merge_by_index(plan, gathered)
        merged = runner.stitch(scene_list, per_scene)
        Path(args.srt).parent.mkdir(parents=True, exist_ok=True)
        Path(args.srt).write_text(asr.compose_srt(merged), encoding="utf-8")
    sharding.barrier()
    model.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
