"""Silero-class VAD on the GPU: batched window scorer (HIP) + the integer post-processing that turns
a probability track into speech regions.

The scorer replaces the per-window TorchScript forward inside ``silero_vad.get_speech_timestamps``
(reference: whisperjav/modules/speech_segmentation/backends/silero_v6.py:205-210); the region logic
mirrors that function's contract (arguments, units, return value ``[{'start': int, 'end': int}]`` in
samples) so the reference's segmenter backends can keep calling "get_speech_timestamps(audio, model,
**kw)" -- see ``get_speech_timestamps`` below, which has exactly that signature.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import hipbind, vad_weights
from .hipbind import check

WINDOW = 512
SR = 16000


class HipSileroScorer:
    """``wj_vad_*``: every stream (scene) is scored concurrently, one workgroup per stream, state reset
    per stream like upstream's ``reset_states()`` per call."""

    def __init__(self, weights: Union[Dict[str, np.ndarray], str, None] = None, device: int = 0,
                 weights_path: Optional[str] = None):
        """``weights``: a parameter dict (``vad_weights`` names), ``"synthetic"`` (seeded random parameters -- an
        explicit opt-in for tests and benchmarks, never a default) or None = the trained Silero parameters:
        ``weights_path`` (.npz / .safetensors / TorchScript .jit|.pt) when given, else the ``silero_vad`` package's
        bundled model (``silero_vad.load_silero_vad()``, the call the reference makes at
        backends/silero_v6.py:143).  There is no silent fallback: without trained parameters this raises."""
        if not torch.cuda.is_available():
            raise hipbind.WjError("no ROCm device visible: the HIP VAD scorer has no CPU fallback")
        self.device = int(device)
        self.dev = torch.device("cuda", device)
        self.ctx = hipbind.context(device)
        self._lib = hipbind.lib()
        blob = vad_weights.pack(vad_weights.resolve(weights, weights_path))
        handle = C.c_void_p()
        check(self._lib.wj_vad_create(self.ctx.handle, blob.ctypes.data_as(C.POINTER(C.c_float)), blob.shape[0],
                                      C.byref(handle)), "wj_vad_create")
        self.handle = handle

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.wj_vad_free(self.handle)
            self.handle = None

    def reset_states(self) -> None:  # upstream API compatibility: state is per call here
        return None

    def scores_device(self, pcm: torch.Tensor, offsets: Sequence[int]) -> List[np.ndarray]:
        n = len(offsets) - 1
        lens = [int(offsets[i + 1]) - int(offsets[i]) for i in range(n)]
        wins = [(ln + WINDOW - 1) // WINDOW for ln in lens]
        poff = np.concatenate([[0], np.cumsum(wins)]).astype(np.int64)
        probs = torch.empty(int(max(1, poff[-1])), dtype=torch.float32, device=self.dev)
        off = (C.c_int64 * (n + 1))(*[int(o) for o in offsets])
        po = (C.c_int64 * (n + 1))(*poff.tolist())
        torch.cuda.current_stream().synchronize()
        check(self._lib.wj_vad_scores(self.handle, C.c_void_p(pcm.data_ptr()), off, po, n, C.c_void_p(probs.data_ptr()),
                                      None), "wj_vad_scores")
        self.ctx.sync()
        host = probs.cpu().numpy()
        return [host[poff[i]:poff[i + 1]].copy() for i in range(n)]

    def scores(self, clips: Sequence[np.ndarray]) -> List[np.ndarray]:
        if len(clips) and all(isinstance(c, torch.Tensor) and c.is_cuda for c in clips):
            # clips already resident in HBM (views of one uploaded recording): gather on the device, no host copy
            offsets = np.concatenate([[0], np.cumsum([int(c.numel()) for c in clips])]).astype(np.int64)
            if offsets[-1] == 0:
                return [np.zeros(0, dtype=np.float32) for _ in clips]
            pcm = torch.cat([c.reshape(-1).to(torch.float32) for c in clips])
            return self.scores_device(pcm, offsets.tolist())
        clips = [c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else c for c in clips]
        arrs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in clips]
        offsets = np.concatenate([[0], np.cumsum([a.shape[0] for a in arrs])]).astype(np.int64)
        flat = np.concatenate(arrs) if arrs else np.zeros(1, dtype=np.float32)
        if flat.shape[0] == 0:
            return [np.zeros(0, dtype=np.float32) for _ in arrs]
        pcm = torch.from_numpy(flat).to(self.dev)
        return self.scores_device(pcm, offsets.tolist())


# --------------------------------------------------------------------------------------------------
# probability track -> speech regions (host, integer sample arithmetic)
# --------------------------------------------------------------------------------------------------
def regions_from_probs(probs: Sequence[float], n_samples: int, *, threshold: float = 0.5,
                       sampling_rate: int = SR, min_speech_duration_ms: int = 250,
                       max_speech_duration_s: float = float("inf"), min_silence_duration_ms: int = 100,
                       speech_pad_ms: int = 30, neg_threshold: Optional[float] = None,
                       min_silence_at_max_speech: int = 98, use_max_poss_sil_at_max_speech: bool = True,
                       window: int = WINDOW) -> List[Dict[str, int]]:
    """Hysteresis segmentation with silero-vad 6.x semantics; returns ``[{'start','end'}]`` in samples."""
    lo = max(threshold - 0.15, 0.01) if neg_threshold is None else neg_threshold
    # upstream compares ``model(chunk, sr).item()`` -- the float32 probability widened to a Python double -- with double thresholds;
    # NumPy 2 would compare a float32 scalar with the threshold ROUNDED to float32 (0.35f < 0.35 is true upstream, false there)
    probs = np.asarray(probs, dtype=np.float32).astype(np.float64).tolist()
    need_speech = sampling_rate * min_speech_duration_ms / 1000.0
    need_silence = sampling_rate * min_silence_duration_ms / 1000.0
    need_silence_at_cap = sampling_rate * min_silence_at_max_speech / 1000.0
    pad = sampling_rate * speech_pad_ms / 1000.0
    cap = sampling_rate * max_speech_duration_s - window - 2 * pad

    out: List[Dict[str, int]] = []
    active = False
    seg_start = 0
    pending_end = 0        # first sample of the silence run that may close the segment
    split_end = 0          # legacy cap handling: last silence long enough to split at
    resume_at = 0
    inner_silences: List[tuple] = []

    def close(end: int, keep: bool = True) -> None:
        nonlocal active, pending_end, split_end, resume_at, inner_silences
        if keep:
            out.append({"start": seg_start, "end": end})
        pending_end = split_end = resume_at = 0
        inner_silences = []

    for idx, p in enumerate(probs):
        t = window * idx
        voiced = p >= threshold
        if voiced and pending_end:
            run = t - pending_end
            if run > need_silence_at_cap:
                inner_silences.append((pending_end, run))
            pending_end = 0
            if resume_at < split_end:
                resume_at = t
        if voiced and not active:
            active, seg_start = True, t
            continue
        if active and t - seg_start > cap:
            if use_max_poss_sil_at_max_speech and inner_silences:
                cut, run = max(inner_silences, key=lambda s: s[1])
                restart = cut + run
                close(cut)
                if restart < cut + t:
                    seg_start = restart
                else:
                    active = False
            elif split_end:
                cut, restart = split_end, resume_at
                close(cut)
                if restart < cut:
                    active = False
                else:
                    seg_start = restart
            else:
                close(t)
                active = False
                continue
        if active and p < lo:
            if not pending_end:
                pending_end = t
            run = t - pending_end
            if not use_max_poss_sil_at_max_speech and run > need_silence_at_cap:
                split_end = pending_end
            if run < need_silence:
                continue
            end = pending_end
            close(end, keep=(end - seg_start) > need_speech)
            active = False
    if active and n_samples - seg_start > need_speech:
        out.append({"start": seg_start, "end": n_samples})

    for i, seg in enumerate(out):
        if i == 0:
            seg["start"] = int(max(0, seg["start"] - pad))
        if i + 1 < len(out):
            nxt = out[i + 1]
            gap = nxt["start"] - seg["end"]
            if gap < 2 * pad:
                seg["end"] += int(gap // 2)
                nxt["start"] = int(max(0, nxt["start"] - gap // 2))
            else:
                seg["end"] = int(min(n_samples, seg["end"] + pad))
                nxt["start"] = int(max(0, nxt["start"] - pad))
        else:
            seg["end"] = int(min(n_samples, seg["end"] + pad))
    return out


def get_speech_timestamps(audio, model: HipSileroScorer, threshold: float = 0.5, sampling_rate: int = SR,
                          min_speech_duration_ms: int = 250, max_speech_duration_s: float = float("inf"),
                          min_silence_duration_ms: int = 100, speech_pad_ms: int = 30, return_seconds: bool = False,
                          neg_threshold: Optional[float] = None, min_silence_at_max_speech: int = 98,
                          use_max_poss_sil_at_max_speech: bool = True, probs: Optional[np.ndarray] = None,
                          **_ignored) -> List[Dict]:
    """Drop-in for ``silero_vad.get_speech_timestamps(audio, model, **kw)`` with the HIP scorer as the
    ``model`` (same keyword names; ``audio`` may be a NumPy array or a torch tensor)."""
    if sampling_rate != SR:
        raise ValueError("the HIP scorer implements the 16 kHz model")
    n_samples = int(audio.numel()) if hasattr(audio, "numel") else int(np.asarray(audio).size)
    if n_samples == 0:
        return []
    if probs is None:       # ``probs``: this clip's window probabilities from a batched scorer call (segment_many)
        if not (hasattr(audio, "is_cuda") and audio.is_cuda):
            if hasattr(audio, "detach"):
                audio = audio.detach().cpu().numpy()
            audio = np.asarray(audio, dtype=np.float32).reshape(-1)
        probs = model.scores([audio])[0]
    segs = regions_from_probs(probs, n_samples, threshold=threshold, sampling_rate=sampling_rate,
                              min_speech_duration_ms=min_speech_duration_ms,
                              max_speech_duration_s=max_speech_duration_s,
                              min_silence_duration_ms=min_silence_duration_ms, speech_pad_ms=speech_pad_ms,
                              neg_threshold=neg_threshold, min_silence_at_max_speech=min_silence_at_max_speech,
                              use_max_poss_sil_at_max_speech=use_max_poss_sil_at_max_speech)
    if return_seconds:
        return [{"start": round(s["start"] / sampling_rate, 1), "end": round(s["end"] / sampling_rate, 1)} for s in segs]
    return segs
