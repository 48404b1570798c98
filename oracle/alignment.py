"""Oracle: word-level timestamp alignment on the CPU (TEST INFRASTRUCTURE, never shipped).

Restates openai-whisper 20250625 ``whisper/timing.py`` -- ``median_filter``, ``dtw_cpu``, ``find_alignment`` --
which CTranslate2 4.7.1 re-implements natively as ``Whisper.align`` for faster-whisper's
``WhisperModel.find_alignment`` (reference call sites: word_timestamps=True reaches
whisperjav/modules/faster_whisper_pro_asr.py:819 and whisperjav/modules/whisper_pro_asr.py:433 through the
``word_timestamps`` kwarg prepared at faster_whisper_pro_asr.py:340-436).

Pinned in tests/test_oracle_alignment.py against the independent implementations importable here:
``transformers.models.whisper.generation_whisper._median_filter`` / ``_dynamic_time_warping``.
PARITY UNPINNED for CTranslate2's native ``align`` (no binary offline): its published algorithm is this one.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from .whisper_ref import WhisperOracle


def median_filter(x: np.ndarray, width: int) -> np.ndarray:
    """Median of a sliding window of odd ``width`` along the last axis, reflect padding (timing.py median_filter)."""
    assert width > 0 and width % 2 == 1
    pad = width // 2
    if x.shape[-1] <= pad:
        return x
    xp = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")
    win = np.lib.stride_tricks.sliding_window_view(xp, width, axis=-1)
    return np.sort(win, axis=-1)[..., pad]


def dtw(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """timing.py ``dtw_cpu`` + ``backtrace`` on a cost matrix x [N tokens, M frames] (float32 accumulation,
    strict-less tie-breaking: diagonal, then up, else left)."""
    N, M = x.shape
    cost = np.full((N + 1, M + 1), np.inf, dtype=np.float32)
    trace = -np.ones((N + 1, M + 1), dtype=np.int8)
    cost[0, 0] = 0
    xf = x.astype(np.float32)
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0, c1, c2 = cost[i - 1, j - 1], cost[i - 1, j], cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = xf[i - 1, j - 1] + c
            trace[i, j] = t
    i, j = N, M
    trace[0, :] = 2
    trace[:, 0] = 1
    out = []
    while i > 0 or j > 0:
        out.append((i - 1, j - 1))
        t = trace[i, j]
        if t == 0:
            i, j = i - 1, j - 1
        elif t == 1:
            i -= 1
        else:
            j -= 1
    out = np.array(out[::-1], dtype=np.int64).reshape(-1, 2)
    return out[:, 0], out[:, 1]


def alignment_matrix(qk: np.ndarray, num_frames: int, medfilt_width: int = 7) -> np.ndarray:
    """qk [heads, tokens, 1500] scaled scores of the alignment heads -> normalised, filtered, head-averaged
    matrix [tokens, num_frames // 2] (find_alignment lines between the forward pass and the DTW)."""
    w = torch.from_numpy(np.ascontiguousarray(qk[:, :, : num_frames // 2], dtype=np.float32))
    w = torch.softmax(w, dim=-1)
    std, mean = torch.std_mean(w, dim=-2, keepdim=True, unbiased=False)
    w = (w - mean) / std
    w = median_filter(w.numpy(), medfilt_width)
    return w.mean(axis=0)


def find_alignment(model: WhisperOracle, xa: torch.Tensor, sot_sequence: Sequence[int], no_timestamps: int,
                   text_tokens: Sequence[int], eot: int, num_frames: int, heads: Sequence[Tuple[int, int]],
                   medfilt_width: int = 7):
    """One window (xa [1, T, D]).  Returns (text_indices, time_indices, text_token_probs, matrix)."""
    tokens = [*sot_sequence, no_timestamps, *text_tokens, eot]
    qks: List[torch.Tensor] = []
    with torch.no_grad():
        logits = model.decoder_logits(torch.tensor([tokens]), xa, cross_qk=qks)[0]
    n0 = len(sot_sequence)
    sampled = logits[n0:, :eot]
    probs = torch.softmax(sampled.float(), dim=-1)
    text_token_probs = probs[np.arange(len(text_tokens)), list(text_tokens)].numpy() if len(text_tokens) else np.zeros(0, np.float32)
    qk = np.stack([qks[l][0, h].numpy() for l, h in heads])          # [heads, tokens, 1500]
    matrix = alignment_matrix(qk, num_frames, medfilt_width)[n0:-1]    # drop the sot sequence rows and the eot row
    ti, fi = dtw(-matrix)
    return ti, fi, text_token_probs.astype(np.float32), matrix
