"""Oracle: Silero-VAD (v5/v6 architecture) window scorer + ``get_speech_timestamps`` state machine
(TEST INFRASTRUCTURE, never shipped).

Restates silero-vad 6.2.1 (pip; pin at /root/reference/uv.lock:3888-3890), which WhisperJAV calls at
whisperjav/modules/speech_segmentation/backends/silero_v6.py:205-210 (and, for the torch.hub tags
v3.1/v4.0, at backends/silero.py:269-273).  The wheel and its TorchScript weights are not available
offline, so

  * the network is restated from the published architecture (16 kHz branch): 64-sample context +
    512-sample chunk -> right reflect-pad 64 -> Conv1d STFT basis [258,1,256] stride 128 ->
    magnitude of the first/second 129 channels -> 4 x (Conv1d k=3 pad=1 + ReLU), strides 1,2,2,1,
    channels 129->128->64->64->128 -> LSTMCell(128) -> ReLU -> Conv1d(128,1,1) -> sigmoid;
    state and context reset per stream (``reset_states``), the last chunk is zero-padded;
  * ``speech_timestamps`` restates ``utils_vad.get_speech_timestamps`` (threshold / neg_threshold
    hysteresis, min speech / min silence, max-speech splitting at the longest inner silence,
    speech_pad with midpoint sharing).

PARITY UNPINNED for the weights (seeded random here) and for the max-speech branch: no upstream
vectors exist offline.  The simple hysteresis path is additionally cross-checked against the
reference's own pure-Python Silero-compatible port (backends/whisperseg.py:419-571) in
tests/test_vad_postprocess.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

WINDOW = 512
CONTEXT = 64


class SileroOracle:
    def __init__(self, w: Dict[str, np.ndarray]):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in w.items()}

    def window_prob(self, x576: torch.Tensor, h: torch.Tensor, c: torch.Tensor):
        w = self.w
        x = F.pad(x576[None, None, :], (0, 64), mode="reflect")          # [1,1,640]
        ft = F.conv1d(x, w["stft.forward_basis_buffer"], stride=128)     # [1,258,4]
        mag = torch.sqrt(ft[:, :129] ** 2 + ft[:, 129:] ** 2)
        y = mag
        for i, stride in enumerate((1, 2, 2, 1)):
            y = F.relu(F.conv1d(y, w[f"encoder.{i}.weight"], w[f"encoder.{i}.bias"], stride=stride, padding=1))
        feat = y[:, :, 0]                                                # [1,128]
        gates = feat @ w["rnn.weight_ih"].T + w["rnn.bias_ih"] + h @ w["rnn.weight_hh"].T + w["rnn.bias_hh"]
        i_g, f_g, g_g, o_g = gates.chunk(4, dim=1)
        c = torch.sigmoid(f_g) * c + torch.sigmoid(i_g) * torch.tanh(g_g)
        h = torch.sigmoid(o_g) * torch.tanh(c)
        out = torch.sigmoid(F.relu(h) @ w["out.weight"].reshape(1, 128).T + w["out.bias"])
        return float(out[0, 0]), h, c

    def probs(self, audio: np.ndarray) -> np.ndarray:
        """Per-window speech probabilities of one stream (fresh state, like reset_states())."""
        audio = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
        n = audio.shape[0]
        h = torch.zeros(1, 128)
        c = torch.zeros(1, 128)
        ctx = torch.zeros(CONTEXT)
        out = []
        with torch.no_grad():
            for start in range(0, n, WINDOW):
                chunk = audio[start:start + WINDOW]
                if chunk.shape[0] < WINDOW:
                    chunk = F.pad(chunk, (0, WINDOW - chunk.shape[0]))
                x = torch.cat([ctx, chunk])
                p, h, c = self.window_prob(x, h, c)
                out.append(p)
                ctx = x[-CONTEXT:]
        return np.array(out, dtype=np.float32)


def speech_timestamps(probs, audio_length_samples: int, threshold: float = 0.5, sampling_rate: int = 16000,
                      min_speech_duration_ms: int = 250, max_speech_duration_s: float = float("inf"),
                      min_silence_duration_ms: int = 100, speech_pad_ms: int = 30,
                      neg_threshold: Optional[float] = None, min_silence_at_max_speech: int = 98,
                      use_max_poss_sil_at_max_speech: bool = True, window_size_samples: int = WINDOW) -> List[Dict[str, int]]:
    """``get_speech_timestamps`` (silero-vad 6.x) on a precomputed probability track; sample units."""
    min_speech = sampling_rate * min_speech_duration_ms / 1000
    pad = sampling_rate * speech_pad_ms / 1000
    max_speech = sampling_rate * max_speech_duration_s - window_size_samples - 2 * pad
    min_silence = sampling_rate * min_silence_duration_ms / 1000
    min_silence_max = sampling_rate * min_silence_at_max_speech / 1000
    if neg_threshold is None:
        neg_threshold = max(threshold - 0.15, 0.01)

    triggered = False
    speeches: List[Dict[str, int]] = []
    cur: Dict[str, int] = {}
    temp_end = 0
    prev_end = next_start = 0
    possible_ends = []

    for i, p in enumerate(probs):
        now = window_size_samples * i
        if p >= threshold and temp_end:
            sil = now - temp_end
            if sil > min_silence_max:
                possible_ends.append((temp_end, sil))
            temp_end = 0
            if next_start < prev_end:
                next_start = now
        if p >= threshold and not triggered:
            triggered = True
            cur["start"] = now
            continue
        if triggered and now - cur["start"] > max_speech:
            if use_max_poss_sil_at_max_speech and possible_ends:
                prev_end, dur = max(possible_ends, key=lambda t: t[1])
                cur["end"] = prev_end
                speeches.append(cur)
                cur = {}
                next_start = prev_end + dur
                if next_start < prev_end + now:
                    cur["start"] = next_start
                else:
                    triggered = False
                prev_end = next_start = temp_end = 0
                possible_ends = []
            else:
                if prev_end:
                    cur["end"] = prev_end
                    speeches.append(cur)
                    cur = {}
                    if next_start < prev_end:
                        triggered = False
                    else:
                        cur["start"] = next_start
                    prev_end = next_start = temp_end = 0
                    possible_ends = []
                else:
                    cur["end"] = now
                    speeches.append(cur)
                    cur = {}
                    prev_end = next_start = temp_end = 0
                    triggered = False
                    possible_ends = []
                    continue
        if p < neg_threshold and triggered:
            if not temp_end:
                temp_end = now
            sil_now = now - temp_end
            if not use_max_poss_sil_at_max_speech and sil_now > min_silence_max:
                prev_end = temp_end
            if sil_now < min_silence:
                continue
            cur["end"] = temp_end
            if cur["end"] - cur["start"] > min_speech:
                speeches.append(cur)
            cur = {}
            prev_end = next_start = temp_end = 0
            triggered = False
            possible_ends = []
            continue

    if cur and audio_length_samples - cur["start"] > min_speech:
        cur["end"] = audio_length_samples
        speeches.append(cur)

    for i, sp in enumerate(speeches):
        if i == 0:
            sp["start"] = int(max(0, sp["start"] - pad))
        if i != len(speeches) - 1:
            gap = speeches[i + 1]["start"] - sp["end"]
            if gap < 2 * pad:
                sp["end"] += int(gap // 2)
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - gap // 2))
            else:
                sp["end"] = int(min(audio_length_samples, sp["end"] + pad))
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - pad))
        else:
            sp["end"] = int(min(audio_length_samples, sp["end"] + pad))
    return speeches
