"""Stand-in TorchScript archives with the SHAPE of the Silero VAD v3.1 / v4.0 JIT models (an INPUT GENERATOR for tests, bench.py and
scripts/ -- like synth.py's audio and weights.synth_weights: nothing in the product path imports it).

The reference's balanced default is ``torch.hub.load("snakers4/silero-vad:v3.1", "silero_vad", onnx=False)``
(/root/reference/whisperjav/modules/speech_segmentation/backends/silero.py:197-206): a scripted module called as
``model(chunk, 16000)`` on 1536-sample windows with ``reset_states()`` between recordings (``:258-273`` through the archive's
``get_speech_timestamps``).  Neither archive can be fetched here, so these modules reproduce the published structure of that
family from its op inventory -- conv-STFT front end (reflect padding, strided conv1d over a fixed Fourier basis, magnitude),
adaptive log-spectrum normalisation (log1p, mean over frequency, reflect padding, smoothing conv, mean over time), an encoder
of depthwise-separable conv blocks with residual projections and strided down-sampling, a 2-layer LSTM whose (h, c) live in
module attributes across calls, and a ReLU -> 1x1 conv -> sigmoid -> mean decoder, plus the 8 kHz / 16 kHz branch on ``sr`` --
with seeded weights, scripted with ``torch.jit.script`` and saved with ``torch.jit.save``: the loader under test
(whisperjav_amd/vad_graph.py) sees a real archive, walks a real inlined graph, and its device probabilities are compared with
THE SAME ARCHIVE executed by torch.jit on the CPU.  ``variant``: "v4" (LSTM, as above) or "v3" (adds a second conv stage, a
Linear layer and tanh/exp-style activations in the decoder to widen the op coverage).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class STFT(nn.Module):
    def __init__(self, filter_length: int = 256, hop_length: int = 64):
        super().__init__()
        self.filter_length = filter_length
        self.hop_length = hop_length
        n = np.arange(filter_length)
        window = 0.5 - 0.5 * np.cos(2 * np.pi * n / filter_length)
        k = np.arange(filter_length // 2 + 1)[:, None]
        basis = np.concatenate([np.cos(2 * np.pi * k * n / filter_length), -np.sin(2 * np.pi * k * n / filter_length)], 0) * window
        self.register_buffer("forward_basis_buffer", torch.from_numpy(basis[:, None, :].astype(np.float32)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.unsqueeze(1)
        pad = int(self.filter_length // 2)
        x = F.pad(x, [pad, pad], mode="reflect")
        f = F.conv1d(x, self.forward_basis_buffer, stride=self.hop_length)
        cutoff = int(self.filter_length // 2 + 1)
        real = f[:, :cutoff, :]
        imag = f[:, cutoff:, :]
        return torch.sqrt(real ** 2 + imag ** 2)


class AdaptiveAudioNormalization(nn.Module):
    def __init__(self, to_pad: int = 3):
        super().__init__()
        self.to_pad = to_pad
        g = np.exp(-0.5 * (np.arange(-to_pad, to_pad + 1) / 1.5) ** 2)
        self.register_buffer("filter_", torch.from_numpy((g / g.sum()).astype(np.float32)).view(1, 1, -1))

    def forward(self, spect: torch.Tensor) -> torch.Tensor:
        spect = torch.log1p(spect * 1048576)
        if len(spect.shape) == 2:
            spect = spect[None, :, :]
        mean = spect.mean(dim=1, keepdim=True)
        mean = F.pad(mean, [self.to_pad, self.to_pad], mode="reflect")
        mean = F.conv1d(mean, self.filter_)
        mean_mean = mean.mean(dim=-1, keepdim=True)
        return spect.add(-mean_mean)


class ConvBlock(nn.Module):
    def __init__(self, cin: int, cout: int, kernel: int = 5, proj: bool = True):
        super().__init__()
        self.dw_conv = nn.Sequential(nn.Conv1d(cin, cin, kernel, padding=kernel // 2, groups=cin), nn.Identity(), nn.ReLU())
        self.pw_conv = nn.Sequential(nn.Conv1d(cin, cout, 1), nn.Identity())
        self.has_proj = proj
        self.proj = nn.Conv1d(cin, cout, 1) if proj else nn.Identity()
        self.activation = nn.ReLU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        residual = x
        x = self.pw_conv(self.dw_conv(x))
        if self.has_proj:
            residual = self.proj(residual)
        x = x + residual
        return self.activation(x)


class VADNet(nn.Module):
    """One sampling rate's network."""

    def __init__(self, filter_length: int, hop_length: int, variant: str):
        super().__init__()
        self.feature_extractor = STFT(filter_length, hop_length)
        self.adaptive_normalization = AdaptiveAudioNormalization()
        bins = filter_length // 2 + 1
        self.first_layer = nn.Sequential(ConvBlock(2 * bins, 16))
        self.encoder = nn.Sequential(nn.Conv1d(16, 16, 1, stride=2), nn.BatchNorm1d(16), nn.ReLU(),
                                     nn.Sequential(ConvBlock(16, 32)),
                                     nn.Conv1d(32, 32, 1, stride=2), nn.BatchNorm1d(32), nn.ReLU(),
                                     nn.Sequential(ConvBlock(32, 32, proj=False)),
                                     nn.Conv1d(32, 32, 1, stride=1), nn.BatchNorm1d(32), nn.ReLU(),
                                     nn.Sequential(ConvBlock(32, 64)))
        self.variant = variant
        self.lstm = nn.LSTM(64, 64, num_layers=2, batch_first=True, dropout=0.1)
        self.mid = nn.Linear(64, 64)
        self.decoder = nn.Sequential(nn.ReLU(), nn.Conv1d(64, 1, 1), nn.Sigmoid())

    def forward(self, x: torch.Tensor, h: torch.Tensor, c: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        x0 = self.feature_extractor(x)
        norm = self.adaptive_normalization(x0)
        x1 = torch.cat([x0, norm], 1)
        x2 = self.first_layer(x1)
        x3 = self.encoder(x2)
        y, (h, c) = self.lstm(x3.permute(0, 2, 1), (h, c))
        if self.variant == "v3":
            y = torch.tanh(self.mid(y)) * torch.exp(-torch.abs(y) * 0.25) + y
        out = self.decoder(y.permute(0, 2, 1))
        return out.mean(dim=2), h, c


class VADStandIn(nn.Module):
    """The archive's top level: ``forward(x, sr)``, ``reset_states()``, state in ``_h`` / ``_c``, 16 kHz and 8 kHz branches."""

    def __init__(self, variant: str = "v4"):
        super().__init__()
        self._model = VADNet(256, 64, variant)
        self._model_8k = VADNet(128, 32, variant)
        self.sample_rates = [8000, 16000]
        self._h = torch.zeros(0)
        self._c = torch.zeros(0)
        self._last_sr = 0
        self._last_batch_size = 0

    @torch.jit.export
    def reset_states(self) -> None:
        self._h = torch.zeros(0)
        self._c = torch.zeros(0)
        self._last_sr = 0
        self._last_batch_size = 0

    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if x.dim() > 2:
            raise ValueError("Too many dimensions for input audio chunk")
        if sr not in self.sample_rates:
            raise ValueError("Supported sampling rates: 8000, 16000")
        if sr / x.shape[1] > 31.25:
            raise ValueError("Input audio chunk is too short")
        batch_size = x.shape[0]
        if self._last_batch_size == 0 or self._last_sr != sr or self._last_batch_size != batch_size:
            self.reset_states()
        if self._h.numel() == 0:
            self._h = torch.zeros(2, batch_size, 64)
            self._c = torch.zeros(2, batch_size, 64)
        if sr == 16000:
            out, h, c = self._model(x, self._h, self._c)
        else:
            out, h, c = self._model_8k(x, self._h, self._c)
        self._h = h
        self._c = c
        self._last_sr = sr
        self._last_batch_size = batch_size
        return out


def _plant_energy_path(net: VADNet, gain: float) -> None:
    """Make channel / unit 0 of every stage carry the window's mean spectral magnitude to the decoder, so the probabilities of
    the seeded network follow the signal energy (speech-like bursts high, near-silence low) while every other channel stays
    random texture: proj rows of the conv blocks pass channel 0 through, the strided 1x1 convs and their BatchNorms are the
    identity on it, LSTM unit 0 of both layers is a leaky threshold on it, the decoder reads it with a large weight."""
    bins = net.feature_extractor.filter_length // 2 + 1
    blocks = [net.first_layer[0], net.encoder[3][0], net.encoder[7][0], net.encoder[11][0]]
    for i, blk in enumerate(blocks):
        blk.pw_conv[0].weight[0].zero_()
        blk.pw_conv[0].bias[0] = 0.0
        if blk.has_proj:
            blk.proj.weight[0].zero_()
            blk.proj.bias[0] = 0.0
            if i == 0:
                blk.proj.weight[0, :bins, 0] = gain / bins
            else:
                blk.proj.weight[0, 0, 0] = 1.0
    for conv_i in (0, 4, 8):
        conv, bn = net.encoder[conv_i], net.encoder[conv_i + 1]
        conv.weight[0].zero_()
        conv.weight[0, 0, 0] = 1.0
        conv.bias[0] = 0.0
        bn.weight[0], bn.bias[0], bn.running_mean[0], bn.running_var[0] = 1.0, 0.0, 0.0, 1.0
    H = 64
    for layer in range(2):
        w_ih, w_hh = getattr(net.lstm, f"weight_ih_l{layer}"), getattr(net.lstm, f"weight_hh_l{layer}")
        b_ih, b_hh = getattr(net.lstm, f"bias_ih_l{layer}"), getattr(net.lstm, f"bias_hh_l{layer}")
        for gate in range(4):                    # rows of unit 0 in the i, f, g, o blocks
            w_ih[gate * H].zero_()
            w_hh[gate * H].zero_()
            b_hh[gate * H] = 0.0
        b_ih[0 * H], b_ih[1 * H], b_ih[3 * H] = 3.0, 0.0, 3.0          # input / output gates open, forget gate at 0.5 (leaky memory)
        w_ih[2 * H, 0] = 2.0 if layer == 0 else 3.0                     # g = tanh(a * x0 - b)
        b_ih[2 * H] = -1.0 if layer == 0 else -0.6
    dec = net.decoder[1]
    dec.weight[0, 0, 0] = 6.0
    dec.bias.fill_(-2.5)
    if net.variant == "v3":
        net.mid.weight[0].zero_()
        net.mid.bias[0] = 0.0


def build(variant: str = "v4", seed: int = 7) -> torch.jit.ScriptModule:
    """A scripted, eval-mode archive with seeded weights; ``_plant_energy_path`` makes its probabilities follow the energy."""
    torch.manual_seed(seed)
    m = VADStandIn(variant)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(max(1, fan_in))))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        for name, b in m.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.2)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        for net in (m._model, m._model_8k):
            net.decoder[1].weight.mul_(2.0)
            _plant_energy_path(net, gain=12.0)
    m.eval()
    return torch.jit.script(m)


def save(path: str, variant: str = "v4", seed: int = 7) -> str:
    torch.jit.save(build(variant, seed), path)
    return path


def reference_probs(archive, audio: np.ndarray, window: int = 1536, sr: int = 16000) -> np.ndarray:
    """The probabilities the archive's own loop produces (utils_vad.get_speech_timestamps' scoring loop: reset_states(), then
    ``model(chunk, sr).item()`` per zero-padded window), executed by torch.jit on the host."""
    model = torch.jit.load(archive, map_location="cpu") if isinstance(archive, str) else archive
    model.reset_states()
    x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
    out: List[float] = []
    with torch.no_grad():
        for s in range(0, len(x), window):
            chunk = x[s: s + window]
            if len(chunk) < window:
                chunk = F.pad(chunk, (0, window - len(chunk)))
            out.append(float(model(chunk, sr).item()))
    return np.asarray(out, dtype=np.float32)


def get_speech_timestamps(audio: torch.Tensor, model, threshold: float = 0.5, sampling_rate: int = 16000, min_speech_duration_ms: int = 250,
                          min_silence_duration_ms: int = 100, window_size_samples: int = 1536, speech_pad_ms: int = 30,
                          return_seconds: bool = False):
    """Stand-in for the hub archive's ``utils_vad.get_speech_timestamps`` at the v3.1 / v4.0 tags (the function the reference
    unpacks from ``torch.hub.load``'s utils and calls at backends/silero.py:258-273): reset the model, score consecutive
    zero-padded windows with ``model(chunk, sampling_rate).item()``, then the trigger / temp_end state machine with
    ``neg_threshold = threshold - 0.15`` and the symmetric padding pass.  Restated from the published source."""
    if not torch.is_tensor(audio):
        audio = torch.Tensor(audio)
    if len(audio.shape) > 1:
        audio = audio.squeeze()
    model.reset_states()
    min_speech_samples = sampling_rate * min_speech_duration_ms / 1000
    min_silence_samples = sampling_rate * min_silence_duration_ms / 1000
    speech_pad_samples = sampling_rate * speech_pad_ms / 1000
    audio_length_samples = len(audio)
    speech_probs = []
    for current_start_sample in range(0, audio_length_samples, window_size_samples):
        chunk = audio[current_start_sample: current_start_sample + window_size_samples]
        if len(chunk) < window_size_samples:
            chunk = F.pad(chunk, (0, int(window_size_samples - len(chunk))))
        speech_probs.append(model(chunk, sampling_rate).item())
    triggered = False
    speeches = []
    current_speech = {}
    neg_threshold = threshold - 0.15
    temp_end = 0
    for i, speech_prob in enumerate(speech_probs):
        if (speech_prob >= threshold) and temp_end:
            temp_end = 0
        if (speech_prob >= threshold) and not triggered:
            triggered = True
            current_speech["start"] = window_size_samples * i
            continue
        if (speech_prob < neg_threshold) and triggered:
            if not temp_end:
                temp_end = window_size_samples * i
            if (window_size_samples * i) - temp_end < min_silence_samples:
                continue
            current_speech["end"] = temp_end
            if (current_speech["end"] - current_speech["start"]) > min_speech_samples:
                speeches.append(current_speech)
            temp_end = 0
            current_speech = {}
            triggered = False
            continue
    if current_speech and (audio_length_samples - current_speech["start"]) > min_speech_samples:
        current_speech["end"] = audio_length_samples
        speeches.append(current_speech)
    for i, speech in enumerate(speeches):
        if i == 0:
            speech["start"] = int(max(0, speech["start"] - speech_pad_samples))
        if i != len(speeches) - 1:
            silence_duration = speeches[i + 1]["start"] - speech["end"]
            if silence_duration < 2 * speech_pad_samples:
                speech["end"] += int(silence_duration // 2)
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - silence_duration // 2))
            else:
                speech["end"] = int(min(audio_length_samples, speech["end"] + speech_pad_samples))
                speeches[i + 1]["start"] = int(max(0, speeches[i + 1]["start"] - speech_pad_samples))
        else:
            speech["end"] = int(min(audio_length_samples, speech["end"] + speech_pad_samples))
    if return_seconds:
        for speech in speeches:
            speech["start"] = round(speech["start"] / sampling_rate, 1)
            speech["end"] = round(speech["end"] / sampling_rate, 1)
    return speeches


def bursty_audio(seconds: float, seed: int = 3, gaps=((2.0, 4.5), (7.0, 8.0), (12.0, 15.5))) -> np.ndarray:
    """Speech-like audio with near-silent stretches (x 0.002), so the stand-in's probabilities cross the thresholds."""
    from whisperjav_amd import synth
    a = synth.speech_like(seconds, seed=seed).copy()
    for s, e in gaps:
        a[int(s * 16000): int(e * 16000)] *= 0.002
    return a
