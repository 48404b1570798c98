"""Silero-VAD (v5/v6, 16 kHz) parameters: synthetic generation, TorchScript import, HBM blob layout.

Blob order (float32, see ``wj_vad_create`` in include/wjhip.h and the ``V_*`` offsets in
csrc/vad.hip).  Every matrix is stored INPUT-major so that the lanes of a wavefront (= output
channels) read consecutive addresses:

    stft    [256 taps][258 ch]        conv1 [129][3][128] + b[128]     conv2 [128][3][64] + b[64]
    conv3   [64][3][64] + b[64]       conv4 [64][3][128] + b[128]
    w_ih^T  [128][512]  w_hh^T [128][512]  b_ih + b_hh [512]           out_w [128]  out_b [1]
"""
from __future__ import annotations

from typing import Dict

import numpy as np

CHANNELS = ((129, 128), (128, 64), (64, 64), (64, 128))
BLOB_FLOATS = 256 * 258 + sum(i * 3 * o + o for i, o in CHANNELS) + 2 * 128 * 512 + 512 + 128 + 1


def synth_weights(seed: int = 4321) -> Dict[str, np.ndarray]:
    """Seeded parameters with the real model's shapes.  The STFT basis is the true windowed Fourier
    basis (as in the published model); the rest is scaled so probabilities spread over (0, 1) and
    react to signal energy, which is enough to exercise the hysteresis state machine."""
    rng = np.random.default_rng(seed)
    n = np.arange(256)
    window = np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / 256))      # sqrt-hann, as in conv-STFT front ends
    k = np.arange(129)[:, None]
    basis = np.concatenate([np.cos(2 * np.pi * k * n / 256), -np.sin(2 * np.pi * k * n / 256)], 0) * window
    w: Dict[str, np.ndarray] = {"stft.forward_basis_buffer": basis[:, None, :].astype(np.float32)}
    for i, (cin, cout) in enumerate(CHANNELS):
        w[f"encoder.{i}.weight"] = (rng.standard_normal((cout, cin, 3)) * (1.2 / np.sqrt(3 * cin))).astype(np.float32)
        w[f"encoder.{i}.bias"] = (rng.standard_normal(cout) * 0.05).astype(np.float32)
    for name in ("weight_ih", "weight_hh"):
        w[f"rnn.{name}"] = (rng.standard_normal((512, 128)) * (1.0 / np.sqrt(128))).astype(np.float32)
    for name in ("bias_ih", "bias_hh"):
        w[f"rnn.{name}"] = (rng.standard_normal(512) * 0.1).astype(np.float32)
    w["out.weight"] = (rng.standard_normal((1, 128, 1)) * 0.9).astype(np.float32)
    w["out.bias"] = np.array([-0.3], dtype=np.float32)
    return w


def from_jit_state_dict(sd) -> Dict[str, np.ndarray]:
    """Map the TorchScript state dict of ``silero_vad.load_silero_vad()`` (16 kHz branch) onto our
    names.  The key names follow the published v5/v6 checkpoints; unverified offline."""
    def g(key):
        return np.asarray(sd[key].detach().cpu().float().numpy() if hasattr(sd[key], "detach") else sd[key],
                          dtype=np.float32)
    out = {"stft.forward_basis_buffer": g("_model.stft.forward_basis_buffer")}
    for i in range(4):
        out[f"encoder.{i}.weight"] = g(f"_model.encoder.{i}.reparam_conv.weight")
        out[f"encoder.{i}.bias"] = g(f"_model.encoder.{i}.reparam_conv.bias")
    for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        out[f"rnn.{name}"] = g(f"_model.decoder.rnn.{name}")
    out["out.weight"] = g("_model.decoder.decoder.2.weight")
    out["out.bias"] = g("_model.decoder.decoder.2.bias")
    return out


V6_KEYS = ("_model.stft.forward_basis_buffer", "_model.encoder.0.reparam_conv.weight", "_model.decoder.rnn.weight_ih",
           "_model.decoder.decoder.2.weight")


def classify_state_dict(sd) -> str:
    """Which Silero generation a TorchScript state dict belongs to: ``"v5/v6"`` (512-sample windows, conv-STFT + 4
    reparametrised conv blocks + one LSTM cell: the network ``csrc/vad.hip`` implements; ``silero_vad`` >= 5,
    reference: backends/silero_v6.py:143) or ``"v3.1/v4.0"`` (the ``torch.hub`` ``snakers4/silero-vad:v3.1`` /
    ``:v4.0`` archives the reference's default back end loads, backends/silero.py:68-72,199-206: 1536-sample windows,
    a different graph).  Decided from the parameter names and shapes alone; raises on anything else."""
    keys = set(sd.keys())
    if all(k in keys for k in V6_KEYS) or "stft.forward_basis_buffer" in keys:
        return "v5/v6"
    names = " ".join(sorted(keys))
    legacy_marks = ("first_layer", "adaptive_normalization", "_model.encoder.0.dw_conv", "_model_8k", "lstm", "transformer")
    if any(m in names for m in legacy_marks) and any("forward_basis_buffer" in k for k in keys):
        return "v3.1/v4.0"
    raise ValueError(f"not a Silero VAD state dict (neither the v5/v6 nor the v3.1/v4.0 parameter names): {sorted(keys)[:8]} ...")


def from_torchscript(path: str):
    """``(generation, state_dict)`` of a Silero TorchScript archive (``silero_vad.jit``, or the ``model.jit`` that
    ``torch.hub.load("snakers4/silero-vad:v3.1", "silero_vad")`` caches under ``~/.cache/torch/hub``)."""
    import torch
    sd = torch.jit.load(path, map_location="cpu").state_dict()
    return classify_state_dict(sd), sd


def load_file(path: str) -> Dict[str, np.ndarray]:
    """Trained parameters from a file: ``.npz`` / ``.safetensors`` holding either this module's names or the
    TorchScript names, or the TorchScript archive itself (``silero_vad.jit``)."""
    low = path.lower()
    if low.endswith(".npz"):
        sd = dict(np.load(path))
    elif low.endswith(".safetensors"):
        from safetensors.numpy import load_file as _load
        sd = _load(path)
    else:
        gen, sd = from_torchscript(path)
        if gen != "v5/v6":
            from .hipbind import WjError
            raise WjError(f"{path}: a Silero {gen} archive (1536-sample windows); the HIP scorer implements the v5/v6 "
                          "network only and will not run a different network under that name")
    if "stft.forward_basis_buffer" in sd:
        return {k: np.asarray(v, dtype=np.float32) for k, v in sd.items()}
    return from_jit_state_dict(sd)


def resolve(weights, weights_path=None) -> Dict[str, np.ndarray]:
    """The scorer's parameter source (see ``vad.HipSileroScorer``): dict | "synthetic" | None (trained weights from
    ``weights_path`` or the ``silero_vad`` package).  Raises when trained parameters were asked for and cannot be had."""
    if isinstance(weights, dict):
        return weights
    if isinstance(weights, str):
        if weights == "synthetic":
            return synth_weights()
        return load_file(weights)
    if weights is not None:
        raise TypeError("weights must be a dict, 'synthetic', a file path or None")
    if weights_path:
        return load_file(weights_path)
    try:
        import silero_vad  # type: ignore
    except ImportError as e:
        raise FileNotFoundError(
            "HipSileroScorer needs trained Silero VAD parameters: pass weights_path=<silero_vad.jit | .npz | "
            ".safetensors> (vad_weights.load_file) or install the silero-vad package; seeded random parameters are "
            "only used when asked for explicitly (weights='synthetic')") from e
    return from_jit_state_dict(silero_vad.load_silero_vad().state_dict())


def pack(w: Dict[str, np.ndarray]) -> np.ndarray:
    """Flatten to the blob layout consumed by ``wj_vad_create``."""
    parts = [np.ascontiguousarray(w["stft.forward_basis_buffer"][:, 0, :].T)]           # [256][258]
    for i, (cin, cout) in enumerate(CHANNELS):
        wt = w[f"encoder.{i}.weight"]                                                     # [out][in][3]
        assert wt.shape == (cout, cin, 3)
        parts.append(np.ascontiguousarray(wt.transpose(1, 2, 0)))                         # [in][3][out]
        parts.append(w[f"encoder.{i}.bias"])
    parts.append(np.ascontiguousarray(w["rnn.weight_ih"].T))                             # [128][512]
    parts.append(np.ascontiguousarray(w["rnn.weight_hh"].T))
    parts.append(w["rnn.bias_ih"] + w["rnn.bias_hh"])
    parts.append(w["out.weight"].reshape(128))
    parts.append(w["out.bias"].reshape(1))
    blob = np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1) for p in parts])
    assert blob.shape[0] == BLOB_FLOATS, (blob.shape, BLOB_FLOATS)
    return blob
