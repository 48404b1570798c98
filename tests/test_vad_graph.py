"""The TorchScript VAD graph loader (whisperjav_amd/vad_graph.py) on the CPU: the LOWERING pinned against torch.jit.

The reference's default segmenter network is a torch.hub TorchScript archive (silero-v3.1, backends/silero.py:197-206) that
cannot be fetched offline; tests/silero_standin.py builds archives of the same structure with torch.jit.script.  Here the
program the loader emits is executed by a NumPy restatement of the device executor (tests/vad_graph_ref.py) and compared with
the archive run by torch.jit -- graph walk, constant folding, strides, the state loop; the HIP kernels themselves are compared
with the same archives in tests/test_gpu_vad_graph.py."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import silero_standin as S, vad_graph_ref as R
from whisperjav_amd import vad_graph as vg


@pytest.mark.parametrize("variant", ["v4", "v3"])
@pytest.mark.parametrize("window,sr", [(1536, 16000), (512, 16000), (1024, 16000), (768, 8000)])
def test_lowered_program_equals_torch_jit(variant, window, sr):
    """Probabilities of consecutive windows (state carried) to float32 rounding: every window size the v3.1 / v4.0 archives take,
    both sampling-rate branches, both stand-in variants (conv blocks + BatchNorm + LSTM; v3 adds Linear / tanh / exp / abs)."""
    m = S.build(variant, seed=7)
    p = vg.lower(m, window, sr)
    audio = S.bursty_audio(6.0)[:: (2 if sr == 8000 else 1)].copy()
    ref = S.reference_probs(m, audio, window, sr)
    got = R.run_stream(p, audio)
    assert ref.shape == got.shape and ref.min() < 0.3 < 0.6 < ref.max()
    assert np.abs(ref - got).max() < 2e-6
    kinds = {t.split()[0].split(".")[0] for t in p.listing}
    assert {"ew", "conv1d", "pad", "mean", "lstm"} <= kinds and (variant == "v4" or "linear" in kinds)
    assert p.state_floats == 2 * 2 * 64 and len(p.state_init) == 2 and all(np.all(v == 0) for _, v in p.state_init)


def test_a_saved_archive_lowers_like_the_live_module(tmp_path):
    path = S.save(str(tmp_path / "silero_vad.jit"), "v4", seed=8)
    a, b = vg.lower(vg.load_archive(path), 1536), vg.lower(S.build("v4", seed=8), 1536)
    assert a.words == b.words and np.array_equal(a.const_blob(), b.const_blob())


class _Stateless(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv1d(1, 4, 9, stride=4, padding=4)
        self.head = nn.Linear(4, 1)

    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        y = torch.nn.functional.leaky_relu(self.conv(x.view(1, 1, -1)), 0.1)
        return torch.sigmoid(self.head(y.mean(dim=2)))


def test_a_stateless_scorer_lowers_without_state():
    torch.manual_seed(1)
    m = torch.jit.script(_Stateless().eval())
    p = vg.lower(m, 512)
    assert p.state_floats == 0
    x = S.bursty_audio(1.0)
    ref = np.array([float(m(torch.from_numpy(x[i: i + 512]), 16000)) for i in range(0, 16000 - 511, 512)], dtype=np.float32)
    assert np.abs(R.run_stream(p, x[: len(ref) * 512]) - ref).max() < 1e-6


class _Softmax(nn.Module):
    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        return torch.softmax(x.view(1, -1), dim=1)[:, :1]


class _DataDependent(nn.Module):
    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        if bool(x.abs().max() > 0.5):
            return x[:1] * 0.0
        return x[:1] * 0.0 + 1.0


class _StateOutsideTheLstm(nn.Module):
    def __init__(self):
        super().__init__()
        self.prev = torch.zeros(1)

    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        out = torch.sigmoid(x.mean().view(1) + self.prev)
        self.prev = out
        return out


class _TooShort(nn.Module):
    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        if x.shape[-1] < 1000:
            raise ValueError("Input audio chunk is too short")
        return x[:1]


@pytest.mark.parametrize("module,match", [(_Softmax, "aten::softmax"), (_DataDependent, "depends on the audio|not lowered"),
                                          (_StateOutsideTheLstm, "state"), (_TooShort, "too short")])
def test_refusals_name_the_op(module, match):
    """Nothing is approximated: an op outside the table, control flow on the audio, state that is not an LSTM's, or a window the
    archive itself rejects all raise LoweringError with the op / the archive's message and the source line."""
    m = torch.jit.script(module().eval())
    with pytest.raises(vg.LoweringError, match=match):
        vg.lower(m, 512)
    with pytest.raises(vg.LoweringError, match="TorchScript"):
        vg.lower(module(), 512)


def test_standin_get_speech_timestamps_equals_the_restated_state_machine():
    """tests/silero_standin.get_speech_timestamps (the hub utils' v3.1 / v4.0 function restated) and
    ``vad.regions_from_probs(..., neg_threshold=threshold - 0.15, max_speech=inf, window=1536)`` -- the route the segmenter takes
    when only the archive file is given -- are the same function of the probabilities."""
    from whisperjav_amd import vad
    from whisperjav_amd.segmenters import _ReplayModel
    rng = np.random.default_rng(5)
    for trial in range(20):
        n_win = int(rng.integers(3, 80))
        probs = np.clip(rng.random(n_win) * 1.4 - 0.2, 0, 1).astype(np.float32)
        if trial % 3 == 0:
            probs = np.repeat(probs[: max(1, n_win // 4)], 4)[:n_win]
        n = (len(probs) - 1) * 1536 + int(rng.integers(1, 1537))
        kw = dict(threshold=float(rng.choice([0.125, 0.25, 0.5])), min_speech_duration_ms=int(rng.choice([90, 150, 250])),
                  min_silence_duration_ms=int(rng.choice([100, 300])), speech_pad_ms=int(rng.choice([30, 700])))
        want = S.get_speech_timestamps(torch.zeros(n), _ReplayModel(probs, 1536), sampling_rate=16000, window_size_samples=1536, **kw)
        got = vad.regions_from_probs(probs, n, sampling_rate=16000, max_speech_duration_s=float("inf"), neg_threshold=kw["threshold"] - 0.15,
                                     window=1536, **kw)
        assert got == want, (trial, kw)
