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


class _OpZoo(nn.Module):
    """Ops of the loader's table that the silero-shaped stand-ins do not use: every one lowered and executed (NumPy executor)."""

    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv1d(2, 6, 4, stride=3, padding=2, dilation=2)
        self.c2 = nn.Conv1d(6, 6, 3, groups=3, bias=False)
        self.lin = nn.Linear(6, 3)
        self.register_buffer("scale", torch.tensor([0.5, 2.0, 1.5]).view(1, 3, 1))

    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        a = x.view(1, 2, -1)                                            # two "channels" of 256 samples
        a = torch.nn.functional.pad(a, [3, 1], mode="constant", value=0.25)
        a = torch.nn.functional.pad(a, [2, 2], mode="replicate")
        b = torch.nn.functional.silu(self.c1(a))
        b = torch.nn.functional.hardtanh(self.c2(b), -0.8, 0.9)
        b = b[:, :, 1:-1:2]                                             # strided slice
        c = self.lin(b.transpose(1, 2))                                 # [1, T, 3]
        c = c.permute(0, 2, 1) * self.scale                             # broadcast constant
        c = torch.clamp(c, min=-2.0) + torch.clamp_max(c, 1.0) * 0.5
        d = (1.0 - c.square()).abs().rsqrt().clamp(max=5.0)
        e = c.sum(dim=2, keepdim=True).expand(1, 3, c.shape[2]) / 7.0
        f = torch.cat([d, e, torch.nn.functional.leaky_relu(c, 0.2)], dim=1)        # [1, 9, T]
        g = f.flatten(1).select(0, 0)                                   # [9 T]
        h = (g.reciprocal().neg().exp() - g * 0.01).mean()
        return torch.sigmoid(h.view(1))


def test_every_op_of_the_table_lowers_and_executes():
    torch.manual_seed(3)
    m = torch.jit.script(_OpZoo().eval())
    p = vg.lower(m, 512)
    rng = np.random.default_rng(2)
    audio = (rng.standard_normal(512 * 6) * 0.3).astype(np.float32)
    with torch.no_grad():
        ref = np.array([float(m(torch.from_numpy(audio[i: i + 512]), 16000)) for i in range(0, len(audio), 512)], dtype=np.float32)
    got = R.run_stream(p, audio)
    assert np.abs(got - ref).max() < 5e-6, (got, ref)
    kinds = " ".join(p.listing)
    for op in ("pad.constant", "pad.replicate", "ew.silu", "ew.hardtanh", "linear", "ew.clamp", "ew.pow_scalar", "ew.leaky_relu", "ew.exp", "mean"):
        assert op in kinds, op


def test_the_device_scorer_refuses_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from whisperjav_amd import hipbind
    with pytest.raises(hipbind.WjError, match="no CPU fallback"):
        vg.HipGraphVadScorer(S.build("v4", seed=7))


class _InPlaceOnASlice(nn.Module):
    """``z.add_(1)`` on a slice of y: the base tensor y must see the write (it used to be lowered to a silently wrong
    program); ``relu_`` on the whole tensor afterwards, then a second view of the mutated tensor."""

    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        y = x.unsqueeze(0) * 2
        z = y[:, :10]
        z.add_(1.0)
        w = y[:, 5:40:3]
        w.mul_(w)
        y.relu_()
        return torch.sigmoid(y.mean(1) + y[:, 7])


class _InPlaceThroughAnotherView(nn.Module):
    def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        y = x.unsqueeze(0) * 2
        y[:, 1:].add_(y[:, :-1])            # reads what other threads of the same instruction write
        return torch.sigmoid(y.mean(1))


def test_inplace_ops_write_through_the_operands_own_view():
    m = torch.jit.script(_InPlaceOnASlice().eval())
    p = vg.lower(m, 512)
    rng = np.random.default_rng(4)
    audio = (rng.standard_normal(512 * 3) * 0.3).astype(np.float32)
    with torch.no_grad():
        ref = np.array([float(m(torch.from_numpy(audio[i: i + 512].copy()), 16000)) for i in range(0, len(audio), 512)], dtype=np.float32)
    got = R.run_stream(p, audio)
    assert np.abs(got - ref).max() < 1e-6, (got, ref)
    with pytest.raises(vg.LoweringError, match="different view"):
        vg.lower(torch.jit.script(_InPlaceThroughAnotherView().eval()), 512)


def test_the_arena_is_laid_out_by_liveness_and_fits_the_lds():
    """The silero-shaped graphs: ~70 KB of arena per 1536-sample window, the staging scratch of the small convolutions included
    (one buffer per tensor would take 248 KB + scratch): two workgroups per compute unit; everything an
    LSTM touches -- and the tensors that cross it -- in the small exchange area; the NumPy executor poisons the arena at every
    stage boundary, so a tensor wrongly left in it would turn the probabilities into NaN (test_lowered_program_equals_torch_jit)."""
    for variant in ("v4", "v3"):
        p = vg.lower(S.build(variant, seed=7), 1536)
        assert p.arena_floats * 4 <= 80 * 1024 and p.unpacked_arena_floats > 3.5 * p.arena_floats
        assert 7 * 64 * 2 <= p.xchg_floats <= 2048
        assert p.input_space == vg.SPACE_ARENA and p.output_space == vg.SPACE_ARENA


def test_region_route_certifies_the_archives_function_or_keeps_it():
    """``region_route="certified"`` (the default): the archive's get_speech_timestamps is run against the restated state machine
    on a battery of probability tracks at the call's parameters; equal -> the restated machine serves the scenes (spot checks
    still replay through the archive's function); a function that is NOT that machine (here: a lower threshold 0.10 below
    instead of 0.15) is detected and keeps serving every scene itself."""
    import functools
    from whisperjav_amd import segmenters

    def deviant(audio, model, threshold: float = 0.5, **kw):
        stamps = S.get_speech_timestamps(audio, model, threshold=threshold + 0.05, **kw)      # moves both thresholds
        return stamps

    rng = np.random.default_rng(9)
    probs = np.repeat(rng.random(40).astype(np.float32), 5)
    n = len(probs) * 1536 - 311
    for gst, want_cert in ((S.get_speech_timestamps, True), (deviant, False)):
        seg = segmenters.HipSileroSpeechSegmenter(version="v3.1", scorer=(object(), gst), threshold=0.5, min_silence_duration_ms=300)
        seg._archive_gst, seg._window, seg._graph_scorer = gst, 1536, object()
        got = seg._graph_regions(np.zeros(n, np.float32), probs, 0.5, 100, 300, 400, spot_check=False)
        want = [dict(t) for t in gst(torch.zeros(n), segmenters._ReplayModel(probs, 1536), threshold=0.5, sampling_rate=16000,
                                     min_speech_duration_ms=100, min_silence_duration_ms=300, speech_pad_ms=400, window_size_samples=1536)]
        assert got == want and len(want) >= 2
        assert list(seg._certified.values()) == [want_cert]
        assert seg.region_stats["certified" if want_cert else "replayed"] == 1
        seg._graph_regions(np.zeros(n, np.float32), probs, 0.5, 100, 300, 400, spot_check=True)
        assert seg.region_stats["spot_checks"] == (1 if want_cert else 0) and seg.region_stats["mismatches"] == 0
    always = segmenters.HipSileroSpeechSegmenter(version="v3.1", scorer=(object(), S.get_speech_timestamps), region_route="archive")
    always._archive_gst, always._window, always._graph_scorer = S.get_speech_timestamps, 1536, object()
    always._graph_regions(np.zeros(n, np.float32), probs, 0.5, 100, 300, 400)
    assert always.region_stats["replayed"] == 1 and not always._certified
