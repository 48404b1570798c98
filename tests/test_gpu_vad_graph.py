"""The reference's DEFAULT segmenter network on the device: a TorchScript archive of the silero v3.1 / v4.0
structure (tests/silero_standin.py -- the real hub archive is unobtainable offline) is lowered by whisperjav_amd/vad_graph.py
and run by csrc/vadgraph.hip; the device probabilities are compared with THE SAME ARCHIVE executed by torch.jit on the CPU
(bar: 1e-5, as for the v6 scorer), the regions through the archive's own get_speech_timestamps (the reference's call,
/root/reference/whisperjav/modules/speech_segmentation/backends/silero.py:258-273), bit for bit."""
import numpy as np
import pytest
import torch

from tests import silero_standin as S

pytestmark = pytest.mark.gpu


def _streams():
    a = S.bursty_audio(16.0, seed=3, gaps=((2.0, 4.5), (7.0, 8.0), (12.0, 14.5)))
    b = S.bursty_audio(6.3, seed=4, gaps=((1.0, 2.2), (4.0, 5.4)))
    return [a, b[: 16000 * 6 + 777], a[5000: 5000 + 1536 * 3], b[:1000], a[16000 * 6: 16000 * 13 + 5]]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("variant,window", [("v4", 1536), ("v3", 1536), ("v4", 512)])
def test_device_probabilities_equal_the_archive_on_the_cpu(hip, variant, window, fused):
    """Five ragged streams in ONE call (a stream shorter than a window, lengths that are and are not multiples of it), state
    carried per stream and reset between streams, host clips and HBM-resident clips: <= 1e-5 on every window probability.
    ``fused``: one launch per stage with the arena in LDS and the LSTM weights in registers (the default for these graphs), or
    round 5's one-launch-per-instruction executor with the general LSTM kernel (the fall-back for arenas beyond the LDS)."""
    from whisperjav_amd import vad_graph
    archive = S.build(variant, seed=7)
    scorer = vad_graph.HipGraphVadScorer(archive, window=window, fused=fused)
    assert scorer.fused == fused and scorer.lstm_in_registers == fused and scorer.n_stages == 2
    assert (0 < scorer.lds_bytes <= 80 * 1024) if fused else scorer.lds_bytes == 0
    clips = _streams()
    got = scorer.scores(clips)
    worst = 0.0
    for c, g in zip(clips, got):
        ref = S.reference_probs(archive, c, window)
        assert g.shape == ref.shape
        worst = max(worst, float(np.abs(g - ref).max()))
    assert worst < 1e-5, worst
    dev = scorer.scores([torch.from_numpy(c).cuda() for c in clips])
    assert all(np.array_equal(a, b) for a, b in zip(dev, got))
    again = scorer.scores(list(reversed(clips)))              # stream order / neighbours do not matter
    assert all(np.array_equal(a, b) for a, b in zip(reversed(again), got))
    scorer.close()


def test_streams_that_straddle_launch_groups_carry_their_state(hip):
    """max_windows_per_launch smaller than a stream: its windows are split over several launch groups, the LSTM state row
    carries it across them -- identical (bit for bit) to the single-group run."""
    from whisperjav_amd import vad_graph
    archive = S.build("v4", seed=7)
    clips = _streams()
    one = vad_graph.HipGraphVadScorer(archive, window=1536)
    ref = one.scores(clips)
    one.close()
    for cap, fused in ((7, None), (64, None), (7, False)):
        small = vad_graph.HipGraphVadScorer(archive, window=1536, max_windows_per_launch=cap, fused=fused)
        got = small.scores(clips)
        small.close()
        if fused is None:
            assert all(np.array_equal(a, b) for a, b in zip(got, ref)), cap
        else:           # the per-instruction executor sums the LSTM's dot products in another order
            assert max(float(np.abs(a - b).max()) for a, b in zip(got, ref)) < 2e-6


def test_the_default_scorer_is_fused_and_allocates_for_the_call(hip):
    """No mode named: the silero-shaped graphs run fused (70 KB of LDS per window, three launches per call), per-window memory
    is allocated by the first call for the windows it scores (round 5 held 4 GB of arenas from create on)."""
    from whisperjav_amd import vad_graph
    scorer = vad_graph.HipGraphVadScorer(S.build("v4", seed=7))
    assert scorer.fused and scorer.lstm_in_registers and scorer.lds_bytes == 4 * scorer.program.arena_floats
    free0 = torch.cuda.mem_get_info()[0]
    got = scorer.scores(_streams())
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20 and sum(len(g) for g in got) > 20
    scorer.close()


@pytest.mark.parametrize("fused", [True, False])
def test_every_op_of_the_loaders_table_on_the_device(hip, fused):
    """The op zoo, the in-place module and the stateless scorer of tests/test_vad_graph.py (strided / broadcast element-wise
    operands, constant / replicate padding, dilated and grouped convolutions, linear, sums, in-place writes through views)
    through both executors: <= 1e-5 of torch."""
    from tests.test_vad_graph import _InPlaceOnASlice, _OpZoo, _Stateless
    from whisperjav_amd import vad_graph
    rng = np.random.default_rng(2)
    audio = (rng.standard_normal(512 * 6 + 100) * 0.3).astype(np.float32)
    for i, cls in enumerate((_OpZoo, _InPlaceOnASlice, _Stateless)):
        torch.manual_seed(3 + i)
        m = torch.jit.script(cls().eval())
        with torch.no_grad():
            ref = np.array([float(m(torch.nn.functional.pad(torch.from_numpy(audio[s: s + 512].copy()), (0, max(0, 512 - len(audio[s: s + 512])))), 16000))
                            for s in range(0, len(audio), 512)], dtype=np.float32)
        scorer = vad_graph.HipGraphVadScorer(m, window=512, fused=fused)
        assert scorer.fused == fused
        got = scorer.scores([audio, audio[:700]])
        scorer.close()
        assert got[0].shape == ref.shape and float(np.abs(got[0] - ref).max()) < 1e-5, (cls.__name__, got[0], ref)
        assert float(np.abs(got[1][:1] - ref[:1]).max()) < 1e-5


@pytest.mark.parametrize("route", ["hub_pair", "archive_file"])
def test_default_segmenter_scores_on_the_device(hip, tmp_path, route):
    """``HipSileroSpeechSegmenter(version="v3.1", ...)`` with the archive's network ON THE DEVICE: through ``scorer=(model,
    utils)`` -- what torch.hub.load returns; the regions come from the archive's own get_speech_timestamps fed by a replay of the
    device probabilities -- and through ``weights_path=<archive>`` (regions from the restated v3.1 / v4.0 state machine).  Both
    must give the segments of the reference's procedure run on the host: the archive's get_speech_timestamps over the JIT model,
    then the reference's sample padding / overlap fix / grouping; ``segment_many`` scores all scenes in one launch group."""
    from whisperjav_amd import segmenters
    archive = S.build("v4", seed=7)
    utils = (S.get_speech_timestamps, None, None, None, None)
    if route == "hub_pair":
        seg = segmenters.HipSileroSpeechSegmenter(version="v3.1", scorer=(archive, utils), threshold=0.5, min_silence_duration_ms=300)
    else:
        path = S.save(str(tmp_path / "model.jit"), "v4", seed=7)
        seg = segmenters.HipSileroSpeechSegmenter(version="v3.1", weights_path=path, threshold=0.5, min_silence_duration_ms=300)
    host = segmenters.HipSileroSpeechSegmenter(version="v3.1", scorer=(archive, utils), threshold=0.5, min_silence_duration_ms=300,
                                               device_scoring=False)                      # round 4's host scoring: the yardstick
    clips = _streams()[:2] + [_streams()[4]]
    pooled = seg.segment_many(clips, 16000)
    assert seg.name == "silero-v3.1-hip+graph" and "lowered" in seg.display_name and host.name.endswith("+hostnet")
    n_seg = 0
    for c, got in zip(clips, pooled):
        want = host.segment(c, sample_rate=16000)
        key = lambda r: [(s.start_sample, s.end_sample) for s in r.segments]       # noqa: E731
        assert key(got) == key(want) and len(got.groups) == len(want.groups)
        assert key(seg.segment(c, sample_rate=16000)) == key(want)
        assert key(seg.segment(torch.from_numpy(c).cuda(), sample_rate=16000)) == key(want)
        n_seg += len(want.segments)
    assert n_seg >= 3
    seg.cleanup()


def test_unsupported_graphs_are_refused_before_any_audio(hip):
    from whisperjav_amd import segmenters, vad_graph
    import torch.nn as nn

    class Odd(nn.Module):
        def forward(self, x: torch.Tensor, sr: int) -> torch.Tensor:
            return torch.softmax(x.view(1, -1), dim=1)[:, :1]

    m = torch.jit.script(Odd().eval())
    seg = segmenters.HipSileroSpeechSegmenter(version="v3.1", scorer=(m, (S.get_speech_timestamps,)))
    with pytest.raises(vad_graph.LoweringError, match="aten::softmax"):
        seg.segment(np.zeros(16000, dtype=np.float32), sample_rate=16000)
