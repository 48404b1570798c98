"""Host beam-search bookkeeping (whisperjav_amd/search.py) against the oracle's literal CTranslate2
restatement, on the CPU: the product code is driven by a NumPy scorer that evaluates the oracle
model, so any disagreement is in the search logic itself (candidate merge, rule state, n-gram bans,
penalty lists, length penalty).

The plain synthetic weights never emit EOT: ``test_beam_search_matches_oracle`` and
``test_openai_beam_search_matches_oracle`` run every hypothesis to ``max_new_tokens`` and cover the
candidate merge only.  The ``*_with_eot`` tests use ``synth_weights(**weights.SPEECHLIKE)`` on clips of different lengths and ASSERT
(through the oracle's trace) that the termination half ran: hypotheses of different lengths, slots
re-filled from the candidates beam..2*beam, the ``round(beam * patience)`` stop, a ragged batch."""
import numpy as np
import pytest
import torch

from oracle import decoding, whisper_ref
from tests import helpers
from whisperjav_amd import dims as pdims, search


class OracleScorer:
    """StepScorer double: same contract as the HIP scorer, arithmetic by the CPU oracle."""

    def __init__(self, oracle, xa, lay, fcfg):
        self.oracle, self.xa, self.lay, self.fcfg = oracle, xa, lay, fcfg

    def open(self, batch, beam):
        idx = torch.arange(batch).repeat_interleave(beam)
        self.dec = whisper_ref.CachedDecoder(self.oracle, self.xa[idx].contiguous())
        self.rows = batch * beam
        self.logits = None

    def step(self, tokens, parents, want_logits):
        if parents is not None:
            self.dec.reorder(torch.from_numpy(np.asarray(parents, dtype=np.int64)))
        with torch.no_grad():
            self.logits = self.dec.step(torch.from_numpy(np.asarray(tokens, dtype=np.int64))[:, None])

    def no_speech(self):
        return torch.softmax(self.logits.float(), -1)[:, self.lay.no_speech].numpy()

    def score(self, k, row_rules, bans, pens, penalty):
        x = self.logits.clone().float()
        tb, lay, cfg = self.lay.timestamp_begin, self.lay, self.fcfg
        for r in range(self.rows):
            for t in pens[r]:
                if t >= 0:
                    x[r, t] = x[r, t] * penalty if x[r, t] < 0 else x[r, t] / penalty
            for t in bans[r]:
                if t >= 0:
                    x[r, t] = float("-inf")
            first, last_ts, penult_ts, floor = (int(v) for v in row_rules[r])
            if len(cfg.suppress_tokens):
                x[r, list(cfg.suppress_tokens)] = float("-inf")
            if first and cfg.suppress_blank:
                x[r, lay.blank] = x[r, lay.eot] = float("-inf")
            if not cfg.without_timestamps:
                x[r, lay.no_timestamps] = float("-inf")
                if last_ts:
                    if penult_ts:
                        x[r, tb:] = float("-inf")
                    else:
                        x[r, :lay.eot] = float("-inf")
                if floor >= 0:
                    x[r, tb:floor] = float("-inf")
                if first:
                    x[r, :tb] = float("-inf")
                    if cfg.max_initial_timestamp_index is not None:
                        x[r, tb + cfg.max_initial_timestamp_index + 1:] = float("-inf")
                lp = torch.log_softmax(x[r], -1)
                if torch.logsumexp(lp[tb:], -1) > lp[:tb].max():
                    x[r, :tb] = float("-inf")
        lp = torch.log_softmax(x, -1)
        vals, ids = torch.topk(lp, k, dim=-1)
        ids = torch.where(torch.isinf(vals), torch.full_like(ids, -1), ids)
        return ids.numpy().astype(np.int32), vals.numpy().astype(np.float32)


@pytest.fixture(scope="module")
def setup():
    d = helpers.small_dims(n_mels=80, d_model=64, heads=1, layers=1, n_vocab=51865)
    oracle, _ = helpers.make_oracle(d, seed=5)
    mel = torch.from_numpy(helpers.synth_mel(2, d.n_mels, seed=2))
    with torch.no_grad():
        xa = oracle.encode(mel)
    return d, oracle, xa


@pytest.mark.parametrize("beam,patience,lp,rep,ngram,no_ts", [
    (2, 1.2, 1.0, 1.5, 3, False),     # the reference's "balanced" defaults (faster_whisper.py:276-315)
    (5, 1.2, 1.0, 1.5, 3, False),     # BASELINE cfg3
    (3, 1.0, 0.0, 1.0, 0, True),
    (4, 2.0, 1.0, 1.3, 2, False),
])
def test_beam_search_matches_oracle(setup, beam, patience, lp, rep, ngram, no_ts):
    d, oracle, xa = setup
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe] + ([toks.no_timestamps] if no_ts else [])
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, without_timestamps=no_ts, max_initial_timestamp_index=0)
    opts = search.SearchOptions(beam_size=beam, patience=patience, length_penalty=lp, repetition_penalty=rep,
                                no_repeat_ngram_size=ngram, suppress_tokens=suppress, without_timestamps=no_ts,
                                max_initial_timestamp_index=0, max_new_tokens=14, num_hypotheses=beam)
    got = search.beam_search(OracleScorer(oracle, xa, lay, fcfg), [prompt, prompt], opts, eot=lay.eot,
                             timestamp_begin=lay.timestamp_begin)
    bcfg = decoding.BeamConfig(beam, patience, lp, rep, ngram, 14)
    for w in range(2):
        ref, nsp = decoding.beam_search(oracle, xa[w:w + 1], prompt, bcfg, fcfg)
        assert got[w].sequences[0] == ref[0][0], (w, got[w].sequences, [r[0] for r in ref])
        assert abs(got[w].scores[0] - ref[0][1]) < 1e-4
        assert abs(got[w].cum_logprobs[0] - ref[0][2]) < 1e-4
        assert abs(got[w].no_speech_prob - nsp) < 1e-6
        n = min(len(got[w].sequences), len(ref))
        assert [s for s in got[w].sequences[:n]] == [r[0] for r in ref[:n]]


def test_rule_helpers():
    tb = 50365
    assert search.timestamp_state([], tb) == (1, 0, 1, -1)
    assert search.timestamp_state([tb + 5], tb) == (0, 1, 1, tb + 6)
    assert search.timestamp_state([tb + 5, 100], tb) == (0, 0, 1, tb + 6)
    assert search.timestamp_state([tb + 5, 100, tb + 9], tb) == (0, 1, 0, tb + 9)
    assert search.timestamp_state([tb + 5, 100, tb + 9, tb + 9], tb) == (0, 1, 1, tb + 10)
    assert search.ngram_bans([1, 2, 3, 1, 2], 3) == [3]
    assert search.ngram_bans([1, 2, 1, 2, 1], 2) == [2]
    assert search.ngram_bans([7, 7, 7], 1) == [7]
    assert search.ngram_bans([1, 2], 3) == []


@pytest.mark.parametrize("beam,patience,lp", [(2, None, None), (5, 2.0, None), (3, 1.0, 1.0)])
def test_openai_beam_search_matches_oracle(setup, beam, patience, lp):
    """fidelity-mode search (openai-whisper BeamSearchDecoder) on the same scorer contract."""
    d, oracle, xa = setup
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    opts = search.SearchOptions(beam_size=beam, patience=patience or 1.0, length_penalty=-1 if lp is None else lp,
                                suppress_tokens=suppress, max_initial_timestamp_index=50, max_new_tokens=12)
    got = search.beam_search_openai(OracleScorer(oracle, xa, lay, fcfg), [prompt, prompt], opts, eot=lay.eot,
                                    timestamp_begin=lay.timestamp_begin)
    for w in range(2):
        seq, total, avg, nsp = decoding.beam_search_openai(oracle, xa[w:w + 1], prompt, beam, patience, lp, 12, fcfg)
        assert got[w].sequences[0] == seq, (w, got[w].sequences[0], seq)
        assert abs(got[w].cum_logprobs[0] - total) < 1e-4
        assert abs(got[w].avg_logprob(0) - avg) < 1e-4
        assert abs(got[w].no_speech_prob - nsp) < 1e-6


# --------------------------------------------------------------------------------------------------
# searches that END: EOT-bearing synthetic weights (weights.EotRamp), windows finishing at different steps
# --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def setup_eot():
    """``weights.SPEECHLIKE``: hypotheses end after a number of tokens that grows with the audio in the window, so the
    four clips (1 .. 6 s of synthetic speech, zero-padded to 30 s like VAD groups) stop 8 .. 23 steps into the search."""
    from oracle import logmel
    from whisperjav_amd import synth, weights as pweights
    d = helpers.small_dims(n_mels=80, d_model=64, heads=1, layers=1, n_vocab=51865)
    oracle, _ = helpers.make_oracle(d, seed=5, **pweights.SPEECHLIKE)
    mel = torch.from_numpy(np.stack([logmel.window_features(synth.speech_like(s, seed=40 + i), 80, "fw")
                                     for i, s in enumerate((1.0, 2.5, 4.0, 6.0))]))
    with torch.no_grad():
        xa = oracle.encode(mel)
    return d, oracle, xa


@pytest.mark.parametrize("beam,patience,lp,rep,ngram,no_ts,max_new,expect_stop", [
    (2, 1.2, 1.0, 1.5, 3, False, 40, {"patience"}),     # the reference's "balanced" defaults
    (5, 1.2, 1.0, 1.5, 3, False, 40, {"patience"}),     # BASELINE cfg3
    (3, 1.0, 0.0, 1.0, 0, True, 40, {"patience"}),
    (4, 2.0, 1.0, 1.3, 2, False, 40, {"patience"}),
    (5, 1.0, 1.0, 1.0, 0, False, 12, {"length"}),       # some hypotheses end with EOT, the rest at the length limit
    (5, 1.2, 1.0, 1.5, 3, False, 15, {"patience", "length"}),   # ragged: some windows stop on patience, others on length
])
def test_beam_search_with_eot_matches_oracle(setup_eot, beam, patience, lp, rep, ngram, no_ts, max_new, expect_stop):
    d, oracle, xa = setup_eot
    B = xa.shape[0]
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe] + ([toks.no_timestamps] if no_ts else [])
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, without_timestamps=no_ts, max_initial_timestamp_index=0)
    opts = search.SearchOptions(beam_size=beam, patience=patience, length_penalty=lp, repetition_penalty=rep,
                                no_repeat_ngram_size=ngram, suppress_tokens=suppress, without_timestamps=no_ts,
                                max_initial_timestamp_index=0, max_new_tokens=max_new, num_hypotheses=24)
    got = search.beam_search(OracleScorer(oracle, xa, lay, fcfg), [prompt] * B, opts, eot=lay.eot,
                             timestamp_begin=lay.timestamp_begin)       # ONE ragged batch of all windows
    bcfg = decoding.BeamConfig(beam, patience, lp, rep, ngram, max_new)
    stops, steps, refills, lens = set(), set(), 0, set()
    for w in range(B):
        tr = {}
        ref, nsp = decoding.beam_search(oracle, xa[w:w + 1], prompt, bcfg, fcfg, trace=tr)
        stops.add(tr["stop"]); steps.add(tr["steps"]); refills += tr["refills"]
        lens.update(len(t) for t, _, _ in ref)
        assert [list(s) for s in got[w].sequences] == [r[0] for r in ref], (w, got[w].sequences, [r[0] for r in ref])
        assert np.allclose(got[w].scores, [r[1] for r in ref], atol=1e-4)
        assert np.allclose(got[w].cum_logprobs, [r[2] for r in ref], atol=1e-4)
        assert abs(got[w].no_speech_prob - nsp) < 1e-6
        assert all(lay.eot not in t for t, _, _ in ref)
    # the branches this test exists for did run (cannot silently regress to "everything runs to max_new_tokens")
    assert stops == expect_stop, stops
    assert min(lens) < max_new and len(lens) > 1, lens
    assert refills > 0
    if "length" not in expect_stop:
        assert len(steps) > 1, steps                                     # windows of the batch stop at different steps


@pytest.mark.parametrize("beam,patience,lp,max_new", [(2, 1.2, None, 40), (5, 2.0, None, 40), (3, 1.0, 1.0, 40), (5, 1.0, None, 17)])
def test_openai_beam_search_with_eot_matches_oracle(setup_eot, beam, patience, lp, max_new):
    """fidelity-mode search on hypotheses that end: finished dict, ``round(beam * patience)`` stop, the
    ``MaximumLikelihoodRanker`` over sequences of different lengths, unfinished beams topped up at the length limit."""
    d, oracle, xa = setup_eot
    B = xa.shape[0]
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    opts = search.SearchOptions(beam_size=beam, patience=patience, length_penalty=-1 if lp is None else lp,
                                suppress_tokens=suppress, max_initial_timestamp_index=50, max_new_tokens=max_new)
    got = search.beam_search_openai(OracleScorer(oracle, xa, lay, fcfg), [prompt] * B, opts, eot=lay.eot,
                                    timestamp_begin=lay.timestamp_begin)
    lens = set()
    for w in range(B):
        seq, total, avg, nsp = decoding.beam_search_openai(oracle, xa[w:w + 1], prompt, beam, patience, lp, max_new, fcfg)
        lens.add(len(seq))
        assert got[w].sequences[0] == seq, (w, got[w].sequences[0], seq)
        assert abs(got[w].cum_logprobs[0] - total) < 1e-4
        assert abs(got[w].avg_logprob(0) - avg) < 1e-4
        assert abs(got[w].no_speech_prob - nsp) < 1e-6
    if max_new == 40:                      # at the length limit the (longer) unfinished beams out-rank the early finishers
        assert min(lens) < max_new, lens


def test_greedy_oracle_ends_with_eot(setup_eot):
    """The oracle's greedy loop on EOT-bearing weights: rows stop at different steps, the cumulative log-prob includes
    the EOT token (``len(token_logprob) == len(tokens) + 1``), finished rows are padded, not advanced."""
    d, oracle, xa = setup_eot
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    res = decoding.greedy_decode(oracle, xa, prompt, 40, decoding.FilterConfig(max_initial_timestamp_index=0))
    lens = [len(t) for t in res.tokens]
    assert max(lens) < 40 and len(set(lens)) > 1, lens
    for r in range(xa.shape[0]):
        assert len(res.token_logprob[r]) == lens[r] + 1
        assert abs(float(res.sum_logprob[r]) - sum(res.token_logprob[r])) < 1e-4
        one = decoding.greedy_decode(oracle, xa[r:r + 1], prompt, 40, decoding.FilterConfig(max_initial_timestamp_index=0))
        assert one.tokens[0] == res.tokens[r]                             # a ragged batch == per-window decodes


@pytest.mark.parametrize("beam,patience,rep,ngram,max_new", [(5, 1.2, 1.5, 3, 40), (5, 1.2, 1.5, 3, 15), (3, 1.0, 1.0, 0, 40)])
def test_oracle_beam_token_logprobs_equal_teacher_forced_rescoring(setup_eot, beam, patience, rep, ngram, max_new):
    """``trace["token_logprobs"]`` of the oracle's CTranslate2 search (what ``wj_whisper_last_beam_token_logprobs`` is held to
    on the GPU) is bookkeeping over parents, refills and the length stop.  Pinned here without that bookkeeping: every
    hypothesis is fed back through the decoder token by token, the same processors and rules are applied to the logits of each
    position, and the log-softmax entry of the token actually taken (then of EOT when the hypothesis ended on one) must be the
    recorded value; the values sum to the hypothesis's cumulative score."""
    d, oracle, xa = setup_eot
    lay = decoding.TokenLayout.for_vocab(d.n_vocab)
    toks = pdims.special_tokens(d.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    fcfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=0)
    bcfg = decoding.BeamConfig(beam, patience, 1.0, rep, ngram, max_new)
    seen_len_stop = seen_eot = 0
    for w in range(xa.shape[0]):
        tr = {}
        hyps, _ = decoding.beam_search(oracle, xa[w:w + 1], prompt, bcfg, fcfg, trace=tr)
        assert len(tr["token_logprobs"]) == len(hyps)
        for (seq, _, cum), lps in zip(hyps[:3], tr["token_logprobs"][:3]):
            assert len(lps) == len(seq) + 1 and abs(sum(lps) - cum) < 1e-4
            ended_on_eot = len(seq) < max_new
            with torch.no_grad():
                dec = decoding.CachedDecoder(oracle, xa[w:w + 1].contiguous())
                for p in prompt[:-1]:
                    dec.step(torch.tensor([[p]]))
                feed, want = prompt[-1], []
                for i, t in enumerate(list(seq) + ([lay.eot] if ended_on_eot else [])):
                    logits = dec.step(torch.tensor([[feed]]))
                    logits = decoding._apply_ct2_processors(logits, [[prompt[-1]] + list(seq[:i])], bcfg)
                    logits = decoding.filter_logits(logits, [list(prompt) + list(seq[:i])], len(prompt), lay, fcfg)
                    want.append(float(torch.log_softmax(logits, dim=-1)[0, t]))
                    feed = t
            if ended_on_eot:
                seen_eot += 1
            else:
                seen_len_stop += 1
                want.append(0.0)                        # the length limit ends the hypothesis: nothing was scored for it
            assert np.allclose(lps, want, atol=2e-4), (w, seq, lps, want)
    assert seen_eot and (seen_len_stop or max_new == 40)
