"""Host-driven token search over the HIP step scorer: CTranslate2-style beam search.

faster-whisper delegates the search to ``ctranslate2.models.Whisper.generate`` (entered from
/root/reference/whisperjav/modules/faster_whisper_pro_asr.py:819 with the kwargs built at :340-436:
``beam_size, patience, length_penalty, repetition_penalty, no_repeat_ngram_size, suppress_blank,
suppress_tokens, max_initial_timestamp``).  CTranslate2 4.7.1 is not vendored in the reference, so this
module restates its published algorithm (``src/decoding.cc`` BeamSearch + logits processors):

  * every step takes the best ``2 * beam`` continuations of a window over (beam x vocabulary);
  * the first ``beam`` of them become the next beams; one that ends in EOT is recorded as a finished
    hypothesis and its slot is refilled from the remaining candidates;
  * a window stops when ``round(beam * patience)`` hypotheses are finished (or at the length limit),
    hypotheses are ranked by ``cum_logprob / len ** length_penalty``;
  * logits processors, applied before the log-softmax: repetition penalty on the distinct generated
    tokens, no-repeat-n-gram bans, token suppression, SuppressBlank and Whisper's timestamp rules.

The vocabulary-wide work (processors, log-softmax, top-k) runs on the GPU (``wj_decode_topk_rules``);
only ``rows x 2*beam`` (id, log-prob) pairs cross PCIe per step.  The bookkeeping below is pure Python
so it is unit-tested on the CPU against the oracle with a NumPy scorer (tests/test_search_host.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Protocol, Sequence, Tuple

import numpy as np


@dataclass
class SearchOptions:
    beam_size: int = 5
    patience: float = 1.0
    length_penalty: float = 1.0
    repetition_penalty: float = 1.0
    no_repeat_ngram_size: int = 0
    suppress_blank: bool = True
    suppress_tokens: Sequence[int] = field(default_factory=tuple)
    without_timestamps: bool = False
    max_initial_timestamp_index: Optional[int] = 50
    max_new_tokens: int = 224
    num_hypotheses: int = 1


@dataclass
class WindowResult:
    sequences: List[List[int]]     # best first, EOT stripped
    scores: List[float]            # cum_logprob / len ** length_penalty (CTranslate2 ``scores``)
    cum_logprobs: List[float]
    no_speech_prob: float

    def avg_logprob(self, index: int = 0) -> float:
        """faster-whisper: ``cum_logprob / (seq_len + 1)``."""
        return self.cum_logprobs[index] / (len(self.sequences[index]) + 1)


class StepScorer(Protocol):
    """What the search needs from the engine (``HipStepScorer`` below; a NumPy double in the tests)."""

    def open(self, batch: int, beam: int) -> None: ...
    def step(self, tokens: np.ndarray, parents: Optional[np.ndarray], want_logits: bool) -> None: ...
    def no_speech(self) -> np.ndarray: ...
    def score(self, k: int, row_rules: np.ndarray, bans: np.ndarray, pens: np.ndarray,
              penalty: float) -> Tuple[np.ndarray, np.ndarray]: ...


def timestamp_state(generated: Sequence[int], timestamp_begin: int) -> Tuple[int, int, int, int]:
    """(first_step, last_was_timestamp, penultimate_was_timestamp, timestamp_floor) of one hypothesis --
    the state ``ApplyTimestampRules`` derives from the sampled tokens."""
    n = len(generated)
    last_ts = n >= 1 and generated[-1] >= timestamp_begin
    penult_ts = n < 2 or generated[-2] >= timestamp_begin
    floor = -1
    for t in reversed(generated):
        if t >= timestamp_begin:
            floor = t if (last_ts and not penult_ts) else t + 1
            break
    return int(n == 0), int(last_ts), int(penult_ts), floor


def ngram_bans(history: Sequence[int], n: int) -> List[int]:
    """Tokens that would complete an n-gram already present in ``history`` (CTranslate2 NoRepeatNgram)."""
    if n <= 0 or len(history) < n:
        return []
    prefix = list(history[len(history) - (n - 1):]) if n > 1 else []
    out = []
    for i in range(len(history) - n + 1):
        if list(history[i:i + n - 1]) == prefix:
            out.append(history[i + n - 1])
    return sorted(set(out))


LAST_TIMING: dict = {}     # wall-clock split of the most recent beam_search call (device step / device scoring / host)
TIMING_LOG: list = []      # ... of every call since the list was last cleared (bench.py --workload cfg3 reports it)


@dataclass
class _Beam:
    tokens: List[int]
    score: float


def beam_search(scorer: StepScorer, prompts: Sequence[Sequence[int]], opts: SearchOptions, *, eot: int,
                timestamp_begin: int, n_text_ctx: int = 448) -> List[WindowResult]:
    """Decode every window (row of ``prompts``; equal lengths) with CTranslate2's beam search."""
    B = len(prompts)
    K = int(opts.beam_size)
    P = len(prompts[0])
    if any(len(p) != P for p in prompts):
        raise ValueError("all prompts must have the same length")
    max_new = min(int(opts.max_new_tokens), n_text_ctx - P)
    if max_new < 1:
        raise ValueError("prompt leaves no room for new tokens")
    max_candidates = int(np.floor(np.float32(K) * np.float32(opts.patience) + np.float32(0.5)))   # C++ std::round, not Python's half-to-even
    n_cand = 2 * K
    R = B * K
    scorer.open(B, K)
    nsp = np.zeros(B, dtype=np.float32)
    for p in range(P - 1):
        col = np.repeat(np.array([pr[p] for pr in prompts], dtype=np.int32), K)
        scorer.step(col, None, want_logits=(p == 0))
        if p == 0:
            nsp = scorer.no_speech()[::K].copy()
    start_tok = [pr[-1] for pr in prompts]
    beams: List[List[_Beam]] = [[_Beam([], 0.0 if b == 0 else float("-inf")) for b in range(K)] for _ in range(B)]
    finished: List[List[Tuple[float, List[int]]]] = [[] for _ in range(B)]
    done = [False] * B
    feed = np.repeat(np.array(start_tok, dtype=np.int32), K)
    parents: Optional[np.ndarray] = None

    import time as _time
    t_step = t_score = t_host = 0.0
    for step in range(max_new):
        _t0 = _time.perf_counter()
        scorer.step(feed, parents, want_logits=True)
        _t1 = _time.perf_counter()
        t_step += _t1 - _t0
        if P == 1 and step == 0:
            nsp = scorer.no_speech()[::K].copy()
        rules = np.zeros((R, 4), dtype=np.int32)
        ban_lists: List[List[int]] = []
        pen_lists: List[List[int]] = []
        for w in range(B):
            for b in range(K):
                gen = beams[w][b].tokens
                rules[w * K + b] = timestamp_state(gen, timestamp_begin)
                hist = [start_tok[w]] + gen
                ban_lists.append(ngram_bans(hist, int(opts.no_repeat_ngram_size)))
                pen_lists.append(sorted(set(hist)) if opts.repetition_penalty != 1.0 else [])
        maxb = max(1, max(len(x) for x in ban_lists))
        maxp = max(1, max(len(x) for x in pen_lists))
        bans = np.full((R, maxb), -1, dtype=np.int32)
        pens = np.full((R, maxp), -1, dtype=np.int32)
        for r in range(R):
            bans[r, :len(ban_lists[r])] = ban_lists[r]
            pens[r, :len(pen_lists[r])] = pen_lists[r]
        _t2 = _time.perf_counter()
        ids, lps = scorer.score(n_cand, rules, bans, pens, float(opts.repetition_penalty))
        _t3 = _time.perf_counter()
        t_score += _t3 - _t2
        t_host += _t2 - _t1

        last_step = step == max_new - 1
        new_parents = np.arange(R, dtype=np.int32)
        new_feed = np.full(R, eot, dtype=np.int32)
        for w in range(B):
            if done[w]:
                continue
            cands: List[Tuple[float, int, int]] = []   # (score, beam, token)
            for b in range(K):
                base = beams[w][b].score
                if base == float("-inf"):
                    continue
                for j in range(n_cand):
                    tok = int(ids[w * K + b, j])
                    if tok < 0:
                        continue
                    cands.append((base + float(lps[w * K + b, j]), b, tok))
            # best 2K over (beam x vocab); ties -> lower flat index (beam-major, then token id)
            cands.sort(key=lambda c: (-c[0], c[1], c[2]))
            cands = cands[:n_cand]
            nxt: List[_Beam] = []
            secondary = K
            for k in range(min(K, len(cands))):
                score, b, tok = cands[k]
                use = cands[k]
                if tok == eot or last_step:
                    seq = beams[w][b].tokens + ([] if tok == eot else [tok])
                    finished[w].append((score, seq))
                    for j in range(secondary, len(cands)):
                        if cands[j][2] != eot:
                            use = cands[j]
                            secondary = j + 1
                            break
                nxt.append(_Beam(beams[w][use[1]].tokens + [use[2]], use[0]))
                new_parents[w * K + k] = w * K + use[1]
                new_feed[w * K + k] = use[2]
            while len(nxt) < K:   # fewer candidates than beams (heavily masked step): dead beams
                nxt.append(_Beam(list(nxt[0].tokens) if nxt else [], float("-inf")))
            beams[w] = nxt
            if last_step or len(finished[w]) >= max_candidates:
                done[w] = True
        t_host += _time.perf_counter() - _t3
        if all(done):
            break
        feed, parents = new_feed, new_parents
    LAST_TIMING.update(step_s=t_step, score_s=t_score, host_s=t_host, steps=step + 1, rows=R)
    TIMING_LOG.append(dict(LAST_TIMING))
    del TIMING_LOG[:-64]

    results: List[WindowResult] = []
    for w in range(B):
        hyps = finished[w]
        if not hyps:   # cannot happen (the last step registers every beam) but stay defensive
            hyps = [(bm.score, bm.tokens) for bm in beams[w] if bm.score > float("-inf")]
        lp = float(opts.length_penalty)
        ranked = sorted(((s / (max(len(t), 1) ** lp) if lp != 0 else s, s, t) for s, t in hyps),
                        key=lambda x: -x[0])[: max(1, int(opts.num_hypotheses))]
        results.append(WindowResult([t for _, _, t in ranked], [n for n, _, _ in ranked], [s for _, s, _ in ranked],
                                    float(nsp[w])))
    return results


def beam_search_openai(scorer: StepScorer, prompts: Sequence[Sequence[int]], opts: SearchOptions, *, eot: int,
                       timestamp_begin: int, n_text_ctx: int = 448) -> List[WindowResult]:
    """openai-whisper's ``BeamSearchDecoder`` + ``MaximumLikelihoodRanker`` (fidelity mode,
    /root/reference/whisperjav/modules/whisper_pro_asr.py:433 -> ``whisper.decoding.DecodingTask``):
    all beams start as copies (score 0), each step every beam proposes its top ``beam + 1`` tokens, candidates
    are de-duplicated by token sequence, the best ``beam`` unfinished ones continue, EOT-terminated ones are
    collected until ``round(beam * patience)`` are finished; ranking by ``sum_logprob / length`` (or the GNMT
    penalty when ``length_penalty`` is given, pass ``length_penalty=None`` via ``opts.length_penalty = -1``)."""
    B, K, P = len(prompts), int(opts.beam_size), len(prompts[0])
    sample_len = min(int(opts.max_new_tokens), n_text_ctx // 2)
    max_candidates = int(round(K * float(opts.patience)))
    R = B * K
    scorer.open(B, K)
    nsp = np.zeros(B, dtype=np.float32)
    for p in range(P - 1):
        col = np.repeat(np.array([pr[p] for pr in prompts], dtype=np.int32), K)
        scorer.step(col, None, want_logits=(p == 0))
        if p == 0:
            nsp = scorer.no_speech()[::K].copy()
    seqs: List[List[List[int]]] = [[[] for _ in range(K)] for _ in range(B)]
    sums = np.zeros((B, K), dtype=np.float64)
    finished: List[dict] = [dict() for _ in range(B)]
    feed = np.repeat(np.array([pr[-1] for pr in prompts], dtype=np.int32), K)
    parents: Optional[np.ndarray] = None
    empty = np.full((R, 1), -1, dtype=np.int32)
    for step in range(sample_len):
        scorer.step(feed, parents, want_logits=True)
        if P == 1 and step == 0:
            nsp = scorer.no_speech()[::K].copy()
        rules = np.zeros((R, 4), dtype=np.int32)
        for w in range(B):
            for b in range(K):
                rules[w * K + b] = timestamp_state(seqs[w][b], timestamp_begin)
        ids, lps = scorer.score(K + 1, rules, empty, empty, 1.0)
        new_parents = np.arange(R, dtype=np.int32)
        new_feed = np.full(R, eot, dtype=np.int32)
        for w in range(B):
            scores, sources = {}, {}
            for b in range(K):
                prefix = seqs[w][b]
                for j in range(K + 1):
                    tok = int(ids[w * K + b, j])
                    if tok < 0:
                        continue
                    key = tuple(prefix + [tok])
                    scores[key] = float(sums[w, b] + float(lps[w * K + b, j]))
                    sources[key] = b
            nxt, newly = [], {}
            for key in sorted(scores, key=scores.get, reverse=True):
                if key[-1] == eot:
                    newly[key] = scores[key]
                else:
                    nxt.append(key)
                    if len(nxt) == K:
                        break
            while len(nxt) < K:     # degenerate step (everything masked): keep copies alive
                nxt.append(nxt[-1] if nxt else tuple(seqs[w][0]) + (eot,))
            for k, key in enumerate(nxt):
                new_parents[w * K + k] = w * K + sources.get(key, 0)
                new_feed[w * K + k] = key[-1]
                sums[w, k] = scores.get(key, float("-inf"))
            seqs[w] = [list(key) for key in nxt]
            for key in sorted(newly, key=newly.get, reverse=True):
                if len(finished[w]) >= max_candidates:
                    break
                finished[w][key] = newly[key]
        feed, parents = new_feed, new_parents
        if all(len(f) >= max_candidates for f in finished):
            break
    results: List[WindowResult] = []
    lp = opts.length_penalty
    for w in range(B):
        cand = dict(finished[w])
        if len(cand) < K:     # not enough finished sequences: add the best unfinished ones (+EOT)
            for b in np.argsort(sums[w])[::-1]:
                cand[tuple(seqs[w][b] + [eot])] = float(sums[w, b])
                if len(cand) >= K:
                    break
        hyps = []
        for key, total in cand.items():
            toks = list(key)
            toks = toks[: toks.index(eot)] if eot in toks else toks
            length = len(toks)
            penalty = float(length) if (lp is None or lp < 0) else ((5 + length) / 6) ** float(lp)
            hyps.append((total / penalty if penalty else float("-inf"), total, toks))
        best = max(range(len(hyps)), key=lambda i: hyps[i][0])   # first maximum, like np.argmax
        order = [best] + [i for i in sorted(range(len(hyps)), key=lambda i: -hyps[i][0]) if i != best]
        order = order[: max(1, int(opts.num_hypotheses))]
        results.append(WindowResult([hyps[i][2] for i in order], [hyps[i][0] for i in order], [hyps[i][1] for i in order],
                                    float(nsp[w])))
    return results


class HipStepScorer:
    """``StepScorer`` over a resident ``engine.HipWhisper`` (windows already encoded)."""

    def __init__(self, model, opts: SearchOptions):
        import ctypes as C
        from . import engine
        self._C = C
        self.model = model
        self._oc = model._opts(engine.DecodeOptions(
            max_new_tokens=opts.max_new_tokens, suppress_blank=opts.suppress_blank,
            without_timestamps=opts.without_timestamps, suppress_tokens=opts.suppress_tokens,
            max_initial_timestamp=None if opts.max_initial_timestamp_index is None
            else opts.max_initial_timestamp_index * 0.02))
        self._rows = 0

    def open(self, batch: int, beam: int) -> None:
        self.model.open(batch, beam)
        self._rows = batch * beam

    def step(self, tokens, parents, want_logits: bool) -> None:
        self.model.step(tokens, parents, want_logits)

    def no_speech(self) -> np.ndarray:
        C = self._C
        out = np.empty(self._rows, dtype=np.float32)
        from .hipbind import check
        check(self.model._lib.wj_decode_no_speech(self.model.handle, self._rows, self.model.tokens.no_speech,
                                                  out.ctypes.data_as(C.POINTER(C.c_float)), None), "wj_decode_no_speech")
        return out

    def score(self, k, row_rules, bans, pens, penalty):
        C = self._C
        from .hipbind import check
        R = self._rows
        row_rules = np.ascontiguousarray(row_rules, dtype=np.int32)
        bans = np.ascontiguousarray(bans, dtype=np.int32)
        pens = np.ascontiguousarray(pens, dtype=np.int32)
        ids = np.empty((R, k), dtype=np.int32)
        lps = np.empty((R, k), dtype=np.float32)
        pi = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
        check(self.model._lib.wj_decode_topk_rules(self.model.handle, R, int(k), C.byref(self._oc), pi(row_rules),
                                                   pi(bans), bans.shape[1], pi(pens), pens.shape[1], float(penalty),
                                                   pi(ids), lps.ctypes.data_as(C.POINTER(C.c_float)), None),
              "wj_decode_topk_rules")
        return ids, lps
