"""Oracle: Whisper token search on the CPU (TEST INFRASTRUCTURE, never shipped).

Restates, over the fp32 ``whisper_ref`` model:

  * the logit filters of openai-whisper 20250625 ``whisper/decoding.py`` -- ``SuppressBlank``,
    ``SuppressTokens``, ``ApplyTimestampRules`` -- which CTranslate2 4.7.1 re-implements for
    faster-whisper's ``WhisperModel.generate`` (reference call sites:
    whisperjav/modules/faster_whisper_pro_asr.py:819, whisperjav/modules/whisper_pro_asr.py:433);
  * greedy decoding (``GreedyDecoder`` at temperature 0 == CTranslate2 ``beam_size=1``):
    log-softmax of the filtered logits, arg-max, cumulative log-probability that includes the
    EOT token, ``avg_logprob = sum / (n_tokens + 1)``;
  * CTranslate2-style beam search with patience / length penalty / repetition penalty /
    no-repeat-ngram (``beam_search``), restated from the published algorithm
    (CTranslate2 ``src/decoding.cc``).  PARITY UNPINNED: no upstream binary or golden vectors are
    available offline, see DESIGN.md.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from .whisper_ref import CachedDecoder, WhisperOracle

NEG_INF = float("-inf")


@dataclass
class TokenLayout:
    eot: int
    sot: int
    no_speech: int
    no_timestamps: int
    timestamp_begin: int
    blank: int = 220

    @staticmethod
    def for_vocab(n_vocab: int) -> "TokenLayout":
        num_lang = n_vocab - 51765 - 1
        sot = 50258
        translate = sot + 1 + num_lang
        return TokenLayout(eot=50257, sot=sot, no_speech=translate + 4, no_timestamps=translate + 5,
                           timestamp_begin=translate + 6)


@dataclass
class FilterConfig:
    suppress_blank: bool = True
    suppress_tokens: Sequence[int] = field(default_factory=tuple)
    without_timestamps: bool = False
    max_initial_timestamp_index: Optional[int] = 50   # 1.0 s / 0.02


def filter_logits(logits: torch.Tensor, history: Sequence[Sequence[int]], sample_begin: int,
                  lay: TokenLayout, cfg: FilterConfig) -> torch.Tensor:
    """Apply Whisper's filters to ``logits`` [R, V]; ``history[r]`` is row r's full token list."""
    out = logits.clone().float()
    tb = lay.timestamp_begin
    for r, toks in enumerate(history):
        sampled = list(toks[sample_begin:])
        row = out[r]
        if cfg.suppress_blank and not sampled:
            row[lay.blank] = NEG_INF
            row[lay.eot] = NEG_INF
        if len(cfg.suppress_tokens):
            row[list(cfg.suppress_tokens)] = NEG_INF
        if cfg.without_timestamps:
            continue
        row[lay.no_timestamps] = NEG_INF
        last_ts = len(sampled) >= 1 and sampled[-1] >= tb
        penult_ts = len(sampled) < 2 or sampled[-2] >= tb
        if last_ts:
            if penult_ts:
                row[tb:] = NEG_INF          # pair closed: no third timestamp
            else:
                row[:lay.eot] = NEG_INF     # pair open: text is not allowed
        stamps = [t for t in sampled if t >= tb]
        if stamps:
            floor = stamps[-1] if (last_ts and not penult_ts) else stamps[-1] + 1
            row[tb:floor] = NEG_INF
        if not sampled:
            row[:tb] = NEG_INF
            if cfg.max_initial_timestamp_index is not None:
                row[tb + cfg.max_initial_timestamp_index + 1:] = NEG_INF
        lp = torch.log_softmax(row, dim=-1)
        if torch.logsumexp(lp[tb:], dim=-1) > lp[:tb].max():
            row[:tb] = NEG_INF
    return out


@dataclass
class GreedyOut:
    tokens: List[List[int]]
    sum_logprob: np.ndarray
    token_logprob: List[List[float]]
    no_speech_prob: np.ndarray

    def avg_logprob(self) -> np.ndarray:
        return np.array([s / (len(t) + 1) for s, t in zip(self.sum_logprob, self.tokens)], dtype=np.float32)


def greedy_decode(model: WhisperOracle, xa: torch.Tensor, prompt: Sequence[int], max_new_tokens: int,
                  cfg: Optional[FilterConfig] = None, forced: Optional[Sequence[Sequence[int]]] = None,
                  processors: Optional["BeamConfig"] = None) -> GreedyOut:
    """Greedy decode every window of ``xa`` [B, T, D].  ``forced`` (teacher forcing) feeds the given
    tokens instead of the arg-max while still reporting what the model would have scored them.
    ``processors`` (its repetition_penalty / no_repeat_ngram_size) applies the ctranslate2 logits
    processors before the Whisper rules, as ``Whisper.generate(beam_size=1, repetition_penalty=...)`` does."""
    cfg = cfg or FilterConfig()
    lay = TokenLayout.for_vocab(model.dims.n_vocab)
    B = xa.shape[0]
    P = len(prompt)
    dec = CachedDecoder(model, xa)
    hist = [list(prompt) for _ in range(B)]
    done = [False] * B
    sums = np.zeros(B, dtype=np.float64)
    tlp: List[List[float]] = [[] for _ in range(B)]
    nsp = np.zeros(B, dtype=np.float32)
    with torch.no_grad():
        logits = None
        for p in range(P):
            logits = dec.step(torch.tensor([[h[p]] for h in hist]))
            if p == 0:
                nsp = torch.softmax(logits.float(), dim=-1)[:, lay.no_speech].numpy().astype(np.float32)
        for i in range(max_new_tokens):
            if processors is not None:
                logits = _apply_ct2_processors(logits, [h[P - 1:] for h in hist], processors)
            filt = filter_logits(logits, hist, P, lay, cfg)
            lp = torch.log_softmax(filt, dim=-1)
            nxt = lp.argmax(dim=-1).tolist()
            for r in range(B):
                if done[r]:
                    hist[r].append(lay.eot)
                    continue
                tok = nxt[r] if forced is None else int(forced[r][i]) if i < len(forced[r]) else lay.eot
                sums[r] += float(lp[r, tok])
                tlp[r].append(float(lp[r, tok]))
                hist[r].append(tok)
                if tok == lay.eot:
                    done[r] = True
            if all(done) or i + 1 == max_new_tokens:
                break
            logits = dec.step(torch.tensor([[h[-1]] for h in hist]))
    toks = []
    for r in range(B):
        seq = hist[r][P:]
        if lay.eot in seq:
            seq = seq[:seq.index(lay.eot)]
        toks.append(seq)
    return GreedyOut(toks, sums.astype(np.float32), tlp, nsp)


# --------------------------------------------------------------------------------------------------
# CTranslate2-style beam search (literal restatement: flatten beam x vocab, top 2*beam, refill)
# --------------------------------------------------------------------------------------------------
@dataclass
class BeamConfig:
    beam_size: int = 5
    patience: float = 1.0
    length_penalty: float = 1.0
    repetition_penalty: float = 1.0
    no_repeat_ngram_size: int = 0
    max_new_tokens: int = 224


def _apply_ct2_processors(logits: torch.Tensor, seqs: Sequence[Sequence[int]], cfg: BeamConfig) -> torch.Tensor:
    """RepetitionPenalty then NoRepeatNgram over the decoded sequences (start token included)."""
    out = logits.clone().float()
    for r, seq in enumerate(seqs):
        if cfg.repetition_penalty != 1.0 and len(seq):
            idx = torch.tensor(sorted(set(seq)))
            vals = out[r, idx]
            out[r, idx] = torch.where(vals < 0, vals * cfg.repetition_penalty, vals / cfg.repetition_penalty)
        n = cfg.no_repeat_ngram_size
        if n > 0 and len(seq) >= n:
            tail = list(seq[len(seq) - n + 1:]) if n > 1 else []
            for i in range(len(seq) - n + 1):
                if list(seq[i:i + n - 1]) == tail:
                    out[r, seq[i + n - 1]] = NEG_INF
    return out


def beam_search(model: WhisperOracle, xa: torch.Tensor, prompt: Sequence[int], bcfg: BeamConfig,
                fcfg: Optional[FilterConfig] = None, trace: Optional[dict] = None):
    """One window (xa [1, T, D]).  Returns (hypotheses best-first [(tokens, score, cum_logprob)], no_speech_prob).
    ``trace`` (a dict) receives which branches of the search ran: ``refills`` (finished slots re-filled from the
    candidates ``beam .. 2*beam``), ``eot_skipped`` (EOT candidates passed over while refilling), ``finish_steps``
    (step index of every finished hypothesis, in registration order), ``stop`` ("patience" | "length"), ``steps``."""
    fcfg = fcfg or FilterConfig()
    lay = TokenLayout.for_vocab(model.dims.n_vocab)
    K, V = bcfg.beam_size, model.dims.n_vocab
    P = len(prompt)
    max_new = min(bcfg.max_new_tokens, model.dims.n_text_ctx - P)
    max_candidates = int(np.floor(np.float32(K) * np.float32(bcfg.patience) + np.float32(0.5)))      # C++ std::round on floats: half away from zero
    with torch.no_grad():
        dec = CachedDecoder(model, xa.expand(K, -1, -1).contiguous())
        nsp = 0.0
        for p in range(P - 1):
            lg = dec.step(torch.full((K, 1), prompt[p]))
            if p == 0:
                nsp = float(torch.softmax(lg[0].float(), -1)[lay.no_speech])
        seqs: List[List[int]] = [[] for _ in range(K)]           # generated tokens per beam
        cums: List[List[float]] = [[] for _ in range(K)]         # cumulative log-prob after each of them
        scores = torch.full((K,), NEG_INF)
        scores[0] = 0.0
        feed = torch.full((K, 1), prompt[-1])
        finished: List = []
        tr = dict(refills=0, eot_skipped=0, finish_steps=[], stop="length", steps=0)
        for step in range(max_new):
            tr["steps"] = step + 1
            logits = dec.step(feed)
            if P == 1 and step == 0:
                nsp = float(torch.softmax(logits[0].float(), -1)[lay.no_speech])
            logits = _apply_ct2_processors(logits, [[prompt[-1]] + s for s in seqs], bcfg)
            hist = [list(prompt) + s for s in seqs]
            logits = filter_logits(logits, hist, P, lay, fcfg)
            lp = torch.log_softmax(logits, dim=-1) + scores[:, None]
            flat = lp.reshape(-1)
            top_s, top_i = torch.topk(flat, 2 * K)
            cand = [(float(s), int(i) // V, int(i) % V) for s, i in zip(top_s, top_i) if s > NEG_INF]
            last = step == max_new - 1
            nxt, secondary = [], K
            for k in range(min(K, len(cand))):
                s, b, t = cand[k]
                use = cand[k]
                if t == lay.eot or last:
                    finished.append((s, seqs[b] + ([] if t == lay.eot else [t]), cums[b] + ([] if t == lay.eot else [s])))
                    tr["finish_steps"].append(step)
                    for j in range(secondary, len(cand)):
                        if cand[j][2] != lay.eot:
                            use, secondary = cand[j], j + 1
                            tr["refills"] += 0 if last else 1
                            break
                        tr["eot_skipped"] += 1
                nxt.append(use)
            if last or len(finished) >= max_candidates:
                tr["stop"] = "length" if last else "patience"
                break
            parents = [b for _, b, _ in nxt] + [0] * (K - len(nxt))
            new_scores = [s for s, _, _ in nxt] + [NEG_INF] * (K - len(nxt))
            new_seqs = [seqs[b] + [t] for _, b, t in nxt] + [list(seqs[0]) for _ in range(K - len(nxt))]
            cums = [cums[b] + [sc] for sc, b, _ in nxt] + [list(cums[0]) for _ in range(K - len(nxt))]
            dec.reorder(torch.tensor(parents))
            seqs, scores = new_seqs, torch.tensor(new_scores)
            feed = torch.tensor([[t] for _, _, t in nxt] + [[lay.eot]] * (K - len(nxt)))
    lpn = bcfg.length_penalty
    ranked = sorted(((s / (max(len(t), 1) ** lpn) if lpn != 0 else s, s, t, c) for s, t, c in finished), key=lambda x: -x[0])
    if trace is not None:
        # per hypothesis (best first): log p of each token under the distribution the search ranked it in, then what ending
        # the sequence added (log p(EOT); 0 at the length limit) -- the differences of the cumulative scores
        tr["token_logprobs"] = [[c[j] - (c[j - 1] if j else 0.0) for j in range(len(c))] + [s - (c[-1] if c else 0.0)] for _, s, _, c in ranked]
        trace.update(tr)
    return [(t, n, s) for n, s, t, _ in ranked], nsp


# --------------------------------------------------------------------------------------------------
# openai-whisper beam search (whisper/decoding.py BeamSearchDecoder + MaximumLikelihoodRanker), literal
# --------------------------------------------------------------------------------------------------
def beam_search_openai(model: WhisperOracle, xa: torch.Tensor, prompt: Sequence[int], beam_size: int,
                       patience: Optional[float], length_penalty: Optional[float], sample_len: int,
                       fcfg: Optional[FilterConfig] = None):
    """One window.  Returns (tokens of the selected hypothesis, sum_logprob, avg_logprob, no_speech_prob)."""
    fcfg = fcfg or FilterConfig()
    lay = TokenLayout.for_vocab(model.dims.n_vocab)
    K, P = beam_size, len(prompt)
    max_candidates = round(K * (patience or 1.0))
    with torch.no_grad():
        dec = CachedDecoder(model, xa.expand(K, -1, -1).contiguous())
        nsp = 0.0
        for p in range(P - 1):
            lg = dec.step(torch.full((K, 1), prompt[p]))
            if p == 0:
                nsp = float(torch.softmax(lg[0].float(), -1)[lay.no_speech])
        tokens = [list(prompt) for _ in range(K)]
        sum_lp = [0.0] * K
        finished: dict = {}
        feed = torch.full((K, 1), prompt[-1])
        for i in range(sample_len):
            logits = dec.step(feed)
            if P == 1 and i == 0:
                nsp = float(torch.softmax(logits[0].float(), -1)[lay.no_speech])
            logits = filter_logits(logits, tokens, P, lay, fcfg)
            logprobs = torch.log_softmax(logits.float(), dim=-1)
            scores, sources, newly = {}, {}, {}
            for j in range(K):
                vals, idx = logprobs[j].topk(K + 1)
                for lp, tok in zip(vals.tolist(), idx.tolist()):
                    if lp == NEG_INF:
                        continue
                    seq = tuple(tokens[j] + [tok])
                    scores[seq] = sum_lp[j] + lp
                    sources[seq] = j
            nxt, src = [], []
            for seq in sorted(scores, key=scores.get, reverse=True):
                if seq[-1] == lay.eot:
                    newly[seq] = scores[seq]
                else:
                    nxt.append(seq)
                    src.append(sources[seq])
                    if len(nxt) == K:
                        break
            while len(nxt) < K:
                nxt.append(nxt[-1]); src.append(src[-1])
            sum_lp = [scores[s] for s in nxt]
            tokens = [list(s) for s in nxt]
            dec.reorder(torch.tensor(src))
            feed = torch.tensor([[s[-1]] for s in nxt])
            for seq in sorted(newly, key=newly.get, reverse=True):
                if len(finished) >= max_candidates:
                    break
                finished[seq] = newly[seq]
            if len(finished) >= max_candidates:
                break
    cand = dict(finished)
    if len(cand) < K:
        for j in np.argsort(sum_lp)[::-1]:
            cand[tuple(tokens[j] + [lay.eot])] = sum_lp[j]
            if len(cand) >= K:
                break
    seqs = [list(s[P:]) for s in cand]
    seqs = [s[: s.index(lay.eot)] if lay.eot in s else s for s in seqs]
    totals = list(cand.values())

    def score(lp, n):
        pen = n if length_penalty is None else ((5 + n) / 6) ** length_penalty
        return lp / pen if pen else NEG_INF
    best = int(np.argmax([score(lp, len(s)) for lp, s in zip(totals, seqs)]))
    return seqs[best], totals[best], totals[best] / (len(seqs[best]) + 1), nsp
