"""Oracle: Qwen3-ASR forward pass in PyTorch-CPU fp32 (TEST INFRASTRUCTURE, never shipped) -- SURVEY.md 8f-3.

The reference's ``--mode qwen`` (BASELINE cfg5) drives the un-vendored ``qwen_asr`` package
(/root/reference/whisperjav/modules/qwen_asr.py:192-193 names the checkpoints ``Qwen/Qwen3-ASR-1.7B`` and
``Qwen/Qwen3-ForcedAligner-0.6B``; the model is loaded at :545-636 and called at :638-757).  Neither the package nor a
checkpoint is available offline, so this restates the published architecture and is pinned against the independent
implementation that IS importable here, ``transformers.models.qwen3_asr`` (tests/test_oracle_qwen3.py: random weights, the
whole path mel -> audio tower -> projector -> Qwen3 decoder -> logits, and greedy generation):

  * audio tower: the mel frames are cut into chunks of ``2 * n_window`` = 100 frames; three 3x3 stride-2 convolutions
    (over frequency x time, GELU) + a linear over (channels x frequency) give 13 tokens per chunk, a 13-position sinusoid
    table is added per chunk, padding tokens are dropped; pre-LN transformer layers whose self-attention runs inside
    windows of ``n_window_infer / (2 n_window)`` chunks (non-causal); ``ln_post``; projector linear-GELU-linear to the
    decoder width;
  * decoder: Qwen3 -- RMSNorm, per-head RMSNorm on q and k, rotary position embedding (rotate-half convention), grouped-
    query causal attention, SwiGLU MLP, tied LM head; the audio embeddings replace the ``<audio>`` placeholder tokens of
    the prompt.

Weights use transformers' state-dict names (``model.audio_tower.conv2d1.weight`` ...
``model.language_model.layers.0.self_attn.q_proj.weight``), which are also the names of the published ``-hf`` checkpoint.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class Qwen3AsrDims:
    # audio tower
    n_mels: int = 128
    a_layers: int = 24
    a_heads: int = 16
    a_ffn: int = 4096
    a_d: int = 1024
    n_window: int = 50
    n_window_infer: int = 800
    conv_hidden: int = 480
    a_max_pos: int = 13
    # decoder
    d: int = 2048
    layers: int = 28
    heads: int = 16
    kv_heads: int = 8
    head_dim: int = 128
    ffn: int = 6144
    vocab: int = 151936
    rope_theta: float = 1000000.0
    rms_eps: float = 1e-6
    audio_token_id: int = 151676
    eos_token_ids: Tuple[int, ...] = (151643, 151645)

    @property
    def freq_bins(self) -> int:        # mel bins after three stride-2 convolutions
        return (((self.n_mels + 1) // 2 + 1) // 2 + 1) // 2


def sinusoid_table(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2).float())
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def post_cnn_length(n: int) -> int:
    """Tokens a chunk of ``n`` valid mel frames yields: three (k=3, s=2, p=1) convolutions; 0 stays 0."""
    for _ in range(3):
        n = (n - 1) // 2 + 1 if n > 0 else 0
    return n


def audio_token_count(n_frames: int, n_window: int = 50) -> int:
    """``_get_feat_extract_output_lengths``: 13 tokens per full 100-frame chunk + the tail chunk's share."""
    chunk = 2 * n_window
    tail = n_frames % chunk
    t = (tail - 1) // 2 + 1
    return ((t - 1) // 2 + 1 - 1) // 2 + 1 + (n_frames // chunk) * 13


def fix_timestamps(raw: Sequence[float]) -> List[int]:
    """The forced aligner's monotonic repair (qwen_asr ``qwen3_forced_aligner.py``; transformers ``_fix_timestamps``): the
    longest non-decreasing subsequence is kept, outlier blocks of <= 2 snap to the nearer good neighbour, longer ones are
    interpolated linearly between the surrounding good values; result truncated to int."""
    data = [float(v) for v in raw]
    n = len(data)
    if n == 0:
        return []
    dp, parent = [1] * n, [-1] * n
    for i in range(1, n):
        for j in range(i):
            if data[j] <= data[i] and dp[j] + 1 > dp[i]:
                dp[i], parent[i] = dp[j] + 1, j
    idx = dp.index(max(dp))
    good = [False] * n
    while idx != -1:
        good[idx] = True
        idx = parent[idx]
    out = list(data)
    i = 0
    while i < n:
        if good[i]:
            i += 1
            continue
        j = i
        while j < n and not good[j]:
            j += 1
        left = next((out[k] for k in range(i - 1, -1, -1) if good[k]), None)
        right = next((out[k] for k in range(j, n) if good[k]), None)
        for pos in range(i, j):
            if j - i <= 2:
                if left is None:
                    out[pos] = right
                elif right is None:
                    out[pos] = left
                else:
                    out[pos] = left if (pos - (i - 1)) <= (j - pos) else right
            elif left is not None and right is not None:
                out[pos] = left + (right - left) / (j - i + 1) * (pos - i + 1)
            else:
                out[pos] = left if left is not None else right
        i = j
    return [int(v) for v in out]


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def repetition_penalised(logits: torch.Tensor, input_ids: Sequence[int], penalty: float) -> torch.Tensor:
    """transformers ``RepetitionPenaltyLogitsProcessor.__call__`` (generation/logits_process.py) for one row: the logit of every
    id in ``input_ids`` (the prompt AND what has been generated) is divided by ``penalty`` when positive, multiplied when
    negative; gather-then-scatter, so an id that occurs twice is still penalised once."""
    if penalty == 1.0 or not len(input_ids):
        return logits
    ids = torch.as_tensor(list(input_ids), dtype=torch.long)
    score = logits[ids]
    score = torch.where(score < 0, score * penalty, score / penalty)
    out = logits.clone()
    out[ids] = score
    return out


def dynamic_token_limit(audio_duration_sec: float, max_new_tokens: int, max_tokens_per_audio_second: float,
                        min_tokens_floor: int = 256) -> int:
    """``QwenASR._compute_dynamic_token_limit`` (modules/qwen_asr.py:414-437): the budget of a clip grows with its duration,
    never under the floor, never over the static limit; rate or duration <= 0 disables the scaling."""
    if max_tokens_per_audio_second <= 0 or audio_duration_sec <= 0:
        return max_new_tokens
    return min(max(min_tokens_floor, int(audio_duration_sec * max_tokens_per_audio_second)), max_new_tokens)


class Qwen3AsrOracle:
    def __init__(self, dims: Qwen3AsrDims, weights: Dict[str, np.ndarray]):
        self.dims = dims
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in weights.items()}
        inv = 1.0 / (dims.rope_theta ** (torch.arange(0, dims.head_dim, 2).float() / dims.head_dim))
        self.inv_freq = inv

    # ---- audio tower ------------------------------------------------------------------------------
    def audio_tokens(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [n_mels, n_frames] (valid frames of ONE clip) -> projected audio embeddings [n_tokens, d]."""
        d, w = self.dims, self.w
        p = "model.audio_tower."
        chunk = 2 * d.n_window
        n = mel.shape[1]
        n_chunks = (n + chunk - 1) // chunk
        x = F.pad(mel, (0, n_chunks * chunk - n))
        x = x.view(d.n_mels, n_chunks, chunk).permute(1, 0, 2)[:, None]                  # [chunks, 1, mels, 100]
        for i in (1, 2, 3):
            x = F.gelu(F.conv2d(x, w[f"{p}conv2d{i}.weight"], w[f"{p}conv2d{i}.bias"], stride=2, padding=1))
        c, ch, fb, ts = x.shape
        x = x.permute(0, 3, 1, 2).reshape(c, ts, ch * fb) @ w[p + "conv_out.weight"].T   # [chunks, 13, a_d]
        x = x + sinusoid_table(d.a_max_pos, d.a_d)[:ts]
        lens = [post_cnn_length(min(chunk, n - i * chunk)) for i in range(n_chunks)]
        h = torch.cat([x[i, : lens[i]] for i in range(n_chunks)], dim=0)                 # packed valid tokens
        # attention windows: n_window_infer / chunk chunks of (max tokens per chunk) tokens
        win = max(lens) * (d.n_window_infer // chunk)
        total = h.shape[0]
        bounds = list(range(0, total, win)) + [total]
        hd = d.a_d // d.a_heads
        for l in range(d.a_layers):
            q = f"{p}layers.{l}."
            y = F.layer_norm(h, (d.a_d,), w[q + "self_attn_layer_norm.weight"], w[q + "self_attn_layer_norm.bias"], 1e-5)
            qs = (y @ w[q + "self_attn.q_proj.weight"].T + w[q + "self_attn.q_proj.bias"]).view(total, d.a_heads, hd)
            ks = (y @ w[q + "self_attn.k_proj.weight"].T + w[q + "self_attn.k_proj.bias"]).view(total, d.a_heads, hd)
            vs = (y @ w[q + "self_attn.v_proj.weight"].T + w[q + "self_attn.v_proj.bias"]).view(total, d.a_heads, hd)
            outs = []
            for a, b in zip(bounds[:-1], bounds[1:]):
                s = torch.einsum("qhd,khd->hqk", qs[a:b], ks[a:b]) * hd ** -0.5
                outs.append(torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), vs[a:b]).reshape(b - a, d.a_d))
            h = h + torch.cat(outs, 0) @ w[q + "self_attn.out_proj.weight"].T + w[q + "self_attn.out_proj.bias"]
            y = F.layer_norm(h, (d.a_d,), w[q + "final_layer_norm.weight"], w[q + "final_layer_norm.bias"], 1e-5)
            y = F.gelu(y @ w[q + "fc1.weight"].T + w[q + "fc1.bias"])
            h = h + y @ w[q + "fc2.weight"].T + w[q + "fc2.bias"]
        h = F.layer_norm(h, (d.a_d,), w[p + "ln_post.weight"], w[p + "ln_post.bias"], 1e-5)
        m = "model.multi_modal_projector."
        h = F.gelu(h @ w[m + "linear_1.weight"].T + w[m + "linear_1.bias"])
        return h @ w[m + "linear_2.weight"].T + w[m + "linear_2.bias"]

    # ---- decoder ------------------------------------------------------------------------------------
    def embed(self, tokens: Sequence[int], audio: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Token embeddings with the ``<audio>`` placeholders replaced, in order, by the rows of ``audio``."""
        ids = torch.as_tensor(list(tokens), dtype=torch.long)
        x = self.w["model.language_model.embed_tokens.weight"][ids].clone()
        if audio is not None:
            mask = ids == self.dims.audio_token_id
            assert int(mask.sum()) == audio.shape[0], (int(mask.sum()), audio.shape)
            x[mask] = audio
        return x

    def _rope(self, x: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
        """x [T, heads, head_dim], pos [T]: rotate-half convention."""
        ang = pos[:, None].float() * self.inv_freq[None, :]
        cos = torch.cat([ang.cos(), ang.cos()], -1)[:, None, :]
        sin = torch.cat([ang.sin(), ang.sin()], -1)[:, None, :]
        half = x.shape[-1] // 2
        rot = torch.cat([-x[..., half:], x[..., :half]], -1)
        return x * cos + rot * sin

    def decoder_layer(self, x: torch.Tensor, l: int, pos0: int, cache: Optional[list]) -> torch.Tensor:
        """x [T, d] = positions pos0 .. pos0+T-1 of ONE sequence; ``cache[l]`` = (k, v) of the earlier positions."""
        d, w = self.dims, self.w
        p = f"model.language_model.layers.{l}."
        T = x.shape[0]
        y = rms_norm(x, w[p + "input_layernorm.weight"], d.rms_eps)
        q = (y @ w[p + "self_attn.q_proj.weight"].T).view(T, d.heads, d.head_dim)
        k = (y @ w[p + "self_attn.k_proj.weight"].T).view(T, d.kv_heads, d.head_dim)
        v = (y @ w[p + "self_attn.v_proj.weight"].T).view(T, d.kv_heads, d.head_dim)
        q = rms_norm(q, w[p + "self_attn.q_norm.weight"], d.rms_eps)
        k = rms_norm(k, w[p + "self_attn.k_norm.weight"], d.rms_eps)
        pos = torch.arange(pos0, pos0 + T)
        q, k = self._rope(q, pos), self._rope(k, pos)
        if cache is not None:
            if cache[l] is not None:
                k, v = torch.cat([cache[l][0], k], 0), torch.cat([cache[l][1], v], 0)
            cache[l] = (k, v)
        g = d.heads // d.kv_heads
        kk, vv = k.repeat_interleave(g, dim=1), v.repeat_interleave(g, dim=1)
        s = torch.einsum("qhd,khd->hqk", q, kk) * d.head_dim ** -0.5
        qpos = pos[:, None]
        kpos = torch.arange(kk.shape[0])[None, :]
        s = s.masked_fill((kpos > qpos)[None], float("-inf"))
        a = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vv).reshape(T, d.heads * d.head_dim)
        x = x + a @ w[p + "self_attn.o_proj.weight"].T
        y = rms_norm(x, w[p + "post_attention_layernorm.weight"], d.rms_eps)
        y = F.silu(y @ w[p + "mlp.gate_proj.weight"].T) * (y @ w[p + "mlp.up_proj.weight"].T)
        return x + y @ w[p + "mlp.down_proj.weight"].T

    def logits(self, x: torch.Tensor, pos0: int = 0, cache: Optional[list] = None, n_layers: Optional[int] = None) -> torch.Tensor:
        """Embeddings [T, d] -> logits [T, vocab] (tied LM head)."""
        L = self.dims.layers if n_layers is None else n_layers
        for l in range(L):
            x = self.decoder_layer(x, l, pos0, cache)
        x = rms_norm(x, self.w["model.language_model.norm.weight"], self.dims.rms_eps)
        return x @ self.w["model.language_model.embed_tokens.weight"].T

    def classify(self, x: torch.Tensor, head_w: torch.Tensor, head_b: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Token classification (the forced aligner's head over time bins): embeddings [T, d] -> logits [T, n_labels] of the
        final-normed hidden states; the aligner reads them at the ``<timestamp>`` marker positions."""
        for l in range(self.dims.layers):
            x = self.decoder_layer(x, l, 0, None)
        x = rms_norm(x, self.w["model.language_model.norm.weight"], self.dims.rms_eps)
        y = x @ head_w.T
        return y if head_b is None else y + head_b

    def greedy(self, prompt: Sequence[int], audio: Optional[torch.Tensor], max_new: int,
               repetition_penalty: float = 1.0) -> Tuple[List[int], List[float]]:
        """Greedy generation until an EOS id (not returned) or ``max_new`` tokens; per-token log-probs (of the penalised
        distribution when ``repetition_penalty`` != 1, as transformers' ``scores``).  The reference's pipeline generates with
        ``repetition_penalty=1.1`` and a per-clip ``max_new`` (pipelines/qwen_pipeline.py:157-158, modules/qwen_asr.py:382-437)."""
        cache: list = [None] * self.dims.layers
        x = self.embed(prompt, audio)
        lg = self.logits(x, 0, cache)[-1]
        out, lps = [], []
        seen = [int(t) for t in prompt]
        pos = len(prompt)
        for _ in range(max_new):
            lg = repetition_penalised(lg, seen, repetition_penalty)
            lp = torch.log_softmax(lg, -1)
            t = int(lp.argmax())
            if t in self.dims.eos_token_ids:
                lps.append(float(lp[t]))
                break
            out.append(t); lps.append(float(lp[t])); seen.append(t)
            lg = self.logits(self.w["model.language_model.embed_tokens.weight"][t][None], pos, cache)[-1]
            pos += 1
        return out, lps
