"""Oracle: Whisper encoder / decoder forward pass in PyTorch-CPU fp32 (TEST INFRASTRUCTURE).

Restates the transformer arithmetic of openai-whisper 20250625 ``whisper/model.py``
(``AudioEncoder``, ``TextDecoder``, ``ResidualAttentionBlock``, ``MultiHeadAttention``)
-- identical math to CTranslate2 4.7.1 ``models::Whisper::encode`` / decoder layers that
faster-whisper drives.  Neither package is vendored in /root/reference; the reference's
call sites are whisperjav/modules/faster_whisper_pro_asr.py:247-253,819-822 and
whisperjav/modules/whisper_pro_asr.py:182,433.

Pinned against ``transformers.models.whisper.WhisperForConditionalGeneration`` (independent
implementation, seeded random weights) in tests/test_oracle_whisper.py.

Weights are a flat ``dict[str, np.ndarray]`` using openai-whisper's state-dict names
(``encoder.conv1.weight`` ... ``decoder.ln.bias``); see ``whisperjav_amd.weights`` for the
blob layout the HIP engine consumes (same names).

``act_round`` lets the oracle emulate the HIP engine's 16-bit rounding points (every GEMM /
attention operand is rounded to bf16 resp. fp16, accumulation stays fp32) so that the 16-bit
compute types can be checked against something tighter than "fp32 +- rounding noise".
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class WhisperDims:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int

    @staticmethod
    def named(name: str) -> "WhisperDims":
        table = {
            # name: (mels, state, heads, layers, vocab)
            "tiny": (80, 384, 6, 4, 51865),
            "base": (80, 512, 8, 6, 51865),
            "small": (80, 768, 12, 12, 51865),
            "medium": (80, 1024, 16, 24, 51865),
            "large-v2": (80, 1280, 20, 32, 51865),
            "large-v3": (128, 1280, 20, 32, 51866),
        }
        m, d, h, l, v = table[name]
        return WhisperDims(m, 1500, d, h, l, v, 448, d, h, l)


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    """Encoder positional table, ``whisper.model.sinusoids``."""
    half = channels // 2
    inc = math.log(max_timescale) / (half - 1)
    inv = np.exp(-inc * np.arange(half, dtype=np.float32)).astype(np.float32)
    t = np.arange(length, dtype=np.float32)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def f16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)


class WhisperOracle:
    """Functional fp32 Whisper over a numpy weight dict."""

    def __init__(self, dims: WhisperDims, weights: Dict[str, np.ndarray],
                 act_round: Optional[Callable[[torch.Tensor], torch.Tensor]] = None):
        self.dims = dims
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
                  for k, v in weights.items()}
        self.rnd = act_round if act_round is not None else (lambda t: t)

    # ---- building blocks -------------------------------------------------
    def _linear(self, x, prefix, bias=True):
        y = self.rnd(x) @ self.rnd(self.w[prefix + ".weight"]).T
        if bias:
            y = y + self.w[prefix + ".bias"]
        return y

    def _ln(self, x, prefix):
        return F.layer_norm(x, (x.shape[-1],), self.w[prefix + ".weight"],
                            self.w[prefix + ".bias"], 1e-5)

    def _attention(self, q, k, v, n_head, causal_from: Optional[int] = None, qk_out: Optional[list] = None):
        """q [B,Tq,D], k/v [B,Tk,D]; scores scaled by 1/sqrt(d_head); fp32 softmax.
        ``qk_out`` (a list) receives the scaled scores [B,H,Tq,Tk] -- what whisper/timing.py's hooks capture."""
        B, Tq, D = q.shape
        Tk = k.shape[1]
        dh = D // n_head
        qh = self.rnd(q).view(B, Tq, n_head, dh).permute(0, 2, 1, 3)
        kh = self.rnd(k).view(B, Tk, n_head, dh).permute(0, 2, 1, 3)
        vh = self.rnd(v).view(B, Tk, n_head, dh).permute(0, 2, 1, 3)
        s = (qh @ kh.transpose(-1, -2)) * (dh ** -0.5)
        if causal_from is not None:
            # query i (absolute position causal_from + i) may see keys <= its position
            qpos = torch.arange(Tq)[:, None] + causal_from
            kpos = torch.arange(Tk)[None, :]
            s = s.masked_fill(kpos > qpos, float("-inf"))
        if qk_out is not None:
            qk_out.append(s)
        p = torch.softmax(s, dim=-1)
        o = self.rnd(p) @ vh
        return o.permute(0, 2, 1, 3).reshape(B, Tq, D)

    # ---- encoder -----------------------------------------------------------
    def encoder_stem(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, n_mels, 3000] -> [B, 1500, D] (conv1+GELU, conv2(s2)+GELU, +pos)."""
        w1 = self.rnd(self.w["encoder.conv1.weight"])
        w2 = self.rnd(self.w["encoder.conv2.weight"])
        x = F.gelu(F.conv1d(self.rnd(mel), w1, self.w["encoder.conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(self.rnd(x), w2, self.w["encoder.conv2.bias"], stride=2, padding=1))
        x = x.permute(0, 2, 1)
        return x + self.w["encoder.positional_embedding"]

    def encoder_block(self, x: torch.Tensor, i: int) -> torch.Tensor:
        p = f"encoder.blocks.{i}."
        h = self._ln(x, p + "attn_ln")
        q = self._linear(h, p + "attn.query")
        k = self._linear(h, p + "attn.key", bias=False)
        v = self._linear(h, p + "attn.value")
        a = self._attention(q, k, v, self.dims.n_audio_head)
        x = x + self._linear(a, p + "attn.out")
        h = self._ln(x, p + "mlp_ln")
        h = F.gelu(self._linear(h, p + "mlp.0"))
        return x + self._linear(h, p + "mlp.2")

    def encode(self, mel, n_layers: Optional[int] = None, final_ln: bool = True) -> torch.Tensor:
        mel = torch.as_tensor(mel, dtype=torch.float32)
        x = self.encoder_stem(mel)
        L = self.dims.n_audio_layer if n_layers is None else n_layers
        for i in range(L):
            x = self.encoder_block(x, i)
        return self._ln(x, "encoder.ln_post") if final_ln else x

    # ---- decoder -----------------------------------------------------------
    def cross_kv(self, xa: torch.Tensor):
        """Per-layer cross-attention K/V of the encoder output, computed once per window."""
        out = []
        for i in range(self.dims.n_text_layer):
            p = f"decoder.blocks.{i}.cross_attn."
            out.append((self._linear(xa, p + "key", bias=False), self._linear(xa, p + "value")))
        return out

    def decoder_logits(self, tokens: torch.Tensor, xa: torch.Tensor, cross=None,
                       n_layers: Optional[int] = None, cross_qk: Optional[list] = None) -> torch.Tensor:
        """Full (uncached) decoder pass: tokens [B,T] int64 -> logits [B,T,V] fp32.
        ``cross_qk`` (a list) receives every layer's scaled cross-attention scores [B,H,T,Tk]."""
        B, T = tokens.shape
        x = self.w["decoder.token_embedding.weight"][tokens] + \
            self.w["decoder.positional_embedding"][:T]
        if cross is None:
            cross = self.cross_kv(xa)
        L = self.dims.n_text_layer if n_layers is None else n_layers
        H = self.dims.n_text_head
        for i in range(L):
            p = f"decoder.blocks.{i}."
            h = self._ln(x, p + "attn_ln")
            q = self._linear(h, p + "attn.query")
            k = self._linear(h, p + "attn.key", bias=False)
            v = self._linear(h, p + "attn.value")
            x = x + self._linear(self._attention(q, k, v, H, causal_from=0), p + "attn.out")
            h = self._ln(x, p + "cross_attn_ln")
            q = self._linear(h, p + "cross_attn.query")
            ck, cv = cross[i]
            x = x + self._linear(self._attention(q, ck, cv, H, qk_out=cross_qk), p + "cross_attn.out")
            h = self._ln(x, p + "mlp_ln")
            h = F.gelu(self._linear(h, p + "mlp.0"))
            x = x + self._linear(h, p + "mlp.2")
        x = self._ln(x, "decoder.ln")
        return self.rnd(x) @ self.rnd(self.w["decoder.token_embedding.weight"]).T


class CachedDecoder:
    """Incremental decoder with a self-attention KV cache (what a real decode loop runs).

    Mathematically identical to ``WhisperOracle.decoder_logits`` on the last position;
    exists so that the CPU baseline timing and the greedy/beam oracles have the same
    per-step cost structure as upstream's cached decoders.
    """

    def __init__(self, model: WhisperOracle, xa: torch.Tensor):
        self.m = model
        self.cross = model.cross_kv(xa)
        L = model.dims.n_text_layer
        self.k: List[Optional[torch.Tensor]] = [None] * L
        self.v: List[Optional[torch.Tensor]] = [None] * L
        self.pos = 0

    def reorder(self, parent: torch.Tensor) -> None:
        self.k = [None if t is None else t[parent] for t in self.k]
        self.v = [None if t is None else t[parent] for t in self.v]

    def step(self, tokens: torch.Tensor, cross_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """tokens [R, T_new] -> logits of the last position [R, V]. ``cross_index`` maps
        each row to its window (rows = windows x beams)."""
        m = self.m
        R, T = tokens.shape
        H = m.dims.n_text_head
        x = m.w["decoder.token_embedding.weight"][tokens] + \
            m.w["decoder.positional_embedding"][self.pos:self.pos + T]
        for i in range(m.dims.n_text_layer):
            p = f"decoder.blocks.{i}."
            h = m._ln(x, p + "attn_ln")
            q = m._linear(h, p + "attn.query")
            k = m._linear(h, p + "attn.key", bias=False)
            v = m._linear(h, p + "attn.value")
            self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], dim=1)
            self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], dim=1)
            a = m._attention(q, self.k[i], self.v[i], H, causal_from=self.pos)
            x = x + m._linear(a, p + "attn.out")
            h = m._ln(x, p + "cross_attn_ln")
            q = m._linear(h, p + "cross_attn.query")
            ck, cv = self.cross[i]
            if cross_index is not None:
                ck, cv = ck[cross_index], cv[cross_index]
            x = x + m._linear(m._attention(q, ck, cv, H), p + "cross_attn.out")
            h = m._ln(x, p + "mlp_ln")
            h = F.gelu(m._linear(h, p + "mlp.0"))
            x = x + m._linear(h, p + "mlp.2")
        self.pos += T
        x = m._ln(x[:, -1], "decoder.ln")
        return m.rnd(x) @ m.rnd(m.w["decoder.token_embedding.weight"]).T
