"""CPU study (oracle only): which 16-bit rounding points move the token log-probs, at the large-v3 geometry.

Each case runs the oracle with fp16 rounding switched on for a subset of
{encoder, decoder} x {GEMM activations, GEMM weights, attention operands} + the logits GEMM, and reports the
teacher-forced log-prob difference to the fp32 oracle on 32 greedy tokens.  It is the design input for the fp16
compute type's "split activation" decode GEMMs (activations enter the matrix cores as hi + lo fp16 pairs).
Writes profiles/r02_precision_ablation_cpu.json.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decoding, logmel, whisper_ref  # noqa: E402
from tests import helpers  # noqa: E402
from whisperjav_amd import dims as pdims, synth, weights as pweights  # noqa: E402


def f16(x):
    return x.to(torch.float16).to(torch.float32)


def ident(x):
    return x


class Ablated(whisper_ref.WhisperOracle):
    """flags: dict phase -> (act, wgt, attn) rounding functions; phase in {"enc", "dec", "logits"}."""

    def __init__(self, dims, weights, flags):
        super().__init__(dims, weights)
        self.flags = flags
        self.phase = "enc"

    def _linear(self, x, prefix, bias=True):
        act, wgt, _ = self.flags[self.phase]
        y = act(x) @ wgt(self.w[prefix + ".weight"]).T
        return y + self.w[prefix + ".bias"] if bias else y

    def _attention(self, q, k, v, n_head, causal_from=None, qk_out=None):
        self.rnd = self.flags[self.phase][2]
        return super()._attention(q, k, v, n_head, causal_from, qk_out)

    def encoder_stem(self, mel):
        act, wgt, _ = self.flags["enc"]
        self.rnd = lambda t: t
        import torch.nn.functional as F
        x = F.gelu(F.conv1d(act(mel), wgt(self.w["encoder.conv1.weight"]), self.w["encoder.conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(act(x), wgt(self.w["encoder.conv2.weight"]), self.w["encoder.conv2.bias"], stride=2, padding=1))
        return x.permute(0, 2, 1) + self.w["encoder.positional_embedding"]

    def encode(self, mel, n_layers=None, final_ln=True):
        self.phase = "enc"
        return super().encode(mel, n_layers, final_ln)

    def cross_kv(self, xa):
        self.phase = "enc"        # the cross K/V projection runs with the encoder (MFMA-bound batch GEMM)
        return super().cross_kv(xa)

    def decoder_logits(self, tokens, xa, cross=None, n_layers=None, cross_qk=None):
        if cross is None:
            cross = self.cross_kv(xa)
        self.phase = "dec"
        act, wgt, _ = self.flags["logits"]
        self.rnd = lambda t: t
        # re-implementation of the tail so the logits GEMM has its own switches
        B, T = tokens.shape
        x = wgt(self.w["decoder.token_embedding.weight"])[tokens] + self.w["decoder.positional_embedding"][:T]
        import torch.nn.functional as F
        H = self.dims.n_text_head
        for i in range(self.dims.n_text_layer):
            p = f"decoder.blocks.{i}."
            h = self._ln(x, p + "attn_ln")
            q = self._linear(h, p + "attn.query"); k = self._linear(h, p + "attn.key", bias=False); v = self._linear(h, p + "attn.value")
            x = x + self._linear(self._attention(q, k, v, H, causal_from=0), p + "attn.out")
            h = self._ln(x, p + "cross_attn_ln")
            q = self._linear(h, p + "cross_attn.query")
            ck, cv = cross[i]
            x = x + self._linear(self._attention(q, ck, cv, H), p + "cross_attn.out")
            h = self._ln(x, p + "mlp_ln")
            h = F.gelu(self._linear(h, p + "mlp.0"))
            x = x + self._linear(h, p + "mlp.2")
        x = self._ln(x, "decoder.ln")
        return act(x) @ wgt(self.w["decoder.token_embedding.weight"]).T


ALL = (f16, f16, f16)
NONE = (ident, ident, ident)
CASES = {
    "all16": dict(enc=ALL, dec=ALL, logits=ALL),
    "enc32": dict(enc=NONE, dec=ALL, logits=ALL),
    "dec32": dict(enc=ALL, dec=NONE, logits=NONE),
    "logits32": dict(enc=ALL, dec=ALL, logits=NONE),
    "attn_ops32": dict(enc=(f16, f16, ident), dec=(f16, f16, ident), logits=ALL),
    "all_gemm_act32": dict(enc=(ident, f16, f16), dec=(ident, f16, f16), logits=(ident, f16, f16)),
    "dec_gemm_act32": dict(enc=ALL, dec=(ident, f16, f16), logits=(ident, f16, f16)),
    "dec_gemm_act32_attn32": dict(enc=ALL, dec=(ident, f16, ident), logits=(ident, f16, f16)),
    "only_weights16": dict(enc=(ident, f16, ident), dec=(ident, f16, ident), logits=(ident, f16, f16)),
}


def main():
    n_new = 32
    dims = pdims.dims_for("large-v3")
    audio = synth.speech_like(30.0, seed=1234)
    mel = torch.from_numpy(logmel.window_features(audio, 128, "fw")[None])
    toks = pdims.special_tokens(dims.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    cfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    out = {"what": __doc__.split("\n\n")[0], "cases": []}
    for exact in ("float16", "none"):
        w = pweights.synth_weights(dims, seed=1234, exact=exact)
        ref = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), w)
        with torch.no_grad():
            enc = ref.encode(mel)
            res = decoding.greedy_decode(ref, enc, prompt, n_new, cfg)
            seq = torch.tensor([prompt + res.tokens[0]], dtype=torch.int64)
            lp_ref = torch.log_softmax(ref.decoder_logits(seq, enc)[0], dim=-1)
            P = len(prompt)
            idx = seq[0, P:]
            pos = torch.arange(P - 1, P - 1 + len(idx))
            for name, flags in CASES.items():
                t0 = time.time()
                em = Ablated(helpers.oracle_dims(dims), w, flags)
                enc_e = em.encode(mel)
                lp_e = torch.log_softmax(em.decoder_logits(seq, enc_e)[0], dim=-1)
                d = (lp_e[pos, idx] - lp_ref[pos, idx]).abs()
                case = {"weights": exact, "case": name, "enc_max_abs": float((enc_e - enc).abs().max()),
                        "token_logprob_max_abs": float(d.max()), "token_logprob_mean_abs": float(d.mean()),
                        "seconds": round(time.time() - t0, 1)}
                out["cases"].append(case)
                print(json.dumps(case), flush=True)
    with open(os.path.join(ROOT, "profiles", "r02_precision_ablation_cpu.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
