"""CPU study (oracle only): what would fp8 (OCP e4m3, per-(window, head) scale) cross-attention K/V cost in parity?

The decode cross-attention streams 245.8 MB of fp16 K/V per window per step and runs at the HBM ceiling, so halving those
bytes is the one remaining lever on the dominant kernel.  This runs the fp32 oracle with ONLY the
cross K/V quantised to fp8 and reports the teacher-forced per-token log-prob error on the large-v3 geometry -- the
number that decides whether it may ship as a default.  Writes profiles/r02_precision_fp8_kv_cpu.json; with
``--speechlike`` (round 3) the weights are ``weights.SPEECHLIKE`` -- PEAKED cross-attention, as in a trained model, and
searches that end -- on a 6 s clip, and the file is profiles/r03_precision_fp8_kv_speechlike_cpu.json.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decoding, logmel, whisper_ref  # noqa: E402
from tests import helpers  # noqa: E402
from whisperjav_amd import dims as pdims, synth, weights as pweights  # noqa: E402


def fp8(x, n_head):
    """[B, T, D] -> per (B, head) scale to the e4m3 range (max 448), round to e4m3, back to fp32."""
    B, T, D = x.shape
    xh = x.view(B, T, n_head, D // n_head)
    scale = xh.abs().amax(dim=(1, 3), keepdim=True).clamp_min(1e-12) / 448.0
    q = (xh / scale).to(torch.float8_e4m3fn).to(torch.float32) * scale
    return q.view(B, T, D)


class Fp8KV(whisper_ref.WhisperOracle):
    which = "kv"

    def cross_kv(self, xa):
        out = []
        for k, v in super().cross_kv(xa):
            H = self.dims.n_text_head
            out.append((fp8(k, H) if "k" in self.which else k, fp8(v, H) if "v" in self.which else v))
        return out


def main():
    speechlike = "--speechlike" in sys.argv
    dims = pdims.dims_for("large-v3")
    audio = synth.speech_like(6.0 if speechlike else 30.0, seed=1234)
    mel = torch.from_numpy(logmel.window_features(audio, 128, "fw")[None])
    toks = pdims.special_tokens(dims.n_vocab)
    prompt = [toks.sot, toks.language_token(pdims.language_index("ja")), toks.transcribe]
    suppress = (1, 2, 7, 8, 9, 10, 14, 25, toks.sot, toks.translate, toks.transcribe, toks.sot_lm, toks.sot_prev, toks.no_speech)
    cfg = decoding.FilterConfig(suppress_tokens=suppress, max_initial_timestamp_index=50)
    w = pweights.synth_weights(dims, seed=1234, exact="float16", **(pweights.SPEECHLIKE if speechlike else {}))
    ref = whisper_ref.WhisperOracle(helpers.oracle_dims(dims), w)
    out = {"what": __doc__.split("\n\n")[0], "weights": "SPEECHLIKE (peaked cross-attention)" if speechlike else "plain", "cases": []}
    with torch.no_grad():
        enc = ref.encode(mel)
        res = decoding.greedy_decode(ref, enc, prompt, 32, cfg)
        seq = torch.tensor([prompt + res.tokens[0]], dtype=torch.int64)
        lp_ref = torch.log_softmax(ref.decoder_logits(seq, enc)[0], dim=-1)
        P = len(prompt)
        idx = seq[0, P:]
        pos = torch.arange(P - 1, P - 1 + len(idx))
        for which in ("kv", "k", "v"):
            em = Fp8KV(helpers.oracle_dims(dims), w)
            em.which = which
            lp_e = torch.log_softmax(em.decoder_logits(seq, enc)[0], dim=-1)
            d = (lp_e[pos, idx] - lp_ref[pos, idx]).abs()
            flips = int((lp_e[pos].argmax(-1) != lp_ref[pos].argmax(-1)).sum())
            case = {"fp8_tensors": which, "token_logprob_max_abs": float(d.max()), "token_logprob_mean_abs": float(d.mean()),
                    "argmax_flips": flips, "tokens": int(len(idx))}
            out["cases"].append(case)
            print(json.dumps(case), flush=True)
    name = "r03_precision_fp8_kv_speechlike_cpu.json" if speechlike else "r02_precision_fp8_kv_cpu.json"
    with open(os.path.join(ROOT, "profiles", name), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
