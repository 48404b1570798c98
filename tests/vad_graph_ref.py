"""NumPy executor of a ``whisperjav_amd.vad_graph.Program`` (TEST INFRASTRUCTURE: never imported by the product).

Runs the lowered instruction stream the way csrc/vadgraph.hip does -- one arena and one exchange area per window, state slots
per stream; the arena is POISONED (NaN) at every stage boundary, as the LDS of the fused stage kernels forgets it -- so the
LOWERING (graph walk, constant folding, strides, state loop) can be pinned against ``torch.jit`` on the CPU without a GPU.  The
HIP kernels are pinned against the archive itself in the ``-m gpu`` tests.
"""
from __future__ import annotations

from typing import List

import numpy as np

from whisperjav_amd import vad_graph as vg


def _view(spaces, words, pos):
    space, off = words[pos], words[pos + 1]
    shape = tuple(words[pos + 2: pos + 2 + vg.MAX_DIMS])
    strides = tuple(words[pos + 2 + vg.MAX_DIMS: pos + 2 + 2 * vg.MAX_DIMS])
    base = spaces[space]
    return np.lib.stride_tricks.as_strided(base[off:], shape=shape, strides=tuple(4 * s for s in strides), writeable=(space != vg.SPACE_CONST))


def _w2f(w: int) -> float:
    return float(np.int32(w).view(np.float32))


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


_EW = {v: k for k, v in vg.EW.items()}


def run_window(p: vg.Program, consts: np.ndarray, state: np.ndarray, chunk: np.ndarray) -> float:
    arena = np.full(p.arena_floats + 8, np.nan, dtype=np.float32)
    xchg = np.full(p.xchg_floats + 8, np.nan, dtype=np.float32)
    spaces = {vg.SPACE_ARENA: arena, vg.SPACE_CONST: consts, vg.SPACE_STATE: state, vg.SPACE_XCHG: xchg}
    spaces[p.input_space][p.input_offset: p.input_offset + p.window] = chunk
    w = p.words
    pos = 0
    while pos < len(w):
        op, n = w[pos], w[pos + 1]
        a = pos + 2
        if op == vg.OP_EW:
            fn, nin, p0, p1 = _EW[w[a]], w[a + 1], _w2f(w[a + 2]), _w2f(w[a + 3])
            out = _view(spaces, w, a + 4)
            ins = [_view(spaces, w, a + 4 + vg.VIEW_WORDS * (1 + i)).astype(np.float32) for i in range(nin)]
            x = ins[0]
            f32 = np.float32
            if fn == "copy": r = x
            elif fn == "add": r = x + ins[1]
            elif fn == "sub": r = x - ins[1]
            elif fn == "mul": r = x * ins[1]
            elif fn == "div": r = x / ins[1]
            elif fn == "relu": r = np.maximum(x, 0)
            elif fn == "sigmoid": r = _sigmoid(x)
            elif fn == "tanh": r = np.tanh(x)
            elif fn == "exp": r = np.exp(x)
            elif fn == "log1p": r = np.log1p(x)
            elif fn == "log": r = np.log(x)
            elif fn == "sqrt": r = np.sqrt(x)
            elif fn == "abs": r = np.abs(x)
            elif fn == "neg": r = -x
            elif fn == "pow_scalar": r = np.power(x, f32(p0))
            elif fn == "add_scalar": r = x + f32(p0)
            elif fn == "mul_scalar": r = x * f32(p0)
            elif fn == "rsub_scalar": r = f32(p0) - x * f32(p1)
            elif fn == "fma": r = x * ins[1] + ins[2]
            elif fn == "clamp": r = np.clip(x, f32(p0), f32(p1))
            elif fn == "leaky_relu": r = np.where(x > 0, x, x * f32(p0))
            elif fn == "silu": r = x * _sigmoid(x)
            elif fn == "hardtanh": r = np.clip(x, f32(p0), f32(p1))
            else: raise AssertionError(fn)
            out[...] = r.astype(np.float32)
        elif op == vg.OP_CONV1D:
            out = _view(spaces, w, a)
            x = _view(spaces, w, a + vg.VIEW_WORDS)
            woff, boff, cout, cin, k, t, tout, stride, padding, dil, groups = w[a + 2 * vg.VIEW_WORDS: a + 2 * vg.VIEW_WORDS + 11]
            cg = cin // groups
            wt = consts[woff: woff + cout * cg * k].reshape(cout, cg, k)
            xin = np.zeros((cin, t + 2 * padding), dtype=np.float32)
            xin[:, padding: padding + t] = x[0, 0] if x.shape[0] == 1 and len(x.shape) == 4 else x.reshape(cin, t)
            og = cout // groups
            res = np.zeros((cout, tout), dtype=np.float32)
            span = (tout - 1) * stride + 1
            for g in range(groups):
                xg = xin[g * cg: (g + 1) * cg]
                unf = np.stack([xg[:, kk * dil: kk * dil + span: stride] for kk in range(k)], axis=1)        # [cg][k][tout]
                res[g * og: (g + 1) * og] = np.einsum("ock,ckt->ot", wt[g * og: (g + 1) * og], unf, optimize=True)
            if boff >= 0:
                res += consts[boff: boff + cout, None]
            out[...] = res.reshape(out.shape)
        elif op == vg.OP_PAD:
            out = _view(spaces, w, a)
            x = _view(spaces, w, a + vg.VIEW_WORDS)
            left, right, mode, val = w[a + 2 * vg.VIEW_WORDS], w[a + 2 * vg.VIEW_WORDS + 1], w[a + 2 * vg.VIEW_WORDS + 2], _w2f(w[a + 2 * vg.VIEW_WORDS + 3])
            pads = [(0, 0)] * (x.ndim - 1) + [(left, right)]
            out[...] = np.pad(x, pads, mode=("constant", "reflect", "edge")[mode], **({"constant_values": val} if mode == 0 else {}))
        elif op == vg.OP_MEAN:
            out = _view(spaces, w, a)
            x = _view(spaces, w, a + vg.VIEW_WORDS)
            r, rstride, inv = w[a + 2 * vg.VIEW_WORDS], w[a + 2 * vg.VIEW_WORDS + 1], _w2f(w[a + 2 * vg.VIEW_WORDS + 2])
            acc = np.zeros(out.shape, dtype=np.float32)
            base_off = w[a + vg.VIEW_WORDS + 1]
            space = spaces[w[a + vg.VIEW_WORDS]]
            for i in range(r):
                xi = np.lib.stride_tricks.as_strided(space[base_off + i * rstride:], shape=out.shape,
                                                     strides=tuple(4 * s for s in w[a + vg.VIEW_WORDS + 2 + vg.MAX_DIMS: a + 2 * vg.VIEW_WORDS]))
                acc = acc + xi
            out[...] = acc * np.float32(inv)
        elif op == vg.OP_LINEAR:
            ospace, ooff, xspace, xoff, woff, boff, rows, nin, nout = w[a: a + 9]
            x = spaces[xspace][xoff: xoff + rows * nin].reshape(rows, nin)
            wt = consts[woff: woff + nout * nin].reshape(nout, nin)
            y = x @ wt.T + (consts[boff: boff + nout] if boff >= 0 else 0.0)
            spaces[ospace][ooff: ooff + rows * nout] = y.reshape(-1).astype(np.float32)
        elif op == vg.OP_LSTM:
            yspace, yoff, xspace, xoff, st_t, st_f, t, nin, hid, layers, hnspace, hnoff, cnspace, cnoff = w[a: a + 14]
            assert yspace == hnspace == cnspace == vg.SPACE_XCHG and xspace in (vg.SPACE_XCHG, vg.SPACE_CONST)
            blobs = w[a + 14: a + 26]
            hslot, cslot, need_hc = w[a + 26], w[a + 27], w[a + 28]
            h = state[hslot: hslot + layers * hid].reshape(layers, hid).copy()
            c = state[cslot: cslot + layers * hid].reshape(layers, hid).copy()
            xs = np.lib.stride_tricks.as_strided(spaces[xspace][xoff:], shape=(t, nin), strides=(4 * st_t, 4 * st_f))
            ys = np.zeros((t, hid), dtype=np.float32)
            for step in range(t):
                inp = xs[step].astype(np.float32)
                for l in range(layers):
                    nin_l = nin if l == 0 else hid
                    wih = consts[blobs[3 * l]: blobs[3 * l] + nin_l * 4 * hid].reshape(nin_l, 4 * hid)
                    whh = consts[blobs[3 * l + 1]: blobs[3 * l + 1] + hid * 4 * hid].reshape(hid, 4 * hid)
                    b = consts[blobs[3 * l + 2]: blobs[3 * l + 2] + 4 * hid]
                    g = inp @ wih + h[l] @ whh + b
                    i_, f_, g_, o_ = g[:hid], g[hid: 2 * hid], g[2 * hid: 3 * hid], g[3 * hid:]
                    c[l] = _sigmoid(f_) * c[l] + _sigmoid(i_) * np.tanh(g_)
                    h[l] = _sigmoid(o_) * np.tanh(c[l])
                    inp = h[l]
                ys[step] = inp
            xchg[yoff: yoff + t * hid] = ys.reshape(-1)
            if need_hc:                              # the kernels write the window's final (h, c) only when an instruction reads them
                xchg[hnoff: hnoff + layers * hid] = h.reshape(-1)
                xchg[cnoff: cnoff + layers * hid] = c.reshape(-1)
            state[hslot: hslot + layers * hid] = h.reshape(-1)
            state[cslot: cslot + layers * hid] = c.reshape(-1)
            arena[:] = np.nan                       # a stage boundary: the next stage's LDS starts undefined
        else:
            raise AssertionError(f"opcode {op}")
        pos += n
    return float(spaces[p.output_space][p.output_offset])


def run_stream(p: vg.Program, audio: np.ndarray) -> np.ndarray:
    consts, state = p.const_blob(), p.state_blob().copy()
    out: List[float] = []
    for s in range(0, len(audio), p.window):
        chunk = np.zeros(p.window, dtype=np.float32)
        seg = audio[s: s + p.window]
        chunk[: len(seg)] = seg
        out.append(run_window(p, consts, state, chunk))
    return np.asarray(out, dtype=np.float32)
