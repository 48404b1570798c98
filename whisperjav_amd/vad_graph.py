"""TorchScript VAD archives on the device: a graph loader that LOWERS a scripted window scorer onto HIP kernels.

The reference's balanced default segmenter is ``silero-v3.1`` (/root/reference/whisperjav/main.py:1867-1876): the TorchScript
archive ``torch.hub.load("snakers4/silero-vad:v3.1", "silero_vad", onnx=False)`` returns (loader
modules/speech_segmentation/backends/silero.py:197-206), called as ``model(chunk, 16000)`` on consecutive 1536-sample windows
with the recurrent state kept inside the module between calls (``:258-273`` through the archive's ``get_speech_timestamps``).
That network is a different graph from the v5/v6 one ``csrc/vad.hip`` hard-codes, and the archive itself is not obtainable
offline -- so instead of hard-coding a second network this module reads WHATEVER the archive's graph says and runs it:

  1. ``lower(module, window, sr)`` walks the archive's inlined ``forward`` graph by abstract interpretation at a fixed
     window shape.  Non-tensor values (ints, lists, bools, strings, module attributes) are evaluated, so every ``prim::If`` /
     ``prim::Loop`` is decided at load time -- sampling-rate branches, dimension checks, "first call" state initialisation;
     tensors derived from the input are symbolic strided views over a per-window arena; parameters and buffers are constants
     (constant sub-expressions are folded with torch at load time, e.g. BatchNorm statistics into a per-channel scale/shift).
     Each tensor op on a symbolic operand becomes one instruction of a small program: strided element-wise ops, conv1d
     (groups / stride / dilation / zero padding), reflect / constant padding, mean over an axis, linear, multi-layer LSTM.
     In-place ops write through the operand's own view (aliases see the write; an in-place op on a view that overlaps another
     operand differently raises).
     Module attributes that the graph WRITES (``prim::SetAttr``: the LSTM's h / c) become per-stream state: the graph is
     walked twice -- the first call from the state ``reset_states()`` leaves, then a steady-state call -- and both walks must
     produce the same instructions (the first one only differs in reading constants where the second reads state).
     Anything else -- an op outside the table, state that flows through something other than the LSTM, data-dependent control
     flow -- raises ``LoweringError`` naming the op and its source line.  Nothing is approximated and nothing runs on the CPU.
  2. Layout (``_Lowerer._layout``): the LSTM instructions cut the program into STAGES.  A tensor that lives inside one stage
     goes to the per-window ARENA -- laid out by liveness (first-fit over the instruction intervals, element-wise results take
     over the slot of an operand that dies there), so the arena of the silero-shaped graphs is ~50 KB and fits the LDS of a
     compute unit; a tensor that crosses a stage boundary or that an LSTM touches goes to the per-window EXCHANGE area in HBM
     (a few KB).
  3. ``HipGraphVadScorer`` hands the program to ``wj_vadg_create`` (csrc/vadgraph.hip) and scores every stream (scene) of a
     call at once: each stage is ONE launch, one workgroup per window with the arena in LDS, running the stage's instructions
     back to back; the LSTM runs one workgroup per stream sequentially over that stream's windows with its weights in
     registers.  Arenas that do not fit the LDS fall back to one launch per instruction over an arena in HBM.

Program encoding (int32 words; floats as their bit patterns) is described next to ``OPCODES`` and mirrored in
csrc/vadgraph.hip; ``tests/vad_graph_ref.py`` holds a NumPy executor of the same program (test infrastructure) so the lowering
is pinned against ``torch.jit`` on the CPU and the kernels against the same archive on the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

MAX_DIMS = 4


class LoweringError(RuntimeError):
    """The archive's graph uses something this loader does not lower (message: op, source line)."""


# ---- program ---------------------------------------------------------------------------------------------------------------
# Every instruction: [opcode, n_words, ...].  A tensor operand ("view") is 2 + 2 * MAX_DIMS words:
#   [space (0 arena / 1 constants / 2 state / 3 exchange), offset in floats, shape[4] (leading 1-padded), strides[4] in floats]
OP_EW, OP_CONV1D, OP_PAD, OP_MEAN, OP_LINEAR, OP_LSTM = 1, 2, 3, 4, 5, 6
OPCODES = {"ew": OP_EW, "conv1d": OP_CONV1D, "pad": OP_PAD, "mean": OP_MEAN, "linear": OP_LINEAR, "lstm": OP_LSTM}
# element-wise function codes (a, b, c = operands; p0, p1 = float parameters)
EW = {"copy": 0, "add": 1, "sub": 2, "mul": 3, "div": 4, "relu": 5, "sigmoid": 6, "tanh": 7, "exp": 8, "log1p": 9, "sqrt": 10,
      "abs": 11, "neg": 12, "pow_scalar": 13, "add_scalar": 14, "mul_scalar": 15, "fma": 16, "clamp": 17, "leaky_relu": 18,
      "log": 19, "rsub_scalar": 20, "silu": 21, "hardtanh": 22}
SPACE_ARENA, SPACE_CONST, SPACE_STATE, SPACE_XCHG = 0, 1, 2, 3
CONV_SCRATCH_MAX = 8192          # floats: conv1d kernels up to this size are staged in the arena by the fused executor
VIEW_WORDS = 2 + 2 * MAX_DIMS


def _f2w(x: float) -> int:
    return int(np.float32(x).view(np.int32))


@dataclass(frozen=True)
class Sym:
    """A strided float32 view: of a per-window buffer (``space`` 0, ``buf`` = buffer id, ``offset`` relative to the buffer: the
    layout pass decides arena / exchange and the base), of the constants, or of a per-stream state slot (``space`` 2,
    ``offset`` = slot base)."""
    space: int
    offset: int
    shape: Tuple[int, ...]
    strides: Tuple[int, ...]
    buf: int = -1

    def like(self, offset: int, shape: Sequence[int], strides: Sequence[int]) -> "Sym":
        return Sym(self.space, int(offset), tuple(int(d) for d in shape), tuple(int(x) for x in strides), self.buf)

    @property
    def numel(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    def is_contiguous(self) -> bool:
        return self.strides == _contig(self.shape)


def _contig(shape: Sequence[int]) -> Tuple[int, ...]:
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= int(d)
    return tuple(reversed(st))


@dataclass
class Program:
    window: int
    sample_rate: int
    arena_floats: int = 0
    consts: List[np.ndarray] = field(default_factory=list)
    const_floats: int = 0
    words: List[int] = field(default_factory=list)
    n_instr: int = 0
    state_floats: int = 0
    state_init: List[Tuple[int, np.ndarray]] = field(default_factory=list)    # (slot offset, initial value)
    input_offset: int = 0
    output_offset: int = 0
    input_space: int = SPACE_ARENA
    output_space: int = SPACE_ARENA
    xchg_floats: int = 0
    unpacked_arena_floats: int = 0            # what one buffer per tensor would take (round 5's layout): reported, not used
    listing: List[str] = field(default_factory=list)

    def const_blob(self) -> np.ndarray:
        return np.concatenate([c.reshape(-1) for c in self.consts]).astype(np.float32) if self.consts else np.zeros(1, np.float32)

    def state_blob(self) -> np.ndarray:
        out = np.zeros(max(1, self.state_floats), dtype=np.float32)
        for off, val in self.state_init:
            out[off: off + val.size] = val.reshape(-1)
        return out

    def signature(self) -> List[Tuple[int, ...]]:
        """Instruction list with operand SPACES and OFFSETS of state / constant reads masked: what the two walks must share."""
        return self._sig


# ---- the abstract interpreter ------------------------------------------------------------------------------------------------
_SCALAR_OPS = {
    "aten::eq": lambda a, b: a == b, "aten::ne": lambda a, b: a != b, "aten::gt": lambda a, b: a > b, "aten::lt": lambda a, b: a < b,
    "aten::ge": lambda a, b: a >= b, "aten::le": lambda a, b: a <= b, "aten::__not__": lambda a: not a,
    "aten::__and__": lambda a, b: a and b, "aten::__or__": lambda a, b: a or b, "aten::__contains__": lambda a, b: b in a,
    "aten::add": lambda a, b, alpha=1: a + b * alpha if alpha != 1 else a + b, "aten::sub": lambda a, b, alpha=1: a - b * alpha if alpha != 1 else a - b,
    "aten::mul": lambda a, b: a * b,
    "aten::div": lambda a, b: a / b, "aten::floordiv": lambda a, b: a // b, "aten::remainder": lambda a, b: a % b,
    "aten::neg": lambda a: -a, "aten::len": lambda a: len(a), "aten::__getitem__": lambda a, i: a[i], "aten::Int": lambda a: int(a),
    "aten::Float": lambda a: float(a), "aten::Bool": lambda a: bool(a), "aten::list": lambda a: list(a), "aten::__is__": lambda a, b: a is b,
    "aten::__isnot__": lambda a, b: a is not b, "aten::pow": lambda a, b: a ** b, "aten::sqrt": lambda a: math.sqrt(a),
    "aten::floor": lambda a: math.floor(a), "aten::ceil": lambda a: math.ceil(a), "aten::abs": lambda a: abs(a),
    "aten::min": lambda a, b: min(a, b), "aten::max": lambda a, b: max(a, b), "aten::str": lambda a: str(a),
    "prim::min": lambda *a: min(*a), "prim::max": lambda *a: max(*a), "aten::ScalarImplicit": lambda a: a.item(), "aten::IntImplicit": lambda a: int(a.item()),
    "aten::FloatImplicit": lambda a: float(a.item()),
}
_IDENTITY_OPS = {"aten::to", "aten::float", "aten::detach", "aten::clone", "aten::contiguous", "aten::dropout", "aten::dropout_",
                 "aten::feature_dropout", "aten::alpha_dropout", "aten::type_as", "aten::requires_grad_"}
_UNARY = {"aten::relu": "relu", "aten::sigmoid": "sigmoid", "aten::tanh": "tanh", "aten::exp": "exp", "aten::log1p": "log1p",
          "aten::sqrt": "sqrt", "aten::abs": "abs", "aten::neg": "neg", "aten::log": "log", "aten::silu": "silu"}
_BINARY = {"aten::add": "add", "aten::sub": "sub", "aten::mul": "mul", "aten::div": "div"}
_INPLACE_OK = {*_UNARY, *_BINARY, "aten::square", "aten::rsqrt", "aten::reciprocal", "aten::leaky_relu", "aten::hardtanh", "aten::clamp",
               "aten::clamp_min", "aten::clamp_max", "aten::pow"}


@dataclass
class _V:
    """A tensor operand of an instruction before the layout pass: expands to the VIEW_WORDS words of a view."""
    sym: Sym
    shape: Tuple[int, ...]
    strides: Tuple[int, ...]


@dataclass
class _O:
    """A contiguous operand named by (space, offset) only (linear / lstm)."""
    sym: Sym


@dataclass
class _Ins:
    name: str
    payload: List[Any]
    reads: List[int]
    writes: List[int]
    inplace_like: Optional[Sym]      # element-wise ops with a fresh result: the result view (it may take over a dying operand's slot)


def _extent(s: Sym) -> Tuple[int, int]:
    return s.offset, s.offset + sum((d - 1) * st for d, st in zip(s.shape, s.strides)) + 1


def _overlap(a: Sym, b: Sym) -> bool:
    """Conservative: the address ranges of two views of one buffer intersect."""
    (a0, a1), (b0, b1) = _extent(a), _extent(b)
    return a0 < b1 and b0 < a1


def _self_overlap(s: Sym) -> bool:
    """Two index tuples of the view may name one element (sorted strides must step over the extent below them)."""
    span = 1
    for st, d in sorted((st, d) for st, d in zip(s.strides, s.shape) if d > 1):
        if st < span:
            return True
        span = st * (d - 1) + span
    return False


class _Lowerer:
    def __init__(self, module: torch.jit.ScriptModule, window: int, sample_rate: int):
        self.module, self.window, self.sr = module, int(window), int(sample_rate)
        g = module.forward.graph.copy()
        torch._C._jit_pass_inline(g)
        self.graph = g
        self.overlay: Dict[Tuple[int, str], Any] = {}         # attributes written by SetAttr during the walks (state and bookkeeping ints)
        self.state_slots: Dict[Tuple[int, str], Tuple[int, Tuple[int, ...]]] = {}   # attr -> (slot offset, shape), fixed after the first walk

    # -- program assembly (one walk) --
    def _begin(self) -> None:
        self.prog = Program(self.window, self.sr)
        self.prog._sig = []
        self.buf_floats: List[int] = []            # per-window buffers of this walk (id -> size in floats)
        self.instrs: List[_Ins] = []
        self.const_index: Dict[int, Tuple[int, np.ndarray]] = {}

    def alloc(self, shape: Sequence[int]) -> Sym:
        shape = tuple(int(d) for d in shape)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        self.buf_floats.append((n + 3) // 4 * 4)          # 16-byte aligned buffers
        return Sym(SPACE_ARENA, 0, shape, _contig(shape), len(self.buf_floats) - 1)

    def const(self, t: Union[torch.Tensor, np.ndarray]) -> Sym:
        a = np.ascontiguousarray(t.detach().cpu().float().numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float32)
        key = id(t)
        hit = self.const_index.get(key)
        if hit is None or hit[1].shape != a.shape or not np.array_equal(hit[1], a):
            off = self.prog.const_floats
            self.prog.consts.append(a)
            self.prog.const_floats += (a.size + 3) // 4 * 4
            if a.size % 4:
                self.prog.consts.append(np.zeros(4 - a.size % 4, np.float32))
            self.const_index[key] = (off, a)
            hit = (off, a)
        return Sym(SPACE_CONST, hit[0], tuple(a.shape), _contig(a.shape))

    def operand(self, v: Any, state_ok: bool = False) -> Sym:
        if isinstance(v, Sym):
            if v.space == SPACE_STATE and not state_ok:      # state changes window by window: only the sequential LSTM kernel may read it
                raise LoweringError("module state is read by an op other than the LSTM's initial state")
            return v
        if isinstance(v, torch.Tensor):
            return self.const(v)
        raise LoweringError(f"a tensor operand was expected, got {type(v).__name__}")

    @staticmethod
    def _view(s: Sym, shape: Optional[Sequence[int]] = None, strides: Optional[Sequence[int]] = None) -> "_V":
        shape = tuple(s.shape if shape is None else shape)
        strides = tuple(s.strides if strides is None else strides)
        if len(shape) > MAX_DIMS:
            raise LoweringError(f"a {len(shape)}-d tensor (this loader handles up to {MAX_DIMS} dimensions)")
        pad = MAX_DIMS - len(shape)
        return _V(s, (*([1] * pad), *[int(d) for d in shape]), (*([0] * pad), *[int(x) for x in strides]))

    def emit(self, name: str, payload: List[Any], text: str, sig: Tuple, reads: Sequence[Sym], writes: Sequence[Sym],
             inplace_like: Optional[Sym] = None) -> None:
        self.instrs.append(_Ins(name, payload, [r.buf for r in reads if r.buf >= 0], [w.buf for w in writes if w.buf >= 0], inplace_like))
        self.prog.n_instr += 1
        self.prog.listing.append(text)
        self.prog._sig.append((name, *sig))

    # -- tensor ops --
    def ew(self, fn: str, ins: Sequence[Any], p0: float = 0.0, p1: float = 0.0, out: Optional[Sym] = None) -> Sym:
        ops = [self.operand(x) for x in ins]
        shape = tuple(np.broadcast_shapes(*[o.shape for o in ops])) if out is None else out.shape
        fresh = out is None
        if fresh:
            out = self.alloc(shape)
        words: List[Any] = [EW[fn], len(ops), _f2w(p0), _f2w(p1), self._view(out)]
        for o in ops:
            pad = len(shape) - len(o.shape)
            if pad < 0:
                raise LoweringError(f"{fn}: operand {o.shape} does not broadcast into {shape}")
            st = [0] * pad + [0 if (d == 1 and shape[pad + i] != 1) else o.strides[i] for i, d in enumerate(o.shape)]
            for i, d in enumerate(o.shape):
                if d != 1 and d != shape[pad + i]:
                    raise LoweringError(f"{fn}: operand {o.shape} does not broadcast into {shape}")
            words.append(self._view(o, shape, st))
            if not fresh and o.buf == out.buf and o.buf >= 0 and _overlap(o, out) and (o.offset, tuple(st)) != (out.offset, tuple(out.strides)):
                # every thread reads its own element and then writes it: only the identical view may be read and written at once
                raise LoweringError(f"{fn}: the result is written over an operand through a different view of the same tensor")
        for _ in range(3 - len(ops)):
            words += [0] * VIEW_WORDS
        self.emit("ew", words, f"ew.{fn} {[o.shape for o in ops]} -> {shape}", (fn, shape, tuple(o.shape for o in ops), float(p0), float(p1)),
                  ops, [out], inplace_like=out if fresh else None)
        return out

    def ew_inplace(self, fn: str, x: Any, others: Sequence[Any] = (), p0: float = 0.0, p1: float = 0.0) -> Sym:
        """``x.op_(...)``: the result goes through x's OWN view, so the base tensor and every other view of it see the write."""
        x = self.operand(x)
        if x.space != SPACE_ARENA or x.buf < 0:
            raise LoweringError(f"{fn}_: an in-place op on a constant or on module state")
        if any(s == 0 and d > 1 for s, d in zip(x.strides, x.shape)) or _self_overlap(x):
            raise LoweringError(f"{fn}_: an in-place op on an expanded / self-overlapping view")
        return self.ew(fn, [x, *others], p0, p1, out=x)

    def materialise(self, s: Sym) -> Sym:
        s = self.operand(s)
        return s if (s.is_contiguous() and s.space == SPACE_ARENA) else self.ew("copy", [s])

    def conv1d(self, x: Any, w: torch.Tensor, b: Optional[torch.Tensor], stride: int, padding: int, dilation: int, groups: int) -> Sym:
        x = self.operand(x)
        if not isinstance(w, torch.Tensor) or (b is not None and not isinstance(b, torch.Tensor)):
            raise LoweringError("conv1d with a weight or bias that depends on the input")
        if len(x.shape) == 2:
            x = x.like(x.offset, (1, *x.shape), (0, *x.strides))
        if len(x.shape) != 3 or x.shape[0] != 1:
            raise LoweringError(f"conv1d input {x.shape}: one window per call ([1, C, T]) expected")
        cout, cin_g, k = (int(d) for d in w.shape)
        cin, t = x.shape[1], x.shape[2]
        if cin != cin_g * groups or cout % groups:
            raise LoweringError(f"conv1d: input channels {cin}, weight {tuple(w.shape)}, groups {groups}")
        tout = (t + 2 * padding - dilation * (k - 1) - 1) // stride + 1
        if tout < 1:
            raise LoweringError(f"conv1d: {t} input positions are too few for kernel {k}")
        out = self.alloc((1, cout, tout))
        wc = self.const(w)
        bc = self.const(b) if b is not None else None
        # small kernels get a scratch tensor that lives for this instruction only: the fused executor stages the weights (rows padded
        # to an odd stride: conflict-free LDS banks) and the bias there with one coalesced sweep instead of chasing them through L2
        # from inside every accumulation chain
        row = cin_g * k
        ws_stride = row | 1
        ws = self.alloc((cout * ws_stride + cout,)) if cout * ws_stride + cout <= CONV_SCRATCH_MAX else None
        words = [self._view(out), self._view(x), wc.offset, bc.offset if bc is not None else -1, cout, cin, k, t, tout, stride, padding,
                 dilation, groups, *([_O(ws), ws_stride] if ws is not None else [-1, -1, 0])]
        self.emit("conv1d", words, f"conv1d {x.shape} * {tuple(w.shape)} s{stride} p{padding} d{dilation} g{groups} -> {out.shape}",
                  (x.shape, tuple(w.shape), b is not None, stride, padding, dilation, groups), [x], [out] + ([ws] if ws is not None else []))
        return out

    def pad_last(self, x: Any, left: int, right: int, mode: str, value: float) -> Sym:
        x = self.operand(x)
        if mode not in ("reflect", "constant", "replicate"):
            raise LoweringError(f"padding mode {mode!r}")
        t = x.shape[-1]
        if mode == "reflect" and (left >= t or right >= t):
            raise LoweringError(f"reflect padding ({left}, {right}) of {t} positions")
        out = self.alloc((*x.shape[:-1], t + left + right))
        words = [self._view(out), self._view(x), left, right, {"constant": 0, "reflect": 1, "replicate": 2}[mode], _f2w(value)]
        self.emit("pad", words, f"pad.{mode} {x.shape} ({left}, {right}) -> {out.shape}", (x.shape, left, right, mode, float(value)), [x], [out])
        return out

    def mean(self, x: Any, dims: Sequence[int], keepdim: bool) -> Sym:
        x = self.operand(x)
        dims = sorted(d % len(x.shape) for d in dims)
        cur = x
        for d in reversed(dims):                       # one axis per instruction (sequential fp32 sums; torch reduces over all at once)
            shape = list(cur.shape)
            r = shape[d]
            shape[d] = 1
            out = self.alloc(shape)
            keep_strides = list(cur.strides)
            rstride = keep_strides[d]
            keep_strides[d] = 0
            words = [self._view(out), self._view(cur, shape, keep_strides), r, rstride, _f2w(1.0 / r)]
            self.emit("mean", words, f"mean {cur.shape} over axis {d} -> {tuple(shape)}", (cur.shape, d), [cur], [out])
            cur = out
        if not keepdim:
            shape = [s for i, s in enumerate(cur.shape) if i not in dims]
            cur = cur.like(cur.offset, shape, _contig(shape))
        return cur

    def linear(self, x: Any, w: torch.Tensor, b: Optional[torch.Tensor]) -> Sym:
        x = self.materialise(x)
        if not isinstance(w, torch.Tensor) or w.dim() != 2 or (b is not None and not isinstance(b, torch.Tensor)):
            raise LoweringError("linear with a weight or bias that depends on the input")
        nout, nin = (int(d) for d in w.shape)
        if x.shape[-1] != nin:
            raise LoweringError(f"linear: input {x.shape}, weight {tuple(w.shape)}")
        rows = x.numel // nin
        out = self.alloc((*x.shape[:-1], nout))
        wc, bc = self.const(w), (self.const(b) if b is not None else None)
        self.emit("linear", [_O(out), _O(x), wc.offset, bc.offset if bc is not None else -1, rows, nin, nout],
                  f"linear {x.shape} * {tuple(w.shape)} -> {out.shape}", (x.shape, tuple(w.shape), b is not None), [x], [out])
        return out

    def lstm(self, x: Any, hx: Sequence[Any], params: Sequence[Any], has_biases: bool, num_layers: int, train: bool, bidirectional: bool,
             batch_first: bool) -> Tuple[Sym, Sym, Sym]:
        if train or bidirectional:
            raise LoweringError("an LSTM in training mode or a bidirectional one")
        if any(not isinstance(p, torch.Tensor) for p in params):
            raise LoweringError("LSTM parameters that depend on the input")
        if self._pending_lstm is not None:
            raise LoweringError("a second LSTM (one recurrent block per archive is lowered)")
        x = self.operand(x)
        if len(x.shape) != 3 or x.shape[0 if batch_first else 1] != 1:
            raise LoweringError(f"LSTM input {x.shape}: one window per call (batch 1) expected")
        t, nin = (x.shape[1], x.shape[2]) if batch_first else (x.shape[0], x.shape[2])
        st_t, st_f = (x.strides[1], x.strides[2]) if batch_first else (x.strides[0], x.strides[2])
        per = 4 if has_biases else 2
        if len(params) != per * num_layers:
            raise LoweringError(f"LSTM with {len(params)} parameter tensors for {num_layers} layers (projections are not lowered)")
        hidden = int(params[1].shape[1])
        if hidden > 128 or nin > 512 or num_layers > 4:
            raise LoweringError(f"LSTM geometry (input {nin}, hidden {hidden}, layers {num_layers}) outside the kernel's limits (512 / 128 / 4)")
        blobs: List[int] = []
        for l in range(num_layers):
            w_ih, w_hh = params[per * l], params[per * l + 1]
            nin_l = nin if l == 0 else hidden
            if tuple(w_ih.shape) != (4 * hidden, nin_l) or tuple(w_hh.shape) != (4 * hidden, hidden):
                raise LoweringError(f"LSTM layer {l}: weights {tuple(w_ih.shape)} / {tuple(w_hh.shape)}")
            bias = (params[per * l + 2] + params[per * l + 3]) if has_biases else torch.zeros(4 * hidden)
            # input-major copies: the lanes of a wavefront are gate rows and read consecutive addresses
            blobs += [self.const(w_ih.t().contiguous()).offset, self.const(w_hh.t().contiguous()).offset, self.const(bias).offset]
        h0, c0 = hx
        want = (num_layers, 1, hidden)
        src = []
        for name, s in (("h", h0), ("c", c0)):
            if isinstance(s, torch.Tensor):
                if tuple(s.shape) != want:
                    raise LoweringError(f"LSTM initial {name} of shape {tuple(s.shape)}, {want} expected")
                src.append(("const", s))
            else:
                s = self.operand(s, state_ok=True)
                if s.space != SPACE_STATE or s.shape != want or not s.is_contiguous():
                    raise LoweringError(f"the LSTM's initial {name} is computed from the input (only module state or constants are lowered)")
                src.append(("state", s))
        y = self.alloc((1, t, hidden) if batch_first else (t, 1, hidden))
        hn, cn = self.alloc(want), self.alloc(want)
        self._pending_lstm = (src, hn, cn)
        # the two state slot offsets are patched in _finish() (they are assigned when the SetAttr that closes the loop is seen), the
        # last word -- does any instruction read the window's final (h, c) -- in _layout()
        words = [_O(y), _O(x), st_t, st_f, t, nin, hidden, num_layers, _O(hn), _O(cn), *blobs, *([0] * (3 * (4 - num_layers))), -1, -1, 0]
        self.emit("lstm", words, f"lstm {x.shape} hidden {hidden} x {num_layers} layers -> {y.shape}", (x.shape, hidden, num_layers, batch_first),
                  [x], [y, hn, cn])
        self._lstm_instr = self.instrs[-1]
        return y, hn, cn

    # -- the walk --
    def run(self, state_from: Optional[Dict[Tuple[int, str], Any]]) -> Program:
        self._begin()
        self._pending_lstm = None
        x = self.alloc((self.window,))
        env: Dict[Any, Any] = {}
        ins = list(self.graph.inputs())
        env[ins[0]] = self.module
        env[ins[1]] = x
        if len(ins) > 2:
            env[ins[2]] = self.sr
        if len(ins) > 3:
            raise LoweringError(f"forward takes {len(ins) - 1} arguments; (chunk, sample_rate) expected")
        outs = self._block(self.graph, env)
        if len(outs) != 1 or not isinstance(outs[0], Sym):
            raise LoweringError("forward must return one tensor computed from the chunk")
        res = self.materialise(outs[0])
        if res.numel != 1:
            raise LoweringError(f"forward returns {res.shape} per window; one probability expected")
        self._finish()
        self._layout(x, res)
        return self.prog

    def _layout(self, x: Sym, res: Sym) -> None:
        """Spaces and offsets of the per-window buffers, then the program words (see the module docstring, step 2)."""
        n = len(self.instrs)
        stage, s = [], 0
        for ins in self.instrs:                                 # an LSTM is a stage boundary (stage -1: it touches the exchange area only)
            stage.append(-1 if ins.name == "lstm" else s)
            s += ins.name == "lstm"
        first: Dict[int, int] = {x.buf: -1}
        last: Dict[int, int] = {x.buf: -1}
        stages: Dict[int, set] = {x.buf: {0}}
        for i, ins in enumerate(self.instrs):
            for b in (*ins.reads, *ins.writes):
                first.setdefault(b, i)
                last[b] = i
                stages.setdefault(b, set()).add(stage[i])
        last[res.buf] = n
        stages[res.buf].add(s)                                  # the probability is read after the last stage
        space: Dict[int, int] = {b: (SPACE_XCHG if (-1 in st or len(st) > 1) else SPACE_ARENA) for b, st in stages.items()}
        base: Dict[int, int] = {}
        xchg = 0
        for b in sorted(space):
            if space[b] == SPACE_XCHG:
                base[b] = xchg
                xchg += self.buf_floats[b]
        # arena: first-fit over the live intervals, in instruction order
        free: List[List[int]] = []          # [offset, size], sorted by offset
        top = 0

        def take(size: int) -> int:
            nonlocal top
            for blk in free:
                if blk[1] >= size:
                    off = blk[0]
                    blk[0] += size
                    blk[1] -= size
                    if blk[1] == 0:
                        free.remove(blk)
                    return off
            if free and free[-1][0] + free[-1][1] == top:        # grow the last free block instead of leaving a hole
                off = free[-1][0]
                top = off + size
                free.pop()
                return off
            off = top
            top += size
            return off

        def give(off: int, size: int) -> None:
            free.append([off, size])
            free.sort()
            i = 0
            while i + 1 < len(free):
                if free[i][0] + free[i][1] == free[i + 1][0]:
                    free[i][1] += free[i + 1][1]
                    del free[i + 1]
                else:
                    i += 1

        owner: Dict[int, int] = {}           # buffer -> the buffer whose slot it took over (in-place element-wise results)
        arena_bufs = [b for b in space if space[b] == SPACE_ARENA]
        by_first: Dict[int, List[int]] = {}
        by_last: Dict[int, List[int]] = {}
        for b in arena_bufs:
            by_first.setdefault(first[b], []).append(b)
            by_last.setdefault(last[b], []).append(b)
        taken_over: set = set()
        for i in range(-1, n + 1):
            ins = self.instrs[i] if 0 <= i < n else None
            for b in by_first.get(i, []):
                donor = None
                out = ins.inplace_like if ins is not None else None
                if out is not None and out.buf == b and out.offset == 0 and out.is_contiguous() and out.numel == int(np.prod(out.shape)):
                    for v in ins.payload:
                        if isinstance(v, _V) and v.sym.buf >= 0 and v.sym.buf != b and space.get(v.sym.buf) == SPACE_ARENA and last[v.sym.buf] == i \
                                and v.sym.buf not in taken_over and self.buf_floats[v.sym.buf] == self.buf_floats[b] \
                                and all(u.sym.offset == 0 and u.strides == self._view(out).strides for u in ins.payload
                                        if isinstance(u, _V) and u.sym.buf == v.sym.buf):
                            donor = v.sym.buf
                            break
                if donor is not None:
                    base[b] = base[donor]
                    taken_over.add(donor)
                else:
                    base[b] = take(self.buf_floats[b])
            for b in by_last.get(i, []):
                if b not in taken_over:
                    give(base[b], self.buf_floats[b])
        p = self.prog
        p.arena_floats = max(4, top)
        p.xchg_floats = max(4, xchg)
        p.unpacked_arena_floats = sum(self.buf_floats)
        p.input_space, p.input_offset = space[x.buf], base[x.buf] + x.offset
        p.output_space, p.output_offset = space[res.buf], base[res.buf] + res.offset
        self.spaces, self.bases = space, base

        def where(sym: Sym) -> Tuple[int, int]:
            return (sym.space, sym.offset) if sym.buf < 0 else (space[sym.buf], base[sym.buf] + sym.offset)

        p.words = []
        for ins in self.instrs:
            if ins.name == "lstm":
                ins.payload[-1] = int(any(b in other.reads for other in self.instrs for b in ins.writes[1:]))
            payload: List[int] = []
            for v in ins.payload:
                if isinstance(v, _V):
                    payload += [*where(v.sym), *v.shape, *v.strides]
                elif isinstance(v, _O):
                    payload += [*where(v.sym)]
                else:
                    payload.append(int(v))
            p.words.extend([OPCODES[ins.name], 2 + len(payload), *payload])

    def _finish(self) -> None:
        """Close the state loop: every attribute that now holds a symbolic tensor must be an LSTM's (hn, cn)."""
        writes = {k: v for k, v in self.overlay.items() if isinstance(v, Sym)}
        if not writes:
            return
        if self._pending_lstm is None:
            raise LoweringError("module state is written, but not by an LSTM (only LSTM (h, c) state is lowered)")
        src, hn, cn = self._pending_lstm
        slots = []
        for (kind, val), out, name in ((src[0], hn, "h"), (src[1], cn, "c")):
            keys = [k for k, v in writes.items() if v == out]
            if len(keys) != 1:
                raise LoweringError(f"the LSTM's final {name} is stored in {len(keys)} module attributes (exactly one expected)")
            key = keys[0]
            if key not in self.state_slots:
                self.state_slots[key] = (sum(int(np.prod(s[1])) for s in self.state_slots.values()), out.shape)
            off, shape = self.state_slots[key]
            if kind == "state" and val.offset != off:
                raise LoweringError(f"the LSTM reads its initial {name} from one attribute and stores the final one in another")
            if kind == "const":
                self.prog.state_init.append((off, np.ascontiguousarray(val.detach().cpu().float().numpy()).reshape(-1)))
            slots.append(off)
        extra = [k for k, v in writes.items() if v not in (hn, cn)]
        if extra:
            raise LoweringError(f"module attribute {extra[0][1]!r} keeps a tensor that is not LSTM state")
        self._lstm_instr.payload[-3:-1] = slots
        self.prog.state_floats = sum(int(np.prod(s[1])) for s in self.state_slots.values())

    def _attr(self, obj: Any, name: str) -> Any:
        key = (id(obj), name)
        if key in self.overlay:            # written by a SetAttr of this or an earlier walk (state slots, bookkeeping ints)
            return self.overlay[key]
        try:
            return getattr(obj, name)
        except Exception as e:
            raise LoweringError(f"attribute {name!r}: {e}") from None

    def _block(self, block, env: Dict[Any, Any]) -> List[Any]:
        for n in block.nodes():
            self._node(n, env)
        return [env[v] for v in block.outputs()] if hasattr(block, "outputs") else []

    def _where(self, n) -> str:
        try:
            return n.sourceRange().strip().splitlines()[0][:160]
        except Exception:
            return "?"

    def _node(self, n, env: Dict[Any, Any]) -> None:
        kind = n.kind()
        ins = [env[v] for v in n.inputs()]
        outs = list(n.outputs())

        def put(*vals):
            if len(vals) != len(outs):
                raise LoweringError(f"{kind}: {len(vals)} results for {len(outs)} outputs ({self._where(n)})")
            for o, v in zip(outs, vals):
                env[o] = v

        if kind == "prim::Constant":
            return put(outs[0].toIValue() if n.hasAttributes() or str(outs[0].type()) == "NoneType" else None)
        if kind == "prim::GetAttr":
            return put(self._attr(ins[0], n.s("name")))
        if kind == "prim::SetAttr":
            self.overlay[(id(ins[0]), n.s("name"))] = ins[1]
            return None
        if kind == "prim::If":
            cond = ins[0]
            if isinstance(cond, torch.Tensor) and cond.numel() == 1:
                cond = bool(cond.item())
            if not isinstance(cond, (bool, int)):
                raise LoweringError(f"control flow that depends on the audio ({self._where(n)})")
            blocks = list(n.blocks())
            return put(*self._block(blocks[0] if cond else blocks[1], env))
        if kind == "prim::Loop":
            max_trip, cond, carried = ins[0], ins[1], list(ins[2:])
            if not isinstance(cond, (bool, int)) or not isinstance(max_trip, int):
                raise LoweringError(f"a loop whose trip count depends on the audio ({self._where(n)})")
            body = list(n.blocks())[0]
            bins = list(body.inputs())
            i = 0
            while cond and i < max_trip:
                if i > 100000:
                    raise LoweringError(f"a loop of more than 100000 iterations ({self._where(n)})")
                env[bins[0]] = i
                for bi, v in zip(bins[1:], carried):
                    env[bi] = v
                res = self._block(body, env)
                cond, carried = res[0], list(res[1:])
                if not isinstance(cond, (bool, int)):
                    raise LoweringError(f"a loop condition that depends on the audio ({self._where(n)})")
                i += 1
            return put(*carried)
        if kind == "prim::RaiseException":
            raise LoweringError(f"the archive raises for this window / sample rate: {ins[0]!r} ({self._where(n)})")
        if kind in ("prim::ListConstruct",):
            return put(list(ins))
        if kind in ("prim::TupleConstruct",):
            return put(tuple(ins))
        if kind in ("prim::TupleUnpack", "prim::ListUnpack"):
            return put(*ins[0])
        if kind == "prim::TupleIndex":
            return put(ins[0][ins[1]])
        if kind in ("prim::Uninitialized",):
            return put(None)
        if kind in ("prim::unchecked_cast", "prim::NumToTensor") or kind in _IDENTITY_OPS:
            return put(ins[0] if kind != "prim::NumToTensor" else torch.tensor(ins[0]))
        if kind == "aten::format":
            return put(str(ins[0]))
        if kind == "prim::dtype":
            return put(6)         # torch.float32
        if kind == "prim::device":
            return put(torch.device("cpu"))
        if kind == "aten::warn" or kind == "prim::Print":
            return None
        has_sym = any(isinstance(v, Sym) for v in ins) or any(isinstance(v, (list, tuple)) and any(isinstance(e, Sym) for e in v) for v in ins)
        if not has_sym:
            return put(*self._concrete(kind, ins, n, len(outs)))
        return put(*self._symbolic(kind, ins, n))

    def _concrete(self, kind: str, ins: List[Any], n, n_out: int) -> Tuple[Any, ...]:
        if not any(isinstance(v, torch.Tensor) for v in ins) and not any(isinstance(v, (list, tuple)) and v and isinstance(v[0], torch.Tensor) for v in ins):
            fn = _SCALAR_OPS.get(kind)
            if fn is not None:
                return (fn(*ins),)
        ns, name = kind.split("::")
        if ns != "aten":
            raise LoweringError(f"op {kind} is not lowered ({self._where(n)})")
        try:
            with torch.no_grad():
                packet = getattr(torch.ops.aten, name)
                try:        # bind by the node's own schema: TorchScript passes keyword-only arguments positionally
                    schema = torch._C.parse_schema(n.schema())
                    args = [v for a, v in zip(schema.arguments, ins) if not a.kwarg_only]
                    kwargs = {a.name: v for a, v in zip(schema.arguments, ins) if a.kwarg_only}
                    res = getattr(packet, schema.overload_name or "default")(*args, **kwargs)
                except (RuntimeError, AttributeError, TypeError):
                    res = packet(*ins)
        except Exception as e:
            fn = _SCALAR_OPS.get(kind)
            if fn is not None:
                return (fn(*ins),)
            raise LoweringError(f"constant folding of {kind} failed: {e} ({self._where(n)})") from None
        if name.endswith("_") and isinstance(ins[0], torch.Tensor):
            return (ins[0],)
        return tuple(res) if (n_out > 1 and isinstance(res, (tuple, list))) else (res,)

    def _symbolic(self, kind: str, ins: List[Any], n) -> Tuple[Any, ...]:
        where = self._where(n)
        base = kind[:-1] if (kind.endswith("_") and not kind.endswith("__")) else kind
        inplace = base != kind
        x = ins[0]
        if inplace and base not in _INPLACE_OK:
            raise LoweringError(f"in-place op {kind} on audio-dependent tensors is not lowered ({where})")

        def E(fn: str, ops: Sequence[Any], p0: float = 0.0, p1: float = 0.0) -> Tuple[Sym]:
            """One element-wise instruction; ``op_`` variants write through the first operand's own view (aliases see the result)."""
            if inplace:
                if ops[0] is not x:
                    raise LoweringError(f"{kind}: the in-place target is not the first operand ({where})")
                return (self.ew_inplace(fn, ops[0], ops[1:], p0, p1),)
            return (self.ew(fn, ops, p0, p1),)

        # shape queries
        if base == "aten::size":
            s = self.operand(x, state_ok=True).shape
            return (list(s) if len(ins) == 1 else s[ins[1]],)
        if base == "aten::dim":
            return (len(self.operand(x, state_ok=True).shape),)
        if base == "aten::numel":
            return (self.operand(x, state_ok=True).numel,)
        if base == "aten::len":
            return (self.operand(x, state_ok=True).shape[0],)
        if base in ("aten::is_floating_point",):
            return (True,)
        # views
        if base == "aten::unsqueeze":
            s = self.operand(x, state_ok=True)
            d = ins[1] % (len(s.shape) + 1)
            return (s.like(s.offset, (*s.shape[:d], 1, *s.shape[d:]), (*s.strides[:d], 0, *s.strides[d:])),)
        if base == "aten::squeeze":
            s = self.operand(x, state_ok=True)
            dims = range(len(s.shape)) if len(ins) == 1 else [ins[1] % len(s.shape)] if isinstance(ins[1], int) else [d % len(s.shape) for d in ins[1]]
            keep = [i for i in range(len(s.shape)) if not (i in dims and s.shape[i] == 1)]
            return (s.like(s.offset, [s.shape[i] for i in keep], [s.strides[i] for i in keep]),)
        if base == "aten::permute":
            s = self.operand(x, state_ok=True)
            p = [d % len(s.shape) for d in ins[1]]
            return (s.like(s.offset, [s.shape[i] for i in p], [s.strides[i] for i in p]),)
        if base in ("aten::transpose", "aten::t"):
            s = self.operand(x, state_ok=True)
            a, b = (0, 1) if base == "aten::t" else (ins[1] % len(s.shape), ins[2] % len(s.shape))
            p = list(range(len(s.shape)))
            p[a], p[b] = p[b], p[a]
            return (s.like(s.offset, [s.shape[i] for i in p], [s.strides[i] for i in p]),)
        if base == "aten::slice":
            s = self.operand(x, state_ok=True)
            d = (ins[1] if len(ins) > 1 and ins[1] is not None else 0) % len(s.shape)
            size = s.shape[d]
            start = 0 if len(ins) < 3 or ins[2] is None else ins[2]
            end = size if len(ins) < 4 or ins[3] is None else ins[3]
            step = 1 if len(ins) < 5 or ins[4] is None else ins[4]
            start = max(0, min(size, start + size if start < 0 else start))
            end = max(start, min(size, end + size if end < 0 else end))
            cnt = (end - start + step - 1) // step
            shape, strides = list(s.shape), list(s.strides)
            shape[d], strides[d] = cnt, s.strides[d] * step
            return (s.like(s.offset + start * s.strides[d], shape, strides),)
        if base == "aten::select":
            s = self.operand(x, state_ok=True)
            d = ins[1] % len(s.shape)
            i = ins[2] % s.shape[d]
            return (s.like(s.offset + i * s.strides[d], s.shape[:d] + s.shape[d + 1:], s.strides[:d] + s.strides[d + 1:]),)
        if base in ("aten::view", "aten::reshape", "aten::flatten"):
            s = self.materialise(x)
            if base == "aten::flatten":
                a = (ins[1] if len(ins) > 1 else 0) % len(s.shape)
                b = (ins[2] if len(ins) > 2 else -1) % len(s.shape)
                shape = [*s.shape[:a], int(np.prod(s.shape[a: b + 1])), *s.shape[b + 1:]]
            else:
                shape = list(ins[1])
                if -1 in shape:
                    shape[shape.index(-1)] = s.numel // max(1, -int(np.prod(shape)))
            if int(np.prod(shape)) != s.numel:
                raise LoweringError(f"{kind}: {s.shape} -> {shape} ({where})")
            return (s.like(s.offset, shape, _contig(shape)),)
        if base == "aten::expand":
            s = self.operand(x, state_ok=True)
            shape = [s.shape[i - (len(ins[1]) - len(s.shape))] if d == -1 else d for i, d in enumerate(ins[1])]
            pad = len(shape) - len(s.shape)
            strides = [0] * pad + [0 if s.shape[i] == 1 and shape[pad + i] != 1 else s.strides[i] for i in range(len(s.shape))]
            return (s.like(s.offset, shape, strides),)
        # element-wise
        if base in _UNARY:
            return E(_UNARY[base], [x])
        if base == "aten::square":
            return E("mul", [x, x])
        if base == "aten::rsqrt":
            return E("pow_scalar", [x], -0.5)
        if base == "aten::reciprocal":
            return E("pow_scalar", [x], -1.0)
        if base == "aten::leaky_relu":
            return E("leaky_relu", [x], float(ins[1]) if len(ins) > 1 else 0.01)
        if base == "aten::hardtanh":
            return E("hardtanh", [x], float(ins[1]) if len(ins) > 1 else -1.0, float(ins[2]) if len(ins) > 2 else 1.0)
        if base in ("aten::clamp", "aten::clamp_min", "aten::clamp_max"):
            lo = ins[1] if base != "aten::clamp_max" else None
            hi = ins[2] if base == "aten::clamp" and len(ins) > 2 else (ins[1] if base == "aten::clamp_max" else None)
            return E("clamp", [x], -3.0e38 if lo is None else float(lo), 3.0e38 if hi is None else float(hi))
        if base == "aten::pow":
            if isinstance(ins[1], (int, float)) and isinstance(x, Sym):
                return E("mul", [x, x]) if float(ins[1]) == 2.0 else E("pow_scalar", [x], float(ins[1]))
            raise LoweringError(f"pow with a tensor exponent ({where})")
        if base in _BINARY:
            a, b = ins[0], ins[1]
            alpha = ins[2] if len(ins) > 2 and base in ("aten::add", "aten::sub") else 1
            if alpha is None:
                alpha = 1
            if base == "aten::div" and len(ins) > 2 and ins[2] is not None:
                raise LoweringError(f"div with rounding_mode ({where})")
            scalar_b, scalar_a = isinstance(b, (int, float)), isinstance(a, (int, float))
            if isinstance(b, torch.Tensor) and b.dim() == 0:
                b, scalar_b = float(b), True
            if isinstance(a, torch.Tensor) and a.dim() == 0:
                a, scalar_a = float(a), True
            if scalar_b:
                v = float(b) * float(alpha)
                if base == "aten::add":
                    return E("add_scalar", [a], v)
                if base == "aten::sub":
                    return E("add_scalar", [a], -v)
                if base == "aten::mul":
                    return E("mul_scalar", [a], float(b))
                if float(b) != 0 and math.log2(abs(float(b))).is_integer():
                    return E("mul_scalar", [a], 1.0 / float(b))
                return E("div", [a, torch.tensor(float(b))])
            if scalar_a:
                if base == "aten::add":
                    return E("add_scalar", [b], float(a))
                if base == "aten::mul":
                    return E("mul_scalar", [b], float(a))
                if base == "aten::sub":
                    return E("rsub_scalar", [b], float(a), float(alpha))
                return E("div", [torch.tensor(float(a)), b])
            if alpha != 1:
                b = self.ew("mul_scalar", [b], float(alpha)) if isinstance(b, Sym) else b * alpha
            return E(_BINARY[base], [a, b])
        if base == "aten::rsub":
            return E("rsub_scalar", [x], float(ins[1]), float(ins[2]) if len(ins) > 2 else 1.0) if isinstance(ins[1], (int, float)) else \
                E("sub", [ins[1], x])
        # structure
        if base == "aten::cat":
            parts, d = [self.operand(p) for p in ins[0]], ins[1]
            d %= len(parts[0].shape)
            shape = list(parts[0].shape)
            shape[d] = sum(p.shape[d] for p in parts)
            out = self.alloc(shape)
            pos = 0
            for p in parts:
                if [s for i, s in enumerate(p.shape) if i != d] != [s for i, s in enumerate(shape) if i != d]:
                    raise LoweringError(f"cat of {[q.shape for q in parts]} along {d} ({where})")
                self.ew("copy", [p], out=out.like(out.offset + pos * out.strides[d], p.shape, out.strides))
                pos += p.shape[d]
            return (out,)
        if base in ("aten::pad", "aten::reflection_pad1d", "aten::constant_pad_nd", "aten::replication_pad1d"):
            pads = list(ins[1])
            mode = {"aten::reflection_pad1d": "reflect", "aten::constant_pad_nd": "constant", "aten::replication_pad1d": "replicate"}.get(base) or (
                ins[2] if len(ins) > 2 and ins[2] is not None else "constant")
            value = (ins[3] if base == "aten::pad" and len(ins) > 3 else ins[2] if base == "aten::constant_pad_nd" and len(ins) > 2 else 0.0) or 0.0
            if len(pads) != 2:
                if any(pads[2:]):
                    raise LoweringError(f"padding {pads}: only the last dimension is lowered ({where})")
                pads = pads[:2]
            return (self.pad_last(x, int(pads[0]), int(pads[1]), str(mode), float(value)),)
        if base in ("aten::mean", "aten::sum"):
            dims = ins[1] if len(ins) > 1 and ins[1] is not None and not isinstance(ins[1], bool) else list(range(len(self.operand(x).shape)))
            dims = [dims] if isinstance(dims, int) else list(dims)
            keep = bool(ins[2]) if len(ins) > 2 and isinstance(ins[2], (bool, int)) else False
            res = self.mean(x, dims, keep)
            if base == "aten::sum":
                s = self.operand(x)
                res = self.ew("mul_scalar", [res], float(np.prod([s.shape[d % len(s.shape)] for d in dims])))
            return (res,)
        if base in ("aten::conv1d", "aten::_convolution", "aten::convolution"):
            one = lambda v: int(v[0]) if isinstance(v, (list, tuple)) else int(v)     # noqa: E731
            if base != "aten::conv1d" and (ins[6] or any(ins[7])):
                raise LoweringError(f"a transposed convolution ({where})")
            groups = ins[6] if base == "aten::conv1d" else ins[8]
            if isinstance(ins[4], str):
                raise LoweringError(f"conv1d padding={ins[4]!r} ({where})")
            return (self.conv1d(x, ins[1], ins[2], one(ins[3]), one(ins[4]), one(ins[5]), int(groups)),)
        if base == "aten::batch_norm":
            w, b, rm, rv, training, eps = ins[1], ins[2], ins[3], ins[4], ins[5], ins[7]
            if training or rm is None or rv is None:
                raise LoweringError(f"batch_norm in training mode / without running statistics ({where})")
            scale = (w if w is not None else 1.0) / torch.sqrt(rv + eps)
            shift = (b if b is not None else 0.0) - rm * scale
            s = self.operand(x)
            bshape = [1] * len(s.shape)
            bshape[1] = s.shape[1]
            return (self.ew("fma", [x, scale.reshape(bshape), shift.reshape(bshape)]),)
        if base == "aten::layer_norm":
            raise LoweringError(f"layer_norm is not lowered ({where})")
        if base == "aten::linear":
            return (self.linear(x, ins[1], ins[2] if len(ins) > 2 else None),)
        if base == "aten::lstm":
            if len(ins) != 9:
                raise LoweringError(f"lstm overload with {len(ins)} arguments (packed sequences are not lowered) ({where})")
            y, hn, cn = self.lstm(x, ins[1], ins[2], bool(ins[3]), int(ins[4]), bool(ins[6]), bool(ins[7]), bool(ins[8]))
            return (y, hn, cn)
        raise LoweringError(f"op {kind} on audio-dependent tensors is not lowered ({where})")


def lower(module: torch.jit.ScriptModule, window: int = 1536, sample_rate: int = 16000) -> Program:
    """The steady-state program of ``module(chunk[window], sample_rate)`` (see the module docstring).  The archive's own
    ``reset_states()`` (when it has one) defines the state a stream starts from."""
    if not isinstance(module, (torch.jit.ScriptModule, torch.jit.RecursiveScriptModule)):
        raise LoweringError(f"a TorchScript module is needed (torch.jit.load / torch.hub.load(..., onnx=False)), got {type(module).__name__}")
    module.eval()
    if hasattr(module, "reset_states"):
        module.reset_states()
    lw = _Lowerer(module, window, sample_rate)
    first = lw.run(None)
    if not any(isinstance(v, Sym) for v in lw.overlay.values()):
        return first                         # stateless scorer
    # steady state: the attributes written by the first walk are now state slots
    for key, (off, shape) in lw.state_slots.items():
        lw.overlay[key] = Sym(SPACE_STATE, off, shape, _contig(shape))
    init = list(first.state_init)
    steady = lw.run(None)
    if steady.signature() != first.signature():
        a, b = first.signature(), steady.signature()
        at = next((i for i, (p, q) in enumerate(zip(a, b)) if p != q), min(len(a), len(b)))
        raise LoweringError(f"the archive computes its first window differently from the following ones (instruction {at}: "
                            f"{a[at] if at < len(a) else None} vs {b[at] if at < len(b) else None})")
    # a third walk must change nothing more (bookkeeping attributes have settled)
    for key, (off, shape) in lw.state_slots.items():
        lw.overlay[key] = Sym(SPACE_STATE, off, shape, _contig(shape))
    again = lw.run(None)
    if again.words != steady.words:
        raise LoweringError("the archive's graph does not reach a steady state after the first window")
    steady.state_init = init
    return steady


_LOWERED_OPS = {
    # element-wise, views, structure (see _Lowerer._symbolic); scalar / list / control ops are evaluated at load time
    *_UNARY, *_BINARY, *_IDENTITY_OPS, *_SCALAR_OPS, "aten::square", "aten::rsqrt", "aten::reciprocal", "aten::leaky_relu", "aten::hardtanh",
    "aten::clamp", "aten::clamp_min", "aten::clamp_max", "aten::pow", "aten::rsub", "aten::size", "aten::dim", "aten::numel", "aten::len",
    "aten::is_floating_point", "aten::unsqueeze", "aten::squeeze", "aten::permute", "aten::transpose", "aten::t", "aten::slice", "aten::select",
    "aten::view", "aten::reshape", "aten::flatten", "aten::expand", "aten::cat", "aten::pad", "aten::reflection_pad1d", "aten::constant_pad_nd",
    "aten::replication_pad1d", "aten::mean", "aten::sum", "aten::conv1d", "aten::_convolution", "aten::convolution", "aten::batch_norm",
    "aten::linear", "aten::lstm", "aten::format", "aten::warn", "aten::zeros", "aten::ones", "aten::tensor", "aten::empty", "aten::full",
    "aten::zeros_like", "aten::ones_like", "aten::arange", "aten::copy", "aten::item",
}


def unsupported_ops(kinds: Sequence[str]) -> List[str]:
    """Of the node kinds of an archive's inlined graph (``{n.kind() for n in graph}``), those this loader has no lowering for --
    whatever their operands turn out to be.  ``prim::`` nodes are structure (constants, attributes, control flow, tuples) and are
    evaluated by the walk; an ``aten::`` op missing here raises ``LoweringError`` only if it meets audio-dependent tensors (constant
    sub-expressions are folded with torch), so a non-empty answer is a WARNING list, an empty one a guarantee."""
    out = []
    for k in kinds:
        base = k[:-1] if (k.endswith("_") and not k.endswith("__")) else k
        if k.startswith("prim::") or base in _LOWERED_OPS:
            continue
        out.append(k)
    return sorted(out)


def load_archive(path: str) -> torch.jit.ScriptModule:
    return torch.jit.load(path, map_location="cpu")


class HipGraphVadScorer:
    """``wj_vadg_*``: a lowered TorchScript window scorer on the device.  Same surface as ``vad.HipSileroScorer`` (``scores`` /
    ``scores_device`` / ``reset_states`` / ``close``); ``window`` is the archive generation's grid (1536 samples for the v3.1 /
    v4.0 hub archives at 16 kHz, utils_vad.get_speech_timestamps' default there)."""

    def __init__(self, archive: Union[str, torch.jit.ScriptModule], window: int = 1536, sample_rate: int = 16000, device: int = 0,
                 max_windows_per_launch: int = 1 << 20, fused: Optional[bool] = None):
        """``fused``: None = one launch per stage with the arena in LDS whenever it fits (the normal case), True = require that,
        False = one launch per instruction over an arena in HBM (the fall-back for arenas beyond the LDS; kept selectable as the
        cross-check of the fused kernels).  Per-window memory is allocated by the first call, for the windows it scores."""
        from . import hipbind
        if not torch.cuda.is_available():
            raise hipbind.WjError("no ROCm device visible: the HIP VAD graph scorer has no CPU fallback")
        module = load_archive(archive) if isinstance(archive, str) else archive
        self.program = lower(module, window, sample_rate)
        self.window, self.sample_rate = int(window), int(sample_rate)
        self.device = int(device)
        self.dev = torch.device("cuda", device)
        self.ctx = hipbind.context(device)
        self._lib = hipbind.lib()
        p = self.program
        words = np.asarray(p.words, dtype=np.int32)
        consts, state = p.const_blob(), p.state_blob()
        handle = C.c_void_p()
        mode = 0 if fused is None else (2 if fused else 1)
        hipbind.check(self._lib.wj_vadg_create(self.ctx.handle, words.ctypes.data_as(C.POINTER(C.c_int32)), len(words), p.n_instr,
                                               consts.ctypes.data_as(C.POINTER(C.c_float)), int(consts.size),
                                               state.ctypes.data_as(C.POINTER(C.c_float)), int(p.state_floats), int(p.arena_floats),
                                               int(p.xchg_floats), int(p.input_space), int(p.input_offset), int(p.output_space),
                                               int(p.output_offset), int(window), int(max(1, max_windows_per_launch)), mode,
                                               C.byref(handle)), "wj_vadg_create")
        self.handle = handle
        info = (C.c_int32 * 8)()
        hipbind.check(self._lib.wj_vadg_info(self.handle, info), "wj_vadg_info")
        self.fused, self.lds_bytes, self.n_stages, self.lstm_in_registers = bool(info[0]), int(info[1]), int(info[2]), bool(info[3])

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._lib.wj_vadg_free(self.handle)
            self.handle = None

    def reset_states(self) -> None:       # upstream API compatibility: every stream of a call starts from the archive's reset state
        return None

    def scores_device(self, pcm: torch.Tensor, offsets: Sequence[int]) -> List[np.ndarray]:
        from . import hipbind
        n = len(offsets) - 1
        lens = [int(offsets[i + 1]) - int(offsets[i]) for i in range(n)]
        wins = [(ln + self.window - 1) // self.window for ln in lens]
        poff = np.concatenate([[0], np.cumsum(wins)]).astype(np.int64)
        probs = torch.empty(int(max(1, poff[-1])), dtype=torch.float32, device=self.dev)
        off = (C.c_int64 * (n + 1))(*[int(o) for o in offsets])
        po = (C.c_int64 * (n + 1))(*poff.tolist())
        torch.cuda.current_stream().synchronize()
        hipbind.check(self._lib.wj_vadg_scores(self.handle, C.c_void_p(pcm.data_ptr()), off, po, n, C.c_void_p(probs.data_ptr()), None),
                      "wj_vadg_scores")
        self.ctx.sync()
        host = probs.cpu().numpy()
        return [host[poff[i]:poff[i + 1]].copy() for i in range(n)]

    def scores(self, clips: Sequence[Union[np.ndarray, torch.Tensor]]) -> List[np.ndarray]:
        if len(clips) and all(isinstance(c, torch.Tensor) and c.is_cuda for c in clips):
            offsets = np.concatenate([[0], np.cumsum([int(c.numel()) for c in clips])]).astype(np.int64)
            if offsets[-1] == 0:
                return [np.zeros(0, dtype=np.float32) for _ in clips]
            return self.scores_device(torch.cat([c.reshape(-1).to(torch.float32) for c in clips]), offsets.tolist())
        arrs = [np.ascontiguousarray(c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else c, dtype=np.float32).reshape(-1) for c in clips]
        offsets = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.int64)
        if offsets[-1] == 0:
            return [np.zeros(0, dtype=np.float32) for _ in arrs]
        pcm = torch.from_numpy(np.concatenate(arrs)).to(self.dev)
        return self.scores_device(pcm, offsets.tolist())
