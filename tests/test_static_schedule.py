"""Static checks of the device code hipcc emitted for the pipelined GEMM kernels (no GPU needed: the gfx950 code object is
cut out of the built ``gemm.o`` and disassembled).

What they guard: the property the round-2 probe found missing in round 1's multi-stage kernels -- LDS-DMA requests must
stay in flight ACROSS the workgroup barrier.  ``__syncthreads()`` carries a fence that hipcc lowers to
``s_waitcnt vmcnt(0) ...`` in front of ``s_barrier`` whenever ``global_load_lds`` is outstanding, which silently turns a
counted-``vmcnt`` ring into a drain-every-step loop (same results, no pipelining).  The kernels use a raw ``s_barrier``
and their own counted waits instead; these tests fail if a rebuild brings the drain back, if the main loops lose their
shape, or if a kernel starts spilling.
"""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
OBJDUMP, READELF = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
pytestmark = pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(READELF)), reason="ROCm LLVM tools not present")

GEMM_ARGS = "EEvNS_8GemmArgsE"
PP64 = "_ZN2wj22gemm_h_big_pp64_kernelIDF16_Li%dE" + GEMM_ARGS            # <f16, EPI>
PP = "_ZN2wj20gemm_h_big_pp_kernelIDF16_Li0ELi%dELi0ELi0E" + GEMM_ARGS    # <f16, EPI_T, NS, ABL 0, PLACE 0>
PPB = "_ZN2wj21gemm_h_big_ppb_kernelIDF16_Li%dELi4E" + GEMM_ARGS             # <f16, EPI, NS 4>: blocked operands (round 4)
MS = "_ZN2wj21gemm_h_tile_ms_kernelIDF16_Li0ELi%dE" + GEMM_ARGS           # <f16, EPI_T, NS>
BIG = "_ZN2wj17gemm_h_big_kernelIDF16_Li0ELi0E" + GEMM_ARGS               # lockstep kernel (drains by design)


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    from whisperjav_amd import build
    build.build(verbose=False)
    obj = build.CSRC / "build" / "gemm.o"
    assert obj.exists()
    tmp = tmp_path_factory.mktemp("co")
    local = tmp / "gemm.o"
    shutil.copy(obj, local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, capture_output=True)
    cos = [p for p in tmp.iterdir() if "gfx950" in p.name]
    assert len(cos) == 1, list(tmp.iterdir())
    return cos[0]


def _disasm(co, symbol):
    out = subprocess.run([OBJDUMP, "-d", f"--disassemble-symbols={symbol}", str(co)], check=True, capture_output=True, text=True).stdout
    ins = [l.split("//")[0].strip() for l in out.splitlines() if l.startswith("\t")]
    assert ins, f"{symbol} not found in the code object"
    return ins


def _before_barriers(ins, window=2):
    return [ins[max(0, k - window):k] for k, l in enumerate(ins) if l.startswith("s_barrier")]


def _is_fence(l):
    # the lowered fence of __syncthreads(): one s_waitcnt that zeroes vmcnt AND lgkmcnt
    return l.startswith("s_waitcnt") and "vmcnt(0)" in l and "lgkmcnt(0)" in l


def test_pairs_kernel_main_loop(code_object):
    ins = _disasm(code_object, PP64 % 0)
    assert sum(l.startswith("v_mfma") for l in ins) == 64          # two MFMA phases of 32, unrolled once
    assert sum(l.startswith("ds_read_b128") for l in ins) == 24
    assert sum("global_load_lds_dwordx4" in l for l in ins) == 16    # prologue + one pair per loop trip
    pre = _before_barriers(ins)
    assert len(pre) == 7       # P, the stagger barrier, 4 in the loop, the balancing one after it
    assert not any(_is_fence(l) for w in pre for l in w), pre
    # the only drains in front of a barrier: the prologue's and the one that ends MEM(2p+1)
    assert sum(any(l == "s_waitcnt vmcnt(0)" for l in w) for w in pre) == 2, pre
    # the MFMA phases are uninterrupted: 32 MFMAs between s_setprio 1 and s_setprio 0, nothing else
    k = ins.index("s_setprio 1")
    body = [l for l in ins[k + 1:ins.index("s_setprio 0", k)] if l != "s_waitcnt lgkmcnt(0)"]   # hipcc's own (already satisfied) wait
    assert len(body) == 32 and all(l.startswith("v_mfma") for l in body), body[:40]


@pytest.mark.parametrize("ns", [3, 4, 5])
def test_ring_kernel_keeps_requests_in_flight_across_barriers(code_object, ns):
    ins = _disasm(code_object, PP % ns)
    pre = _before_barriers(ins, window=3)
    assert len(pre) == 5
    assert not any(_is_fence(l) for w in pre for l in w), pre
    counted = f"s_waitcnt vmcnt({4 * (ns - 2)})"
    assert sum(l == counted for l in ins) == 2, (counted, [l for l in ins if l.startswith("s_waitcnt vmcnt")])
    assert sum(l.startswith("v_mfma") for l in ins) == 32


@pytest.mark.parametrize("epi", [0, 1, 3, 5, 6])
def test_blocked_ring_kernel_keeps_requests_in_flight_across_barriers(code_object, epi):
    """The encoder's default since round 4 (bias, GELU, residual, Q/K heads, transposed V epilogues): the ring schedule over
    blocked operands -- two stages stay in flight across every barrier of the main loop, no fence, one 32-MFMA phase."""
    ins = _disasm(code_object, PPB % epi)
    pre = _before_barriers(ins, window=3)
    assert len(pre) == 5
    assert not any(_is_fence(l) for w in pre for l in w), pre
    loop = ins[:max(k for k, l in enumerate(ins) if l.startswith("s_barrier"))]     # everything before the epilogue
    assert sum(l == "s_waitcnt vmcnt(8)" for l in loop) == 2, [l for l in loop if l.startswith("s_waitcnt vmcnt")]
    assert sum(l.startswith("v_mfma") for l in ins) == 32
    assert sum("global_load_lds_dwordx4" in l for l in ins) == 16      # 3 prologue stages + one per loop trip, 4 requests each
    k = ins.index("s_setprio 1")
    body = [l for l in ins[k + 1:ins.index("s_setprio 0", k)] if l != "s_waitcnt lgkmcnt(0)"]
    assert len(body) == 32 and all(l.startswith("v_mfma") for l in body), body[:40]


@pytest.mark.parametrize("ns", [3, 4, 5])
def test_decode_ring_kernel_has_no_fence(code_object, ns):
    ins = _disasm(code_object, MS % ns)
    pre = _before_barriers(ins)
    assert len(pre) == 1 and not any(_is_fence(l) for l in pre[0]), pre
    assert f"s_waitcnt vmcnt({8 * (ns - 2)})" in ins


def test_lockstep_kernel_is_the_one_that_drains(code_object):
    # the control: __syncthreads() with LDS-DMA in flight does produce the fence this file looks for
    pre = _before_barriers(_disasm(code_object, BIG))
    assert any(_is_fence(l) for w in pre for l in w), pre


@pytest.mark.parametrize("epi,name", [(0, "bias"), (1, "gelu"), (5, "q/k heads"), (8, "cross k/v")])
def test_epilogue_store_width_and_no_per_fragment_waits(code_object, epi, name):
    ins = _disasm(code_object, PP64 % epi)
    # 16-byte stores exist (permlane-swapped fragment pairs) ...
    assert sum(l.startswith("v_permlane16_swap") for l in ins) >= 32
    assert sum(l.startswith("global_store_dwordx4") for l in ins) >= 16
    # ... and the epilogue is not a chain of "wait for everything, then store" blocks any more (round 1: one per fragment)
    tail = ins[max(k for k, l in enumerate(ins) if l.startswith("s_barrier")):]
    assert sum(l.startswith("s_waitcnt vmcnt(0)") for l in tail) <= 2, name


def test_no_spills_in_the_256_tile_kernels(code_object):
    notes = subprocess.run([READELF, "--notes", str(code_object)], check=True, capture_output=True, text=True).stdout
    blocks = re.split(r"\n\s*- \.agpr_count:", notes)
    seen = 0
    for b in blocks:
        m = re.search(r"\.name:\s+(\S+)", b)
        if not m or "gemm_h_big" not in m.group(1):
            continue
        if re.search(r"Li\d+ELi\d+ELi[1-9]\d*ELi\dE", m.group(1)):   # timing-ablation instantiations
            continue
        seen += 1
        size = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1))
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1))
        assert size == 0 and vgpr <= 256, (m.group(1), size, vgpr)
    assert seen >= 40, seen


def test_no_kernel_of_the_library_spills_except_the_known_two(tmp_path):
    """Scratch (private segment) use of EVERY kernel in every translation unit.  Allowed: the register-resident beam top-2K
    kernels, which sit at their 128-VGPR cap (1024 threads per workgroup) and spill 24 / 68 bytes per lane by design."""
    from whisperjav_amd import build
    build.build(verbose=False)
    allowed = {"beam_topk_reg_kernelILi51E": 24, "beam_topk_reg_kernelILi64E": 68}
    total = 0
    for src in build.SOURCES:
        obj = build.CSRC / "build" / (src.replace(".hip", ".o"))
        local = tmp_path / obj.name
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, capture_output=True)
        cos = [p for p in tmp_path.iterdir() if p.name.startswith(obj.name + ".") and "gfx950" in p.name]
        if not cos:          # comm.hip is host code only
            continue
        notes = subprocess.run([READELF, "--notes", str(cos[0])], check=True, capture_output=True, text=True).stdout
        for b in re.split(r"\n\s*- \.agpr_count:", notes):
            m = re.search(r"\.name:\s+(\S+)", b)
            if not m:
                continue
            total += 1
            size = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1))
            limit = max([v for k, v in allowed.items() if k in m.group(1)] or [0])
            assert size <= limit, (m.group(1), size)
    assert total > 500, total


def test_encoder_attention_masks_padding_keys_on_the_last_tile_only(tmp_path):
    """attn_enc_h_kernel: the padding-key mask (64 v_cmp / v_cndmask per 64-key tile) must live in a peeled copy of the tile
    body, not in the main loop -- as a branch inside one loop body hipcc if-converts it onto every tile."""
    from whisperjav_amd import build
    build.build(verbose=False)
    local = tmp_path / "attention.o"
    shutil.copy(build.CSRC / "build" / "attention.o", local)
    subprocess.run([OBJDUMP, "--offloading", str(local)], check=True, capture_output=True)
    co = [p for p in tmp_path.iterdir() if "gfx950" in p.name][0]
    ins = _disasm(co, "_ZN2wj17attn_enc_h_kernelIDF16_Li9EEEvPKtS2_S2_PT_iiii")       # <f16, default variant>
    blocks, cur = [], []
    for l in ins:                       # straight-line runs between branches
        cur.append(l)
        if l.startswith(("s_cbranch", "s_branch")):
            blocks.append(cur)
            cur = []
    blocks.append(cur)
    count = lambda b, k: sum(l.startswith(k) for l in b)
    clean = [b for b in blocks if count(b, "v_mfma") >= 16 and count(b, "v_cndmask") == 0]
    masked = [b for b in blocks if count(b, "v_mfma") >= 16 and count(b, "v_cndmask") >= 16]
    assert clean and masked, [(count(b, "v_mfma"), count(b, "v_cndmask")) for b in blocks if count(b, "v_mfma")]
    assert sum(count(b, "v_mfma") for b in blocks) == 72      # two copies of (16 S + 16 PV + 4 row-sum) MFMAs
