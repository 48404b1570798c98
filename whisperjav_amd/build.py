"""In-tree build of libwjhip.so (hipcc, gfx950 only) and of the C oracle helpers.

``python -m whisperjav_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without
a GPU; the resulting ``whisperjav_amd/csrc/libwjhip.so`` is git-ignored but travels to the GPU
box with the working tree.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libwjhip.so"
SOURCES = ["engine.hip", "gemm.hip", "attention.hip", "norm.hip", "sampler.hip", "logmel.hip", "vad.hip", "vadgraph.hip", "align.hip", "comm.hip", "qwen.hip", "qwen_audio.hip"]
HEADERS = ["common.hpp", "kernels.hpp", "../../include/wjhip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
# attention.hip: MFMA results feed VALU softmax code every key tile; with accumulators in AGPRs the compiler emits
# ~190 v_accvgpr_read/write per tile, the VGPR form of the MFMAs removes all of them (gfx950 has a unified file)
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libwjhip can only be built with the ROCm toolchain")
    return exe


def _stamp(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(Path(p).read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    hdrs = [(CSRC / h).resolve() for h in HEADERS]
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()
    hdr_stamp = _stamp(hdrs)

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        stamp_file = objdir / (src.stem + ".stamp")
        stamp = _stamp([src]) + hdr_stamp
        if not force and obj.exists() and stamp_file.exists() and stamp_file.read_text() == stamp:
            return obj
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{res.stdout}\n{res.stderr}")
        stamp_file.write_text(stamp)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    newest = max(o.stat().st_mtime for o in objs)
    if force or not LIB.exists() or LIB.stat().st_mtime < newest:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs), "-ldl"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB


PROBE_SRC = CSRC.parent.parent / "scripts" / "gemm_probe.hip"
PROBE = CSRC / "gemm_probe"


def build_probe(verbose: bool = True) -> Path:
    """The stand-alone GEMM probe (scripts/gemm_probe.hip): a C-ABI client of libwjhip.so (dlopen), no Python at run time.
    A development tool -- micro-benchmarks and bit-for-bit variant comparisons in a few seconds of GPU time."""
    if PROBE.exists() and PROBE.stat().st_mtime >= PROBE_SRC.stat().st_mtime:
        return PROBE
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", str(PROBE_SRC), "-o", str(PROBE), "-I", str(CSRC.parent.parent / "include"), "-ldl"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for the probe:\n{res.stdout}\n{res.stderr}")
    return PROBE


if __name__ == "__main__":
    path = build(force="--force" in sys.argv)
    build_probe()
    print(path)
