"""Scene-parallel sharding across the GPUs of one node.

The reference has no multi-GPU code at all; its work units -- auditok scenes
/ VAD groups -- are independent (``condition_on_previous_text=False``,
/root/reference/whisperjav/config/components/asr/faster_whisper.py:296; VAD state resets per
``segment()`` call), so the path shards with NO data-path collective:

  * one process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on the GPU box,
    ``gloo`` in the CPU tests);
  * ONE broadcast of the packed weight blob at start-up (large-v3 bf16 = 3.1 GB; 7 xGMI links x
    ~153 GB/s make this well under a second) -- every rank then holds the whole model;
  * scenes are assigned longest-processing-time-first so the ranks finish together;
  * results (token ids / timestamps, a few KB per scene) are gathered as Python objects.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


log = logging.getLogger("whisperjav_amd")


@dataclass(frozen=True)
class RankInfo:
    rank: int
    world: int
    local_rank: int


def rank_info() -> RankInfo:
    return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                    int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend: Optional[str] = None) -> RankInfo:
    """Join the process group described by the torchrun environment (no-op for a single process)."""
    info = rank_info()
    if info.world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(info.local_rank)
            kwargs["device_id"] = torch.device("cuda", info.local_rank)
        dist.init_process_group(backend=backend, rank=info.rank, world_size=info.world, **kwargs)
    return info


def broadcast_blob(blob: Optional[torch.Tensor], offsets: Optional[np.ndarray], device: torch.device,
                   src: int = 0) -> Tuple[torch.Tensor, np.ndarray]:
    """Rank ``src`` owns the packed weights (uint8 host or device tensor + int64 offsets); every rank
    returns a device copy.  The only collective of the whole path."""
    info = rank_info()
    if info.world == 1 or not dist.is_initialized():
        assert blob is not None and offsets is not None
        return blob.to(device), np.asarray(offsets, dtype=np.int64)
    if os.environ.get("WJ_BCAST_CABI") == "1" and device.type == "cuda":
        return broadcast_blob_cabi(blob, offsets, device, src)
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if info.rank == src:
        assert blob is not None and offsets is not None
        meta[0], meta[1] = blob.numel(), len(offsets)
    dist.broadcast(meta, src=src)
    nbytes, n_off = int(meta[0]), int(meta[1])
    off_t = torch.zeros(n_off, dtype=torch.int64, device=device)
    if info.rank == src:
        off_t.copy_(torch.from_numpy(np.asarray(offsets, dtype=np.int64)))
        dev_blob = blob.to(device)
    else:
        dev_blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(off_t, src=src)
    dist.broadcast(dev_blob, src=src)
    return dev_blob, off_t.cpu().numpy()


def broadcast_blob_cabi(blob: Optional[torch.Tensor], offsets: Optional[np.ndarray], device: torch.device,
                        src: int = 0) -> Tuple[torch.Tensor, np.ndarray]:
    """``broadcast_blob`` with the payload moved by libwjhip's own RCCL entry points (``wj_comm_unique_id`` /
    ``wj_comm_init`` / ``wj_bcast_weights``, include/wjhip.h) -- what a host without PyTorch would call.  The 128-byte
    communicator id and the sizes travel through the existing process group (any backend) as Python objects.
    Selected by ``WJ_BCAST_CABI=1``; the torch.distributed path stays the default (it is the one the multi-GPU bench of
    the driver exercises)."""
    import ctypes as C
    from . import hipbind
    info = rank_info()
    lib = hipbind.lib()
    ctx = hipbind.context(device.index or 0)
    meta = None
    if info.rank == src:
        assert blob is not None and offsets is not None
        uid = C.create_string_buffer(128)
        hipbind.check(lib.wj_comm_unique_id(uid), "wj_comm_unique_id")
        meta = (uid.raw, int(blob.numel()), np.asarray(offsets, dtype=np.int64))
    meta = broadcast_object(meta, src=src)
    uid_raw, nbytes, offs = meta
    dev_blob = blob.to(device) if info.rank == src else torch.empty(nbytes, dtype=torch.uint8, device=device)
    comm = C.c_void_p()
    hipbind.check(lib.wj_comm_init(ctx.handle, info.world, info.rank, uid_raw, C.byref(comm)), "wj_comm_init")
    torch.cuda.current_stream().synchronize()
    global LAST_COMM_RANKS
    try:
        n = C.c_int(0)
        rc = int(lib.wj_comm_count(comm, C.byref(n)))     # what RCCL says it built
        if rc == 0:
            LAST_COMM_RANKS = int(n.value)
        elif rc == hipbind.WJ_E_UNSUPPORTED:               # a librccl without ncclCommCount: the size is unknown, not wrong
            LAST_COMM_RANKS = 0
            log.warning("wj_comm_count: this RCCL has no ncclCommCount; communicator size not verified")
        else:
            hipbind.check(rc, "wj_comm_count")
        if rc == 0 and LAST_COMM_RANKS != info.world:
            raise hipbind.WjError(f"RCCL built a communicator of {LAST_COMM_RANKS} ranks for a world of {info.world}")
        hipbind.check(lib.wj_bcast_weights(comm, C.c_void_p(dev_blob.data_ptr()), nbytes, src, None), "wj_bcast_weights")
    finally:
        lib.wj_comm_destroy(comm)
    return dev_blob, offs


LAST_COMM_RANKS = 0      # ncclCommCount of the communicator the last broadcast_blob_cabi built on this rank


def assign_lpt(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first partition of work units (scene durations) over ``world`` ranks.
    Deterministic: ties go to the lower rank / lower index, so every rank computes the same plan."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    plan: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(i)
        load[r] += float(costs[i])
    for p in plan:
        p.sort()
    return plan


def gather_objects(local: Any, dst: int = 0) -> Optional[List[Any]]:
    """Collect small per-rank Python results on ``dst`` (scene transcripts)."""
    info = rank_info()
    if info.world == 1 or not dist.is_initialized():
        return [local]
    out: Optional[List[Any]] = [None] * info.world if info.rank == dst else None
    dist.gather_object(local, out, dst=dst)
    return out


def merge_by_index(plan: List[List[int]], per_rank: List[Dict[int, Any]]) -> List[Any]:
    """Restore the original unit order (the reference's SRTStitcher sorts by scene start the same way,
    /root/reference/whisperjav/modules/srt_stitching.py:35)."""
    n = sum(len(p) for p in plan)
    merged: List[Any] = [None] * n
    for r, idxs in enumerate(plan):
        for i in idxs:
            merged[i] = per_rank[r][i]
    return merged


def barrier() -> None:
    if dist.is_initialized() and rank_info().world > 1:
        dist.barrier()


def max_over_ranks(value: float, device: torch.device) -> float:
    if not dist.is_initialized() or rank_info().world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def broadcast_object(obj: Any, src: int = 0) -> Any:
    """Small Python object (the scene list) from ``src`` to every rank."""
    info = rank_info()
    if info.world == 1 or not dist.is_initialized():
        return obj
    box = [obj if info.rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def scene_parallel(scenes: Sequence[Tuple[float, float]], work) -> Optional[List[Any]]:
    """BASELINE cfg4's flow: every rank holds the same scene list ``[(start_s, end_s)]``; scenes are
    assigned longest-first, rank r calls ``work(scene_index)`` for its own scenes only, rank 0 returns all results in
    scene order (the order the reference's SRTStitcher restores, srt_stitching.py:35); other ranks return None.
    No collective touches the data path: one object gather at the end."""
    info = rank_info()
    plan = assign_lpt([float(b) - float(a) for a, b in scenes], info.world)
    mine = {i: work(i) for i in plan[info.rank]}
    gathered = gather_objects(mine, dst=0)
    if info.rank != 0:
        return None
    return merge_by_index(plan, gathered)
