"""Scene-parallel sharding: world_size-2 gloo processes on the CPU (the GPU path uses the same code with
backend nccl = RCCL).  Covers the single weight broadcast, the LPT plan and the gather/merge of results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from whisperjav_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    info = sharding.init_distributed("gloo")
    assert (info.rank, info.world) == (rank, world)
    dev = torch.device("cpu")
    blob = offsets = None
    if rank == 0:
        g = torch.Generator().manual_seed(7)
        blob = torch.randint(0, 255, (1 << 16,), dtype=torch.uint8, generator=g)
        offsets = np.arange(0, 1 << 16, 256, dtype=np.int64)[:40]
    got_blob, got_off = sharding.broadcast_blob(blob, offsets, dev)
    # every rank computes the same deterministic plan and works only on its own scenes
    durations = [29.0, 3.5, 12.0, 28.5, 7.25, 0.8, 19.0, 19.0, 5.5]
    plan = sharding.assign_lpt(durations, world)
    mine = {i: f"scene{i}:rank{rank}:{durations[i]}" for i in plan[rank]}
    gathered = sharding.gather_objects(mine, dst=0)
    t = sharding.max_over_ranks(1.0 + rank, dev)
    sharding.barrier()
    if rank == 0:
        merged = sharding.merge_by_index(plan, gathered)
        torch.save({"blob_sum": int(got_blob.sum()), "off": got_off, "merged": merged, "plan": plan, "tmax": t},
                   os.path.join(out_dir, "rank0.pt"))
    else:
        torch.save({"blob_sum": int(got_blob.sum()), "off": got_off, "tmax": t}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_roundtrip(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert r0["blob_sum"] == r1["blob_sum"] and np.array_equal(r0["off"], r1["off"]) and len(r0["off"]) == 40
    assert r0["tmax"] == r1["tmax"] == 2.0
    assert [m.split(":")[0] for m in r0["merged"]] == [f"scene{i}" for i in range(9)]
    ranks = {int(m.split("rank")[1].split(":")[0]) for m in r0["merged"]}
    assert ranks == {0, 1}


def test_lpt_plan_properties():
    rng = np.random.default_rng(0)
    for world in (1, 2, 4, 8):
        costs = rng.uniform(0.5, 29.0, size=300).tolist()
        plan = sharding.assign_lpt(costs, world)
        assert sorted(i for p in plan for i in p) == list(range(300))          # a partition
        loads = [sum(costs[i] for i in p) for p in plan]
        assert max(loads) - min(loads) <= max(costs) + 1e-9                     # LPT balance bound
        assert plan == sharding.assign_lpt(costs, world)                        # deterministic
    assert sharding.assign_lpt([], 4) == [[], [], [], []]
    assert sharding.assign_lpt([5.0], 2) == [[0], []]


def test_single_process_paths_are_noops():
    os.environ.pop("WORLD_SIZE", None); os.environ.pop("RANK", None)
    blob = torch.arange(64, dtype=torch.uint8)
    b, o = sharding.broadcast_blob(blob, np.array([0, 32]), torch.device("cpu"))
    assert torch.equal(b, blob) and o.tolist() == [0, 32]
    assert sharding.gather_objects({"a": 1}) == [{"a": 1}]
    assert sharding.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def _scene_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from whisperjav_amd import sharded_transcribe
    sharding.init_distributed("gloo")
    scenes = sharding.broadcast_object([(0.0, 29.0), (30.0, 33.5), (40.0, 52.0), (60.0, 88.5), (90.0, 97.25)] if rank == 0 else None)
    audio = np.arange(16000 * 100, dtype=np.float32)
    seen = []

    def transcribe_scene(clip, start_s):
        seen.append(start_s)
        assert clip[0] == int(start_s * 16000)          # the right slice of the recording
        return [{"start": start_s + 1.0, "end": start_s + 2.0, "text": f"rank{rank}@{start_s}", "avg_logprob": -0.1},
                {"start": start_s + 0.2, "end": start_s + 0.9, "text": "first", "avg_logprob": -0.2}]
    merged = sharded_transcribe.transcribe_scenes(audio, 16000, scenes, transcribe_scene)
    torch.save({"merged": merged, "seen": seen}, os.path.join(out_dir, f"scene_rank{rank}.pt"))
    sharding.barrier()
    torch.distributed.destroy_process_group()


def test_scene_parallel_driver_two_ranks(tmp_path):
    """cfg4's control flow on two gloo ranks: scene list broadcast, LPT split, every scene transcribed exactly once,
    rank 0 holds all segments sorted by absolute start."""
    port = _free_port()
    mp.spawn(_scene_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "scene_rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "scene_rank1.pt", weights_only=False)
    assert r1["merged"] is None and len(r0["merged"]) == 10
    assert sorted(r0["seen"] + r1["seen"]) == [0.0, 30.0, 40.0, 60.0, 90.0] and r0["seen"] and r1["seen"]
    starts = [s["start"] for s in r0["merged"]]
    assert starts == sorted(starts) and starts[0] == pytest.approx(0.2)
    assert {s["text"].split("@")[0] for s in r0["merged"] if "@" in s["text"]} == {"rank0", "rank1"}


def test_bench_launcher_spawns_its_own_ranks_gloo_dry_run():
    """``python bench.py --gpus 2`` outside torchrun becomes the launcher of two ranks; the CPU dry run (``--simulate``,
    gloo) exercises the launcher, the weight-blob broadcast, the deterministic LPT plan of ``--strong``, the barrier +
    max-over-ranks timing and the gather, and prints ONE JSON line with n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--simulate", "--strong", "--steps", "3",
                          "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong" and line["config"]["backend"] == "gloo"
    assert sum(line["config"]["scenes_per_rank"]) == 40
    # the dry run also drives the REAL RecordingTranscriber / HipFasterWhisperProASR control flow over a stub engine
    # (detect -> LPT share -> pooled transcribe -> gather -> stitch) and asserts the merged transcript equals the one-rank transcript
    flow = line["config"]["control_flow"]
    assert flow["equal_to_one_rank_transcript"] is True and flow["ranks"] == 2 and flow["scenes"] >= 40
    assert sum(flow["segments_per_rank"]) == flow["segments"] and min(flow["segments_per_rank"]) > 0
    assert all(0.0 <= v < 0.05 for v in flow["lpt_imbalance_max_over_mean_minus_1"].values())
    # asking for more GPUs than the node has is an error, not a silent single-rank run
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, timeout=120, env=env, cwd=root)
    assert res.returncode != 0 and "GPU(s) visible" in (res.stderr + res.stdout)
