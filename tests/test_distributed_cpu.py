"""N > 1 path on CPU: world_size-2 gloo processes exercise the same work split + collective bench.py uses on GPUs
(cpu_ray_tracer_amd.spp_window / tile_partition / allreduce_accumulator), with the CPU oracle standing in for the
kernels (this is a test; the product path never does that)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ASSETS, REPO, load_crt, scene_path

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(%(repo)r, "tests")); sys.path.insert(0, os.path.join(%(repo)r, "oracle"))
from conftest import load_crt, scene_path, ASSETS
import orc
crt = load_crt()
mode, out = sys.argv[1], sys.argv[2]
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
W, H, F = 96, 64, 3
o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
o.renderer_init(W, H)
if mode == "weak":
    o.set_spp(crt.spp_window(rank, F))
    o.render(F, 1)
else:
    tiles = (W // 16) * (H // 16)
    first, stride, count = crt.tile_partition(rank, world, tiles)
    # the oracle takes contiguous tile ranges: render the owned tiles one by one
    for f in range(F):
        o.set_spp(1 + f)
        for i in range(count):
            o.set_spp(1 + f); o.set_tile_range(first + i * stride, 1); o.render(1, 1)
acc = torch.from_numpy(o.accumulator())
if mode == "strong_reduce":
    crt.reduce_accumulator(acc, dist, 0)          # bench.py's default collective: ncclReduce to rank 0
else:
    crt.allreduce_accumulator(acc, dist)
if rank == 0:
    np.save(out, acc.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def _run(mode, tmp_path, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(repo=REPO, port=port))
    out = str(tmp_path / ("acc_%s.npy" % mode))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), mode, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-2000:]
    return np.load(out)


def test_partition_arithmetic():
    crt = load_crt()
    for world in (1, 2, 3, 8):
        for tiles in (1, 7, 24, 3600):
            owned = []
            for r in range(world):
                first, stride, count = crt.tile_partition(r, world, tiles)
                owned += [first + i * stride for i in range(count)]
            assert sorted(owned) == list(range(tiles))
    assert [crt.spp_window(r, 64) for r in range(4)] == [1, 65, 129, 193]


def test_weak_scaling_two_ranks_gloo(orc, tmp_path):
    """each rank renders its own window of frames; the all-reduced accumulator equals the 2F-frame render up to the
    association of the float sums (<= 1e-4 per sample, the north-star gate)"""
    got = _run("weak", tmp_path, 29611)
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(96, 64)
    o.render(6, 2)
    want = o.accumulator()
    assert np.abs(got - want).max() / 6 <= 1e-4
    assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("mode,port", [("strong", 29612), ("strong_reduce", 29613)])
def test_tile_split_two_ranks_gloo(orc, tmp_path, mode, port):
    """tile ownership split: every pixel is non-zero on one rank only, so the reduced image (all_reduce, or reduce to rank 0 — bench.py's
    default) is the single-rank image exactly"""
    got = _run(mode, tmp_path, port)
    o, _ = orc.load_scene(scene_path("bunny_scene.xml"), 0, ASSETS)
    o.renderer_init(96, 64)
    o.render(3, 2)
    assert np.array_equal(got, o.accumulator())


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("crt_bench", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_bench_workload_text_and_spawn_command():
    """the N > 1 workload string of both splits is built without a formatting error; the self-spawn command targets 127.0.0.1"""
    b = _bench()
    t = b.workload_text("bunny_scene.xml", 0, 1280, 720, 64, 20, 1, 2, "tiles", "RCCL", "reduce")
    assert "dealt round-robin over 2 ranks" in t and "ONE RCCL reduce of the float4 accumulator" in t
    f = b.workload_text("bunny_scene.xml", 0, 1280, 720, 64, 20, 1, 8, "frames", "RCCL", "all_reduce")
    assert "every one of the 8 ranks renders its own 20 windows" in f and "ONE RCCL all-reduce" in f
    assert "ranks" not in b.workload_text("bunny_scene.xml", 0, 1280, 720, 64, 20, 1, 1, "tiles", "RCCL", "reduce")
    cmd = b.spawn_command(4, ["--gpus", "4", "--steps", "20"], port=29777)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "20"]


def test_bench_spawns_its_own_ranks():
    """`bench.py --gpus 2` with no WORLD_SIZE in the environment starts its two ranks itself (gloo rendezvous on 127.0.0.1 here); without a GPU
    every rank stops at the product's "no CPU path" message — after the ranks have found each other — and the launcher hands the failure on"""
    env = dict(os.environ, CRT_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"): env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU-marked twin of this test runs the whole job")
    assert r.returncode != 0
    assert "no CPU path for the product (rank 0 of 2)" in r.stderr and "no CPU path for the product (rank 1 of 2)" in r.stderr, r.stderr[-2000:]


def test_choose_collective_agrees_across_ranks(tmp_path):
    """the collective is decided once, before the warm-up, identically on every rank (probe + all_reduce(MIN) of a success flag)"""
    script = tmp_path / "w.py"
    script.write_text(r"""
import importlib.util, os, sys, torch, torch.distributed as dist
spec = importlib.util.spec_from_file_location("crt_bench", os.path.join(%r, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:29641", rank=int(os.environ["RANK"]), world_size=2)
got = b.choose_collective(dist, torch, "cpu", "reduce")
assert got in ("reduce", "all_reduce")
flags = [None, None]; dist.all_gather_object(flags, got)
assert flags[0] == flags[1], flags
assert b.choose_collective(dist, torch, "cpu", "all_reduce") == "all_reduce"
dist.barrier(); dist.destroy_process_group()
""" % REPO)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", OMP_NUM_THREADS="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-2000:]
