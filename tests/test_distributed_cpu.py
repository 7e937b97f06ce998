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
