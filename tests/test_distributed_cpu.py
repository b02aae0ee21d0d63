"""N>1 path on CPU: world_size-2 (and 3) gloo processes shard the image by interleaved row
strips, each produces its own rows, one gather assembles the image on rank 0, which must
equal the 1-rank image byte for byte.  The per-rank rows come from the CPU oracle here
(there is no GPU in this container); on the GPU box tests/test_gpu_parity.py checks the
same invariant with the HIP renderer behind rtiow_set_shard."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT, compact


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, S, B, strip, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracingincuda_amd as rt
    from raytracingincuda_amd.distributed import StripGather
    from tests.oracle_lib import Oracle
    orc = Oracle()
    sc = compact(orc.build_scene(3, 32))
    cam = rt.camera(32, W, H, S, B)
    g = StripGather(W, H, rank, world, strip, torch.float32, "cpu")
    rows = rt.shard_rows(H, rank, world, strip)
    local = g.local_view().numpy()
    for k, row in enumerate(rows):                      # stand-in for the HIP renderer's shard
        local[k] = orc.render(32, sc, cam, 1227, int(row), int(row) + 1)[0][0]
    full = g.gather()
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,strip", [(2, 40, 8), (3, 37, 4)])
def test_strip_gather_equals_single_rank_image(world, H, strip, tmp_path, oracle, native):
    W, S, B = 48, 2, 6
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, S, B, strip, out), nprocs=world, join=True)
    got = np.load(out)
    want, _ = oracle.render(32, compact(oracle.build_scene(3, 32)), native.camera(32, W, H, S, B), 1227)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_strip_gather_single_rank(native):
    from raytracingincuda_amd.distributed import StripGather
    g = StripGather(8, 20, 0, 1, 8, torch.float32, "cpu")
    g.local_view().copy_(torch.arange(20 * 8 * 3, dtype=torch.float32).reshape(20, 8, 3))
    assert torch.equal(g.gather(), torch.arange(20 * 8 * 3, dtype=torch.float32).reshape(20, 8, 3))


def _one_rank_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracingincuda_amd.distributed import StripGather
    g = StripGather(8, 20, 0, 1, 4, torch.float32, "cpu", always_collective=True)
    g.local_view().copy_(torch.arange(20 * 8 * 3, dtype=torch.float32).reshape(20, 8, 3))
    full = g.gather()                                   # dist.gather in a one-rank group (what bench.py does under torchrun at N=1)
    np.save(out_path, full.numpy())
    dist.destroy_process_group()


def test_one_rank_group_still_runs_the_collective(tmp_path, native):
    out = str(tmp_path / "one.npy")
    mp.spawn(_one_rank_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    assert np.array_equal(np.load(out), np.arange(20 * 8 * 3, dtype=np.float32).reshape(20, 8, 3))
