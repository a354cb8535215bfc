"""N > 1 path on CPU: world_size-2 gloo process group, per-rank seed blocks, one all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
    import multigpu
    import oracle_binding as ob
    import pyptgpu as pt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, passes = 24, 16, multigpu.split_passes(6, world)
    sc = pt.LoadedScene("cornell-large", w, h)
    seeds = pt.seed_block(w, h, rank)
    assert multigpu.rank_seed_skip(rank, w, h) == rank * 2 * w * h
    # the oracle stands in for the device here: this test is about the rank logic and the collective
    r = ob.Render(sc, seeds, ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(passes)
    t = torch.from_numpy(r.colors.copy().reshape(-1))
    scale = multigpu.allreduce_mean_(t, world)
    combined = t.numpy() * np.float32(scale)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), combined)
    np.save(os.path.join(out_dir, f"own{rank}.npy"), r.colors)
    np.save(os.path.join(out_dir, f"seeds{rank}.npy"), seeds)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_disjoint_seeds_and_allreduce(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
    import multigpu
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]
    own = [np.load(tmp_path / f"own{r}.npy") for r in range(world)]
    seeds = [np.load(tmp_path / f"seeds{r}.npy") for r in range(world)]
    assert (got[0] == got[1]).all()                       # every rank holds the same combined image
    assert not (seeds[0] == seeds[1]).all()               # disjoint seed streams
    assert not np.allclose(own[0], own[1])                # ... so different samples
    want = multigpu.combine_on_host(own).reshape(-1)
    assert np.allclose(got[0], want, rtol=0, atol=1e-7)
    # rank 0 alone is the single-GPU render
    import pyptgpu as pt
    assert (seeds[0] == pt.seed_block(24, 16, 0)).all()
    assert multigpu.split_passes(1024, 8) == 128 and multigpu.split_passes(10, 4) == 3
