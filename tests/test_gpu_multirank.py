"""Two ranks on the one GPU of the test box (gloo carries the CUDA tensors; RCCL needs one GPU per
rank): per-rank seed blocks -> device render -> in-place all-reduce of the colour buffer through
the zero-copy alias -> ptgpu_scale_and_repack.  Checked against the oracle's two renders combined."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu
W, H, PASSES = 40, 24, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, backend="gloo"):
    sys.path.insert(0, os.path.join(REPO, "path-tracer_amd"))
    import torch
    import torch.distributed as dist
    import multigpu
    import pyptgpu as pt
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = pt.LoadedScene("cornell-large", W, H)
    g = pt.Gpu(0, 0, math=pt.MATH_STRICT)
    g.set_scene(sc)
    g.resize(W, H, pt.seed_block(W, H, rank))
    g.render(PASSES)
    g.finish()
    t = torch.as_tensor(multigpu.DeviceArray(g.colors_device_ptr(), pt.COLOR_STRIDE_FLOATS * W * H), device="cuda:0")
    scale = multigpu.allreduce_mean_(t, world)
    torch.cuda.synchronize()
    g.scale_and_repack(scale)
    np.save(os.path.join(out_dir, f"colors{rank}.npy"), g.colors())
    np.save(os.path.join(out_dir, f"pixels{rank}.npy"), g.pixels())
    g.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu(tmp_path):
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import ctypes as C
    import multigpu
    import oracle_binding as ob
    import pyptgpu as pt
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = [np.load(tmp_path / f"colors{r}.npy") for r in range(2)]
    pix = [np.load(tmp_path / f"pixels{r}.npy") for r in range(2)]
    assert got[0].tobytes() == got[1].tobytes() and (pix[0] == pix[1]).all()
    sc = pt.LoadedScene("cornell-large", W, H)
    own = [ob.Render(sc, pt.seed_block(W, H, r), ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(PASSES).colors
           for r in range(2)]
    want = multigpu.combine_on_host(own)
    assert np.allclose(got[0], want, rtol=0, atol=1e-7)
    # the repacked 8-bit image is the pack of the combined colours
    c = got[0].reshape(H, W, 3)[::-1].reshape(-1, 3)
    for i in (0, 7, W * H // 2, W * H - 1):
        v = pt.Vec3(*[float(x) for x in c[i]])
        z = pt.Vec3(0, 0, 0)
        assert ob.oracle().oracle_accumulate(C.byref(z), C.byref(v), 0, sc.scene.gamma) == pix[0][i]


def test_rccl_group_of_one(tmp_path):
    # the box has one GPU, so RCCL can only be rehearsed with a group of one: the same calls as N > 1
    # (communicator on the device, in-place all-reduce of the aliased colour buffer, repack)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_binding as ob
    import pyptgpu as pt
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), "nccl"), nprocs=1, join=True)
    got = np.load(tmp_path / "colors0.npy")
    sc = pt.LoadedScene("cornell-large", W, H)
    want = ob.Render(sc, pt.seed_block(W, H, 0), ob.SEEDS_KERNEL_LAYOUT, 0).pass_oracle(PASSES).colors
    assert got.tobytes() == want.tobytes()
