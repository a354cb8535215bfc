"""Multi-GPU split of a progressive render: passes are independent given their seed
pair, so rank g renders its share of the passes from its own block of rand()-compatible seeds into
a private running-mean buffer, and the buffers are combined with ONE all-reduce (RCCL over xGMI when
the process group is "nccl"; gloo in the CPU tests).  No other data-path collective exists.

Equal pass counts per rank make the mean of the per-rank means the global mean:
    mean = (1/G) * sum_g mean_g
"""
import numpy as np


def rank_seed_skip(rank, width, height):
    """Rank g takes the g-th consecutive block of 2*W*H rand() outputs (rank 0 = single GPU)."""
    return rank * 2 * width * height


def split_passes(total_passes, world):
    """Passes per rank; the combine step needs them equal, so the total is rounded up."""
    return (total_passes + world - 1) // world


class DeviceArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def allreduce_mean_(colors_tensor, world):
    """In-place: colors <- sum over ranks; returns the scale (1/world) still to be applied."""
    import torch.distributed as dist
    if world > 1 or (dist.is_available() and dist.is_initialized()):   # a group of one still goes through RCCL
        dist.all_reduce(colors_tensor, op=dist.ReduceOp.SUM)
    return 1.0 / world


def combine_on_host(per_rank_means):
    """Reference combine used by the tests: float32 sum in rank order, then one multiply."""
    acc = np.zeros_like(per_rank_means[0], dtype=np.float32)
    for m in per_rank_means:
        acc = acc + m.astype(np.float32)
    return acc * np.float32(1.0 / len(per_rank_means))
