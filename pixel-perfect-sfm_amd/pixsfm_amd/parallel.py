"""Multi-GPU partitioning of the hot path: one process per GPU.

BA: points -- with all their observations, patches and reference descriptors -- are
partitioned over the ranks; images and cameras are replicated.  Each rank forms its partial
reduced camera system S_r = U_r - sum_p W_p T_p W_p^T and right-hand side; ONE all-reduce
(sum) of the packed [S | rhs] buffer over RCCL/xGMI per linear solve, plus small all-reduces
of diag(U)/g_c per linearisation and of 8 scalars per LM attempt.  Every rank then solves
the (replicated) reduced system and back-substitutes its own points.
KA: tracks are independent (edges are intra-track only), so problems are simply dealt out
to ranks and no collective is needed during the solve.

torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) is plumbing.
"""
import numpy as np


def balanced_ranges(weights, world):
    """Contiguous ranges [lo, hi) of items with ~equal total weight (observations per point)."""
    w = np.asarray(weights, dtype=np.int64)
    csum = np.concatenate([[0], np.cumsum(w)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csum, total * r / world, side="left")))
    cuts.append(len(w))
    cuts = np.maximum.accumulate(np.minimum(cuts, len(w)))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def shard_ba_problem(problem, rank, world):
    """Return (shard dict, global ids of the shard's points).  The shard keeps every image and
    camera and re-indexes points / patches locally."""
    obs_point = np.asarray(problem["obs_point"])
    n_points = len(problem["xyz"])
    counts = np.bincount(obs_point, minlength=n_points)
    lo, hi = balanced_ranges(counts, world)[rank]
    pt_ids = np.arange(lo, hi)
    sel = np.nonzero((obs_point >= lo) & (obs_point < hi))[0]
    shard = dict(problem)
    shard["obs_image"] = np.asarray(problem["obs_image"])[sel]
    shard["obs_point"] = (obs_point[sel] - lo).astype(np.int32)
    patch_ids = np.asarray(problem["obs_patch"])[sel]
    shard["obs_patch"] = np.arange(len(sel), dtype=np.int64)
    for k in ("patches", "corners", "scales"):
        if k in problem and problem[k] is not None:
            shard[k] = np.asarray(problem[k])[patch_ids]
    shard["xyz"] = np.asarray(problem["xyz"])[lo:hi].copy()
    shard["refs"] = np.asarray(problem["refs"])[lo:hi].copy()
    return shard, pt_ids


class _CudaArrayView:
    """Wraps a raw device pointer so torch.as_tensor can alias it (no copy)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def make_allreduce(group=None):
    """all-reduce(sum) callback for BAProblem.solve over torch.distributed (RCCL on MI355X).
    The collective is enqueued on torch's current stream, which must be the stream the
    engine's Context was created on."""
    import torch
    import torch.distributed as dist

    def allreduce(ptr, count):
        t = torch.as_tensor(_CudaArrayView(ptr, count), device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    return allreduce


def assign_problems_to_ranks(problem_sizes, world):
    """KA: longest-processing-time assignment of independent sub-problems (sizes = #edges) to
    ranks; returns rank index per problem."""
    order = np.argsort(-np.asarray(problem_sizes, dtype=np.int64), kind="stable")
    load = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(problem_sizes), dtype=np.int32)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += int(problem_sizes[i])
    return owner
