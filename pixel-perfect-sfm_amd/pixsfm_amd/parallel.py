"""Multi-GPU partitioning of the hot path: one process per GPU.

BA: points -- with all their observations, patches and reference descriptors -- are
partitioned over the ranks; images and cameras are replicated.  Direct solver: each rank forms
its partial reduced camera system S_r = U_r - sum_p W_p T_p W_p^T and right-hand side, ONE
all-reduce (sum) of the packed [S | rhs] buffer per linear solve; iterative solver (> 1000
images): one all-reduce of a camera-sized vector per conjugate-gradient iteration.  Plus small
all-reduces of diag(U) | g_c per linearisation and of 16 scalars per LM attempt.  Every rank
then holds the (replicated) camera step and back-substitutes its own points.
KA and reference extraction: sub-problems / points are independent (edges are intra-track only,
base/src/parallel_optimizer.h:77-211 is the reference's own shape), so they are dealt out to the
ranks; no collective during the solve, one gather of disjoint rows at the end.

The collective of the BA path is native: an RCCL communicator owned by the engine's Context
(pxr_comm_init, ncclAllReduce on the context's stream).  torch.distributed is only the side
channel that carries the 128-byte communicator id and the host-side gathers -- and, in the CPU
tests (backend "gloo"), the transport behind the callback form of the collective.
"""
import os

import numpy as np


# ---- process group plumbing ----------------------------------------------------------------
def world(group=None):
    """(rank, world size) of torch.distributed, or (0, 1) when it is not initialised."""
    try:
        import torch.distributed as dist
    except ImportError:
        return 0, 1
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def init_native_comm(ctx, group=None):
    """Give `ctx` its RCCL communicator: rank 0 draws the id (ncclGetUniqueId), torch.distributed carries the
    128 bytes to the other ranks, every rank joins ncclCommInitRank.  Returns (rank, world size)."""
    import torch.distributed as dist
    rank, n = world(group)
    if n == 1:
        return rank, n
    box = [ctx.comm_unique_id() if rank == 0 else None]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group)
    ctx.comm_init(box[0], rank, n)
    return rank, n


def local_device():
    """HIP device of this process: PXR_DEVICE if set, else LOCAL_RANK (one process per GPU), else 0."""
    import os
    for key in ("PXR_DEVICE", "LOCAL_RANK"):
        if os.environ.get(key, "") != "":
            return int(os.environ[key])
    return 0


def ensure_collective(ctx, group=None):
    """Make `ctx` able to take part in the BA collective of `group`.  Backend nccl (one GPU per rank): the context
    gets its native RCCL communicator and None is returned (pxr_ba_solve calls ncclAllReduce itself).  Any other
    backend (gloo: tests, ranks sharing a GPU): the rank is recorded on the context and the all-reduce CALLBACK
    to pass to BAProblem.solve is returned.  One rank: None."""
    import torch.distributed as dist
    rank, n = world(group)
    if n == 1:
        return None
    if dist.get_backend(group) == "nccl":
        if ctx.comm_rank() != (rank, n):
            init_native_comm(ctx, group)
        return None
    if ctx.comm_rank() != (rank, n):
        ctx.comm_set_rank(rank, n)
    return make_allreduce(group, ctx)


def allgather_rows(array, group=None):
    """Every rank contributes an (n_r, ...) array (n_r may differ, trailing shape and dtype agree); returns the list of
    all ranks' arrays on every rank.  Two tensor collectives (row counts, rows padded to the longest) -- no pickling."""
    import torch
    import torch.distributed as dist
    rank, n = world(group)
    array = np.ascontiguousarray(array)
    if n == 1:
        return [array]
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(n)]
    dist.all_gather(counts, torch.tensor([array.shape[0]], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    if longest == 0:
        return [array[:0].copy() for _ in range(n)]
    padded = np.zeros((longest,) + array.shape[1:], dtype=array.dtype)
    padded[:array.shape[0]] = array
    mine = torch.from_numpy(padded).to(dev)
    parts = [torch.empty_like(mine) for _ in range(n)]
    dist.all_gather(parts, mine, group=group)
    return [parts[r][:counts[r]].cpu().numpy() for r in range(n)]


def gather_references(local, group=None):
    """Union of per-rank {point3D_id: Reference} maps with disjoint keys on every rank.  What a Reference carries
    (features/src/references.h:29-72: source observation, 1 x C descriptor, optionally the per-observation descriptors,
    their costs and the visible track) travels as flat arrays -- ids, descriptor rows, per-reference lengths and the
    concatenated variable-length parts -- through allgather_rows (the same collectives whatever form a rank holds); the result
    is a features.ReferenceMap (arrays, objects on demand) when no reference carries variable-length parts, else a dict."""
    from .api.features import ReferenceMap
    rank, n = world(group)
    if n == 1:
        return local if isinstance(local, ReferenceMap) else dict(local)
    packed = local.arrays() if isinstance(local, ReferenceMap) else None
    if packed is not None:                       # the extractor's array form: nothing to flatten
        ids, refs, my_chan = packed[0], None, (packed[2].shape[1] if len(packed[0]) else 0)
    else:
        ids = sorted(local)
        refs = [local[p] for p in ids]
        my_chan = max([r.channels for r in refs], default=0)
    # channel count agreed over the ranks (a rank may hold no reference at all)
    chan = int(allreduce_host(np.array([my_chan], dtype=np.int64), group, op="max")[0])
    head = np.zeros((len(ids), 6), dtype=np.int64)          # id, image_id, point2D_idx, #observations, #costs, #track
    desc = np.zeros((len(ids), chan), dtype=np.float64)
    obs, costs, track = np.zeros((0, chan)), np.zeros(0), np.zeros((0, 2), np.int64)
    if refs is None:
        if len(ids):
            head[:, 0], head[:, 1:3], desc[:] = packed[0], packed[1], packed[2]
    else:
        for k, (p, r) in enumerate(zip(ids, refs)):
            head[k] = (p, r.source[0], r.source[1], len(r.observations), len(r.costs), len(r.track))
            desc[k] = r.descriptor.reshape(-1)
        obs = np.concatenate([np.asarray(o, dtype=np.float64).reshape(1, chan) for r in refs for o in r.observations]
                             or [np.zeros((0, chan))])
        costs = np.array([c for r in refs for c in r.costs], dtype=np.float64)
        track = np.array([tuple(e) for r in refs for e in r.track], dtype=np.int64).reshape(-1, 2)
    parts = [allgather_rows(a, group) for a in (head, desc, obs, costs, track)]
    if all(h[:, 3:].sum() == 0 for h in parts[0]):
        h, d = np.concatenate(parts[0]), np.concatenate(parts[1])
        return ReferenceMap(h[:, 0], h[:, 1:3], d)
    return _unpack_references(parts)


def _unpack_references(parts):
    from .api.features import Reference
    out = {}
    for h, d, o, c, t in zip(*parts):
        o_at = np.concatenate([[0], np.cumsum(h[:, 3])])
        c_at = np.concatenate([[0], np.cumsum(h[:, 4])])
        t_at = np.concatenate([[0], np.cumsum(h[:, 5])])
        for k in range(len(h)):
            out[int(h[k, 0])] = Reference(int(h[k, 1]), int(h[k, 2]), d[k].copy(),
                                          observations=[o[i].copy() for i in range(o_at[k], o_at[k + 1])],
                                          costs=c[c_at[k]:c_at[k + 1]].tolist(),
                                          track=[(int(a), int(b)) for a, b in t[t_at[k]:t_at[k + 1]]])
    return out


class _CudaArrayView:
    """Wraps a raw device pointer so torch.as_tensor can alias it (no copy)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def make_allreduce(group=None, ctx=None):
    """all-reduce(sum) CALLBACK for BAProblem.solve over torch.distributed -- the transport-agnostic form of the
    collective (gloo in the tests, or any backend torch offers).  On MI355X prefer init_native_comm(): the
    solver then calls ncclAllReduce itself and no Python runs inside the LM loop.

    nccl backend: the collective is enqueued on torch's current stream, which must be the stream the engine's
    Context was created on.  gloo backend (several ranks sharing one GPU / CPU transport): the buffer is staged
    through the host; pass `ctx` so the copies are ordered with the engine's stream."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)

    def allreduce(ptr, count):
        t = torch.as_tensor(_CudaArrayView(ptr, count), device="cuda")
        if backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return
        if ctx is not None:
            ctx.sync()
        else:
            torch.cuda.synchronize()
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        torch.cuda.synchronize()

    return allreduce


def allreduce_host(array, group=None, op="sum"):
    """Reduce a host numpy array over the ranks (in place; gathers of disjoint rows are sums with zeros elsewhere)."""
    import torch
    import torch.distributed as dist
    rank, n = world(group)
    if n == 1 and not (os.environ.get("PXR_FORCE_COLLECTIVE", "0") not in ("", "0") and dist.is_available() and dist.is_initialized()):
        return array          # (PXR_FORCE_COLLECTIVE=1: a one-rank group still goes through the backend -- tests)
    rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}[op]
    t = torch.from_numpy(array)
    if dist.get_backend(group) == "nccl":
        d = t.cuda()
        dist.all_reduce(d, op=rop, group=group)
        t.copy_(d.cpu())
    else:
        dist.all_reduce(t, op=rop, group=group)
    return array


# ---- BA: points sharded, cameras replicated ---------------------------------------------------
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
    if problem.get("refs") is not None:
        shard["refs"] = np.asarray(problem["refs"])[lo:hi].copy()
    shard["obs_ids"] = sel
    return shard, pt_ids


def gather_rows(local_rows, row_ids, n_rows, group=None):
    """Every rank contributes `local_rows` for the disjoint `row_ids`; returns the (n_rows, ...) array of all ranks'
    rows on every rank (rows nobody owns stay zero).  One all-reduce(sum) with zeros elsewhere: exact."""
    local_rows = np.asarray(local_rows)
    full = np.zeros((int(n_rows),) + local_rows.shape[1:], dtype=local_rows.dtype)
    full[np.asarray(row_ids, dtype=np.int64)] = local_rows
    return allreduce_host(full, group)


# ---- KA: independent sub-problems dealt to the ranks ---------------------------------------------------
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


def shard_ka_problem(problem, rank, world):
    """The sub-problems of a flat KA problem (ka_engine.KAProblem's dict) owned by `rank`, as a flat problem of its
    own: nodes / edges / patches re-indexed, sub-problems renumbered 0 .. n_local - 1.  Returns
    (shard dict, global node ids of the shard's nodes).  Nodes outside every sub-problem (label -1) belong to nobody."""
    node_problem = np.asarray(problem["node_problem"], dtype=np.int64)
    edge_src = np.asarray(problem["edge_src"], dtype=np.int64)
    edge_dst = np.asarray(problem["edge_dst"], dtype=np.int64)
    n_prob = int(node_problem.max()) + 1 if len(node_problem) else 0
    ep = node_problem[edge_src] if len(edge_src) else np.zeros(0, np.int64)
    sizes = np.bincount(ep[ep >= 0], minlength=n_prob) if n_prob else np.zeros(0, np.int64)
    unary_node = problem.get("unary_node")
    if unary_node is not None and len(unary_node):
        up = node_problem[np.asarray(unary_node, dtype=np.int64)]
        sizes = sizes + np.bincount(up[up >= 0], minlength=n_prob)
    owner = assign_problems_to_ranks(sizes, world)
    mine = np.nonzero(owner == rank)[0]
    new_label = np.full(n_prob + 1, -1, dtype=np.int64)
    new_label[mine] = np.arange(len(mine))
    node_sel = np.nonzero((node_problem >= 0) & (new_label[node_problem] >= 0))[0]
    new_node = np.full(len(node_problem), -1, dtype=np.int64)
    new_node[node_sel] = np.arange(len(node_sel))
    edge_sel = np.nonzero(new_node[edge_src] >= 0)[0] if len(edge_src) else np.zeros(0, np.int64)
    node_patch = np.asarray(problem["node_patch"], dtype=np.int64)[node_sel]
    shard = dict(kp=np.asarray(problem["kp"], dtype=np.float64)[node_sel].copy(),
                 node_patch=np.arange(len(node_sel), dtype=np.int64),
                 node_const=np.asarray(problem["node_const"], dtype=np.uint8)[node_sel],
                 node_problem=new_label[node_problem[node_sel]].astype(np.int32),
                 edge_src=new_node[edge_src[edge_sel]].astype(np.int32),
                 edge_dst=new_node[edge_dst[edge_sel]].astype(np.int32),
                 edge_w=np.asarray(problem["edge_w"], dtype=np.float64)[edge_sel],
                 patch_ids=node_patch, edge_ids=edge_sel)
    if problem.get("node_track") is not None:
        shard["node_track"] = np.asarray(problem["node_track"], dtype=np.int64)[node_sel]
    if len(edge_sel) and (shard["edge_dst"] < 0).any():
        raise ValueError("an edge connects two different sub-problems")
    for k in ("patches", "corners", "scales"):
        if k in problem and problem[k] is not None:
            shard[k] = np.asarray(problem[k])[node_patch]
    if unary_node is not None and len(unary_node):
        usel = np.nonzero(new_node[np.asarray(unary_node, dtype=np.int64)] >= 0)[0]
        shard["unary_node"] = new_node[np.asarray(unary_node, dtype=np.int64)[usel]].astype(np.int32)
        shard["unary_ref"] = np.asarray(problem["unary_ref"])[usel]
        if problem.get("unary_w") is not None:
            shard["unary_w"] = np.asarray(problem["unary_w"])[usel]
    return shard, node_sel


def ka_solve_sharded(ctx, problem, cfg, loss, bound=4.0, options=None, group=None, arena=None):
    """KA over the ranks of `group`: this rank solves its share of the sub-problems on its GPU (pxr_ka_solve), then
    the refined keypoints (disjoint rows) are gathered.  `problem` is the GLOBAL flat problem on every rank (host
    arrays; `patches` may be omitted when `arena` -- an arena holding this rank's patches in shard order, built by
    the caller from shard["patch_ids"] -- is given).  Returns (keypoints of all nodes, summed summary dict)."""
    from .engine import PatchArena
    from .ka_engine import KAProblem
    rank, n = world(group)
    shard, node_ids = shard_ka_problem(problem, rank, n)
    own_arena = arena is None
    if own_arena:
        arena = PatchArena.from_numpy(ctx, shard["patches"], shard["corners"], shard["scales"])
    kp = np.asarray(problem["kp"], dtype=np.float64).copy()
    summary = dict(iterations=0, num_successful=0, initial_cost=0.0, final_cost=0.0, total_ms=0.0)
    if len(node_ids):
        ka = KAProblem(ctx, arena, shard)
        total, _ = ka.solve(cfg, loss, bound=bound, options=options)
        local = ka.keypoints()
        for k in summary:
            summary[k] = total[k]
    else:
        local = np.zeros((0, 2))
    if own_arena:
        arena.close()
    if n > 1:
        owned = np.zeros(len(kp), dtype=np.float64)
        owned[node_ids] = 1.0
        gathered = gather_rows(local, node_ids, len(kp), group)
        owned = allreduce_host(owned, group)
        kp = np.where(owned[:, None] > 0, gathered, kp)
        s = np.array([summary["iterations"], summary["num_successful"], summary["initial_cost"], summary["final_cost"]],
                     dtype=np.float64)
        s = allreduce_host(s, group)
        t = allreduce_host(np.array([summary["total_ms"]]), group)      # summed device time (AccumulateSummaries)
        summary.update(iterations=int(s[0]), num_successful=int(s[1]), initial_cost=float(s[2]), final_cost=float(s[3]),
                       total_ms=float(t[0]))
    else:
        kp[node_ids] = local
    return kp, summary


def compute_references_sharded(ctx, arena, shard, pt_ids, n_points, cfg, loss, iters=100, group=None):
    """Reference extraction on this rank's point shard (pxr_ba_compute_references, independent per point) followed by
    a gather of the 128-double references, for hosts that need every reference everywhere (the BA itself does not:
    observations follow their point).  Returns (refs of all points, chosen observation per point as GLOBAL ids)."""
    from .engine import BAProblem
    ba = BAProblem(ctx, arena, shard)
    chosen, _ = ba.compute_references(cfg, loss, iters=iters)
    refs = ba.d["refs"].download()
    glob = np.where(chosen >= 0, np.asarray(shard["obs_ids"])[np.maximum(chosen, 0)], -1).astype(np.float64)
    all_refs = gather_rows(refs, pt_ids, n_points, group)
    all_obs = gather_rows(glob + 1.0, pt_ids, n_points, group) - 1.0          # rows nobody owns: -1
    return all_refs, all_obs.astype(np.int64)
