"""Device-resident keypoint-adjustment problem (pxr_ka_view) and its solve / eval calls."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KaView, check
from .engine import lm_options


def _csr(labels, n_groups):
    """Stable grouping: returns (ptr[n_groups + 1], ids sorted by label then by id)."""
    labels = np.asarray(labels, dtype=np.int64)
    order = np.argsort(labels, kind="stable")
    counts = np.bincount(labels, minlength=n_groups)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return ptr, order.astype(np.int32)


def pack_tracks_into_problems(track_labels, max_per_problem, track_edge_counts=None):
    """Sub-problem label per graph node: whole tracks are packed into sub-problems holding at most
    `max_per_problem` keypoints (or edges, when `track_edge_counts[track]` gives each track's weight).
    Drop-in for `find_problem_labels` of pixsfm/keypoint_adjustment/main.py:13-57 -- same (labels, sizes)
    return value -- which A15's ParallelOptimizer (base/src/parallel_optimizer.h:77-211) and our
    one-workgroup-per-sub-problem solver both take.  The rule being reproduced:
      * tracks are visited from the heaviest to the lightest, equal weights in order of first appearance
        (edge counts: ascending track id);
      * a track of weight >= the capacity opens a sub-problem of its own;
      * any other track goes to the first sub-problem with room, the scan starting at the sub-problem the
        previous track of the SAME weight went to (and at sub-problem 0 for the first track of a weight);
        without room anywhere it opens a new one;
      * capacity -1 = the heaviest track.
    tests/test_cabi_and_host.py checks it against vectors produced by the reference's own function."""
    labels = np.asarray(track_labels, dtype=np.int64).reshape(-1)
    if track_edge_counts is None:
        track_ids, first_seen, weight = np.unique(labels, return_index=True, return_counts=True)
        visit = np.lexsort((first_seen, -weight))             # weight descending, ties by first appearance
    else:
        weight = np.asarray(track_edge_counts, dtype=np.int64).reshape(-1)
        track_ids = np.arange(len(weight), dtype=np.int64)
        visit = np.argsort(-weight, kind="stable")
    n_tracks = len(track_ids)
    if n_tracks == 0:
        raise ValueError("no tracks to pack")
    if track_ids[0] < 0 or track_ids[-1] >= n_tracks or (len(labels) and (labels.min() < 0 or labels.max() >= n_tracks)):
        raise ValueError("track labels must be the contiguous ids 0 .. n_tracks - 1")
    capacity = int(weight.max()) if max_per_problem == -1 else int(max_per_problem)
    occupancy = []                                            # keypoints (edges) per sub-problem
    problem_of_track = np.full(n_tracks, -1, dtype=np.int64)
    cursor, weight_class = 0, None
    for t in visit:
        w = int(weight[t])
        if w != weight_class:
            weight_class, cursor = w, 0
        slot = len(occupancy)
        if w < capacity:
            slot = cursor
            while slot < len(occupancy) and occupancy[slot] + w > capacity:
                slot += 1
        if slot == len(occupancy):
            occupancy.append(0)
        occupancy[slot] += w
        problem_of_track[track_ids[t]] = slot
        cursor = slot
    return problem_of_track[labels].tolist(), occupancy


CHUNK_KPS = 50          # keypoints per chunk of a large label group (the size the solve kernel's LDS caches are made for)
CHUNK_FROM = 96         # label groups with more nodes than this are chunked (KA_NODE_CACHE of csrc/pxr_ka.hip)
MAX_CHUNKS = 448        # chunks of ONE group that a launch keeps resident (7/8 of 2 workgroups x 256 CUs)


def chunk_label_groups(node_problem, node_track, chunk_kps=CHUNK_KPS, chunk_from=CHUNK_FROM, max_chunks=MAX_CHUNKS):
    """Label groups (one Ceres problem each) -> chunks of whole tracks for the solve kernel (pxr_ka_view.d_prob_group).
    Returns (chunk label per node, group id per chunk).  Groups of at most `chunk_from` nodes stay one chunk; tracks are taken
    in order of first appearance, a chunk is closed when the next
    track would not fit `chunk_kps` keypoints (a longer track is a chunk of its own); a group that would need more than `max_chunks`
    chunks gets larger chunks (split_in_subproblems = false on a whole scene).  Nodes labelled -1 stay -1."""
    node_problem = np.asarray(node_problem, dtype=np.int64)
    node_track = np.asarray(node_track, dtype=np.int64)
    out = np.full(len(node_problem), -1, dtype=np.int64)
    groups = []
    n_chunks = 0
    for g in range(int(node_problem.max()) + 1 if len(node_problem) else 0):
        idx = np.flatnonzero(node_problem == g)
        if len(idx) == 0:
            groups.append(g); n_chunks += 1                               # an empty group keeps its (empty) sub-problem
            continue
        if len(idx) <= chunk_from:
            out[idx] = n_chunks; groups.append(g); n_chunks += 1
            continue
        tr = node_track[idx]
        _, first, inv, cnt = np.unique(tr, return_index=True, return_inverse=True, return_counts=True)
        order = np.argsort(first, kind="stable")                          # tracks by first appearance
        chunk_of_track = np.empty(len(order), dtype=np.int64)
        cap = chunk_kps
        for _ in range(8):
            fill, local = 0, 0
            for t in order:
                if fill > 0 and fill + cnt[t] > cap:
                    local += 1; fill = 0
                chunk_of_track[t] = local
                fill += cnt[t]
            if local + 1 <= max_chunks:
                break
            # more chunks than one launch keeps resident: larger chunks (beyond ~96 keypoints a chunk's metadata no longer fits
            # the kernel's LDS caches and its LM state moves to global memory -- slower per chunk, still one workgroup each)
            cap = max(cap + 1, int(np.ceil(len(idx) * 1.05 / max_chunks)))
        if local + 1 > max_chunks:                                        # (tracks too long to balance): the single-workgroup form
            out[idx] = n_chunks; groups.append(g); n_chunks += 1
            continue
        out[idx] = n_chunks + chunk_of_track[inv]
        groups += [g] * (local + 1)
        n_chunks += local + 1
    return out, np.asarray(groups, dtype=np.int32)


class KAProblem:
    """problem: dict with kp (n,2), node_patch, node_const, node_problem, edge_src, edge_dst, edge_w.
    Optional node_track (n,): the track of every node.  With it, a label group (node_problem) of more than CHUNK_FROM nodes is
    handed to the solver as chunks of whole tracks that share the group's trust region (pxr_ka_view.d_prob_group): the
    decisions of ONE Ceres problem, on as many workgroups as the group has chunks.  `problem_group` then maps the solver's
    sub-problems to the caller's labels; summaries(...) aggregates per label.
    An edge belongs to the sub-problem of its source node (edges are intra-track and a track
    lives in exactly one sub-problem, keypoint_adjustment/main.py:13-57).
    Optional unary reference terms (localization QKA, query_keypoint_optimizer.h:122-139):
    unary_node (m,), unary_ref (m, C), unary_w (m,) or None; a term belongs to its node's sub-problem."""

    def __init__(self, ctx, arena, problem, max_chunks=MAX_CHUNKS):
        self.ctx, self.arena = ctx, arena
        self._problem = problem
        self.d = {}
        self._build(max_chunks)

    def _build(self, max_chunks):
        ctx, arena = self.ctx, self.arena
        g = self._problem
        self.n_nodes = len(g["kp"])
        self.n_edges = len(g["edge_src"])
        node_problem = np.asarray(g["node_problem"], dtype=np.int64)
        if self.n_nodes and node_problem.min() < -1:
            raise ValueError("problem labels are >= 0, or -1 for nodes outside every sub-problem (run_subset)")
        self.n_labels = int(node_problem.max()) + 1 if self.n_nodes else 0
        self.problem_group = None
        if g.get("node_track") is not None and self.n_nodes and np.bincount(node_problem[node_problem >= 0], minlength=1).max() > CHUNK_FROM:
            node_problem, self.problem_group = chunk_label_groups(node_problem, g["node_track"], max_chunks=max_chunks)
            if len(self.problem_group) == self.n_labels:
                self.problem_group = None                       # nothing was split
        self.n_problems = int(node_problem.max()) + 1 if self.n_nodes else 0
        if self.problem_group is not None:
            self.n_problems = len(self.problem_group)
        edge_src = np.asarray(g["edge_src"], dtype=np.int32)
        edge_dst = np.asarray(g["edge_dst"], dtype=np.int32)
        if self.n_edges and not np.array_equal(node_problem[edge_src], node_problem[edge_dst]):
            raise ValueError("an edge connects two different sub-problems")
        # label -1: the node / edge takes part in no sub-problem (the extra group is dropped)
        node_ptr, nodes = _csr(np.where(node_problem < 0, self.n_problems, node_problem), self.n_problems + 1)
        node_ptr, nodes = node_ptr[:-1], nodes[:node_ptr[-2]]
        ep = node_problem[edge_src] if self.n_edges else np.zeros(0, np.int64)
        edge_ptr, edges = _csr(np.where(ep < 0, self.n_problems, ep), self.n_problems + 1)
        edge_ptr, edges = edge_ptr[:-1], edges[:edge_ptr[-2]]
        self.n_unary = len(g["unary_node"]) if g.get("unary_node") is not None else 0
        kp_dev = self.d.get("kp")          # (a rebuild with other chunks keeps the keypoints the caller has on the device)
        self.d = {
            "kp": kp_dev if kp_dev is not None else ctx.to_device(g["kp"], np.float64),
            "node_patch": ctx.to_device(g["node_patch"], np.int64),
            "node_const": ctx.to_device(g["node_const"], np.uint8),
            "edge_src": ctx.to_device(edge_src, np.int32),
            "edge_dst": ctx.to_device(edge_dst, np.int32),
            "edge_w": ctx.to_device(g["edge_w"], np.float64),
            "node_ptr": ctx.to_device(node_ptr, np.int64), "nodes": ctx.to_device(nodes, np.int32),
            "edge_ptr": ctx.to_device(edge_ptr, np.int64), "edges": ctx.to_device(edges, np.int32),
        }
        d = self.d
        self.view = KaView(self.n_nodes, d["kp"].ptr, d["node_patch"].ptr, d["node_const"].ptr, self.n_edges,
                           d["edge_src"].ptr, d["edge_dst"].ptr, d["edge_w"].ptr, self.n_problems,
                           d["node_ptr"].ptr, d["nodes"].ptr, d["edge_ptr"].ptr, d["edges"].ptr,
                           0, None, None, None, None, None, None)
        if self.problem_group is not None:
            d["prob_group"] = ctx.to_device(self.problem_group, np.int32)
            self.view.d_prob_group = d["prob_group"].ptr
        if self.n_unary:
            unary_node = np.asarray(g["unary_node"], dtype=np.int32)
            unary_ref = np.ascontiguousarray(g["unary_ref"], dtype=np.float64)
            if unary_ref.shape != (self.n_unary, arena.C):
                raise ValueError("unary_ref must be (n_unary, CHANNELS)")
            if unary_node.min() < 0 or unary_node.max() >= self.n_nodes:
                raise ValueError("unary_node out of range")
            up = node_problem[unary_node]
            u_ptr, u_ids = _csr(np.where(up < 0, self.n_problems, up), self.n_problems + 1)
            u_ptr, u_ids = u_ptr[:-1], u_ids[:u_ptr[-2]]
            d["unary_node"] = ctx.to_device(unary_node, np.int32)
            d["unary_ref"] = ctx.to_device(unary_ref, np.float64)
            d["unary_ptr"] = ctx.to_device(u_ptr, np.int64)
            d["unary_ids"] = ctx.to_device(u_ids, np.int32)
            v = self.view
            v.n_unary = self.n_unary
            v.d_unary_node, v.d_unary_ref = d["unary_node"].ptr, d["unary_ref"].ptr
            v.d_prob_unary_ptr, v.d_prob_unary = d["unary_ptr"].ptr, d["unary_ids"].ptr
            if g.get("unary_w") is not None:
                d["unary_w"] = ctx.to_device(g["unary_w"], np.float64)
                v.d_unary_w = d["unary_w"].ptr
        self.problem_sizes = np.diff(edge_ptr)

    def eval(self, cfg, loss, materialize=False, out=None):
        """Per-edge costs (+ residuals / Jacobians with `materialize`); `out` re-uses a device array for the costs."""
        ctx = self.ctx
        cost = out if out is not None else ctx.empty((self.n_edges,), np.float64)
        r = J1 = J2 = None
        if materialize:
            r = ctx.empty((self.n_edges, self.arena.C), np.float64)
            J1 = ctx.empty((self.n_edges, self.arena.C, 2), np.float64)
            J2 = ctx.empty((self.n_edges, self.arena.C, 2), np.float64)
        check(ctx.lib.pxr_ka_eval(ctx.handle, self.arena.handle, C.byref(self.view), C.byref(cfg), C.byref(loss),
                                  cost.ptr, r.ptr if r else None, J1.ptr if J1 else None, J2.ptr if J2 else None),
              "pxr_ka_eval")
        return cost, r, J1, J2

    def solve(self, cfg, loss, bound=4.0, options=None, per_problem=False):
        """Refines the keypoints in place on the device; returns (total summary dict, per-problem list|None)."""
        ctx = self.ctx
        opts = options or lm_options(parameter_tolerance=1e-5)
        total = _lib.LMSummary()
        for attempt in range(3):
            arr = (_lib.LMSummary * max(1, self.n_problems))() if per_problem else None
            try:
                check(ctx.lib.pxr_ka_solve(ctx.handle, self.arena.handle, C.byref(self.view), C.byref(cfg), C.byref(loss),
                                           C.c_double(bound), C.byref(opts), C.cast(arr, C.c_void_p) if arr else None,
                                           C.byref(total)), "pxr_ka_solve")
                break
            except _lib.PixsfmHipError as e:
                # a chunked label group needs all its workgroups resident at once; how many a launch keeps resident depends on the
                # chunks' LDS footprint, which the library knows only after its set-up pass: fewer, larger chunks and again
                import re
                m = re.search(r"resident \((\d+) workgroups\)", str(e))
                if self.problem_group is None or m is None or attempt == 2:
                    raise
                self._build(max(1, int(m.group(1)) * 7 // 8 - 1) if attempt == 0 else 1)
        per = [arr[i].as_dict() for i in range(self.n_problems)] if per_problem else None
        if per is not None and self.problem_group is not None:
            per = self.summaries_per_label(per)
        return total.as_dict(), per

    def summaries_per_label(self, per_chunk):
        """per-chunk summaries -> one per label group of the caller: the group's counts (every chunk reports them), costs and
        unknowns added up"""
        out = [None] * self.n_labels
        for s, gid in zip(per_chunk, self.problem_group):
            if out[gid] is None:
                out[gid] = dict(s)
            else:
                for k in ("initial_cost", "final_cost", "num_camera_unknowns", "linear_iterations"):
                    if k in s:
                        out[gid][k] += s[k]
        return out

    def keypoints(self):
        return self.d["kp"].download()
