"""CPU, world_size 2 over gloo: the host side of the multi-GPU paths -- the real sharding and gather
code of pixsfm_amd.parallel run in two processes.  The solves themselves need the GPU: tests/test_multi_rank_gpu.py
runs the same worker with pxr_ka_solve / pxr_ba_solve inside."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _multi_rank_launch import run_ranks  # noqa: E402
import _multi_rank_worker as worker  # noqa: E402


def test_ka_shards_partition_the_problem_and_gather_is_exact(tmp_path):
    from pixsfm_amd import parallel
    prob = worker.ka_problem()
    n = len(prob["kp"])
    res = run_ranks("ka_plumbing", tmp_path, world=2)
    ids = [r["node_ids"] for r in res]
    assert len(np.intersect1d(ids[0], ids[1])) == 0 and len(ids[0]) + len(ids[1]) == n     # every node owned once
    assert len(ids[0]) > 0 and len(ids[1]) > 0
    edges = np.concatenate([r["edge_ids"] for r in res])
    assert np.array_equal(np.sort(edges), np.arange(len(prob["edge_src"])))                 # every edge on one rank
    n_prob = int(prob["node_problem"].max()) + 1
    assert int(res[0]["n_local_problems"][0]) + int(res[1]["n_local_problems"][0]) == n_prob
    want = prob["kp"] + (np.arange(n)[:, None] + 1) * np.array([1e-3, -2e-3])
    for r in res:                                                                           # the gather is bit-exact
        assert np.array_equal(r["kp"], want) and np.array_equal(r["owned"], np.ones(n))
    # whole sub-problems stay together, and the load is balanced by edge count
    for rank in range(2):
        shard, node_ids = parallel.shard_ka_problem(prob, rank, 2)
        assert set(prob["node_problem"][node_ids]) .isdisjoint(set(prob["node_problem"][np.setdiff1d(np.arange(n), node_ids)]))
        assert np.array_equal(prob["edge_w"][shard["edge_ids"]], shard["edge_w"])
        assert np.array_equal(node_ids[shard["edge_src"]], prob["edge_src"][shard["edge_ids"]])
    loads = [len(r["edge_ids"]) for r in res]
    assert max(loads) <= 1.2 * min(loads)


def test_ba_point_shards_cover_all_observations():
    from pixsfm_amd import parallel
    prob = worker.ba_problem()
    seen = []
    for rank in range(3):
        shard, pt_ids = parallel.shard_ba_problem(prob, rank, 3)
        assert np.array_equal(prob["obs_point"][shard["obs_ids"]], pt_ids[shard["obs_point"]])
        assert np.array_equal(prob["patches"][prob["obs_patch"][shard["obs_ids"]]], shard["patches"])
        assert np.array_equal(shard["qvec"], prob["qvec"]) and len(shard["image_camera"]) == len(prob["image_camera"])
        seen.append(shard["obs_ids"])
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(len(prob["obs_image"])))
