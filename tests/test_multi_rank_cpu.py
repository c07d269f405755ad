"""CPU, world_size 2 over gloo: the host side of the multi-GPU paths -- the real sharding and gather
code of pixsfm_amd.parallel run in two processes.  The solves themselves need the GPU: tests/test_zz_multi_rank_gpu.py
runs the same worker with pxr_ka_solve / pxr_ba_solve inside."""
import os
import sys

import numpy as np
import pytest

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


def test_references_gather_over_two_and_three_ranks(tmp_path):
    """Real Reference objects (source / track / observations / costs) through parallel.gather_references on gloo: every
    rank ends with the union, field for field (the round-2 regression: un-pickling TrackElementTuple in all_gather_object)."""
    want = worker.pack_references(worker.make_references())
    for world in (2, 3):
        d = tmp_path / ("w%d" % world)
        d.mkdir()
        res = run_ranks("refs_gather", d, world=world)
        keep = np.arange(len(want["ids"])) % world != 1 if world == 3 else np.ones(len(want["ids"]), bool)      # rank 1 of 3 contributed nothing
        full = worker.make_references()
        expect = worker.pack_references({p: full[p] for p in want["ids"][keep]})
        for r in res:
            for k in expect:
                assert np.array_equal(r[k], expect[k]), k
            assert r["typed"].all()


def test_api_containers_survive_pickle_and_deepcopy():
    import copy
    import pickle
    from pixsfm_amd.api import features
    refs = worker.make_references()
    for clone in (pickle.loads(pickle.dumps(refs)), copy.deepcopy(refs)):
        a, b = worker.pack_references(refs), worker.pack_references(clone)
        assert all(np.array_equal(a[k], b[k]) for k in a) and b["typed"].all()
    e = pickle.loads(pickle.dumps(features.TrackElementTuple(3, 4)))
    assert (e.image_id, e.point2D_idx) == (3, 4) and e == (3, 4) and isinstance(e, features.TrackElementTuple)
    t = copy.deepcopy(features.TrackList([features.TrackElementTuple(1, 2)]))
    assert t.length() == 1 and t.elements[0].image_id == 1
    patch = features.FeaturePatch(np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4), (5, 6), (0.5, 0.25))
    fmap = features.FeatureMap(); fmap.patches[9] = patch
    fset = features.FeatureSet({"a.jpg": fmap})
    man = pickle.loads(pickle.dumps(features.FeatureManager([fset])))
    got = man.fsets[0].fmaps["a.jpg"].patches[9] if hasattr(man, "fsets") else None
    if got is not None:
        assert np.array_equal(got.data, patch.data) and np.array_equal(got.corner, patch.corner) and np.array_equal(got.scale, patch.scale)


def test_reference_map_is_a_lazy_dict_of_references():
    """features.ReferenceMap (what ReferenceExtractor.run returns by default): dict behaviour, objects on demand, edits and
    assignments visible to descriptor_matrix (what the optimiser reads), pickle round trip."""
    import pickle
    from pixsfm_amd.api import features
    rng = np.random.default_rng(0)
    ids, src, desc = [5, 9, 12], np.array([[1, 10], [2, 20], [3, 30]]), rng.normal(size=(3, 8))
    m = features.ReferenceMap(ids, src, desc)
    assert len(m) == 3 and sorted(m) == ids and 9 in m and 7 not in m and list(m.keys()) == ids
    r = m[9]
    assert isinstance(r, features.Reference) and tuple(r.source) == (2, 20) and np.array_equal(r.descriptor.reshape(-1), desc[1]) and m[9] is r
    with pytest.raises(KeyError):
        m[7]
    assert np.array_equal(m.descriptor_matrix([12, 5]), desc[[2, 0]])
    r.descriptor = np.full((1, 8), 2.0)                       # an edited object wins
    m[40] = features.Reference(4, 44, np.arange(8.0))         # so does an assigned one, also under a new key
    assert np.array_equal(m.descriptor_matrix([9, 40]), np.stack([np.full(8, 2.0), np.arange(8.0)])) and len(m) == 4
    a = m.arrays()
    assert a[0].tolist() == [5, 9, 12, 40] and np.array_equal(a[2][1], np.full(8, 2.0)) and a[1][3].tolist() == [4, 44]
    m2 = pickle.loads(pickle.dumps(m))
    assert sorted(m2) == sorted(m) and np.array_equal(m2.descriptor_matrix([9, 40, 5]), m.descriptor_matrix([9, 40, 5]))
    assert {k: v.channels for k, v in m.items()} == {5: 8, 9: 8, 12: 8, 40: 8}
    assert len(features.ReferenceMap([], np.zeros((0, 2)), np.zeros((0, 8)))) == 0


def test_array_form_references_gather_as_a_reference_map(tmp_path):
    """ReferenceMap on every rank (the extractor's default output) -> gather_references -> one ReferenceMap with all rows."""
    res = run_ranks("refmap_gather", tmp_path, world=2)
    for r in res:
        assert r["is_map"][0] == 1 and r["ids"].tolist() == list(range(3, 3 + 11))
        assert np.array_equal(r["desc"], np.arange(11 * 4, dtype=np.float64).reshape(11, 4)) and np.array_equal(r["src"][:, 1], 7 * np.arange(11))


# ---- world size 8: the node the SCALE run uses; counts that are not multiples of 8, empty shards --------
def _ragged_ba_scene(n_points, seed=3):
    """Flat BA arrays with heavy-tailed track lengths (2 .. 60 observations per point), observations sorted by point."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(60, 2 + rng.geometric(0.25, n_points)).astype(np.int64)
    lens[rng.integers(0, n_points, n_points // 50)] = 60                      # a few very long tracks
    obs_point = np.repeat(np.arange(n_points), lens).astype(np.int32)
    n_obs = len(obs_point)
    return dict(obs_point=obs_point, obs_image=rng.integers(0, 16, n_obs).astype(np.int32), obs_patch=np.arange(n_obs, dtype=np.int64),
                xyz=rng.normal(size=(n_points, 3)), refs=rng.normal(size=(n_points, 4)), image_camera=np.arange(16, dtype=np.int32),
                qvec=np.tile([1.0, 0, 0, 0], (16, 1)), tvec=np.zeros((16, 3)), cam_model=np.zeros(16, np.int32),
                cam_params=np.ones((16, 3)), patches=rng.normal(size=(n_obs, 1, 1, 4)).astype(np.float16),
                corners=np.zeros((n_obs, 2), np.int32), scales=np.ones((n_obs, 2)))


def test_ba_shards_at_world_8_are_balanced_by_observations():
    """SURVEY 8e's risk: a partition by POINTS leaves the ranks with different numbers of observations on ragged tracks (and
    the LM iteration waits for the slowest).  shard_ba_problem cuts the point range by cumulative OBSERVATIONS: on 20 003 points
    with 2 .. 60 observations each the largest shard is within 2 % of the mean; every observation is owned exactly once."""
    from pixsfm_amd import parallel
    prob = _ragged_ba_scene(20003)
    n_obs = len(prob["obs_point"])
    loads, seen, pts = [], [], []
    for rank in range(8):
        shard, pt_ids = parallel.shard_ba_problem(prob, rank, 8)
        assert np.array_equal(prob["obs_point"][shard["obs_ids"]], pt_ids[shard["obs_point"]])
        assert np.array_equal(shard["refs"], prob["refs"][pt_ids]) and np.array_equal(shard["patches"], prob["patches"][shard["obs_ids"]])
        loads.append(len(shard["obs_ids"])); seen.append(shard["obs_ids"]); pts.append(pt_ids)
    assert np.array_equal(np.sort(np.concatenate(seen)), np.arange(n_obs))
    assert np.array_equal(np.concatenate(pts), np.arange(20003))                            # contiguous, disjoint, complete
    assert max(loads) <= 1.02 * (n_obs / 8.0), (loads, n_obs / 8.0)
    by_points = [np.isin(prob["obs_point"], np.arange(20003)[r * 2501:(r + 1) * 2501]).sum() for r in range(8)]
    assert max(by_points) > max(loads)                                                       # (what equal point counts would give)


def test_ba_shards_at_world_8_with_fewer_points_than_ranks():
    from pixsfm_amd import parallel
    prob = _ragged_ba_scene(5)
    owned = []
    for rank in range(8):
        shard, pt_ids = parallel.shard_ba_problem(prob, rank, 8)
        assert len(shard["xyz"]) == len(pt_ids) and len(shard["obs_point"]) == len(shard["obs_ids"])
        assert len(pt_ids) == 0 or shard["obs_point"].max() == len(pt_ids) - 1
        owned.append(pt_ids)
    assert np.array_equal(np.sort(np.concatenate(owned)), np.arange(5)) and sum(len(o) == 0 for o in owned) >= 3    # empty shards exist


@pytest.mark.parametrize("n_tracks", [61, 3])
def test_ka_shards_at_world_8(n_tracks):
    """Sub-problems dealt to 8 ranks by residual-block count (a count that is not a multiple of 8; fewer sub-problems than
    ranks: empty shards): every node / edge owned once, whole sub-problems stay together, loads balanced."""
    from pixsfm_amd import parallel, synthetic_ka
    prob = synthetic_ka.make_ka_problem(n_tracks=n_tracks, track_len=5, seed=5, channels=8, patch_size=8, max_kps_per_problem=10)
    n, n_prob = len(prob["kp"]), int(prob["node_problem"].max()) + 1
    nodes, edges, loads, n_local = [], [], [], 0
    for rank in range(8):
        shard, node_ids = parallel.shard_ka_problem(prob, rank, 8)
        nodes.append(node_ids); edges.append(shard["edge_ids"]); loads.append(len(shard["edge_ids"]))
        if len(node_ids):
            n_local += int(shard["node_problem"].max()) + 1
            assert np.array_equal(node_ids[shard["edge_src"]], prob["edge_src"][shard["edge_ids"]])
            assert np.array_equal(node_ids[shard["edge_dst"]], prob["edge_dst"][shard["edge_ids"]])
            outside = np.setdiff1d(np.arange(n), node_ids)
            assert set(prob["node_problem"][node_ids]).isdisjoint(set(prob["node_problem"][outside]))
        else:
            assert len(shard["edge_ids"]) == 0 and len(shard["kp"]) == 0
    assert np.array_equal(np.sort(np.concatenate(nodes)), np.arange(n)) and np.array_equal(np.sort(np.concatenate(edges)), np.arange(len(prob["edge_src"])))
    assert n_local == n_prob
    if n_prob >= 16:
        sizes = np.bincount(prob["node_problem"][prob["edge_src"]], minlength=n_prob)
        assert max(loads) <= np.mean(loads) + sizes.max()                                   # longest-processing-time bound
    else:
        assert sum(l == 0 for l in loads) >= 8 - n_prob


def test_gathers_over_eight_ranks(tmp_path):
    """gather_rows (KA keypoints), gather_references (Reference objects and the array-backed ReferenceMap) on 8 gloo ranks."""
    prob = worker.ka_problem()
    n = len(prob["kp"])
    (tmp_path / "ka").mkdir(); (tmp_path / "refs").mkdir(); (tmp_path / "map").mkdir()
    res = run_ranks("ka_plumbing", tmp_path / "ka", world=8, timeout=900)
    want = prob["kp"] + (np.arange(n)[:, None] + 1) * np.array([1e-3, -2e-3])
    for r in res:
        assert np.array_equal(r["kp"], want) and np.array_equal(r["owned"], np.ones(n))
    assert sum(int(r["n_local_problems"][0]) for r in res) == int(prob["node_problem"].max()) + 1
    res = run_ranks("refs_gather", tmp_path / "refs", world=8, timeout=900)
    expect = worker.pack_references(worker.make_references())
    for r in res:
        for k in expect:
            assert np.array_equal(r[k], expect[k]), k
    res = run_ranks("refmap_gather", tmp_path / "map", world=8, timeout=900)
    for r in res:
        assert int(r["is_map"][0]) == 1 and np.array_equal(r["ids"], np.arange(3, 14))
        assert np.array_equal(r["desc"], np.arange(44, dtype=np.float64).reshape(11, 4))
