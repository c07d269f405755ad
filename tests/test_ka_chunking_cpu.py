"""Host logic of the label groups that span several workgroups (pxr_ka_view.d_prob_group): ka_engine.chunk_label_groups cuts a
large label group -- ONE Ceres problem of the reference (keypoint_adjustment/main.py:189-202) -- into chunks of whole tracks."""
import numpy as np


def test_chunks_are_whole_tracks_in_consecutive_non_decreasing_groups():
    from pixsfm_amd.ka_engine import CHUNK_FROM, CHUNK_KPS, chunk_label_groups
    rng = np.random.default_rng(0)
    lens = rng.integers(2, 12, 300)
    track = np.repeat(np.arange(300), lens)
    label = np.where(track < 150, 0, np.where(track < 158, 1, np.where(track < 290, 2, 3))).astype(np.int64)
    perm = rng.permutation(len(track))                   # nodes of a track need not be adjacent
    track, label = track[perm], label[perm]
    label[rng.choice(len(label), 7, replace=False)] = -1
    chunk, group = chunk_label_groups(label, track)
    assert (chunk[label < 0] == -1).all() and (chunk[label >= 0] >= 0).all()
    assert np.all(np.diff(group) >= 0) and sorted(set(group.tolist())) == [0, 1, 2, 3]
    for c in range(len(group)):
        idx = np.flatnonzero(chunk == c)
        assert (label[idx] == group[c]).all()
        n_tracks = len(set(track[idx].tolist()))
        assert len(idx) <= CHUNK_KPS or n_tracks == 1 or (group == group[c]).sum() == 1      # <= 50 keypoints unless unsplit / one long track
    for t in range(300):                                 # a track lives in one chunk
        assert len(set(chunk[(track == t) & (label >= 0)].tolist())) <= 1
    small = [g for g in range(4) if (label == g).sum() <= CHUNK_FROM]
    assert small and all((group == g).sum() == 1 for g in small)


def test_a_group_beyond_the_resident_limit_gets_larger_chunks():
    from pixsfm_amd.ka_engine import chunk_label_groups
    track = np.repeat(np.arange(10000), 10)
    chunk, group = chunk_label_groups(np.zeros(100000, np.int64), track, max_chunks=448)
    sizes = np.bincount(chunk)
    assert len(group) <= 448 and sizes.sum() == 100000 and sizes.max() <= 240
    chunk2, group2 = chunk_label_groups(np.zeros(100000, np.int64), track, max_chunks=100)
    assert len(group2) <= 100


def test_nothing_to_cut():
    from pixsfm_amd.ka_engine import chunk_label_groups
    track = np.repeat(np.arange(40), 5)
    label = (track // 10).astype(np.int64)               # 4 groups of 50 keypoints
    chunk, group = chunk_label_groups(label, track)
    assert np.array_equal(chunk, label) and group.tolist() == [0, 1, 2, 3]
