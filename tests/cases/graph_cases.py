"""Seeded match graphs for the labelling (pixsfm/base/src/graph.cc:126-256).  Inputs only."""
import numpy as np


def cases():
    """Seeded match graphs: (name, pairs (P, 2) image indices, list of (matches (m, 2), sims (m,))).  Random matches
    between few features force the conflict case of graph.cc:126-206 (two keypoints of one image competing for a track);
    coarse similarities force ties in the edge order and in the root scores."""
    rng = np.random.default_rng(314159)
    out = []
    for i in range(24):
        n_img = int(rng.integers(3, 9))
        per_img = int(rng.choice([6, 15, 40]))
        digits = int(rng.choice([1, 2, 6]))
        pairs, mm = [], []
        for a in range(n_img):
            for b in range(a + 1, n_img):
                if rng.random() < 0.15:
                    continue
                m = int(rng.integers(1, 2 * per_img))
                matches = np.stack([rng.integers(0, per_img, m), rng.integers(0, per_img, m)], 1).astype(np.int64)
                sims = np.round(rng.uniform(0.2, 1.0, m), digits)
                if i % 5 == 4 and rng.random() < 0.5:          # reversed pair order (b, a): out-matches of later images
                    pairs.append((b, a)); matches = matches[:, ::-1].copy()
                else:
                    pairs.append((a, b))
                mm.append((matches, sims))
        out.append(("graph%02d" % i, np.array(pairs, np.int32), mm))
    # a chain that must split: features 0 of images 0..3 matched in a cycle with one image twice
    pairs = np.array([(0, 1), (1, 2), (2, 0), (0, 2)], np.int32)
    mm = [(np.array([[0, 0]]), np.array([0.9])), (np.array([[0, 0]]), np.array([0.8])),
          (np.array([[0, 1]]), np.array([0.7])), (np.array([[1, 0]]), np.array([0.95]))]
    out.append(("conflict", pairs, mm))
    return out
