"""Equal-size clustering of the training cameras into focal blocks (host side, one shot at the stage switch).

The reference clusters with SpectralEqualSizeClustering(nclusters=k, nneighbors=10 %, equity_fraction=1, seed=1234) over
the camera-position distance matrix (gfnerf/perssampler.py:216-242 -> gfnerf/cluster/spectral_equal_size_clustering.py,
sklearn); SURVEY 2a marks that file OUT of scope (host-side, one-shot).  What the hot path needs from it is its RESULT
contract -- `cameras_labels [n_cams, 1] int64`, every cluster non-empty (perssampler.py:239-242) and sizes equal up to one
(equity_fraction = 1) -- which ray routing (each GPU draws rays only from its block's cameras) and render
routing (perssampler.py:138-165,369-376) consume.  This module produces labels with that contract by capacity-constrained
k-means on the camera positions: farthest-point seeds, then Lloyd rounds whose assignment step hands every cluster exactly
its quota, cameras taken in order of how much they prefer their best centre over their second best.  Deterministic in
(positions, k, seed); numpy only.
"""
import numpy as np


def cluster_sizes(n, k):
    """Quota per cluster: n // k each, the first n % k one more (_optimal_cluster_sizes of the reference)."""
    base, extra = divmod(int(n), int(k))
    return [base + (1 if i < extra else 0) for i in range(k)]


def _assign_with_quota(d, quota):
    """d [n,k] distances -> labels with exactly quota[j] points in cluster j.  Greedy by regret: points whose best centre
    beats their second best by the most choose first; a full cluster drops out of everyone's list."""
    n, k = d.shape
    labels = np.full(n, -1, np.int64)
    left = np.asarray(quota, np.int64).copy()
    open_ = np.ones(k, bool)
    todo = np.arange(n)
    while todo.size:
        dd = np.where(open_[None, :], d[todo], np.inf)
        order = np.argsort(dd, axis=1)
        best = order[:, 0]
        best_d = dd[np.arange(todo.size), best]
        second_d = dd[np.arange(todo.size), order[:, 1]] if k > 1 else np.full(todo.size, np.inf)
        regret = np.where(np.isfinite(second_d), second_d - best_d, np.inf)
        # take, per cluster, its `left` highest-regret candidates of this round; the rest try again
        rank = np.lexsort((todo, -regret))  # stable: ties by index
        taken = np.zeros(todo.size, bool)
        for i in rank:
            j = best[i]
            if left[j] > 0:
                labels[todo[i]] = j
                left[j] -= 1
                taken[i] = True
        open_ = left > 0
        todo = todo[~taken]
    return labels


def equal_size_clusters(positions, k, seed=1234, iters=20):
    """positions [n,3] -> labels int64 [n] in [0, k), cluster sizes = cluster_sizes(n, k)."""
    p = np.asarray(positions, np.float64)
    n = p.shape[0]
    if not (1 <= k <= n):
        raise ValueError(f"equal_size_clusters: need 1 <= k <= n, got k={k}, n={n}")
    quota = cluster_sizes(n, k)
    rng = np.random.default_rng(seed)
    centres = [p[int(rng.integers(n))]]
    for _ in range(1, k):  # farthest-point seeding
        dmin = np.min(np.linalg.norm(p[:, None, :] - np.asarray(centres)[None], axis=-1), axis=1)
        centres.append(p[int(np.argmax(dmin))])
    centres = np.asarray(centres)
    labels = None
    for _ in range(iters):
        d = np.linalg.norm(p[:, None, :] - centres[None], axis=-1)
        new = _assign_with_quota(d, quota)
        if labels is not None and np.array_equal(new, labels):
            break
        labels = new
        centres = np.stack([p[labels == j].mean(0) for j in range(k)])
    # canonical numbering: clusters ordered by their first camera (labels then do not depend on the seeding order)
    first = [int(np.nonzero(labels == j)[0][0]) for j in range(k)]
    remap = np.empty(k, np.int64)
    remap[np.argsort(first)] = np.arange(k)
    return remap[labels]


def cameras_of(labels, block):
    """Indices of the training cameras in `block`'s cluster."""
    return np.nonzero(np.asarray(labels).reshape(-1) == int(block))[0]
