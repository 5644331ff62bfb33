"""ProcOctree (PersSampler.cpp:154-417): the product's closed-form host implementation against the oracle's
statement-by-statement restatement, on built trees with random invalidation / visit patterns.  CPU only (host code)."""
import numpy as np
import pytest

from helpers import small_scene


def _random_state(nodes, rng, kill):
    n = nodes.shape[0]
    nodes = nodes.view(np.uint8).copy().view(nodes.dtype)  # raw-byte copy keeps the padding bytes defined
    leaves = np.nonzero((nodes["is_leaf_node"] == 1) & (nodes["trans_idx"] >= 0))[0]
    dead = rng.random(leaves.size) < kill
    nodes["trans_idx"][leaves[dead]] = -1          # what K6 / K7 do
    w = rng.integers(-100, 5000, n).astype(np.int64)
    a = rng.integers(-100, 5000, n).astype(np.int64)
    v = rng.integers(0, 12, n).astype(np.int64)
    return nodes, w, a, v


def _check_tree(nodes):
    assert nodes["parent"][0] == -1
    for u in range(nodes.shape[0]):
        for c in nodes["childs"][u]:
            if c >= 0:
                assert nodes["parent"][c] == u and c > u
        if nodes["is_leaf_node"][u]:
            assert np.all(nodes["childs"][u] == -1)


@pytest.mark.parametrize("kill", [0.0, 0.3, 0.8, 1.0])
@pytest.mark.parametrize("mode", ["compact", "subdivide", "brute"])
def test_proc_octree_matches_oracle(oracle, kill, mode):
    import gfnerf_amd
    from gfnerf_amd import ops

    sc = small_scene()
    base = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    rng = np.random.default_rng(int(kill * 10) + len(mode))
    nodes, w, a, v = _random_state(base, rng, kill)
    flags = dict(compact=dict(compact=True, subdivide=False, brute_force=False),
                 subdivide=dict(compact=True, subdivide=True, brute_force=False),
                 brute=dict(compact=True, subdivide=True, brute_force=True))[mode]
    want_n, want_w, want_a = oracle.proc_octree(nodes, w, a, v, **flags)
    got_n, got_w, got_a = ops.octree_proc_host(nodes, w, a, v, **flags)
    got_n = got_n.view(oracle.TREE_NODE_DTYPE)
    assert got_n.shape == want_n.shape
    assert got_n.tobytes() == want_n.tobytes()   # every byte of the checkpoint blob, padding included
    assert np.array_equal(got_w, want_w) and np.array_equal(got_a, want_a)
    _check_tree(got_n)
    if kill == 1.0:
        assert got_n.shape[0] == 1  # only the root survives
    if mode == "brute" and kill < 1.0:
        assert got_n.shape[0] > base.shape[0] * (1 - kill) * 2


def test_proc_octree_twice_is_stable(oracle):
    """Compacting a compacted tree changes nothing (the reference runs it every compact_freq iterations)."""
    import gfnerf_amd
    from gfnerf_amd import ops

    sc = small_scene()
    rng = np.random.default_rng(3)
    nodes, w, a, v = _random_state(sc["nodes"].view(oracle.TREE_NODE_DTYPE), rng, 0.5)
    n1, w1, a1 = ops.octree_proc_host(nodes, w, a, v)
    n2, w2, a2 = ops.octree_proc_host(n1, w1, a1, np.zeros_like(w1))
    assert n1.tobytes() == n2.tobytes() and np.array_equal(w1, w2) and np.array_equal(a1, a2)


def test_proc_octree_rejects_malformed_blob(oracle):
    import gfnerf_amd
    from gfnerf_amd import _lib, ops

    sc = small_scene()
    nodes = sc["nodes"].copy().view(oracle.TREE_NODE_DTYPE)  # raw-byte copy: a structured copy leaves the padding bytes undefined
    nodes["parent"][5] = 3 if nodes["parent"][5] != 3 else 4
    z = np.zeros(nodes.shape[0], np.int64)
    with pytest.raises(_lib.GfnError):
        ops.octree_proc_host(nodes, z, z, z)
    with pytest.raises(_lib.GfnError):  # the reference never runs without the compaction pass
        ops.octree_proc_host(sc["nodes"], z, z, z, compact=False, subdivide=True)
