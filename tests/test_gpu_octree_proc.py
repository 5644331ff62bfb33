"""ProcOctree on the DEVICE (csrc/octree_proc.hip, gfn_octree_proc) against the oracle's statement-by-statement restatement
of PersSampler.cpp:154-417 and against the product's host closed form, byte for byte (padding included), on the same 12
invalidation / visit patterns x modes as tests/test_octree_proc.py -- plus chained calls the way UpdateOctNodes issues them at
a milestone (subdivide, MarkInvisibleNodes, compact: PersSampler_cuda.cu:667-676), a larger tree, and the malformed-blob
error.  The tree never leaves HBM: the only read-back is the new node count."""
import numpy as np
import pytest

from helpers import small_scene
from test_octree_proc import _check_tree, _random_state

pytestmark = pytest.mark.gpu

FLAGS = dict(compact=dict(subdivide=False, brute_force=False), subdivide=dict(subdivide=True, brute_force=False),
             brute=dict(subdivide=True, brute_force=True))


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _run_device(ops, nodes, w, a, v, **flags):
    n_t, w_t, a_t = ops.octree_proc(_dev(nodes.view(np.uint8).reshape(-1)), _dev(w), _dev(a), _dev(v), **flags)
    return n_t.cpu().numpy(), w_t.cpu().numpy(), a_t.cpu().numpy()


@pytest.mark.parametrize("kill", [0.0, 0.3, 0.8, 1.0])
@pytest.mark.parametrize("mode", ["compact", "subdivide", "brute"])
def test_device_proc_octree_is_byte_identical(oracle, kill, mode):
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import ops

    sc = small_scene()
    base = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    rng = np.random.default_rng(int(kill * 10) + len(mode))
    nodes, w, a, v = _random_state(base, rng, kill)
    want_n, want_w, want_a = oracle.proc_octree(nodes, w, a, v, compact=True, **FLAGS[mode])
    host_n, host_w, host_a = ops.octree_proc_host(nodes, w, a, v, compact=True, **FLAGS[mode])
    got_n, got_w, got_a = _run_device(ops, nodes, w, a, v, **FLAGS[mode])
    assert got_n.tobytes() == want_n.tobytes() == host_n.tobytes()
    assert np.array_equal(got_w, want_w) and np.array_equal(got_a, want_a)
    _check_tree(got_n.view(oracle.TREE_NODE_DTYPE))
    if kill == 1.0:
        assert got_w.shape[0] == 1


def test_device_proc_octree_milestone_sequence_and_larger_tree(oracle):
    """subdivide -> K7-like invalidation -> compact, three times over (the tree grows past 20k nodes), device against host at every
    step; compacting a compacted tree changes nothing."""
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import ops

    sc = small_scene()
    rng = np.random.default_rng(11)
    nodes, w, a, v = _random_state(sc["nodes"].view(oracle.TREE_NODE_DTYPE), rng, 0.2)
    v[:] = 9  # every leaf was visited often enough to split
    cur = (nodes.view(np.uint8).reshape(-1), w, a, v)
    sizes = []
    for rnd in range(3):
        for flags in (FLAGS["subdivide"], FLAGS["compact"]):
            n_np = cur[0].view(oracle.TREE_NODE_DTYPE)
            host = ops.octree_proc_host(n_np, cur[1], cur[2], cur[3], compact=True, **flags)
            got = _run_device(ops, n_np, cur[1], cur[2], cur[3], **flags)
            assert got[0].tobytes() == host[0].tobytes() and np.array_equal(got[1], host[1]) and np.array_equal(got[2], host[2])
            nn = got[0].view(oracle.TREE_NODE_DTYPE).copy()
            leaves = np.nonzero((nn["is_leaf_node"] == 1) & (nn["trans_idx"] >= 0))[0]
            nn["trans_idx"][leaves[rng.random(leaves.size) < 0.15]] = -1  # MarkInvisibleNodes / K6 between the calls
            m = nn.shape[0]
            cur = (nn.view(np.uint8).reshape(-1), got[1], got[2], np.full(m, 9, np.int64))
            sizes.append(m)
    assert sizes[0] > sc["n_nodes"] and max(sizes) > 20000, sizes
    n_np = cur[0].view(oracle.TREE_NODE_DTYPE)
    once = _run_device(ops, n_np, cur[1], cur[2], cur[3], **FLAGS["compact"])
    twice = _run_device(ops, once[0].view(oracle.TREE_NODE_DTYPE), once[1], once[2], np.zeros_like(once[1]), **FLAGS["compact"])
    assert once[0].tobytes() == twice[0].tobytes() and np.array_equal(once[1], twice[1])
    _check_tree(once[0].view(oracle.TREE_NODE_DTYPE))


def test_device_proc_octree_rejects_malformed_blob(oracle):
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import _lib, ops

    sc = small_scene()
    nodes = sc["nodes"].copy().view(oracle.TREE_NODE_DTYPE)
    nodes["parent"][5] = 3 if nodes["parent"][5] != 3 else 4
    z = np.zeros(nodes.shape[0], np.int64)
    with pytest.raises(_lib.GfnError):
        _run_device(ops, nodes, z, z, z)


def test_sampler_proc_octree_stays_on_the_device(oracle, monkeypatch):
    """The PersSampler mirror's ProcOctree and HotPath.proc_octree go through the device kernel: no tensor of the tree is
    copied to the host (Tensor.cpu is made to fail for the duration), and the result equals the host closed form."""
    import torch

    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import hotpath, ops

    sc = small_scene()
    hp = hotpath.HotPath(sc["nodes"], sc["transes"], n_cams=sc["c2w"].shape[0], cfg=hotpath.HotPathConfig(log2_T=12), seed=1)
    rng = np.random.default_rng(5)
    nodes, w, a, v = _random_state(sc["nodes"].view(oracle.TREE_NODE_DTYPE), rng, 0.4)
    hp.nodes = _dev(nodes.view(np.uint8).reshape(-1)); hp.weight_stats = _dev(w); hp.alpha_stats = _dev(a); hp.visit_cnt = _dev(v)
    want = ops.octree_proc_host(nodes, w, a, v, compact=True, subdivide=True)

    def no_cpu(self, *args, **kwargs):
        raise AssertionError("the tree was copied to the host")

    monkeypatch.setattr(torch.Tensor, "cpu", no_cpu)
    hp.proc_octree(compact=True, subdivide=True)
    monkeypatch.undo()
    assert hp.n_nodes == want[0].size // 128 and hp.nodes.cpu().numpy().tobytes() == want[0].tobytes()
    assert np.array_equal(hp.weight_stats.cpu().numpy(), want[1]) and int(hp.visit_cnt.sum()) == 0 and hp.visit_cnt.numel() == hp.n_nodes
    assert hp.nodes_compact.shape[0] * hp.nodes_compact.element_size() * (hp.nodes_compact.numel() // hp.nodes_compact.shape[0]) >= hp.n_nodes * 64
