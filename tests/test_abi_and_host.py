"""CPU-only checks: the C-ABI library loads and exports every symbol include/*.h declares, the
product refuses to run without a device, and the host-side octree builder is sane."""
import os
import re
import sys

import numpy as np
import pytest

from helpers import ROOT, small_scene


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gfnerf_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import gfnerf_amd
    from gfnerf_amd import _lib

    names = _declared_functions()
    assert len(names) >= 20
    handle = _lib.lib()
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/gfnerf_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    # the library, the header and the binding agree on the ABI revision (a mismatch is refused at load time, see below)
    header = open(os.path.join(ROOT, "include", "gfnerf_amd.h")).read()
    abi = int(re.search(r"#define\s+GFN_ABI_VERSION\s+(\d+)", header).group(1))
    assert handle.gfn_version() == abi == _lib.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_revision(monkeypatch):
    """exported signatures changed while gfn_version() stood still.  Now the loader compares gfn_version() with the revision
    _lib.py was written for and raises instead of calling through shifted arguments."""
    import gfnerf_amd  # noqa: F401
    from gfnerf_amd import _lib

    _lib.lib()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    monkeypatch.setenv("GFN_NO_REBUILD", "1")
    with pytest.raises(_lib.GfnError, match="ABI revision"):
        _lib.lib()


def test_no_cpu_fallback():
    """Without a gfx950 device the product must fail loudly, never route to a CPU path."""
    import torch

    import gfnerf_amd
    from gfnerf_amd import _lib, ops

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert _lib.lib().gfn_device_count() == 0
    with pytest.raises(_lib.GfnError):
        _lib.require_device()
    with pytest.raises(_lib.GfnError):
        ops.sh4(torch.zeros(4, 3))  # CPU tensor -> refused
    # nothing in the product package may reference the oracle
    pkg = os.path.join(ROOT, "gf-nerf_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "libgfn_oracle" not in src, f"{f} touches the oracle"


def test_level_scales_match_oracle(oracle):
    import gfnerf_amd
    from gfnerf_amd import ops

    assert np.array_equal(ops.level_scales(), oracle.level_scales())
    assert ops.level_scales()[0] == 8.0 and ops.level_scales()[15] == 1024.0


def test_octree_builder_layout_and_invariants(oracle):
    sc = small_scene()
    nodes = sc["nodes"].view(oracle.TREE_NODE_DTYPE)
    tr = sc["transes"].view(oracle.TRANS_INFO_DTYPE)
    assert nodes.shape[0] == sc["n_nodes"] > 50 and tr.shape[0] == sc["n_trans"] > 10
    assert nodes["parent"][0] == -1 and nodes["side_len"][0] == 512.0
    # children: half the side, centred on the octant; parent links consistent
    for u in range(nodes.shape[0]):
        for st, v in enumerate(nodes["childs"][u]):
            if v < 0:
                continue
            assert nodes["parent"][v] == u
            assert nodes["side_len"][v] == nodes["side_len"][u] * 0.5
            off = np.array([((st >> 2) & 1) - 0.5, ((st >> 1) & 1) - 0.5, (st & 1) - 0.5], np.float32)
            np.testing.assert_allclose(nodes["center"][v], nodes["center"][u] + nodes["side_len"][u] * 0.5 * off)
    valid = nodes["trans_idx"] >= 0
    assert np.all(nodes["is_leaf_node"][valid] == 1)
    assert sorted(nodes["trans_idx"][valid]) == list(range(tr.shape[0]))
    # each TransInfo belongs to its leaf and warps the leaf into a bounded, step-normalised frame
    rng = np.random.default_rng(0)
    for u in np.nonzero(valid)[0][:20]:
        t = tr[nodes["trans_idx"][u]]
        np.testing.assert_array_equal(t["center"], nodes["center"][u])
        assert t["side_len"] == nodes["side_len"][u] and t["dis_summary"] > 0
        pts = (rng.random((64, 3), dtype=np.float32) - 0.5) * t["side_len"] + t["center"]
        w = np.stack([oracle.warp_point(tr[nodes["trans_idx"][u]:nodes["trans_idx"][u] + 1], p) for p in pts])
        assert np.isfinite(w).all() and np.abs(w - w.mean(0)).max() < 1.5  # (w+1.5)/3 stays inside [0,1]
        J = oracle.warp_jac(tr[nodes["trans_idx"][u]:nodes["trans_idx"][u] + 1], pts[0])
        assert np.isfinite(J).all() and abs(np.linalg.det(J.astype(np.float64))) > 0


def test_octree_builder_is_deterministic():
    import gfnerf_amd
    from gfnerf_amd import ops, scene

    c2w, intri, bounds = scene.make_cameras(n_side=4, seed=3)
    a = ops.octree_build(c2w, intri, bounds, max_depth=7, seed=5, threads=2)
    b = ops.octree_build(c2w, intri, bounds, max_depth=7, seed=5, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_primes_are_prime_and_in_range():
    import gfnerf_amd
    from gfnerf_amd import scene

    p = scene.make_primes(40, seed=2)
    assert p.shape == (16, 40, 3) and p.dtype == np.int32
    assert p.min() >= (1 << 28) and p.max() < (1 << 30)
    flat = p.reshape(-1).astype(np.int64)[:300]
    small = [q for q in range(2, 1 << 15) if all(q % r for r in range(2, int(q ** 0.5) + 1))]
    for v in flat:
        assert all(v % q for q in small if q * q <= v)


def test_gf_scheduler_matches_reference_curve(golden):
    """Product scheduler (host logic) against the LR curve captured from the reference's LambdaLR."""
    import torch

    import gfnerf_amd
    from gfnerf_amd.gfnerf.schedulers import GFNerfExponentialDecaySchedulerConfig, lr_multiplier

    for tag in ("default", "warm"):
        lr_final, max_steps, warm, n_split, per_split, init = golden[f"sched_{tag}_cfg"]
        cfg = GFNerfExponentialDecaySchedulerConfig(lr_final=float(lr_final), max_steps=int(max_steps), warmup_steps=int(warm),
                                                    n_split_dataset=int(n_split), steps_per_split_dataset=int(per_split),
                                                    steps_perssampler_init=int(init))
        want = golden[f"sched_{tag}_lr"]
        got = [1e-2 * lr_multiplier(s, 1e-2, cfg) for s in range(want.shape[0])]
        np.testing.assert_allclose(got, want, rtol=1e-12)
        # and through the same LambdaLR wiring the reference uses
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=1e-2)
        sched = cfg.setup().get_scheduler(opt, 1e-2)
        for s in range(120):
            assert abs(opt.param_groups[0]["lr"] - want[s]) <= 1e-12 * want[s]
            opt.step(); sched.step()


def test_matrix_kernels_do_not_spill():
    """The MFMA kernels are sized to the register file (256 VGPRs at two waves per SIMD, 512 at one): a build that spills
    to scratch is a performance bug, and on this toolchain the spilling 512-thread form of the fused forward also
    produced wrong lookups intermittently.  Cross-compiles the three sources and reads the resource remarks."""
    import concurrent.futures as cf
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "gf-nerf_amd", "csrc")

    def remarks(src):
        cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-c",
               os.path.join(csrc, src), "-o", os.devnull, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stderr

    srcs = ("fused_fwd.hip", "field_x3.hip", "wgrad_x3.hip", "batch.hip", "lm_lookup.hip")
    with cf.ThreadPoolExecutor(5) as ex:
        outs = dict(zip(srcs, ex.map(remarks, srcs)))
    for src, text in outs.items():
        spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", text)]
        scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)]
        assert spills and all(s == 0 for s in spills), (src, spills)
        assert all(s == 0 for s in scratch), (src, scratch)


def test_choose_tile_divides_the_frame():
    """render.choose_tile: tiles that divide the image exactly and hold whole 8 x 4 patches, near 512 Ki rays."""
    import importlib.util
    import math  # noqa: F401

    src = open(os.path.join(ROOT, "gf-nerf_amd", "render.py")).read()
    ns = {}
    exec("import math\n" + src[src.index("MAX_TILE_RAYS ="):src.index("def frame_shard")], ns)  # the module itself imports the HIP library
    choose = ns["choose_tile"]
    for (h, w) in ((1080, 1920), (864, 1152), (2160, 3840), (64, 128), (720, 1280), (480, 640)):
        tw, th = choose(h, w)
        assert w % tw == 0 and h % th == 0 and tw % 8 == 0 and th % 4 == 0, (h, w, tw, th)
        assert tw * th <= max(h * w, 1) and (tw * th >= 0.4 * min(524288, h * w))
    assert choose(1080, 1920) == (960, 540) and choose(1080, 1920, target_rays=131072) == (384, 360)
    assert choose(1080, 1917)[0] == 128  # no width divides 1917 into multiples of 8: fixed-width tiles with ragged borders
    # a frame with few divisors must not become ONE tile beyond the sampler's 32-bit indexing (n_rays * 1024 < 2^31), and a caller's
    # memory cap is respected
    tw, th = choose(1052, 2008)
    assert tw * th <= ns["MAX_TILE_RAYS"] and 2008 % tw == 0 and 1052 % th == 0, (tw, th)
    tw, th = choose(1080, 1920, max_rays=150000)
    assert tw * th <= 150000 and 1920 % tw == 0 and 1080 % th == 0


def test_bench_line_stays_under_two_thousand_characters():
    """The driver keeps the tail of the log: bench.compact_line must fit every sub-record's headline figures into < 2,000 characters (the
    full records go to stderr as [bench-details])."""
    import json

    sys.path.insert(0, ROOT)
    import bench

    f = 1234567.891234
    out = dict(metric="train_rays_per_sec", value=f, unit="rays/s", n_gpus=1, steps=50, warmup=5, ms_per_step=1.35164321, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f32 (split-f16 x3 MFMA decoder, f16 table)", data="synthetic", samples_per_sec=f * 300, batch_samples_per_sec=f * 400, ranks_seen=1,
               config=dict(workload="GF-NeRF global+focal train step, 4 focal blocks resident (1 active/step), 4096 rays x 192 samples, Rubble-scale synthetic cameras",
                           rays_per_batch=4096, mean_samples_per_ray=192.51234, log2_table_size=21, n_cameras=1600, n_tree_nodes=10809, stage="focal", early_T=1e-4,
                           alive_sample_frac=0.5681234, evaluated_sample_frac=0.6749412, backend=None, loss=1.683412, launch="hipGraph replay (8 captured graphs)",
                           parallelism="single GPU", two_phase=dict(K=130), allreduce_selfcheck=None),
               roofline=dict(bound="hbm", kernel="lm_lookup_kernel<true, false> (+ <true, true>: the second table accumulates)", achieved=1242.01234, peak=8000.0, unit="GB/s",
                             frac=0.1552512, traffic=1033260646, avg_kernel_ms=0.2828112, alg_bytes_per_launch=351253980, launches_per_step=2, mfma_busy_frac=0.3612),
               global_stage=dict(ms_per_step=1.926412, value=f, launch_short="graph", replicas_bit_identical=None, wgrad_operands="f16 rne+dither (dza1 hi + lo')", x3_ms_per_step=2.002512),
               render=dict(value=51.655123, roofline=dict(samples_per_sec=f * 1000, mfma_busy_frac=0.3039), x3_ms_per_mpix=55.123456,
                           colour_x2_vs_x3=dict(max_abs_rgb=4.1127e-06, accumulation_and_depth_same_bits=True)),
               plugin_step_focal=dict(value=f, ms_per_step=1.817712), plugin_step_global=dict(value=f, ms_per_step=2.028512), batch_assembly=dict(ms_per_batch=0.1041123),
               psnr=dict(psnr_fast=33.668123, psnr_reference=33.565123, delta=0.1026612, se=0.1582212, n_seeds=3, decoder_delta=1.2716e-06, study_delta=0.012345, study_se=0.048712,
                         study_n_seeds=148),
               trained_state=dict(value=f, ms_per_step=1.583012, alive_sample_frac=0.8626212, evaluated_sample_frac=1.0, two_phase_K=1024, roofline_frac=0.2323312,
                                  global_ms_per_step=1.782412, global_rays_per_sec=f, render_ms_per_mpix=35.123456, render_evaluated_frac=0.912345,
                                  render_x3_ms_per_mpix=38.123456, render_colour_x2_max_abs_rgb=0.00016761),
               plugin_step_trained=dict(value=f, ms_per_step=1.712345), rays8192=dict(value=f * 1.5, global_rays_per_sec=f * 1.4, ms_per_step=2.1),
               cpu_reference_field=dict(train_rays_per_sec=4227.212), cpu_baseline=dict(value=712.0812, unit="rays/s", cores=256, kind="port",
                                                                                        sample="2048 rays (392816 samples) of the same batch, oracle fwd+bwd (no Adam), 2.9 s; all stages OpenMP"))
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < 2000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    c = line["config"]
    for key in ("workload", "trained_rays_per_sec", "trained_ms_per_step", "psnr", "psnr_reference", "psnr_delta", "psnr_se", "psnr_n_seeds", "psnr_study_delta", "psnr_study_se",
                "psnr_study_n_seeds", "render_ms_per_mpix", "global_ms_per_step", "plugin_ms_per_step", "trained_render_ms_per_mpix", "plugin_trained_ms_per_step",
                "rays8192_rays_per_sec", "rays8192_global_rays_per_sec", "render_decoder", "trained_render_x3_ms_per_mpix", "trained_render_x2_max_drgb",
                "global_x3_ms_per_step", "global_wgrad_operands", "plugin_global_ms_per_step"):
        assert key in c, key
    assert line["roofline"]["kernel"] == "lm_lookup_kernel<true, false>" and set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"}


def test_profile_summaries_take_the_newest_run_of_a_merged_directory(tmp_path):
    """tools/prof_summary.py (what turns rocprofv3 output into profiles/*.csv): gpurun merges every call's files into the same local directory, named by
    process id, so a directory may hold several runs' files -- the summary must be the NEWEST run's (round 5: the first file found was an older run's)
    -- and tools/trace_tail.py's per-step window divides by the dispatches of the kernel it is told marks a step."""
    import csv
    import subprocess
    import time

    d = tmp_path / "kt" / "runc"
    d.mkdir(parents=True)
    head = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]
    for pid, avg in ((111, "100.0"), (222, "200.0")):
        with open(d / f"{pid}_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(head)
            w.writerow(["void (anonymous namespace)::k(int)", "4", "800", avg, "100", "1", "9"])
        time.sleep(0.05)
    os.utime(d / "111_kernel_stats.csv", (1, 1))  # the older run
    out = tmp_path / "stats.csv"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), "stats", str(tmp_path / "kt"), str(out)])
    rows = list(csv.DictReader(open(out)))
    assert rows == [dict(kernel="k(int)", calls="4", total_ns="800", avg_ns="200.0", percent="100", min_ns="1", max_ns="9")]
    p = tmp_path / "pmc" / "runc"
    p.mkdir(parents=True)
    for pid, val in ((5, "1000"), (6, "3000")):
        with open(p / f"{pid}_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writerow(["void (anonymous namespace)::k(int)", "FETCH_SIZE", val])
    os.utime(p / "5_counter_collection.csv", (1, 1))
    out2 = tmp_path / "pmc.csv"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), "pmc_many", str(out2), str(tmp_path / "pmc")])
    r2 = list(csv.DictReader(open(out2)))
    assert len(r2) == 1 and r2[0]["kernel"] == "k(int)" and float(r2[0]["FETCH_SIZE_mean"]) == 3000.0
    # trace_tail: the last 1 ms of a trace, per step = per dispatch of the marker kernel
    t = tmp_path / "trace"
    t.mkdir()
    with open(t / "7_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Start_Timestamp", "End_Timestamp", "Kernel_Name", "Queue_Id"])
        w.writerow([0, 100, "old", "1"])  # outside the window
        for i in range(4):
            w.writerow([10_000_000 + i * 100_000, 10_000_000 + i * 100_000 + 50_000, "step_marker", "1"])
            w.writerow([10_000_000 + i * 100_000 + 50_000, 10_000_000 + i * 100_000 + 80_000, "other", "1"])
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_tail.py"), str(t), "1", "auto:step_marker"], capture_output=True, text=True, check=True).stdout
    assert "/ 4:" in res and "old" not in res, res
    lines = [ln.split() for ln in res.splitlines()[1:]]
    got = {ln[-1]: float(ln[0]) for ln in lines}
    assert abs(got["step_marker"] - 0.05) < 1e-9 and abs(got["other"] - 0.03) < 1e-9, res


def test_committed_counters_belong_to_the_kernel_sources_in_the_tree():
    """bench.py reports roofline.traffic, request_rate, mfma_busy_frac and the inference kernel's occupancy from profiles/pmc_traffic.json ONLY while the
    sha256 of the kernel sources recorded with the counters equals the tree's (a number measured on another kernel is not reported at all).  The committed
    file must therefore be the committed sources': an edit to csrc/lm_lookup.hip, field_x3.hip, fused_fwd.hip, x3_device.h or hash_device.h after the last
    counter run would silently drop those fields from the driver's bench line -- this fails instead, and says which passes to run again
    (tools/r06_profiles.sh pmc | pmc_render | busy, then tools/sync_profiles.py)."""
    import hashlib
    import json

    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    groups = pmc["source_sha256"]
    assert set(groups) == {"lm_lookup", "decoder", "render"}
    for group, files in groups.items():
        for rel, want in files.items():
            got = hashlib.sha256(open(os.path.join(ROOT, "gf-nerf_amd", rel), "rb").read()).hexdigest()
            assert got == want, f"profiles/pmc_traffic.json [{group}] was collected on another {rel}: run the counter passes again"
    # and the figures the line takes from it are there
    for key in ("lm_lookup:focal", "mfma_busy:field_fwd:focal", "mfma_busy:render", "valu_busy:render", "kernel_ms:render", "tile_rays:render", "mfma_per_tile:render"):
        assert pmc.get(key) is not None, key
    assert pmc["mfma_per_tile:render"] == 160.0  # the renderer's default decoder form (colour_x2): head layer 2 issues 64 MFMAs instead of 96
