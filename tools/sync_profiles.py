"""Copy the summaries of a tools/rNN_profiles.sh run (gpurun_out/rNNp: modes traces + pmc) into profiles/rNN_* and refresh profiles/pmc_traffic.json
    python tools/sync_profiles.py gpurun_out/r05p r05
    python tools/sync_profiles.py gpurun_out/r06p r06 render     after a change to the inference kernel / x3_device.h that leaves the training kernels' code as it
                                                                was (modes render_traces + pmc_render + busy): the bench record, the render traces and counters, and
                                                                the decoder kernels' matrix-pipe occupancy passes (their own table, profiles/rNN_pmc_{focal,global}_busy_per_kernel.csv;
                                                                the focal / global traces and traffic tables of the full run stay)
(what bench.py reads for roofline.traffic / request_rate / mfma_busy_frac; the hashes of the kernel sources the counters were
collected on are recorded beside them, and bench.py reports none of it for other sources)."""
import csv, glob, hashlib, json, os, shutil, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05p"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r05"
PREV = "r%02d" % (int(TAG[1:]) - 1)
PARTIAL = len(sys.argv) > 3 and sys.argv[3] == "render"
py = sys.executable
for t in (() if PARTIAL else ("focal_graph", "global_graph")):
    subprocess.check_call([py, "tools/prof_summary.py", "stats", f"{R}/kt_{t}", f"profiles/{TAG}_{t}_kernel_stats.csv"])
    shutil.copy(f"{R}/{t}_under_rocprof.json", f"profiles/{TAG}_{t}_bench_under_rocprof.json")
subprocess.check_call([py, "tools/prof_summary.py", "stats", f"{R}/kt_render", f"profiles/{TAG}_render_kernel_stats.csv"])
shutil.copy(f"{R}/render_under_rocprof.json", f"profiles/{TAG}_render_bench_under_rocprof.json")
if not PARTIAL:
    subprocess.check_call([py, "tools/prof_summary.py", "stats", f"{R}/kt_plugin_focal", f"profiles/{TAG}_plugin_focal_kernel_stats.csv"])
    shutil.copy(f"{R}/plugin_focal_under_rocprof.txt", f"profiles/{TAG}_plugin_focal_under_rocprof.txt")
if not PARTIAL and os.path.exists(f"{R}/kt_plugin_trained"):  # the focal iteration on the scene trained through the plugin surface (the last 300 ms of the run: per step)
    subprocess.check_call([py, "tools/prof_summary.py", "stats", f"{R}/kt_plugin_trained", f"profiles/{TAG}_plugin_trained_kernel_stats.csv"])
    shutil.copy(f"{R}/plugin_trained_under_rocprof.txt", f"profiles/{TAG}_plugin_trained_under_rocprof.txt")
    shutil.copy(f"{R}/plugin_trained_window.txt", f"profiles/{TAG}_plugin_trained_window.txt")
    if os.path.exists(f"{R}/plugin_trained_timeline.txt"):
        shutil.copy(f"{R}/plugin_trained_timeline.txt", f"profiles/{TAG}_plugin_trained_timeline.txt")
shutil.copy(f"{R}/bench.json", f"profiles/{TAG}_bench.json")
shutil.copy(f"{R}/bench_details.json", f"profiles/{TAG}_bench_details.json")
if PARTIAL:  # (the occupancy passes alone, into their own tables)
    subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", f"profiles/{TAG}_pmc_focal_busy_per_kernel.csv"] + sorted(glob.glob(f"{R}/pmc_focal_SQ_VALU_MFMA_BUSY*/")))
    subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", f"profiles/{TAG}_pmc_global_busy_per_kernel.csv"] + sorted(glob.glob(f"{R}/pmc_global_SQ_VALU_MFMA_BUSY*/")))
else:
    subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", f"profiles/{TAG}_pmc_focal_per_kernel.csv"] + sorted(glob.glob(f"{R}/pmc_focal_*/")))
    subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", f"profiles/{TAG}_pmc_global_per_kernel.csv"] + sorted(glob.glob(f"{R}/pmc_global_*/")))
subprocess.check_call([py, "tools/prof_summary.py", "stats", f"{R}/kt_render_eager", f"profiles/{TAG}_render_eager_kernel_stats.csv"])
shutil.copy(f"{R}/render_eager_under_rocprof.json", f"profiles/{TAG}_render_eager_bench_under_rocprof.json")
subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", f"profiles/{TAG}_pmc_render_per_kernel.csv"] + sorted(glob.glob(f"{R}/pmc_render_*/")))
p = "profiles/pmc_traffic.json"
old = json.load(open(p))
j = {k: old[k] for k in ("gather_request_ceiling_per_s", "l2_gather_ceiling_per_s")}
rounds = [k for k in old if len(k) == 3 and k[0] == "r" and k[1:].isdigit()]
for k in rounds:
    j[k] = old[k]
if PREV not in old:  # the previous round's figures move under their key
    j[PREV] = {k: v for k, v in old.items() if k not in rounds + ["gather_request_ceiling_per_s", "l2_gather_ceiling_per_s"]}
load = lambda pp: {r["kernel"]: r for r in csv.DictReader(open(pp))}
f, g = load(f"profiles/{TAG}_pmc_focal_per_kernel.csv"), load(f"profiles/{TAG}_pmc_global_per_kernel.csv")
fb, gb = (load(f"profiles/{TAG}_pmc_focal_busy_per_kernel.csv"), load(f"profiles/{TAG}_pmc_global_busy_per_kernel.csv")) if PARTIAL else (f, g)
tr = lambda r: int(round((2 * float(r["FETCH_SIZE_mean"]) + float(r["WRITE_SIZE_mean"])) * 1024))
def one(d, sub):
    ks = [k for k in d if sub in k]
    assert len(ks) == 1, (sub, ks)
    return d[ks[0]]
busy = lambda r: round(float(r["SQ_VALU_MFMA_BUSY_CYCLES_mean"]) / (float(r["GRBM_GUI_ACTIVE_mean"]) * 128.0), 4)  # MFMA-busy cycles / SIMD cycles
# the focal step's lm_lookup launches: phase A of the two tables (the phase-B launches return at once: their counters are ~0)
lm = sorted([r for k, r in f.items() if "lm_lookup" in k], key=lambda r: -float(r["FETCH_SIZE_mean"]))
n_launch_rows = len(lm)
j["n_side"] = 40
j["_note"] = ("HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from separate rocprofv3 --pmc passes of the " + TAG + " build on the default workload "
              "(1,600 cameras, T = 2^21, own primes per table, early_T 1e-4, phased forward): profiles/" + TAG + "_pmc_{focal,global,render}_per_kernel.csv.  2 x: a request moves a 128-byte "
              "line and is tallied as 64 bytes (profiles/r02_pmc_fetch_calibration.csv).  lm_lookup rows are means over ALL launches of a kernel instantiation, "
              f"phase-B launches that return at once included ({n_launch_rows} instantiations): bench.py uses the per-launch figures of the instantiation with the larger FETCH.")
# per-launch means include the gated phase-B launches (no traffic): scale to the launches that work.  launches column = per pass
def lm_live(r, key):
    return float(r[key])
j["lm_lookup:focal"] = sum(tr(r) for r in lm) // max(len(lm), 1)
j["lm_lookup:focal:fetch_requests"] = int(round(sum(float(r["TCC_EA0_RDREQ_mean"]) for r in lm) / len(lm)))
j["lm_lookup:focal:l1_miss_requests"] = int(round(sum(float(r["TCP_TCC_READ_REQ_sum_mean"]) for r in lm) / len(lm)))
hit, miss = sum(float(r["TCC_HIT_mean"]) for r in lm), sum(float(r["TCC_MISS_mean"]) for r in lm)
j["lm_lookup:focal:l2_hit_rate"] = round(hit / (hit + miss), 3)
j["lm_lookup:focal:launch_mix"] = "mean over phase-A and (empty) phase-B launches of each instantiation: multiply by 2 for a phase-A launch"
for key, d, sub in (("bwd_scatter:focal", f, "field_bwd_dx"), ("grad_reduce:focal", f, "grad_accumulate"), ("adam:focal", f, "adam_ex"), ("field_fwd:focal", f, "field_fwd_x3"),
                    ("bwd_scatter:global", g, "field_bwd_dx"), ("wgrad:global", g, "wgrad_x3_layers"), ("field_fwd:global", g, "field_fwd_x3"),
                    ("grad_reduce:global", g, "grad_accumulate"), ("adam:global", g, "adam_ex")):
    try:
        j[key] = tr(one(d, sub))
    except AssertionError as e:
        print("skip", key, e)
glm = [r for k, r in g.items() if "lm_lookup" in k]
if glm:
    j["lm_lookup:global"] = j["hash_fwd:global"] = sum(tr(r) for r in glm) // len(glm)
j["mfma_busy:field_fwd:focal"] = busy(one(fb, "field_fwd_x3"))
j["mfma_busy:bwd_scatter:focal"] = busy(one(fb, "field_bwd_dx"))
j["mfma_busy:field_fwd:global"] = busy(one(gb, "field_fwd_x3"))
j["mfma_busy:bwd_scatter:global"] = busy(one(gb, "field_bwd_dx"))
j["mfma_busy:wgrad:global"] = busy(one(gb, "wgrad_x3_layers"))
if PARTIAL:
    j["_note_partial"] = ("the traffic figures of the focal / global steps' kernels are the full run's (" + TAG + "_pmc_{focal,global}_per_kernel.csv: lm_lookup.hip, hash_device.h, grad_part.hip, "
                          "render.hip unchanged since); after csrc/x3_device.h changed (inference-only template parameters: the training kernels' code objects are the same "
                          "instructions, checked by disassembly) the decoder kernels' matrix-pipe occupancy was measured again (" + TAG + "_pmc_{focal,global}_busy_per_kernel.csv) "
                          "and the inference kernel's traces and counters collected anew (" + TAG + "_render_*, " + TAG + "_pmc_render_per_kernel.csv)")
# the inference kernel (one eager launch per 138,240-ray tile): matrix / vector pipe occupancy, what it decodes, its duration
rd = load(f"profiles/{TAG}_pmc_render_per_kernel.csv")
rk = one(rd, "fused_render_patch_kernel")
fv = lambda key: float(rk[key + "_mean"])
simd_cycles = fv("GRBM_GUI_ACTIVE") * 128.0  # GUI_ACTIVE sums the 8 XCDs; 1,024 SIMDs
j["mfma_busy:render"] = round(fv("SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles, 4)
j["valu_busy:render"] = round(4.0 * fv("SQ_ACTIVE_INST_VALU") / simd_cycles, 4)  # SQ_ACTIVE_INST_* count quad-cycles
j["wait_frac:render"] = round(fv("SQ_WAIT_INST_ANY") / fv("SQ_WAVE_CYCLES"), 4)
j["mfma_valu_coexec_frac:render"] = round(fv("SQ_VALU_MFMA_COEXEC_CYCLES") / fv("SQ_VALU_MFMA_BUSY_CYCLES"), 4)
MFMA_PER_TILE = 160.0 if int(TAG[1:]) >= 6 else 216.0  # MFMAs per 32-sample tile of the inference decoder (round 6: the image's embedding folded into the bias, 24 less;
                                                       # colour_x2, the renderer's default: head layer 2 issues 64 instead of 96)
j["decoded_samples_per_launch:render"] = int(round(fv("SQ_INSTS_MFMA") / MFMA_PER_TILE * 32.0))
j["mfma_per_tile:render"] = MFMA_PER_TILE
# (per-launch figures belong to the tile the eager run rendered: the renderer's default for a full-HD frame, bench.py scales by its own tile)
_src = open("gf-nerf_amd/render.py").read()
_ns = {}
exec("import math\n" + _src[_src.index("MAX_TILE_RAYS = "):_src.index("def frame_shard")], _ns)
_tw, _th = _ns["choose_tile"](1080, 1920)
j["tile_rays:render"] = _tw * _th
j["valu_insts_per_tile:render"] = round(fv("SQ_INSTS_VALU") / (fv("SQ_INSTS_MFMA") / MFMA_PER_TILE), 1)
j["l1_miss_requests:render"] = int(round(fv("TCP_TCC_READ_REQ_sum")))
j["l2_hit_rate:render"] = round(fv("TCC_HIT") / (fv("TCC_HIT") + fv("TCC_MISS")), 3)
j["traffic:render"] = tr(rk)
# the march of a render tile (quad of lanes per ray; the same eager run): bound by vector-instruction issue
mq = one(rd, "march_group_kernel<4")
j["valu_busy:march_quad:render"] = round(4.0 * float(mq["SQ_ACTIVE_INST_VALU_mean"]) / (float(mq["GRBM_GUI_ACTIVE_mean"]) * 128.0), 4)
j["valu_insts_per_launch:march_quad:render"] = int(round(float(mq["SQ_INSTS_VALU_mean"])))
ks = {r["kernel"]: r for r in csv.DictReader(open(f"profiles/{TAG}_render_eager_kernel_stats.csv"))}
j["kernel_ms:render"] = round(float(one(ks, "fused_render_patch_kernel")["avg_ns"]) / 1e6, 4)
sha = lambda q: hashlib.sha256(open(os.path.join("gf-nerf_amd", q), "rb").read()).hexdigest()
j["source_sha256"] = {"lm_lookup": {q: sha(q) for q in ("csrc/lm_lookup.hip", "csrc/hash_device.h")},
                      "decoder": {q: sha(q) for q in ("csrc/field_x3.hip", "csrc/x3_device.h")},
                      "render": {q: sha(q) for q in ("csrc/fused_fwd.hip", "csrc/x3_device.h", "csrc/hash_device.h")}}
json.dump(j, open(p, "w"), indent=1)
for k in sorted(j):
    if ":" in k and not isinstance(j[k], dict):
        print(k, j[k])
b = json.load(open(f"profiles/{TAG}_bench.json"))
print("bench:", b["ms_per_step"], b["value"], b["roofline"]["frac"], b["config"])
