"""Copy the summaries of a tools/r02_profiles.sh run (gpurun_out/r02p) into profiles/r02_* and refresh profiles/pmc_traffic.json."""
import csv, json, os, shutil, subprocess, sys, glob
R = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02p"
py = sys.executable
for t in ("focal_graph", "focal_eager", "global_graph", "global_eager"):
    subprocess.check_call([py, "tools/prof_summary.py", "stats", f"{R}/kt_{t}", f"profiles/r02_{t}_kernel_stats.csv"])
    shutil.copy(f"{R}/{t}_under_rocprof.json", f"profiles/r02_{t}_bench_under_rocprof.json")
subprocess.check_call([py, "tools/prof_summary.py", "stats", f"{R}/kt_render", "profiles/r02_render_kernel_stats.csv"])
shutil.copy(f"{R}/render_under_rocprof.json", "profiles/r02_render_bench_under_rocprof.json")
shutil.copy(f"{R}/bench.json", "profiles/r02_bench.json")
subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", "profiles/r02_pmc_focal_per_kernel.csv"] + sorted(glob.glob(f"{R}/pmc_focal_*/")))
subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", "profiles/r02_pmc_global_per_kernel.csv"] + sorted(glob.glob(f"{R}/pmc_global_*/")))
subprocess.check_call([py, "tools/prof_summary.py", "pmc_many", "profiles/r02_pmc_fetch_calibration.csv"] + sorted(glob.glob(f"{R}/calib_*/")))
shutil.copy(f"{R}/calib_FETCH_SIZE.txt", "profiles/r02_fetch_calibration_timings.txt")
p = "profiles/pmc_traffic.json"
j = json.load(open(p))
load = lambda pp: {r["kernel"]: r for r in csv.DictReader(open(pp))}
f, g = load("profiles/r02_pmc_focal_per_kernel.csv"), load("profiles/r02_pmc_global_per_kernel.csv")
tr = lambda r: int(round((2 * float(r["FETCH_SIZE_mean"]) + float(r["WRITE_SIZE_mean"])) * 1024))
def one(d, sub):
    ks = [k for k in d if sub in k]
    assert len(ks) == 1, (sub, ks)
    return d[ks[0]]
lm = [r for k, r in f.items() if "lm_lookup" in k]
j["lm_lookup:focal"] = sum(tr(r) for r in lm) // 2
j["lm_lookup:focal:fetch_requests"] = int(round(sum(float(r["TCC_EA0_RDREQ_mean"]) for r in lm) / 2))
j["lm_lookup:focal:l1_miss_requests"] = int(round(sum(float(r["TCP_TCC_READ_REQ_sum_mean"]) for r in lm) / 2))
hit, miss = sum(float(r["TCC_HIT_mean"]) for r in lm), sum(float(r["TCC_MISS_mean"]) for r in lm)
j["lm_lookup:focal:l2_hit_rate"] = round(hit / (hit + miss), 3)
for key, d, sub in (("bwd_scatter:focal", f, "field_bwd_dx"), ("grad_reduce:focal", f, "grad_accumulate"), ("adam:focal", f, "adam_ex"), ("field_fwd:focal", f, "field_fwd_x3"),
                    ("lm_lookup:global", g, "lm_lookup"), ("hash_fwd:global", g, "lm_lookup"), ("bwd_scatter:global", g, "field_bwd_dx"), ("wgrad:global", g, "wgrad_x3_layers"),
                    ("field_fwd:global", g, "field_fwd_x3"), ("grad_reduce:global", g, "grad_accumulate"), ("adam:global", g, "adam_ex"), ("fused_fwd:global", g, "fused_fwd_kernel")):
    j[key] = tr(one(d, sub))
json.dump(j, open(p, "w"), indent=1)
for c in ("FETCH_SIZE_mean", "WRITE_SIZE_mean", "TCC_HIT_mean", "TCC_MISS_mean", "TCC_EA0_RDREQ_mean", "TCP_TCC_READ_REQ_sum_mean"):
    print(c, sum(float(r[c]) for r in lm) / 2)
b = json.load(open("profiles/r02_bench.json"))
print("bench:", b["ms_per_step"], b["value"], b["roofline"]["frac"], b["roofline"]["kernel_ms"], b["global_stage"]["ms_per_step"], b["global_stage"]["value"],
      b["render"]["value"], b["cpu_baseline"]["value"])
