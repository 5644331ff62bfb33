"""Condense rocprofv3 output (gpurun_out/...) into the small CSVs committed under profiles/.

    python tools/prof_summary.py stats <dir> <out.csv>          # --kernel-trace --stats run
    python tools/prof_summary.py pmc <fetch_dir> <write_dir> <out.csv>   # two --pmc passes (FETCH_SIZE, WRITE_SIZE)
    python tools/prof_summary.py pmc_many <out.csv> <dir>...             # any number of --pmc passes, every counter found
"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def stats(d, out):
    f = max(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)  # (a merged gpurun_out/ keeps earlier runs' files: the newest)
    rows = list(csv.DictReader(open(f)))
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc_table(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def pmc(fd, wd, out):
    fe, wr = pmc_table(fd, "FETCH_SIZE"), pmc_table(wd, "WRITE_SIZE")
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_mean_raw", "WRITE_SIZE_KB_mean"])
        for k in sorted(fe, key=lambda k: -sum(fe[k])):
            wv = wr.get(k, [0.0])
            w.writerow([short(k), len(fe[k]), round(sum(fe[k]) / len(fe[k]), 1), round(sum(wv) / len(wv), 1)])



def pmc_many(dirs, out):
    """pmc_many <out.csv> <dir>...: mean per launch of every counter found in several --pmc passes (one row per kernel)."""
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        found = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        for f in ([max(found, key=os.path.getmtime)] if found else []):  # (one pass per directory: the newest, as in stats())
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc for c in acc[k]})
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "launches"] + [c + "_mean" for c in counters])
        order = sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values()))
        for k in order:
            n = max(len(v) for v in acc[k].values())
            w.writerow([k, n] + [round(sum(acc[k][c]) / len(acc[k][c]), 1) if acc[k][c] else "" for c in counters])

if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmc_many":
        pmc_many(sys.argv[3:], sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])

