"""Condense rocprofv3 output (gpurun_out/...) into the small CSVs committed under profiles/.

    python tools/prof_summary.py stats <dir> <out.csv>          # --kernel-trace --stats run
    python tools/prof_summary.py pmc <fetch_dir> <write_dir> <out.csv>   # two --pmc passes (FETCH_SIZE, WRITE_SIZE)
"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def stats(d, out):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "percent", "min_ns", "max_ns"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def pmc_table(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
