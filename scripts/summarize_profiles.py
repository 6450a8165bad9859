"""Turn rocprofv3 output directories into the committed summaries under profiles/.

    python scripts/summarize_profiles.py stats <dir> <out.csv> [steps]
        <dir> holds *_kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`;
        writes the per-kernel table (calls, total, average, percentage) and, with `steps`, ms per step.
    python scripts/summarize_profiles.py pmc <fetch_dir> <write_dir> <out.json>
        the two directories hold *_counter_collection.csv of separate `--pmc FETCH_SIZE` and
        `--pmc WRITE_SIZE` passes of the same command.  Per kernel: launches, average counter values (KB)
        and HBM bytes per launch, with FETCH_SIZE doubled (the gfx950 correction of
        /opt/skills/guides/MI355X_MICROARCH.md: wide coalesced reads are tallied at half their size).
"""
import csv
import glob
import json
import os
import sys


def find(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not f:
        raise SystemExit("no %s under %s" % (pat, d))
    return f[-1]


def short(name):
    return name.split("(")[0].replace("void ", "")[:110]


def stats(d, out, steps=None):
    rows = list(csv.DictReader(open(find(d, "*_kernel_stats.csv"))))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"] + (["ms_per_step"] if steps else []))
        for r in rows:
            tot = float(r["TotalDurationNs"]) / 1e3
            line = [short(r["Name"]), r["Calls"], "%.1f" % tot, "%.2f" % (float(r["AverageNs"]) / 1e3),
                    "%.2f" % float(r["Percentage"])]
            if steps:
                line.append("%.4f" % (tot / 1e3 / steps))
            w.writerow(line)
    print("wrote", out, len(rows), "kernels")


def pmc(fd, wd, out):
    def load(d, counter):
        acc = {}
        for r in csv.DictReader(open(find(d, "*_counter_collection.csv"))):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        return acc
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {}
    for k, (n, v) in fe.items():
        wv = wr.get(k, [n, 0.0])
        f_kb, w_kb = v / n, wv[1] / max(wv[0], 1)
        res[k] = {"launches": n, "FETCH_SIZE_KB_avg": f_kb, "WRITE_SIZE_KB_avg": w_kb,
                  "hbm_bytes_per_launch_corrected": (2.0 * f_kb + w_kb) * 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
