"""Turn rocprofv3 output directories into the committed summaries under profiles/.

    python scripts/summarize_profiles.py stats <dir> <out.csv> [steps]
        <dir> holds *_kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`;
        writes the per-kernel table (calls, total, average, percentage) and, with `steps`, ms per step --
        `steps` = EVERY step the traced command ran (warm-up + timed: round 2 divided by the timed steps only).
    python scripts/summarize_profiles.py pmc <fetch_dir> <write_dir> <out.json>
        the two directories hold *_counter_collection.csv of separate `--pmc FETCH_SIZE` and
        `--pmc WRITE_SIZE` passes of the same command.  Per kernel: launches, average counter values (KB)
        and HBM bytes per launch, with FETCH_SIZE doubled (the gfx950 correction of
        /opt/skills/guides/MI355X_MICROARCH.md: wide coalesced reads are tallied at half their size).
    python scripts/summarize_profiles.py trace <dir> <out.csv> <steps>
        <dir> holds *_kernel_trace.csv of `rocprofv3 --kernel-trace -- python3 bench.py --steps N ...`; writes the
        per-dispatch timeline (start, end relative to the step's first launch, in us) of the LAST step and, in
        the header comment, what bench.py's roofline uses: the length of the UNION of the MFMA convolution
        launches' intervals, their summed durations and the step's span.
    python scripts/summarize_profiles.py gbps <stats.csv> <hbm.json> <out.csv>
        HBM bytes per launch (pmc summary, FETCH doubled) / average launch duration (stats summary) per kernel.
    python scripts/summarize_profiles.py sq <dir> <out.csv>
        <dir> holds *_counter_collection.csv of one `--pmc SQ_...` pass; per kernel: launches and the average of
        every counter, plus MFMA busy / CU busy and LDS conflict ratios where the inputs are present.
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


MFMA_KERNELS = ("conv_haloq", "conv_halo_kernel", "conv_igemm_kernel", "conv_rf_kernel", "conv_rfn_kernel", "wgrad9",
                "wgrad_kernel", "gemm1x1", "conv_big", "wgrad1x1")


def is_mfma_conv(name):
    return any(k in name for k in MFMA_KERNELS)


def trace(d, out, steps):
    rows = [r for r in csv.DictReader(open(find(d, "*_kernel_trace.csv"))) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at the image pack of the first layer (pack_input_kernel); keep the last complete one
    starts = [i for i, r in enumerate(rows) if "pack_input" in r["Kernel_Name"]]
    if len(starts) < 2:
        raise SystemExit("no step boundaries found")
    # the last step ends where the optimizer kernel of that step ends
    lo = starts[-1]
    hi = len(rows)
    step = rows[lo:hi]
    for j, r in enumerate(step):
        if "adam" in r["Kernel_Name"] or "momentum" in r["Kernel_Name"]:
            hi = lo + j + 1
            break
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in step
                if is_mfma_conv(r["Kernel_Name"]))
    union, cur0, cur1 = 0, None, None
    for a, b in iv:
        if cur1 is None or a > cur1:
            if cur1 is not None:
                union += cur1 - cur0
            cur0, cur1 = a, b
        elif b > cur1:
            cur1 = b
    if cur1 is not None:
        union += cur1 - cur0
    span = int(step[-1]["End_Timestamp"]) - t0
    with open(out, "w", newline="") as f:
        f.write("# last step of %s: %d dispatches, span %.1f us; MFMA convolution launches: %d, union of intervals "
                "%.1f us, sum of durations %.1f us\n" % (os.path.basename(d.rstrip("/")), len(step), span / 1e3, len(iv),
                                                         union / 1e3, sum(b - a for a, b in iv) / 1e3))
        w = csv.writer(f)
        w.writerow(["start_us", "end_us", "dur_us", "stream", "mfma_conv", "grid", "wg", "lds", "vgpr", "kernel"])
        for r in step:
            a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            w.writerow(["%.2f" % (a / 1e3), "%.2f" % (b / 1e3), "%.2f" % ((b - a) / 1e3), r.get("Stream_Id", ""),
                        int(is_mfma_conv(r["Kernel_Name"])), r["Grid_Size_X"], r["Workgroup_Size_X"],
                        r["LDS_Block_Size"], r["VGPR_Count"], short(r["Kernel_Name"])])
    print("wrote", out, "union %.1f us over %d launches, span %.1f us" % (union / 1e3, len(iv), span / 1e3))


def step_between(d, out, marker):
    """one steady-state step = the dispatches between the last two occurrences of a kernel that runs once per step
    (ResNet swap: rn_conv7_fwd_kernel): dispatch count, span, busy time (union of all intervals), per-kernel totals"""
    rows = [r for r in csv.DictReader(open(find(d, "*_kernel_trace.csv"))) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit("marker kernel seen fewer than twice")
    step = rows[marks[-2]:marks[-1]]
    t0 = int(step[0]["Start_Timestamp"])
    span = int(rows[marks[-1]]["Start_Timestamp"]) - t0
    iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in step)
    union, cur0, cur1 = 0, None, None
    for a, b in iv:
        if cur1 is None or a > cur1:
            if cur1 is not None:
                union += cur1 - cur0
            cur0, cur1 = a, b
        elif b > cur1:
            cur1 = b
    union += cur1 - cur0
    acc = {}
    for r in step:
        k = short(r["Kernel_Name"])
        v = acc.setdefault(k, [0, 0])
        v[0] += 1
        v[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    with open(out, "w", newline="") as f:
        f.write("# one step of %s (between two %s): %d dispatches, span %.1f us, busy (union of intervals) %.1f us, sum of "
                "durations %.1f us\n" % (os.path.basename(d.rstrip("/")), marker, len(step), span / 1e3, union / 1e3,
                                         sum(b - a for a, b in iv) / 1e3))
        w = csv.writer(f)
        w.writerow(["kernel", "calls_per_step", "total_us", "avg_us"])
        for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, n, "%.1f" % (t / 1e3), "%.2f" % (t / 1e3 / n)])
    print("wrote", out, "%d dispatches, span %.1f us, busy %.1f us" % (len(step), span / 1e3, union / 1e3))


def sq(d, out):
    acc = {}
    names = []
    for r in csv.DictReader(open(find(d, "*_counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        a = acc.setdefault(k, {})
        v = a.setdefault(c, [0, 0.0])
        v[0] += 1
        v[1] += float(r["Counter_Value"])
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        extra = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in names and "SQ_BUSY_CU_CYCLES" in names:
            extra.append("mfma_busy_per_cu_busy")
        if "SQ_LDS_BANK_CONFLICT" in names and "SQ_LDS_IDX_ACTIVE" in names:
            extra.append("lds_conflict_frac")
        if "SQ_WAIT_ANY" in names and "SQ_WAVE_CYCLES" in names:
            extra.append("wait_any_frac")
        w.writerow(["kernel", "launches"] + names + extra)
        for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get(names[0], [0, 0])[1]):
            n = max(v[0] for v in a.values())
            avg = {c: (a[c][1] / a[c][0] if c in a else 0.0) for c in names}
            row = [k, n] + ["%.0f" % avg[c] for c in names]
            for e in extra:
                if e == "mfma_busy_per_cu_busy":
                    row.append("%.3f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / max(avg["SQ_BUSY_CU_CYCLES"], 1)))
                elif e == "lds_conflict_frac":
                    row.append("%.3f" % (avg["SQ_LDS_BANK_CONFLICT"] / max(avg["SQ_LDS_IDX_ACTIVE"], 1)))
                else:
                    row.append("%.3f" % (avg["SQ_WAIT_ANY"] / max(avg["SQ_WAVE_CYCLES"], 1)))
            w.writerow(row)
    print("wrote", out, len(acc), "kernels")


def gbps(stats_csv, hbm_json, out):
    st = {r["kernel"]: r for r in csv.DictReader(open(stats_csv))}
    hb = json.load(open(hbm_json))
    rows = []
    for k, v in hb.items():
        r = st.get(k)
        if r is None:
            continue
        us = float(r["avg_us"])
        b = v["hbm_bytes_per_launch_corrected"]
        rows.append((float(r["total_us"]), k, r["calls"], us, b, b / (us * 1e-6) / 1e9 if us > 0 else 0.0))
    rows.sort(reverse=True)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "avg_us", "hbm_MB_per_launch(FETCHx2+WRITE)", "GB_per_s", "frac_of_8TBps"])
        for _t, k, calls, us, b, g in rows:
            w.writerow([k, calls, "%.2f" % us, "%.2f" % (b / 1e6), "%.0f" % g, "%.3f" % (g / 8000.0)])
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    elif sys.argv[1] == "step":
        step_between(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "gbps":
        gbps(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
