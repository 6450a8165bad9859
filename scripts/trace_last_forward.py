"""Dev tool: print the launches of the LAST forward in a rocprofv3 kernel trace of scripts/trace_forward.py
(start offset us, duration us, kernel) with the span and the busy time: python scripts/trace_last_forward.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pack_input" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
busy = 0
for r in rows[a:b]:
    s = int(r["Start_Timestamp"]) - t0
    e = int(r["End_Timestamp"]) - t0
    busy += e - s
    print("%8.1f %7.1f  %s" % (s / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
print("span us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "busy us", busy / 1e3, "launches", b - a)
