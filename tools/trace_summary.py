"""Per-sort kernel timeline from a rocprofv3 kernel-trace CSV: python tools/trace_summary.py trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into sorts at every rsx_generate_kernel
sorts, cur = [], []
for r in rows:
    name = r["Kernel_Name"]
    if "generate" in name:
        if cur: sorts.append(cur)
        cur = []
    elif cur is not None and ("rsx" in name or "fill" in name.lower() or "memset" in name.lower()):
        cur.append(r)
if cur: sorts.append(cur)
for s in sorts[-2:]:
    t0 = int(s[0]["Start_Timestamp"]); prev_end = t0
    print("--- sort: total %.1f us" % ((int(s[-1]["End_Timestamp"]) - t0) / 1e3))
    for r in s:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].split("(")[0]
        nm = nm[nm.find("rsx"):][:70] if "rsx" in nm else nm[:70]
        print("  gap %5.1f  dur %6.1f  %s  grid=%s wg=%s" % ((st - prev_end) / 1e3, (en - st) / 1e3, nm, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
        prev_end = en
