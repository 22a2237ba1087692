"""Per-launch timeline of a kernel trace (rocprofv3 --kernel-trace csv): start, duration, gap to the previous kernel."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = None; prev_end = None
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:36]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_generate" in n: t0 = s
    if t0 is None: continue
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{(s - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {n} grid {r['Grid_Size_X']}")
    prev_end = max(prev_end or 0, e)
