"""Per-kernel HBM traffic of one rendered frame from the rocprofv3 --pmc passes of scripts/pmc_traffic.sh."""
import collections, csv, glob, json, os, re, sys
d = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "render"
width = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
build = sys.argv[5] if len(sys.argv) > 5 else None
# FETCH_SIZE counts fabric-side read REQUESTS at 64 B each: a 128-byte request (wide coalesced streams) is tallied at half its
# size, a 64-byte one (a lane's child-pair gather that misses the L2) at its size.  Factors measured on known byte counts:
# profiles/r03/fetch_size_calibration.json (scripts/fetch_calib.sh): stream 2.00, 64-byte gathers 1.01, 128-byte gathers 1.96.
GATHER_KERNELS = ("k_extend", "k_shadow", "k_mixed", "k_tail", "k_trace")
def fetch_factor(kernel):
    return 1.0 if kernel.startswith(GATHER_KERNELS) else 2.0
agg = collections.defaultdict(float)
calls = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+)(<[0-9]>)?", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)  # instantiations of one kernel (k_extend_lean<1>: first bounce, <0>: later ones) are one entry
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
kernels = {}
for k in sorted({k for k, _ in agg}):
    n = len(calls.get((k, "FETCH_SIZE"), ())) or 1
    fetch_raw = agg.get((k, "FETCH_SIZE"), 0.0) * 1024.0   # FETCH_SIZE / WRITE_SIZE are in KiB
    write = agg.get((k, "WRITE_SIZE"), 0.0) * 1024.0
    fetch = fetch_factor(k) * fetch_raw
    hit, miss = agg.get((k, "TCC_HIT_sum"), 0.0), agg.get((k, "TCC_MISS_sum"), 0.0)
    kernels[k] = {"launches": n, "fetch_bytes_raw": fetch_raw, "fetch_factor": fetch_factor(k), "fetch_bytes": fetch, "write_bytes": write,
                  "fetch_requests_per_launch": fetch_raw / 64.0 / n,
                  "hbm_bytes_per_launch": (fetch + write) / n,
                  "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None}
print(json.dumps({
    "workload": {"name": "config5" if wl.startswith("config5") else wl, "instanced": wl == "config5i", "levels": 9, "width": width, "spp": spp, "depth": 16},
    "build": build,
    "note": "fetch_bytes = FETCH_SIZE x the factor calibrated for the kernel's access shape (profiles/r03/fetch_size_calibration.json: "
            "1.0 for the traversal kernels' 64-byte gathers, 2.0 for streaming kernels); the counter sits on the fabric side of L2 and "
            "also counts Infinity-Cache hits, so this is an upper bound on HBM bytes. One frame = scripts/prof_render.py 64.",
    "kernels": kernels}, indent=1))
