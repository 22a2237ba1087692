#!/bin/bash
# usage (on the GPU box): bash scripts/fetch_calib.sh OUTDIR -> gpurun_out/OUTDIR/fetch_size_calibration.{txt,json}
# FETCH_SIZE (one --pmc pass of its own) of scripts/microbench/fetch_calib: known bytes per access shape, table of 2 GiB read once.
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
timeout -k 10 120 $R/scripts/microbench/fetch_calib 2048 > $D/fetch_calib_run.txt 2>&1 || { echo "plain run failed"; tail -3 $D/fetch_calib_run.txt; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $D/fs -o fs --output-format csv -- $R/scripts/microbench/fetch_calib 2048 > $D/fs.log 2>&1 || { echo "pmc pass failed"; tail -5 $D/fs.log; exit 1; }
python3 - $D <<'PY'
import csv, glob, json, os, re, sys
d = sys.argv[1]
req = {}
for line in open(os.path.join(d, "fetch_calib_run.txt")):
    m = re.match(r"(k_\w+)\s+requested\s+(\d+) bytes\s+([\d.]+) ms", line)
    if m: req[m.group(1)] = (float(m.group(2)), float(m.group(3)))
got = {}
for f in glob.glob(os.path.join(d, "fs", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_stream|k_gather)(<(\d+), *(\d+)>)?", r["Kernel_Name"])
        if not m or r["Counter_Name"] != "FETCH_SIZE": continue
        name = "k_stream" if m.group(1) == "k_stream" else {("64", "4"): "k_gather64", ("48", "3"): "k_gather48", ("128", "8"): "k_gather128", ("64", "1"): "k_gather16"}[(m.group(3), m.group(4))]
        got[name] = got.get(name, 0.0) + float(r["Counter_Value"]) * 1024.0  # KiB
out = {"note": "FETCH_SIZE (KiB x 1024) against the bytes each kernel requests once from a 2 GiB table; factor = requested / counter: "
               "what to multiply FETCH_SIZE by for that access shape", "shapes": {}}
for k, (b, ms) in req.items():
    if k in got:
        out["shapes"][k] = {"requested_bytes": b, "fetch_size_bytes": got[k], "factor": round(b / got[k], 4), "ms": ms,
                            "requested_GBs": round(b / ms / 1e6, 1), "counter_GBs": round(got[k] / ms / 1e6, 1)}
json.dump(out, open(os.path.join(d, "fetch_size_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $D/fs
