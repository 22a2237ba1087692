#!/bin/bash
# per-launch k_classify / k_shade durations of ONE 64-spp frame of the product library (RT_LIST_PATHS etc. from the environment)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shade_probe; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python3 $R/scripts/prof_render.py 64 9 render 1920 > $O/run.log 2>&1 || { tail -3 $O/run.log; exit 1; }
python3 - $O/kt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for name in ("k_classify", "k_shade"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if name in r["Kernel_Name"]]
    print(name, "launches", len(d), "first five ms:", [round(x, 3) for x in d[:5]], "sum", round(sum(d), 2))
PY
rm -rf $O/kt
