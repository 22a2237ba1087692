#!/bin/bash
# per-launch counters of k_shade over ONE 64-spp frame: vector instructions per wave, waves, busy cycles (is a listed launch's
# wave longer because it executes more, or because it waits more?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shade_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $O/p -o p --output-format csv -- python3 $R/scripts/prof_render.py 64 9 render 1920 > $O/run.log 2>&1 || { tail -3 $O/run.log; exit 1; }
python3 - $O/p <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_shade" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (d, c) in enumerate(list(by.items())[:6]):
    w = c.get("SQ_WAVES", 1)
    print("k_shade launch", i, "waves %.0f" % w, "VALU/wave %.0f" % (c.get("SQ_INSTS_VALU", 0) / w), "VMEM_RD/wave %.1f" % (c.get("SQ_INSTS_VMEM_RD", 0) / w),
          "wave cycles/wave %.0f" % (4 * c.get("SQ_WAVE_CYCLES", 0) / w), "waiting %.2f" % (c.get("SQ_WAIT_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1))),
          "VALU active/wave %.0f" % (4 * c.get("SQ_ACTIVE_INST_VALU", 0) / w))
PY
rm -rf $O/p
