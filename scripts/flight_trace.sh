#!/bin/bash
# kernel trace of bench.py with frames in flight, plain and with the forced-distributed path: which queue and stream
# the large launches of successive frames ran on and whether they overlapped
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/flight_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/plain -o kt --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/plain.json 2> $O/plain.err || exit 1
RT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/dist -o kt --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/dist.json 2> $O/dist.err || exit 1
cd $R
python3 - <<'PY'
import csv, glob
for name in ("plain", "dist"):
    f = glob.glob(f"gpurun_out/flight_trace/{name}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void k_extend") or "k_extend" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    print(name, "k_extend launches:", len(rows), "columns:", [c for c in rows[0].keys() if "ueue" in c or "tream" in c or "Thread" in c])
    for r in rows[-40:]:
        print("  q", r.get("Queue_Id"), "s", r.get("Stream_Id"), "tid", r.get("Thread_Id"),
              f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.2f} .. {(int(r["End_Timestamp"]) - t0) / 1e6:9.2f} ms')
PY
rm -rf $O/plain $O/dist
