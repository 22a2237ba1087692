"""Aggregate rocprofv3 counter_collection.csv files: sum of each counter per kernel name."""
import csv, sys, collections, re
agg = collections.defaultdict(float); calls = collections.defaultdict(set)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_]+(<[a-z, ]+>)?|__amd_[A-Za-z_]+)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:40]
        agg[(k, r['Counter_Name'])] += float(r['Counter_Value'])
        calls[k].add(r['Dispatch_Id'])
for k in sorted(calls):
    print(f"== {k}  dispatches={len(calls[k])}")
    for (kk, c), v in sorted(agg.items()):
        if kk == k and v: print(f"   {c:36s} {v:18.0f}")
