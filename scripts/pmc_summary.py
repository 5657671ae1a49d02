#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (scripts/pmc_passes.sh) per kernel: mean counter per dispatch."""
import csv, glob, os, sys, collections, json
d = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, '*', '*', '*counter_collection.csv')) + glob.glob(os.path.join(d, '*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
        res[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(os.path.join(d, '*', '*', '*kernel_trace.csv')) + glob.glob(os.path.join(d, '*', '*kernel_trace.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:40]
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = {}
for k in res:
    out[k] = {c: sum(v) / len(v) for c, v in res[k].items()}
    out[k]['dispatches_per_pass'] = len(next(iter(res[k].values())))
    if dur[k]:
        out[k]['mean_ns_under_pmc'] = sum(dur[k]) / len(dur[k])
print(json.dumps(out, indent=1, sort_keys=True))
