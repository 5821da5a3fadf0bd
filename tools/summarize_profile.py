#!/usr/bin/env python3
"""Summarise the csv output of tools/profile_round.sh: per-kernel durations from the kernel trace, and per-kernel
averages of every PMC counter (per dispatch).  usage: python tools/summarize_profile.py gpurun_out/<dir> [name-substring]"""
import csv, glob, sys, collections
root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "hnsw"
def short(n):
    n = n.replace("void hnswgpu::(anonymous namespace)::", "").replace("hnswgpu::(anonymous namespace)::", "")
    return n.split("(")[0]
for f in glob.glob(f"{root}/kt/**/*_kernel_stats.csv", recursive=True):
    print("# kernel trace (--kernel-trace --stats): calls, average us, min us, max us, % of GPU time")
    for r in csv.DictReader(open(f)):
        if sub in r["Name"] or float(r["Percentage"]) > 0.5:
            print(f"  {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}  {short(r['Name'])}")
acc = collections.defaultdict(list)
for f in glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)  # (dispatch, kernel, counter) -> value summed over the rows of a dispatch
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[(k, c)].append(v)
print("# PMC counters, average per dispatch (separate --pmc passes)")
for (k, c), v in sorted(acc.items()):
    print(f"  {k:45s} {c:24s} avg {sum(v)/len(v):14.6g}  (min {min(v):.6g}, max {max(v):.6g}, n={len(v)})")
