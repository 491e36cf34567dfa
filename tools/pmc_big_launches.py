"""Per kernel, the counters of its LARGEST launches only (the step's launch, not the few-robot launches of a reset chain):
average per launch over the dispatches whose grid is at least half the kernel's largest.
usage: python tools/pmc_big_launches.py dir [dir...]"""
import collections
import csv
import glob
import sys

rows = []
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
biggest = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:44]
    biggest[k] = max(biggest[k], int(r["Grid_Size"]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:44]
    if int(r["Grid_Size"]) * 2 < biggest[k]:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    if not (k.startswith("k_") or k.startswith("void k_")):
        continue
    print("%s   (grid %d)" % (k, biggest[k]))
    for c in sorted(agg[k]):
        print("    %-40s %16.0f /launch  (%d launches)" % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
