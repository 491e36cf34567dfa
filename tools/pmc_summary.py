"""Summarise rocprofv3 --pmc CSV output: per kernel, counter averages per launch and per wave.
usage: python tools/pmc_summary.py gpurun_out/pmc_dir [more dirs...]"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
for k in sorted(agg):
    if not k.startswith("k_") and not k.startswith("void k_"):
        continue
    waves = agg[k].get("SQ_WAVES", 0) / max(cnt[k].get("SQ_WAVES", 1), 1)
    print("%-28s waves/launch %d" % (k, waves))
    for c in sorted(agg[k]):
        per = agg[k][c] / cnt[k][c]
        print("    %-28s %14.0f /launch %12.1f /wave" % (c, per, per / waves if waves else 0))
