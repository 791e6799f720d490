"""Aggregate rocprofv3 --pmc csv output: mean counter value per kernel.  usage: pmc_summary.py DIR [DIR ...]"""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "rgcn" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
