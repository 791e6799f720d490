"""Aggregate rocprofv3 --pmc csv output: mean counter value per kernel.
    pmc_summary.py DIR [DIR ...]                  text table (rgcn kernels)
    pmc_summary.py --traffic-json FETCH_DIR WRITE_DIR   HBM bytes per launch, JSON (profiles/*_pmc_traffic.json)
gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE under-reports 16-B/lane coalesced
reads by 2x, WRITE_SIZE is exact, both in KiB: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import glob
import json
import re
import sys


def collect(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


if sys.argv[1] == "--traffic-json":
    acc = collect(sys.argv[2:])
    out = {"command": "cd /tmp && rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, separate pass) --output-format csv -- "
                      "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline   (tools/profile_round.sh)",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of 16-B/lane "
                         "coalesced reads; WRITE_SIZE exact; unit KiB)",
           "kernels": {}}
    for k, cs in acc.items():
        if "rgcn" not in k:
            continue
        f, w = cs.get("FETCH_SIZE", []), cs.get("WRITE_SIZE", [])
        if not f or not w:
            continue
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        out["kernels"][k] = {"FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm, "launches": len(f),
                             "hbm_bytes_per_launch": (2 * fm + wm) * 1024}
    print(json.dumps(out, indent=1))
else:
    acc = collect(sys.argv[1:])
    for k, cs in acc.items():
        if "rgcn" not in k:
            continue
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} mean {sum(v) / len(v):18.1f}  (n={len(v)})")
