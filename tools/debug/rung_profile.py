"""One ladder rung of bench.py under rocprofv3 --kernel-trace --stats:  rocprofv3 ... -- python3 tools/debug/rung_profile.py AM-like"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
for name, n, e, r, din, dout, nb in bench.LADDER:
    if any(w in name for w in sys.argv[1:]):
        ms, plan_s, msg, st = bench.gpu_rung(n, e, r, din, dout, dev, steps=10, warmup=3, num_bases=nb, skew="skew" in name)
        print(name, ms, st)
