"""A/B of the layer paths on the ladder rungs of bench.py: RGCN_PATH=ring | ep | auto per subprocess.
    python tools/debug/rung_timing.py [rung substring ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import torch
    import bench
    dev = torch.device("cuda:0")
    want = sys.argv[1:]
    for name, n, e, r, din, dout, nb in bench.LADDER:
        if want and not any(w in name for w in want):
            continue
        ms, plan_s, msg, st = bench.gpu_rung(n, e, r, din, dout, dev, graph=e <= 8_000_000, num_bases=nb, skew="skew" in name,
                                             steps=10 if "skew" in name else 20)
        print(json.dumps({"rung": name, "path_env": os.environ.get("RGCN_PATH"), "ms": round(ms, 4),
                          "ms_hipgraph": None if msg is None else round(msg, 4), "plan_s": round(plan_s, 3), "plan": st}), flush=True)


if __name__ == "__main__":
    if os.environ.get("RT_CHILD"):
        child()
    else:
        for path in os.environ.get("RT_PATHS", "ring,ep").split(","):
            subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, RT_CHILD="1", RGCN_PATH=path), check=False)
