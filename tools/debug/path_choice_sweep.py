"""Is eplan.choose_path right?  For a grid of graph shapes: the layer step (forward + backward through the module, replayed from a
hipGraph where launch-bound) on the tile kernels, on the edge-parallel path, and what 'auto' picks.
    python tools/debug/path_choice_sweep.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # n, e, r, in, out, skew
    (20_000, 400_000, 8, 64, 64, False), (50_000, 500_000, 32, 64, 64, False), (200_000, 2_000_000, 64, 64, 64, False),
    (100_000, 1_000_000, 100, 32, 32, False), (1_000_000, 4_000_000, 200, 32, 32, False), (300_000, 3_000_000, 16, 16, 16, False),
    (500_000, 5_000_000, 32, 64, 64, True), (50_000, 2_000_000, 4, 64, 64, False), (2_000_000, 20_000_000, 90, 64, 16, False),
]


def child():
    sys.path.insert(0, ROOT)
    import torch
    import bench
    dev = torch.device("cuda:0")
    for n, e, r, din, dout, skew in SHAPES:
        ms, plan_s, msg, st = bench.gpu_rung(n, e, r, din, dout, dev, graph=e <= 4_000_000, skew=skew, steps=10, warmup=3)
        print(json.dumps({"shape": [n, e, r, din, dout, skew], "env": os.environ.get("RGCN_PATH"), "ms": round(msg if msg is not None else ms, 4),
                          "path": st["path"]}), flush=True)


if __name__ == "__main__":
    if os.environ.get("PS_CHILD"):
        child()
    else:
        res = {}
        for path in ("ring", "ep", "auto"):
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PS_CHILD="1", RGCN_PATH=path),
                                 capture_output=True, text=True).stdout
            for line in out.splitlines():
                if line.startswith("{"):
                    d = json.loads(line)
                    res.setdefault(tuple(d["shape"]), {})[path] = (d["ms"], d["path"])
        print("# n, e, R', in, out, skew | tile kernels ms | edge-parallel ms | auto ms (its choice)   [replayed from a hipGraph up to 4M edges]")
        for k, v in res.items():
            best = min(v["ring"][0], v["ep"][0])
            flag = "" if v["auto"][0] <= 1.15 * best else "   <-- auto is %.0f %% slower than the better path" % (100 * (v["auto"][0] / best - 1))
            print(k, "| %.4f | %.4f | %.4f (%s/%s)%s" % (v["ring"][0], v["ep"][0], v["auto"][0], v["auto"][1]["fwd"], v["auto"][1]["dx"], flag))
