"""A/B timing of library variants (tools/debug/build_variant.sh) at the headline size, one subprocess per variant.

    python tools/debug/variant_timing.py base onechain ...        (names under scaling_rgcn_training_amd/_build/variants/)

Per variant: median HIP-event time of the forward launch, the dX launch, the tile-major dW launch and the root-only dW
pass, plus a checksum of every result so that a variant that changes the numbers shows up next to its time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from scaling_rgcn_training_amd import _lib, plan as P
    import bench
    n, e, r = int(os.environ.get("VT_N", 10_000_000)), int(os.environ.get("VT_E", 100_000_000)), 32
    which = os.environ.get("VT_WHICH", "fwd,dx,dw").split(",")
    dev = torch.device("cuda:0")
    ei, et, x, dg, w, root = bench.synthetic_on_device(n, e, r, 64, 64, dev)
    if os.environ.get("VT_UNIQ") == "1":        # no two edges share (dst, relation): no run of equal destinations anywhere
        key = torch.randperm(n * r, device=dev)[:e]
        ei = torch.stack([ei[0], key // r])
        et = key % r
        del key
    tile, chunk = P.choose_layout(n, e, r, 64, 64)
    tile = int(os.environ.get("VT_TILE", tile))
    chunk = int(os.environ.get("VT_CHUNK", chunk))
    kflags = int(os.environ.get("VT_FLAGS", 0))
    plans = P.build_graph_plans_device(ei, et, n, r, tile, chunk=chunk, dw_tiles="dw" in which, split=int(os.environ.get("VT_SPLIT", "0")))
    del ei, et

    def t(fn, reps=int(os.environ.get("VT_REPS", 12))):
        fn(); fn(); torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return ts[len(ts) // 2], ts[0]

    def cs(v):
        return "%.9e" % v.double().abs().sum().item()

    bias = torch.zeros(64, device=dev)
    out = torch.empty(n, 64, device=dev)
    if "fwd" in which:
        pk = _lib.pack_weights(w, root, False)
        ps = _lib.plan_struct(plans.fwd)
        m, lo = t(lambda: _lib.fwd(ps, x, 64, pk, bias, out, 64, 0, kflags))
        print("  fwd        %.3f ms (min %.3f)  checksum %s" % (m, lo, cs(out)), flush=True)
    if "dx" in which:
        pkt = _lib.pack_weights(w, root, True)
        pst = _lib.plan_struct(plans.bwd)
        m, lo = t(lambda: _lib.bwd_dx(pst, dg, 64, pkt, out, 64, None, kflags))
        print("  dx         %.3f ms (min %.3f)  checksum %s" % (m, lo, cs(out)), flush=True)
    if "dw" in which:
        dw, dr, db = torch.empty_like(w), torch.empty_like(root), torch.empty(64, device=dev)
        psd, psf = _lib.plan_struct(plans.dw), _lib.plan_struct(plans.fwd)
        m, lo = t(lambda: _lib.bwd_dw_tiles(psd, plans.dw_walk, x, 64, dg, 64, dw, kflags))
        print("  dw tiles   %.3f ms (min %.3f)  checksum %s" % (m, lo, cs(dw)), flush=True)
        m, lo = t(lambda: _lib.bwd_dw(psf, x, 64, dg, 64, None, dr, db, _lib.FLAG_DW_ROOT_ONLY))
        print("  dw root    %.3f ms (min %.3f)  checksum %s %s" % (m, lo, cs(dr), cs(db)), flush=True)


if __name__ == "__main__":
    if os.environ.get("VT_CHILD"):
        child()
    else:
        for name in sys.argv[1:]:
            so = os.path.join(ROOT, "scaling_rgcn_training_amd", "_build", "variants", name + ".so")
            print("variant", name, flush=True)
            env = dict(os.environ, VT_CHILD="1", RGCN_LIB=so)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
