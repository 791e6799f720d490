"""Parity criterion shared by tests/ and __graft_entry__.smoke().  TEST INFRASTRUCTURE ONLY.

Two bounds, both enforced wherever the fp32 CPU loop can run (``cpu32`` given):

(1)  |actual - ref| <= 1e-5 + 1e-5 |ref| + 4 u cond        (u = 2^-24, fp32 unit roundoff), elementwise.
     1e-5 (atol = rtol) is the tolerance BASELINE.json states.  ``cond`` is the same sum evaluated on absolute
     values (sum_e |t_e|): the forward error of ANY fp32 summation of n terms is bounded by about n u sum|t|, so an
     element that sums ~10^4 O(1) terms with cancellation (AIFB summary hubs: in-degree up to 11,825; d_weight:
     thousands of edges per relation) cannot meet a flat 1e-5 against float64 -- the reference's own fp32
     index_add / mm does not either.  4 u cond stays below 1e-5 for every ordinary node (cond < 40).

(2)  max_i ( |actual_i - ref_i| - 1e-5 (1 + |ref_i|) )  <=  2 max_i |cpu32_i - ref_i|      per tensor.
     BASELINE.json asks for results "within 1e-5 of the reference CPU path"; that path is fp32 (PyG's loop over
     ATen kernels, restated by ``rgcn_oracle.rgcn_conv_loop``), so ITS error against float64 on the same input is
     the only legitimate slack over the flat 1e-5: a kernel ten times less accurate than ATen fails (2) even where
     the a-priori bound (1) would let it through.  (Round 2 / 3 allowed the tile-major d_weight kernel and the streaming
     d_root / d_bias kernel ``cpu_factor`` = 4, then 2.5: a wave added ALL rows of its range into one accumulator, and the
     bf16 x 3 form's MFMA truncated every product under it -- a bias.  Round 4 folds the accumulator into the wave's slab
     every 128 units with alternating signs (csrc/rgcn_dw_tile.hip): the exception is gone, 2 x holds everywhere.)

Every call records how much of the slack over flat 1e-5 was used (``SLACK_LOG``); tests/conftest.py prints the
worst cases in the terminal summary.
"""
import numpy as np

from . import rgcn_oracle as O

U32 = 2.0 ** -24
SLACK_LOG = []   # (what, worst excess over flat 1e-5/1e-5 [<= 0: flat criterion met], cpu32 worst error or None)


def assert_close(actual, ref, cond=None, what="", cpu32=None, cpu_factor=2.0):
    actual = np.asarray(actual, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    flat = 1e-5 + 1e-5 * np.abs(ref)
    tol = flat
    if cond is not None:
        tol = tol + 4 * U32 * np.asarray(cond, dtype=np.float64)
    err = np.abs(actual - ref)
    bad = ~(err <= tol)  # also catches NaN
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.size} outside tolerance, worst excess "
                           f"{float(np.nanmax(err - tol)):.3e}, max err {float(np.nanmax(err)):.3e}")
    excess = float(np.max(err - flat)) if err.size else 0.0
    cpu_err = None
    if cpu32 is not None:
        cpu_err = float(np.max(np.abs(np.asarray(cpu32, dtype=np.float64) - ref))) if err.size else 0.0
        assert excess <= cpu_factor * cpu_err, (f"{what}: error exceeds flat 1e-5 by {excess:.3e}, more than {cpu_factor:g} x the fp32 "
                                                f"CPU loop's own worst error {cpu_err:.3e} on this tensor")
    SLACK_LOG.append((what, excess, cpu_err))


def abs_condition(x, ei, et, w_full, root, bias, dg, aggr="mean"):
    """(out_cond, grads_cond): the layer and its gradients evaluated on absolute values."""
    a = lambda t: None if t is None else np.abs(np.asarray(t, dtype=np.float64))
    return O.rgcn_conv_segments(a(x), np.asarray(ei), np.asarray(et), a(w_full), a(root), a(bias), a(dg), aggr=aggr)


def cpu32_reference(x, ei, et, w_full, root, bias, dg, aggr="mean"):
    """The reference-style CPU path in fp32: PyG's per-relation loop (``rgcn_conv_loop``) under autograd on float32
    tensors, dense [R, in, out] weights.  Returns (out, {'x','weight','root','bias'}) as numpy float32."""
    import torch
    t = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=torch.float32).clone().requires_grad_(True)
    xt, wt, rt, bt = t(x), t(w_full), t(root), t(bias)
    ei_t, et_t = torch.as_tensor(np.asarray(ei)).long(), torch.as_tensor(np.asarray(et)).long()
    out = O.rgcn_conv_loop(xt, ei_t, et_t, wt, rt, bt, aggr=aggr)
    out.backward(torch.as_tensor(np.asarray(dg), dtype=torch.float32))
    g = lambda a: None if a is None else (a.grad.numpy() if a.grad is not None else np.zeros(tuple(a.shape), np.float32))
    return out.detach().numpy(), {"x": g(xt), "weight": g(wt), "root": g(rt), "bias": g(bt)}
