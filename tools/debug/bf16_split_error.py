#!/usr/bin/env python3
"""Numerics of a split-precision (bf16 x k) contraction against the 1e-5 criterion, on the headline layer's
distribution (x ~ N(0,1), W ~ U(+-sqrt(6/4096)), root ~ U(+-sqrt(6/128)), ten in-edges per node), CPU / numpy:
x = h + m (+ l) with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); products of two bf16 values are exact in the
fp32 MFMA accumulator, so only the split and the dropped cross terms cost accuracy.  DESIGN.md 4.4 quotes the table."""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def split(x, parts):
    out, r = [], x.astype(np.float32)
    for _ in range(parts):
        h = bf16(r)
        out.append(h)
        r = (r - h).astype(np.float32)
    return out


n, deg = 20000, 10
W = rng.uniform(-np.sqrt(6 / 4096), np.sqrt(6 / 4096), (64, 64)).astype(np.float32)
R = rng.uniform(-np.sqrt(6 / 128), np.sqrt(6 / 128), (64, 64)).astype(np.float32)
X = rng.standard_normal((n * deg, 64)).astype(np.float32)
X0 = rng.standard_normal((n, 64)).astype(np.float32)
ref = (X.astype(np.float64) @ W.astype(np.float64)).reshape(n, deg, 64).sum(1) + X0.astype(np.float64) @ R.astype(np.float64)


def run(parts, terms):
    xs, x0s, ws, rs = split(X, parts), split(X0, parts), split(W, parts), split(R, parts)
    acc, acc0 = np.zeros((n * deg, 64), np.float32), np.zeros((n, 64), np.float32)
    for i, j in terms:
        acc += xs[i] @ ws[j]
        acc0 += x0s[i] @ rs[j]
    return acc.reshape(n, deg, 64).sum(1, dtype=np.float32) + acc0


def report(name, got):
    err = np.abs(got.astype(np.float64) - ref)
    flat = 1e-5 + 1e-5 * np.abs(ref)
    print(f"{name:36s} max err {err.max():.2e}  rms {np.sqrt((err ** 2).mean()):.2e}  worst excess over 1e-5 (1 + |ref|) "
          f"{np.max(err - flat):+.2e}  elements outside {np.mean(err > flat) * 100:.3f} %")


report("exact fp32 (what the kernels run)", (X @ W).reshape(n, deg, 64).sum(1, dtype=np.float32) + X0 @ R)
report("plain bf16, 1 product", run(1, [(0, 0)]))
report("bf16 x2 split, 3 products", run(2, [(0, 0), (0, 1), (1, 0)]))
report("bf16 x2 split, 4 products", run(2, [(0, 0), (0, 1), (1, 0), (1, 1)]))
report("bf16 x3 split, 6 products", run(3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]))
report("bf16 x3 split, 9 products", run(3, [(i, j) for i in range(3) for j in range(3)]))
