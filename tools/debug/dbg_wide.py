import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import rgcn_oracle as O
from scaling_rgcn_training_amd import _lib, plan as P
from scaling_rgcn_training_amd.conv import tile_for, _rows16, _round4
dev = torch.device('cuda:0')
din, dout = int(sys.argv[1]), int(sys.argv[2])
n, e, r = 600, 4000, 3
ei, et = O.synthetic_graph(n, e, r, seed=1)
w, root, bias = O.synthetic_params(r, din, dout, seed=3)
g = torch.Generator().manual_seed(11)
x = torch.randn(n, din, generator=g)
ref, _ = O.rgcn_conv_segments(x.numpy(), ei.numpy(), et.numpy(), w.numpy(), root.numpy(), bias.numpy())
tile = tile_for(din, dout, n, e, r)
plans = P.build_graph_plans(ei.to(dev), et.to(dev), n, r, tile)
xd = _rows16(x.to(dev), din)
out = torch.full((n, _round4(dout)), float('nan'), device=dev)
_lib.fwd(_lib.plan_struct(plans.fwd), xd, din, _lib.pack_weights(w.to(dev), root.to(dev), False), bias.to(dev), out, dout)
torch.cuda.synchronize()
err = np.abs(out[:, :dout].cpu().numpy() - ref)
bad = err > 1e-4
print('tile', tile, 'n_tiles', plans.fwd.n_tiles, 'chunks', plans.fwd.n_chunks, 'bad', bad.sum(), 'of', bad.size)
rows = np.nonzero(bad.any(1))[0]
print('bad rows', rows[:40], 'count', len(rows))
print('bad cols hist', bad.sum(0))
# which chunk/relation do bad rows' edges belong to
p = plans.fwd
dl = p.slot_dstl.cpu().numpy().reshape(-1, 64); ct = p.chunk_tile.cpu().numpy(); cr = p.chunk_rel.cpu().numpy(); cc = p.chunk_cnt.cpu().numpy()
tp = p.tile_ptr.cpu().numpy()
for t in range(p.n_tiles):
    print('tile', t, 'chunks', tp[t], tp[t+1], 'rels', cr[tp[t]:tp[t+1]], 'cnt', cc[tp[t]:tp[t+1]])
for rr in rows[:6]:
    t, l = rr // tile, rr % tile
    hits = [(c, int((dl[c] == l).sum()), int(np.nonzero(dl[c] == l)[0][0])) for c in range(tp[t], tp[t+1]) if (dl[c][:cc[c]] == l).any()]
    print('row', rr, 'err', err[rr].max(), 'chunks (id, nrows, first slot):', hits)
# attribute errors to slots
src = p.slot_src.cpu().numpy().reshape(-1, 64); sw = p.slot_w.cpu().numpy().reshape(-1, 64)
W = np.concatenate([w.numpy(), root.numpy()[None]], 0).astype(np.float64)
o = out[:, :dout].cpu().numpy().astype(np.float64)
for rr in rows[:8]:
    t, l = rr // tile, rr % tile
    e = o[rr] - ref[rr]
    terms = []
    for c in range(tp[t], tp[t+1]):
        for s_ in range(cc[c]):
            if dl[c][s_] == l:
                terms.append((c, s_, sw[c][s_] * (x.numpy()[src[c][s_]].astype(np.float64) @ W[cr[c]])))
    desc = []
    for (c, s_, tv) in terms:
        k = float(e @ tv / (tv @ tv))
        desc.append((c, s_, round(k, 3)))
    res = e - sum(round(float(e @ tv / (tv @ tv))) * tv for (_, _, tv) in terms)
    print('row', rr, 'proj of error on each term (chunk, slot, coeff):', desc, 'residual', np.abs(res).max())
