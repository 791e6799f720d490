"""Host-side mirror of the reference interface: Data bag, metric helpers, parameter contracts,
distributed range arithmetic.  CPU only."""
import numpy as np
import torch

from scaling_rgcn_training_amd import dist as rdist
from scaling_rgcn_training_amd.conv import RGCNConv
from scaling_rgcn_training_amd.data import Data
from scaling_rgcn_training_amd.layers import Emb_ATT_Layers, Emb_Layers, Emb_MLP_Layers
from scaling_rgcn_training_amd import trainer as T


def test_data_bag_attribute_assignment_and_to():
    d = Data(edge_index=torch.zeros(2, 3, dtype=torch.long))
    d.edge_type = torch.zeros(3, dtype=torch.long)
    d.note = "x"
    e = d.to("cpu")
    assert e is not d and torch.equal(e.edge_index, d.edge_index) and e.note == "x"
    assert set(e.keys()) == {"edge_index", "edge_type", "note"}


def test_rgcnconv_parameter_shapes_and_registration_order():
    c = RGCNConv(63, 16, 89)
    assert [n for n, _ in c.named_parameters()] == ["weight", "root", "bias"]
    assert c.weight.shape == (89, 63, 16) and c.root.shape == (63, 16) and c.bias.shape == (16,) and c.comp is None
    assert torch.all(c.bias == 0) and c.weight.abs().max() <= (6 / (63 + 16)) ** 0.5
    b = RGCNConv(32, 32, 45, num_bases=30)
    assert [n for n, _ in b.named_parameters()] == ["weight", "comp", "root", "bias"]
    assert b.weight.shape == (30, 32, 32) and b.comp.shape == (45, 30)
    assert b.effective_weight().shape == (45, 32, 32)
    k = RGCNConv(12, 8, 5, num_blocks=4, root_weight=False, bias=False)
    assert k.weight.shape == (5, 4, 3, 2) and k.root is None and k.bias is None
    w = k.effective_weight()
    assert w.shape == (5, 12, 8) and torch.equal(w[:, 0:3, 0:2], k.weight[:, 0]) and torch.all(w[:, 0:3, 2:] == 0)
    import pytest
    with pytest.raises(ValueError):
        RGCNConv(8, 8, 3, num_bases=2, num_blocks=2)
    with pytest.raises(ValueError):
        RGCNConv(200, 8, 3)


def test_model_wrappers_state_dict_keys_and_override_contract():
    m = Emb_Layers(5, 16, 4, 20, 63, None)
    assert list(m.state_dict().keys()) == ["embedding.weight", "rgcn1.weight", "rgcn1.root", "rgcn1.bias",
                                           "rgcn2.weight", "rgcn2.root", "rgcn2.bias"]
    # kaiming_uniform_(fan_in) on a [R,in,out] tensor: bound sqrt(6 / (in*out))  
    assert m.rgcn1.weight.abs().max() <= (6 / (63 * 16)) ** 0.5 + 1e-7
    old = m.rgcn1.weight
    m.override_params(torch.ones(5, 63, 16), torch.ones(16), torch.ones(63, 16),
                      torch.ones(5, 16, 4), torch.ones(4), torch.ones(16, 4), grad=False)
    assert m.rgcn1.weight is not old and not m.rgcn1.weight.requires_grad and not m.rgcn2.root.requires_grad
    m.reset_embedding(7, 63)
    assert m.embedding.weight.shape == (7, 63)
    m.load_embedding(torch.zeros(9, 63), freeze=True)
    assert not m.embedding.weight.requires_grad
    a = Emb_ATT_Layers(5, 16, 4, None, 63, 3)
    a.load_embedding(torch.zeros(3, 9, 63), freeze=False)
    assert a.embedding.requires_grad and a.att.num_heads == 3
    p = Emb_MLP_Layers(5, 16, 4, 9, 21, 3)
    assert p.lin1.in_features == 63 and p.lin1.out_features == round(63 * 2 / 3 + 4) and p.lin2.out_features == 21


def test_metrics_match_definitions():
    y = np.array([[1, 0, 0], [0, 1, 0], [0, 1, 0], [0, 0, 1]])
    p = np.array([[1, 0, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1]])
    # per-class F1: c0 2*1/(2+1+0)=2/3, c1 2*1/(2+0+1)=2/3, c2 1
    assert abs(T._f1(y, p, "macro") - (2 / 3 + 2 / 3 + 1) / 3) < 1e-12
    assert abs(T._f1(y, p, "weighted") - (1 * 2 / 3 + 2 * 2 / 3 + 1 * 1) / 4) < 1e-12
    lf, act = T.get_losst("AIFB")
    assert lf is T.bce_loss and act is torch.sigmoid
    lf, act = T.get_losst("MUTAG")
    assert lf is T.ce_loss and act is T.do_nothing
    lf, act = T.get_losst("MUTAG", sumModel=True)
    assert lf is T.bce_loss


def test_rank_blocks_tile_aligned_and_cover():
    from scaling_rgcn_training_amd.conv import DistContext
    for n, tile, world, pieces in ((1000, 256, 2, 1), (10_000_000, 384, 8, 4), (300, 128, 4, 1), (5, 256, 2, 1),
                                   (100_000, 64, 2, 4)):
        pr = rdist.piece_rows(n, tile, world, pieces)
        assert pr % tile == 0 and pr * world * pieces >= n
        seen = np.zeros(n, dtype=np.int32)
        for r in range(world):
            c = DistContext(None, r, world, pr, pieces)
            for s_ in range(pieces):
                b, e = c.node_range(s_, n)
                assert b % tile == 0 or b == n
                seen[b:e] += 1
        assert np.all(seen == 1)                     # every node owned by exactly one (rank, piece)


def test_choose_layout_respects_lds_and_amortises_chunks():
    """plan.choose_layout is what the product uses (conv.layout_for): both directions of the layer must fit the
    160 KiB LDS with at least two ring slots, and the headline graph gets 128-slot chunks on 352-node tiles."""
    from scaling_rgcn_training_amd.plan import ACC_PAD, CHUNKS, LDS_BYTES, choose_layout, padded_width
    for args in ((10_000_000, 100_000_000, 32, 64, 64), (8243, 49838, 89, 63, 16), (1000, 9000, 5, 128, 128),
                 (100, 0, 3, 8, 8), (5000, 10 ** 6, 2, 64, 128), (23644, 150000, 45, 63, 16), (1_500_000, 6_000_000, 267, 32, 32)):
        t, c = choose_layout(*args)
        kp, np_ = padded_width(args[3]), padded_width(args[4])
        assert t % 16 == 0 and t >= 64 and c in CHUNKS
        for acc_w, ring_w in ((np_, kp), (kp, np_)):        # forward, dX
            assert (t + 1) * (acc_w + ACC_PAD) * 4 + 2 * c * (ring_w + 2) * 4 <= LDS_BYTES
    assert choose_layout(10_000_000, 100_000_000, 32, 64, 64) == (352, 128)


def test_trainer_device_data_caches_edges_only():
    """``Trainer._device_data`` keeps only the edge tensors on the device between calls (the plan cache keys on
    their identity); labels and split indices assigned after a first ``train`` (a new fold: graphs/dataset.py:30-35) must be
    what the next call sees."""
    from scaling_rgcn_training_amd.data import Data
    from scaling_rgcn_training_amd.trainer import Trainer

    class G:
        pass

    g = G()
    g.training_data = Data(edge_index=torch.zeros(2, 5, dtype=torch.long), edge_type=torch.zeros(5, dtype=torch.long))
    g.training_data.x_train = torch.tensor([0, 1])
    g.training_data.y_train = torch.tensor([[1.0], [0.0]])
    tr = Trainer(None, 4, 1, 4, 0.01, 0.0, verbose=False)
    tr.device = torch.device("cpu")
    a = tr._device_data(g)
    g.training_data.x_train = torch.tensor([2, 3])              # re-split on the same graph
    g.training_data.y_train = torch.tensor([[0.0], [1.0]])
    b = tr._device_data(g)
    assert b.edge_index is a.edge_index and b.edge_type is a.edge_type        # same device edge tensors: plans stay cached
    assert torch.equal(b.x_train, torch.tensor([2, 3])) and torch.equal(b.y_train, torch.tensor([[0.0], [1.0]]))
    g.training_data.edge_type = torch.ones(5, dtype=torch.long)               # a different graph tensor: moved again
    c = tr._device_data(g)
    assert torch.equal(c.edge_type, torch.ones(5, dtype=torch.long))
