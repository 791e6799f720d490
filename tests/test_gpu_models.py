"""SURVEY rows a5 / a7 / f-3 on the GPU against CPU twins (tests/twins.py): the MLP and attention heads in front of
the two R-GCN layers (reference model/layers.py:49-66, 90-112), the ``Trainer.train`` loop itself
(model/modelTrainer.py:41-74) and the summary -> original embedding transfer (model/embeddingTricks.py)."""
import types

import numpy as np
import pytest
import torch

from oracle import rgcn_oracle as O
from tests.twins import cpu_twin

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _graph(n, e, r, c, seed, n_train=500, n_val=200):
    from scaling_rgcn_training_amd.data import Data
    ei, et = O.synthetic_graph(n, e, r, seed=seed)
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(n, generator=g)
    y = torch.nn.functional.one_hot(torch.randint(0, c, (n,), generator=g), c).float()
    d = Data(edge_index=ei)
    d.edge_type = et
    d.x_train, d.x_val, d.x_test = perm[:n_train], perm[n_train:n_train + n_val], perm[n_train + n_val:n_train + 2 * n_val]
    d.y_train, d.y_val, d.y_test = y[d.x_train], y[d.x_val], y[d.x_test]
    return d


@pytest.mark.parametrize("kind", ["mlp", "att"])
def test_mlp_and_attention_models_match_cpu_twin(kind):
    """forward (eval mode: the attention dropout is off) and every parameter gradient of Emb_MLP_Layers /
    Emb_ATT_Layers on the GPU against the CPU twin with identical parameters"""
    from scaling_rgcn_training_amd.layers import Emb_ATT_Layers, Emb_MLP_Layers
    n, e, r, c, emb, hid, s = 1500, 12000, 7, 4, 21, 16, 3
    data = _graph(n, e, r, c, seed=11)
    torch.manual_seed(3)
    if kind == "mlp":
        model = Emb_MLP_Layers(r, hid, c, n, emb, s)
        model.load_embedding(torch.randn(n, s * emb), freeze=False)
    else:
        model = Emb_ATT_Layers(r, hid, c, n, emb, s)
        model.load_embedding(torch.randn(s, n, emb), freeze=False)
    twin = cpu_twin(model)
    model = model.to(DEV).eval()
    twin.eval()
    dd = data.to(DEV)
    out = model(dd, torch.sigmoid)
    ref = twin(data, torch.sigmoid)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    w = torch.randn(n, c, generator=torch.Generator().manual_seed(2))
    (out * w.to(DEV)).sum().backward()
    (ref * w).sum().backward()
    gp, tp = dict(model.named_parameters()), dict(twin.named_parameters())
    assert set(gp) == set(tp)
    for k in gp:
        a, b = gp[k].grad.cpu().numpy(), tp[k].grad.numpy()
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(b).max())), err_msg=k)


@pytest.mark.parametrize("loss_kind", ["bce", "ce"])
def test_trainer_train_loss_list_matches_cpu_twin(loss_kind):
    """``Trainer.train`` (eval forward + train forward + backward + Adam per epoch, model/modelTrainer.py:41-74) run as
    is on the GPU model and on its CPU twin: loss list, accuracy list and final parameters agree"""
    from scaling_rgcn_training_amd.layers import Emb_Layers
    from scaling_rgcn_training_amd.trainer import Trainer, bce_loss, ce_loss, do_nothing
    n, e, r, c, emb, hid = 2000, 16000, 9, 5, 63, 16
    data = _graph(n, e, r, c, seed=5)
    graph = types.SimpleNamespace(training_data=data)
    torch.manual_seed(1)
    model = Emb_Layers(r, hid, c, n, emb, None)
    twin = cpu_twin(model)
    loss_f, act = (bce_loss, torch.sigmoid) if loss_kind == "bce" else (ce_loss, do_nothing)
    tr = Trainer(None, hid, epochs=12, emb_dim=emb, lr=0.01, weight_d=5e-5, verbose=False)
    tr.device = DEV
    g_acc, g_loss, g_f1w, g_f1m = tr.train(model, graph, loss_f, act, sum_graph=False)
    tr.device = torch.device("cpu")
    c_acc, c_loss, c_f1w, c_f1m = tr.train(twin, graph, loss_f, act, sum_graph=False)
    assert len(g_loss) == 12 and len(g_acc) == 12
    np.testing.assert_allclose(g_loss, c_loss, rtol=2e-4, atol=2e-5)
    # accuracies are step functions of the outputs: allow one validation row to flip
    assert np.max(np.abs(np.array(g_acc) - np.array(c_acc))) <= 1.0 / 200 + 1e-12
    assert g_loss[-1] < g_loss[0]
    # (Adam divides by sqrt(v): where a gradient is ~0 its rounding noise moves a weight by up to lr per epoch -- a few of the
    # 126,000 embedding values; bounded, and everything else agrees)
    for k, v in model.state_dict().items():
        a, b = v.cpu().numpy(), twin.state_dict()[k].numpy()
        off = ~np.isclose(a, b, rtol=5e-3, atol=5e-4)
        assert off.mean() <= 1e-3 and np.abs(a - b).max() <= 0.012, (k, int(off.sum()), float(np.abs(a - b).max()))


def test_embedding_transfer_then_gpu_forward_matches_cpu_twin():
    """f-3: summary embeddings -> original-graph embedding through the index tensors of graphs.*_embeddings
    (reference model/embeddingTricks.py:8-49), loaded into the model, forward on the GPU == CPU twin"""
    from scaling_rgcn_training_amd import graphs as G
    from scaling_rgcn_training_amd.layers import Emb_Layers
    n, e, r, c, emb, hid = 900, 7000, 5, 3, 12, 16
    data = _graph(n, e, r, c, seed=8)
    rng = np.random.default_rng(0)
    org = G.Graph("org")
    org.num_nodes, org.nodes = n, [f"<n{i}>" for i in range(n)]
    org.node_to_enum = {s: i for i, s in enumerate(org.nodes)}
    sums = []
    for k, m in enumerate((40, 75)):
        sg = G.Graph(f"sum{k}")
        sg.num_nodes, sg.nodes = m, [f"<s{k}_{i}>" for i in range(m)]
        sg.node_to_enum = {s: i for i, s in enumerate(sg.nodes)}
        sg.embedding = torch.randn(m, emb, generator=torch.Generator().manual_seed(k))
        # every original node but a few maps to one summary node
        sg.orgNode2sumNode_dict = {org.nodes[i]: sg.nodes[int(rng.integers(m))] for i in range(n) if i % 97 != 5}
        sums.append(sg)
    torch.manual_seed(4)
    emb_sum = G.sum_embeddings(org, sums, emb)
    assert emb_sum.shape == (n, emb)
    mapped = [i for i in range(n) if i % 97 != 5]
    want = sum(sg.embedding[[sg.node_to_enum[sg.orgNode2sumNode_dict[org.nodes[i]]] for i in mapped]] for sg in sums)
    np.testing.assert_allclose(emb_sum[mapped].numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    model = Emb_Layers(r, hid, c, n, emb, None)
    model.load_embedding(emb_sum, freeze=True)
    twin = cpu_twin(model)
    out = model.to(DEV)(data.to(DEV), torch.sigmoid)
    np.testing.assert_allclose(out.detach().cpu().numpy(), twin(data, torch.sigmoid).detach().numpy(), rtol=1e-5, atol=1e-5)
