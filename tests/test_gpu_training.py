"""End-to-end on the GPU: the reference's experiment flow (summary pre-training -> embedding + weight transfer ->
training on the original graph) through the HIP layer, and training-curve parity of the two-layer model against
a CPU twin built on the oracle with identical initial parameters."""
import os

import numpy as np
import pytest
import torch

from oracle import rgcn_oracle as O

pytestmark = pytest.mark.gpu


def test_reference_flow_on_test_dataset_runs_and_transfers():
    from scaling_rgcn_training_amd import graphs as G
    from scaling_rgcn_training_amd.layers import Emb_ATT_Layers, Emb_Layers, Emb_MLP_Layers
    from scaling_rgcn_training_amd.trainer import Trainer
    from tests.conftest import GOLDEN_DIR
    t = os.path.join(GOLDEN_DIR, "TEST")
    data = G.Dataset(os.path.join(t, "TEST_complete.nt"), os.path.join(t, "attr", "sum"), os.path.join(t, "attr", "map"))
    data.init_dataset()
    torch.manual_seed(0)
    cfg = dict(dataset="TEST", e_trans=True, e_freeze=False, w_trans=True, w_grad=True, num_sums=3, e_viz=False, sum="attr")
    tr = Trainer(data, hidden_l=16, epochs=6, emb_dim=63, lr=0.01, weight_d=5e-5, verbose=False)
    tr.train_summaries(cfg)
    assert all(sg.embedding is not None and sg.embedding.shape == (sg.num_nodes, 63) for sg in data.sumGraphs)
    for layers, trick, exp in ((Emb_Layers, G.sum_embeddings, "summation"), (Emb_MLP_Layers, G.concat_embeddings, "mlp"),
                               (Emb_ATT_Layers, G.stack_embeddings, "attention"), (Emb_Layers, None, "baseline")):
        acc, loss, f1w, f1m, tacc, tf1w, tf1m, model = tr.train_original(layers, trick, cfg, exp)
        assert len(loss["loss"]) == 6 and all(np.isfinite(loss["loss"])) and len(acc["accuracy"]) == 6
        assert 0.0 <= tacc <= 1.0
        if exp == "summation":   # weights were transferred from the summary model and then trained
            assert model.rgcn1.weight.shape == tr.sumModel.rgcn1.weight.shape


def test_two_layer_training_curve_matches_cpu_twin():
    from scaling_rgcn_training_amd.data import Data
    from scaling_rgcn_training_amd.layers import Emb_Layers
    dev = torch.device("cuda:0")
    n, e, r, emb, hid, c = 2500, 20000, 9, 63, 16, 5
    ei, et = O.synthetic_graph(n, e, r, seed=3)
    g = torch.Generator().manual_seed(1)
    y = torch.nn.functional.one_hot(torch.randint(0, c, (n,), generator=g), c).float()
    train_idx = torch.randperm(n, generator=g)[:600]
    torch.manual_seed(0)
    model = Emb_Layers(r, hid, c, n, emb, None)
    params0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    data = Data(edge_index=ei)
    data.edge_type = et

    def twin_forward(p):
        h = O.rgcn_conv_loop(p["embedding.weight"], ei, et, p["rgcn1.weight"], p["rgcn1.root"], p["rgcn1.bias"]).relu()
        return torch.sigmoid(O.rgcn_conv_loop(h, ei, et, p["rgcn2.weight"], p["rgcn2.root"], p["rgcn2.bias"]))

    # CPU twin: the same parameters trained through the oracle's loop form (float32, autograd)
    p = {k: v.clone().requires_grad_(True) for k, v in params0.items()}
    opt_c = torch.optim.Adam([p[k] for k, _ in model.named_parameters()], lr=0.01, weight_decay=5e-5)
    cpu_losses = []
    for _ in range(15):
        opt_c.zero_grad()
        loss = torch.nn.functional.binary_cross_entropy(twin_forward(p)[train_idx], y[train_idx])
        loss.backward()
        opt_c.step()
        cpu_losses.append(loss.item())
    # GPU: the drop-in model
    model = model.to(dev)
    dd = data.to(dev)
    opt_g = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-5)
    yd, td = y.to(dev), train_idx.to(dev)
    gpu_losses = []
    for _ in range(15):
        opt_g.zero_grad()
        loss = torch.nn.functional.binary_cross_entropy(model(dd, torch.sigmoid)[td], yd[td])
        loss.backward()
        opt_g.step()
        gpu_losses.append(loss.item())
    np.testing.assert_allclose(gpu_losses, cpu_losses, rtol=2e-4, atol=2e-5)
    assert gpu_losses[-1] < gpu_losses[0]
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), p[k].detach().numpy(), rtol=5e-3, atol=5e-4)


def test_trainer_hipgraph_epochs_match_eager_epochs():
    """``Trainer.train`` with the epoch replayed from hipGraphs (eval forward; zero_grad + forward + loss + backward + Adam
    step) against the same loop run eagerly from the same initial state: loss list, validation metrics and final
    parameters agree (Adam's device-side step count changes the bias corrections in the last fp32 bits only), the warm-up
    epoch the capture needs leaves no trace, and both follow the CPU twin's curve."""
    import copy
    from scaling_rgcn_training_amd.data import Data
    from scaling_rgcn_training_amd.layers import Emb_Layers
    from scaling_rgcn_training_amd.trainer import Trainer, bce_loss
    from tests.twins import cpu_twin
    n, e, r, emb, hid, c = 3000, 24000, 11, 63, 16, 4
    ei, et = O.synthetic_graph(n, e, r, seed=5)
    g = torch.Generator().manual_seed(2)
    y = torch.nn.functional.one_hot(torch.randint(0, c, (n,), generator=g), c).float()
    perm = torch.randperm(n, generator=g)
    data = Data(edge_index=ei)
    data.edge_type = et
    data.x_train, data.y_train = perm[:500], y[perm[:500]]
    data.x_val, data.y_val = perm[500:700], y[perm[500:700]]

    class _Graph:
        pass

    torch.manual_seed(0)
    model0 = Emb_Layers(r, hid, c, n, emb, None)
    runs = {}
    for mode in (False, True):
        gobj = _Graph()
        gobj.training_data = data
        tr = Trainer(None, hid, epochs=12, emb_dim=emb, lr=0.01, weight_d=5e-5, verbose=False, hipgraph=mode)
        model = copy.deepcopy(model0)
        acc, losses, f1w, f1m = tr.train(model, gobj, bce_loss, torch.sigmoid, sum_graph=False)
        assert tr.last_train_mode == ("hipgraph" if mode else "eager")
        runs[mode] = (acc, losses, f1w, {k: v.detach().cpu() for k, v in model.state_dict().items()})
        assert all(q.grad is not None for q in model.parameters())
    np.testing.assert_allclose(runs[True][1], runs[False][1], rtol=1e-5, atol=1e-6)
    assert runs[True][0] == runs[False][0] and len(runs[True][0]) == 12
    for k in runs[True][3]:
        np.testing.assert_allclose(runs[True][3][k].numpy(), runs[False][3][k].numpy(), rtol=1e-4, atol=1e-5)
    # the CPU twin (oracle convolutions, unfused tail) trained by the same loop
    twin = cpu_twin(model0)
    opt = torch.optim.Adam(twin.parameters(), lr=0.01, weight_decay=5e-5)
    cpu_losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = bce_loss(twin(data, torch.sigmoid)[data.x_train], data.y_train)
        loss.backward()
        opt.step()
        cpu_losses.append(loss.item())
    np.testing.assert_allclose(runs[True][1], cpu_losses, rtol=2e-4, atol=2e-5)
