"""BASELINE.json config 5 at AIFB scale: attribute-summary pre-training -> embedding + weight transfer -> attention model
on the full graph (reference main.py:92-95 `-exp attention`, model/modelTrainer.py:76-116, model/layers.py:49-66), through
the HIP layer on the GPU against the SAME flow run on the CPU through the oracle.

Data: the three attribute summaries and node maps the reference ships for AIFB (graphs/AIFB/attr/{sum,map}: 44 / 418 / 359
summary nodes, 49,838 edges each) and an original graph of the real size (8,243 nodes, 49,838 edges, 89 relation ids)
re-sampled so that its attribute summaries are exactly those files -- the reference does not ship AIFB_complete.nt
(tests/golden/config5/make_aifb_attr.py has the construction; class labels are synthetic).  parity unpinned, as everywhere: the
CPU side is the oracle's restatement of PyG's loop, not PyG."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
EPOCHS, EMB, HID = 10, 63, 16
TWIN_EPOCHS = 2      # the CPU twin walks 89 relations x 4 graphs per epoch through the oracle loop: ~12 s per epoch on the GPU box's host share


def _dataset(tmp_path):
    from scaling_rgcn_training_amd import graphs as G
    from tests.aifb_attr import write_dataset
    org, sums, maps = write_dataset(str(tmp_path))
    data = G.Dataset(org, sums, maps)
    data.init_dataset()
    return data


def _flow(trainer_cls, data, dropout, epochs=EPOCHS):
    """summary pre-training on the three summaries, then the attention experiment on the original graph"""
    from scaling_rgcn_training_amd import graphs as G
    from scaling_rgcn_training_amd.layers import Emb_ATT_Layers
    cfg = dict(dataset="AIFB", e_trans=True, e_freeze=False, w_trans=True, w_grad=True, num_sums=3)
    torch.manual_seed(0)
    tr = trainer_cls(data, hidden_l=HID, epochs=epochs, emb_dim=EMB, lr=0.01, weight_d=5e-5, verbose=False)
    sum_losses = []
    real_train = tr.train

    def recording_train(model, graph, loss_f, activation, sum_graph=True):
        out = real_train(model, graph, loss_f, activation, sum_graph)
        if sum_graph:
            sum_losses.append(out[1])
        return out

    tr.train = recording_train
    tr.train_summaries(cfg)

    def att_layers(*a):
        m = Emb_ATT_Layers(*a)
        m.att.dropout = dropout
        return m

    acc, loss, f1w, f1m, tacc, tf1w, tf1m, model = tr.train_original(att_layers, G.stack_embeddings, cfg, "attention")
    return dict(sum_losses=sum_losses, sum_emb=[g.embedding.detach().cpu().numpy() for g in data.sumGraphs],
                loss=loss["loss"], acc=acc["accuracy"], f1w=f1w["f1 weighted"], test=(tacc, tf1w, tf1m), model=model, trainer=tr)


def test_config5_shapes(tmp_path):
    data = _dataset(tmp_path)
    org = data.orgGraph
    assert org.num_nodes == 8243 and org.training_data.edge_index.shape == (2, 49838) and 2 * len(org.relations) + 1 == 89
    assert [g.num_nodes for g in data.sumGraphs] == [44, 418, 359]
    assert all(g.training_data.edge_index.shape == (2, 49838) for g in data.sumGraphs)
    assert int(org.training_data.edge_type.max()) == 87          # relation id 2R = 88 stays unused


def test_config5_attention_transfer_flow_matches_cpu_twin(tmp_path):
    from scaling_rgcn_training_amd.trainer import Trainer
    from tests.twins import make_cpu_twin_trainer
    data_g = _dataset(tmp_path)
    data_c = copy.deepcopy(data_g)
    # dropout off in both runs: the two devices draw from different generators (the reference's 0.2 is exercised below)
    gpu = _flow(Trainer, data_g, dropout=0.0, epochs=TWIN_EPOCHS)
    cpu = _flow(make_cpu_twin_trainer(Trainer), data_c, dropout=0.0, epochs=TWIN_EPOCHS)
    assert gpu["trainer"].last_train_mode == "hipgraph" and cpu["trainer"].last_train_mode == "eager"
    # summary pre-training: one model trained on the three summaries in turn (hub graphs: in-degree up to 11,825)
    assert len(gpu["sum_losses"]) == 3
    for k, (a, b) in enumerate(zip(gpu["sum_losses"], cpu["sum_losses"])):
        np.testing.assert_allclose(a, b, rtol=5e-4, atol=5e-5, err_msg=f"summary graph {k} loss curve")
    for k, (a, b) in enumerate(zip(gpu["sum_emb"], cpu["sum_emb"])):
        np.testing.assert_allclose(a, b, rtol=5e-3, atol=5e-4, err_msg=f"summary graph {k} trained embedding")
    # the attention model on the original graph, initialised from the transferred embeddings and weights
    np.testing.assert_allclose(gpu["loss"], cpu["loss"], rtol=5e-4, atol=5e-5)
    assert len(gpu["loss"]) == TWIN_EPOCHS and gpu["loss"][-1] < gpu["loss"][0]
    np.testing.assert_allclose(gpu["f1w"], cpu["f1w"], atol=0.02)
    np.testing.assert_allclose(gpu["acc"], cpu["acc"], atol=0.02)
    np.testing.assert_allclose(gpu["test"], cpu["test"], atol=0.02)            # end-to-end accuracy parity (config 5)
    # final parameters: Adam divides by sqrt(v), so where a gradient is ~0 its rounding noise moves a weight by up to lr per
    # epoch -- a handful of the embedding's 1.6M values and a few entries of the small attention tensors; nothing drifts
    # further than the optimiser's steps can cover, and the large tensors agree to 1e-3 in 99.8 % of their entries
    for k, v in gpu["model"].state_dict().items():
        a, b = v.cpu().numpy(), cpu["model"].state_dict()[k].numpy()
        # (Adam's first steps move a weight by lr whatever the gradient's size: where a gradient is rounding noise its SIGN may
        # differ between the two devices -- 2 lr apart per epoch at worst; the share of such entries is what the next check bounds)
        assert np.abs(a - b).max() <= 2 * 0.01 * TWIN_EPOCHS, (k, float(np.abs(a - b).max()))
        if a.size >= 10000:
            off = ~np.isclose(a, b, rtol=1e-2, atol=1e-3)
            assert off.mean() <= 2e-3, (k, int(off.sum()))


def test_config5_with_the_reference_dropout_trains(tmp_path):
    """the configuration as the reference runs it (attention dropout 0.2 active during training, hipGraph replays draw fresh
    masks): finite, decreasing loss"""
    from scaling_rgcn_training_amd.trainer import Trainer
    r = _flow(Trainer, _dataset(tmp_path), dropout=0.2)
    assert np.isfinite(r["loss"]).all() and r["loss"][-1] < r["loss"][0]
    assert len(set(np.round(r["loss"], 7))) == EPOCHS       # no two epochs alike: the replayed dropout masks differ
