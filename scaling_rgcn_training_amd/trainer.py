"""Full-batch training loop with the reference's ``Trainer`` API
(/root/reference/model/modelTrainer.py:15-116) and the loss / activation / metric helpers of
/root/reference/model/evaluation.py, re-provided so the HIP layer can be dropped into the same
experiment flow.  SURVEY.md 8f-2: the step either side of the hot path.

Differences from the reference, all on the host side of the path:
* ``evaluate`` keeps predictions on the device and makes ONE host copy of the few labelled rows, then
  computes accuracy / F1 with numpy (the reference builds a CPU ``torch.zeros`` and hands device tensors
  to sklearn, which breaks on a GPU: evaluation.py:20,29);
* validation forwards run under ``torch.no_grad()`` by default (the reference's eval forward builds an
  autograd graph it never uses: modelTrainer.py:53-55); pass ``eval_no_grad=False`` for the exact behaviour;
* the per-epoch ``.item()`` host sync of the loss is kept (modelTrainer.py:68) because the returned loss
  list is part of the API.
The loop contract itself is the reference's: per epoch on the original graph 1 eval forward + 1 train
forward + 1 backward through the two RGCN layers, Adam(lr, weight_decay), one optimizer step per epoch.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Callable, Dict, List, Tuple, Union

import numpy as np
import torch
from torch import Tensor, nn

from .layers import Emb_Layers


# ---- model/evaluation.py equivalents -------------------------------------------------------------
def do_nothing(x: Tensor) -> Tensor:
    return x


def ce_loss(pred: Tensor, targets: Tensor) -> Tensor:
    return nn.functional.cross_entropy(pred, targets.argmax(-1))


def bce_loss(pred: Tensor, targets: Tensor) -> Tensor:
    return nn.functional.binary_cross_entropy(pred, targets)


def get_losst(dataset: str, sumModel: bool = False) -> Tuple[Callable, Callable]:
    """BCE + sigmoid for summary models and AIFB (multi-label), CE(argmax) + identity otherwise
    (reference evaluation.py:44-48)."""
    if sumModel or dataset == "AIFB":
        return bce_loss, torch.sigmoid
    return ce_loss, do_nothing


def _f1(y_true: np.ndarray, y_pred: np.ndarray, average: str) -> float:
    """F1 over the label columns of multilabel-indicator arrays with zero_division=0, 'weighted' by
    support or 'macro' (what sklearn.f1_score computes for the reference's call)."""
    tp = np.logical_and(y_true == 1, y_pred == 1).sum(0).astype(np.float64)
    fp = np.logical_and(y_true == 0, y_pred == 1).sum(0).astype(np.float64)
    fn = np.logical_and(y_true == 1, y_pred == 0).sum(0).astype(np.float64)
    denom = 2 * tp + fp + fn
    f1 = np.where(denom > 0, 2 * tp / np.maximum(denom, 1), 0.0)
    if average == "macro":
        return float(f1.mean()) if f1.size else 0.0
    support = (y_true == 1).sum(0).astype(np.float64)
    return float((f1 * support).sum() / support.sum()) if support.sum() > 0 else 0.0


def evaluate(model: nn.Module, activation: Callable, training_data, x: Tensor, y: Tensor,
             report: bool = False, no_grad: bool = True) -> Tuple[float, float, float]:
    """accuracy (exact-match over label rows), weighted F1, macro F1 on rows ``x`` against ``y``."""
    ctx = torch.no_grad() if no_grad else torch.enable_grad()
    with ctx:
        pred = model(training_data, activation)
        rows = pred[x.to(pred.device)]
        if activation is not torch.sigmoid:
            hard = torch.zeros_like(rows).scatter_(1, rows.argmax(1, keepdim=True), 1.0)
        else:
            hard = torch.round(rows)
    y_pred = hard.detach().to("cpu").numpy().astype(np.int64)
    y_true = np.asarray(y.detach().to("cpu").numpy()).astype(np.int64)
    acc = float((y_pred == y_true).all(axis=1).mean()) if len(y_true) else 0.0
    f1_w, f1_m = _f1(y_true, y_pred, "weighted"), _f1(y_true, y_pred, "macro")
    if report:
        print(f"test rows {len(y_true)}: accuracy {acc:.4f}  f1 weighted {f1_w:.4f}  f1 macro {f1_m:.4f}")
    return acc, f1_w, f1_m


# ---- model/modelTrainer.py equivalent ---------------------------------------------------------------
class Trainer:
    """``Trainer(data, hidden_l, epochs, emb_dim, lr, weight_d)`` with ``train_summaries``,
    ``train_original``, ``train`` and ``transfer_weights`` as in the reference.  ``data`` is any object
    with the reference ``Dataset`` attributes used here: ``sumGraphs``, ``orgGraph`` (each with
    ``relations``, ``num_nodes``, ``training_data``, ``embedding``) and ``num_classes``."""

    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")

    def __init__(self, data, hidden_l: int, epochs: int, emb_dim: int, lr: float, weight_d: float,
                 eval_no_grad: bool = True, verbose: bool = True):
        self.data = data
        self.hidden_l, self.epochs, self.emb_dim, self.lr, self.weight_d = hidden_l, epochs, emb_dim, lr, weight_d
        self.sumModel: nn.Module = None
        self.eval_no_grad = eval_no_grad
        self.verbose = verbose

    def transfer_weights(self, orgModel: nn.Module, grad: bool) -> None:
        s = self.sumModel
        orgModel.override_params(s.rgcn1.weight.clone(), s.rgcn1.bias.clone(), s.rgcn1.root.clone(),
                                 s.rgcn2.weight.clone(), s.rgcn2.bias.clone(), s.rgcn2.root.clone(), grad)
        if self.verbose:
            print("weight transfer done")

    def _device_data(self, graph):
        """``graph.training_data`` on the training device, moved ONCE per (graph, device): the layer's graph plans are
        cached on the identity of the edge tensors, so a fresh copy per call (what ``Data.to`` returns) would rebuild
        them for the final test evaluation and keep the old copies alive."""
        cached = getattr(graph, "_device_data", None)
        if cached is None or cached[0] != self.device or cached[1] is not graph.training_data:
            cached = (self.device, graph.training_data, graph.training_data.to(self.device))
            try:
                graph._device_data = cached
            except AttributeError:
                pass
        return cached[2]

    def train(self, model: nn.Module, graph, loss_f: Callable, activation: Callable,
              sum_graph: bool = True) -> Tuple[List[float], List[float], List[float], List[float]]:
        model = model.to(self.device)
        training_data = self._device_data(graph)
        optimizer = torch.optim.Adam(model.parameters(), lr=self.lr, weight_decay=self.weight_d)
        accuracies, losses, f1_ws, f1_ms = [], [], [], []
        targets = training_data.y_train.to(torch.float32)
        for epoch in range(self.epochs):
            if not sum_graph:
                model.eval()
                acc, f1_w, f1_m = evaluate(model, activation, training_data, training_data.x_val,
                                           training_data.y_val, no_grad=self.eval_no_grad)
                if self.verbose:
                    print(f"Accuracy on validation set = {acc}")
                accuracies.append(acc)
                f1_ws.append(f1_w)
                f1_ms.append(f1_m)
            model.train()
            optimizer.zero_grad()
            out = model(training_data, activation)
            output = loss_f(out[training_data.x_train], targets)
            output.backward()
            optimizer.step()
            loss_value = output.item()
            losses.append(loss_value)
            if self.verbose and epoch % 10 == 0:
                print(f"Epoch: {epoch}, Loss: {loss_value:.4f}")
        return accuracies, losses, f1_ws, f1_ms

    def train_summaries(self, configs: Dict[str, Union[bool, str, int, float]]) -> None:
        loss_f, activation = get_losst(configs["dataset"], sumModel=True)
        first = self.data.sumGraphs[0]
        self.sumModel = Emb_Layers(2 * len(first.relations.keys()) + 1, self.hidden_l, self.data.num_classes,
                                   first.num_nodes, self.emb_dim, len(self.data.sumGraphs))
        for sumGraph in self.data.sumGraphs:
            self.sumModel.reset_embedding(sumGraph.num_nodes, self.emb_dim)
            self.train(self.sumModel, sumGraph, loss_f, activation, sum_graph=True)
            sumGraph.embedding = self.sumModel.embedding.weight.clone()

    def train_original(self, org_layers, embedding_trick: Callable, configs: Dict[str, Union[bool, str, int, float]],
                       exp: str):
        acc, loss, f1_w, f1_m = defaultdict(list), defaultdict(list), defaultdict(list), defaultdict(list)
        org = self.data.orgGraph
        orgModel = org_layers(2 * len(org.relations.keys()) + 1, self.hidden_l, self.data.num_classes, org.num_nodes,
                              self.emb_dim, configs["num_sums"])
        if exp != "baseline" and configs["e_trans"]:
            embedding = embedding_trick(org, self.data.sumGraphs, self.emb_dim)
            orgModel.load_embedding(embedding, freeze=configs["e_freeze"])
        if exp != "baseline" and configs["w_trans"]:
            self.transfer_weights(orgModel, configs["w_grad"])
        loss_f, activation = get_losst(configs["dataset"], sumModel=False)
        acc["accuracy"], loss["loss"], f1_w["f1 weighted"], f1_m["f1 macro"] = self.train(
            orgModel, org, loss_f, activation, sum_graph=False)
        td = self._device_data(org)
        test_acc, test_f1_w, test_f1_m = evaluate(orgModel, activation, td, td.x_test, td.y_test, report=self.verbose)
        return acc, loss, f1_w, f1_m, test_acc, test_f1_w, test_f1_m, orgModel
