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


def _metrics(pred: Tensor, activation: Callable, x: Tensor, y: Tensor, report: bool = False) -> Tuple[float, float, float]:
    """accuracy / weighted F1 / macro F1 of the predictions ``pred`` (all nodes, on the device) on rows ``x`` against ``y``:
    hard labels on the device, ONE host copy of the labelled rows, numpy metrics."""
    rows = pred.detach()[x.to(pred.device)]
    if activation is not torch.sigmoid:
        hard = torch.zeros_like(rows).scatter_(1, rows.argmax(1, keepdim=True), 1.0)
    else:
        hard = torch.round(rows)
    y_pred = hard.to("cpu").numpy().astype(np.int64)
    y_true = np.asarray(y.detach().to("cpu").numpy()).astype(np.int64)
    acc = float((y_pred == y_true).all(axis=1).mean()) if len(y_true) else 0.0
    f1_w, f1_m = _f1(y_true, y_pred, "weighted"), _f1(y_true, y_pred, "macro")
    if report:
        print(f"test rows {len(y_true)}: accuracy {acc:.4f}  f1 weighted {f1_w:.4f}  f1 macro {f1_m:.4f}")
    return acc, f1_w, f1_m


def evaluate(model: nn.Module, activation: Callable, training_data, x: Tensor, y: Tensor,
             report: bool = False, no_grad: bool = True) -> Tuple[float, float, float]:
    """accuracy (exact-match over label rows), weighted F1, macro F1 on rows ``x`` against ``y``."""
    ctx = torch.no_grad() if no_grad else torch.enable_grad()
    with ctx:
        pred = model(training_data, activation)
    return _metrics(pred, activation, x, y, report)


# ---- model/modelTrainer.py equivalent ---------------------------------------------------------------
class _CaptureFailed(RuntimeError):
    pass


class Trainer:
    """``Trainer(data, hidden_l, epochs, emb_dim, lr, weight_d)`` with ``train_summaries``,
    ``train_original``, ``train`` and ``transfer_weights`` as in the reference.  ``data`` is any object
    with the reference ``Dataset`` attributes used here: ``sumGraphs``, ``orgGraph`` (each with
    ``relations``, ``num_nodes``, ``training_data``, ``embedding``) and ``num_classes``."""

    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")

    # graphs up to this many edges train from replayed hipGraphs when ``hipgraph="auto"``: there the epoch is bound by
    # the host submitting its ~60 launches, not by the GPU (AIFB-shaped layer: 0.47 ms eager, 0.20 ms replayed)
    HIPGRAPH_AUTO_MAX_EDGES = 2_000_000

    def __init__(self, data, hidden_l: int, epochs: int, emb_dim: int, lr: float, weight_d: float,
                 eval_no_grad: bool = True, verbose: bool = True, hipgraph="auto"):
        self.data = data
        self.hidden_l, self.epochs, self.emb_dim, self.lr, self.weight_d = hidden_l, epochs, emb_dim, lr, weight_d
        self.sumModel: nn.Module = None
        self.eval_no_grad = eval_no_grad
        self.verbose = verbose
        # True / False / "auto": capture the epoch (eval forward; zero_grad + forward + loss + backward + Adam step) into
        # two hipGraphs and replay them -- the layer path never synchronises or allocates through the library, so it is
        # capturable as it is.  ``last_train_mode`` records what the last ``train`` call did ("hipgraph" / "eager").
        self.hipgraph = hipgraph
        self.last_train_mode = None

    def transfer_weights(self, orgModel: nn.Module, grad: bool) -> None:
        s = self.sumModel
        orgModel.override_params(s.rgcn1.weight.clone(), s.rgcn1.bias.clone(), s.rgcn1.root.clone(),
                                 s.rgcn2.weight.clone(), s.rgcn2.bias.clone(), s.rgcn2.root.clone(), grad)
        if self.verbose:
            print("weight transfer done")

    def _device_data(self, graph):
        """``graph.training_data`` on the training device.  Only the EDGE tensors are cached on the graph (keyed on the
        identity and version of the host tensors): the layer's graph plans are cached on the identity of the device edge
        tensors, so a fresh copy per call (what ``Data.to`` returns) would rebuild them for every ``train`` and for the final
        test evaluation.  Everything else -- labels, split indices, whatever a caller assigned since the last call
        (graphs/dataset.py:30-35,53-54) -- is moved afresh on every call, as the reference's ``.to(device)`` does
        (model/modelTrainer.py:43)."""
        data = graph.training_data
        key = (self.device, id(data.edge_index), data.edge_index._version, id(data.edge_type), data.edge_type._version)
        cached = getattr(graph, "_device_edges", None)
        if cached is None or cached[0] != key:
            cached = (key, data.edge_index.to(self.device), data.edge_type.to(self.device), data.edge_index, data.edge_type)
            try:
                graph._device_edges = cached
            except AttributeError:
                pass
        out = type(data)()
        for k, v in data.__dict__.items():
            if k == "edge_index":
                v = cached[1]
            elif k == "edge_type":
                v = cached[2]
            elif torch.is_tensor(v):
                v = v.to(self.device)
            setattr(out, k, v)
        return out

    def _want_hipgraph(self, training_data, model: nn.Module = None) -> bool:
        if self.device.type != "cuda" or not self.eval_no_grad:
            return False
        # edge-partitioned layers issue asynchronous RCCL collectives and wait on their handles: not captured
        if model is not None and any(getattr(m, "dist", None) is not None for m in model.modules()):
            return False
        if self.hipgraph == "auto":
            return int(training_data.edge_type.shape[0]) <= self.HIPGRAPH_AUTO_MAX_EDGES
        return bool(self.hipgraph)

    def train(self, model: nn.Module, graph, loss_f: Callable, activation: Callable,
              sum_graph: bool = True) -> Tuple[List[float], List[float], List[float], List[float]]:
        model = model.to(self.device)
        training_data = self._device_data(graph)
        targets = training_data.y_train.to(torch.float32)
        if self._want_hipgraph(training_data, model):
            try:
                out = self._train_hipgraph(model, training_data, targets, loss_f, activation, sum_graph)
                self.last_train_mode = "hipgraph"
                return out
            except _CaptureFailed as err:       # nothing was trained yet: the state was restored before the capture
                if self.hipgraph is True:
                    raise
                if self.verbose:
                    print(f"hipGraph capture not possible ({err}); training eagerly")
        self.last_train_mode = "eager"
        optimizer = torch.optim.Adam(model.parameters(), lr=self.lr, weight_decay=self.weight_d)
        accuracies, losses, f1_ws, f1_ms = [], [], [], []
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

    def _train_hipgraph(self, model, training_data, targets, loss_f, activation, sum_graph):
        """The same epoch loop with its GPU work replayed from two hipGraphs: ``g_eval`` (validation forward in eval mode,
        no autograd) and ``g_train`` (zero_grad + forward + loss + backward + Adam step, ``capturable=True``: the step
        count lives on the device).  Capture needs the lazy state in place first -- graph plans, Adam moments, allocator
        pools -- so ONE eager epoch runs on a side stream before the capture and is then UNDONE (parameters restored,
        Adam state zeroed in place): the captured loop starts from exactly the state an eager ``train`` starts from.
        Per epoch the host does two replays, one host copy of the validation rows and the ``.item()`` of the loss
        (both part of the reference's API: modelTrainer.py:55,68)."""
        dev = self.device
        params = [q for q in model.parameters()]
        saved = [q.detach().clone() for q in params]
        # the warm-up epoch also advances what is not a parameter: module buffers and the device's RNG stream (the attention
        # model's dropout) -- both are put back, so that the captured loop starts where an eager ``train`` starts.  (Inside
        # the replayed graphs the dropout masks come from torch's graph-safe Philox offsets: same distribution as the eager
        # loop's, not the same stream -- loss curves of Emb_ATT_Layers with dropout agree statistically, not bit for bit.)
        buffers = [b for b in model.buffers()]
        saved_buffers = [b.detach().clone() for b in buffers]
        rng_state = torch.cuda.get_rng_state(dev)
        optimizer = torch.optim.Adam(params, lr=self.lr, weight_decay=self.weight_d, capturable=True)
        idx_train = training_data.x_train.to(dev)

        def train_step():
            optimizer.zero_grad(set_to_none=True)
            out = model(training_data, activation)
            loss = loss_f(out[idx_train], targets)
            loss.backward()
            optimizer.step()
            return loss

        def eval_forward():
            with torch.no_grad():
                return model(training_data, activation)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        try:
            with torch.cuda.stream(side):
                if not sum_graph:
                    model.eval()
                    eval_forward()
                model.train()
                train_step()
                # undo the warm-up epoch: parameters back, Adam moments and step count zeroed IN PLACE (their tensors are
                # what the captured optimizer step reads and writes)
                with torch.no_grad():
                    for q, q0 in zip(params, saved):
                        q.copy_(q0)
                    for b, b0 in zip(buffers, saved_buffers):
                        b.copy_(b0)
                    for st in optimizer.state.values():
                        for v in st.values():
                            if torch.is_tensor(v):
                                v.zero_()
                torch.cuda.set_rng_state(rng_state, dev)
                optimizer.zero_grad(set_to_none=True)
                g_eval = g_train = pred = loss = None
                if not sum_graph:
                    model.eval()
                    g_eval = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_eval, stream=side):
                        pred = eval_forward()
                model.train()
                g_train = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_train, stream=side):
                    loss = train_step()
        except RuntimeError as err:
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for q, q0 in zip(params, saved):
                    q.copy_(q0)
            for q in params:
                q.grad = None
            raise _CaptureFailed(str(err).splitlines()[0] if str(err) else type(err).__name__) from err
        torch.cuda.current_stream(dev).wait_stream(side)
        accuracies, losses, f1_ws, f1_ms = [], [], [], []
        for epoch in range(self.epochs):
            if not sum_graph:
                g_eval.replay()
                acc, f1_w, f1_m = _metrics(pred, activation, training_data.x_val, training_data.y_val)
                if self.verbose:
                    print(f"Accuracy on validation set = {acc}")
                accuracies.append(acc)
                f1_ws.append(f1_w)
                f1_ms.append(f1_m)
            g_train.replay()
            loss_value = loss.item()
            losses.append(loss_value)
            if self.verbose and epoch % 10 == 0:
                print(f"Epoch: {epoch}, Loss: {loss_value:.4f}")
        model.train()
        # the gradients live in the graph's private pool: hand the caller ordinary tensors
        for q in params:
            if q.grad is not None:
                q.grad = q.grad.detach().clone()
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
        # (as in the reference the model is still in train mode here: modelTrainer.py:57,113 -- only the attention model,
        # whose dropout stays active, can tell)
        test_acc, test_f1_w, test_f1_m = evaluate(orgModel, activation, td, td.x_test, td.y_test, report=self.verbose)
        return acc, loss, f1_w, f1_m, test_acc, test_f1_w, test_f1_m, orgModel
