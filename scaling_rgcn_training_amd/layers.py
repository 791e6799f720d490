"""The three model wrappers of the reference (/root/reference/model/layers.py), re-provided over the
HIP ``RGCNConv``: same constructor argument order, same attribute names (``embedding``, ``rgcn1``,
``rgcn2``, ``att`` / ``lin1`` / ``lin2``), same ``forward(training_data, activation)``,
``reset_embedding`` / ``load_embedding`` / ``override_params`` methods and the same state_dict keys
(``embedding.weight, rgcn1.{weight,root,bias}, rgcn2.{...}``; SURVEY.md 5 "Checkpoint").

Parameter initialisation follows the reference's RNG draw order:
``Embedding.normal_`` -> rgcn1 glorot(weight), glorot(root) -> rgcn2 glorot(weight), glorot(root) ->
kaiming_uniform_(rgcn1.weight, mode='fan_in') -> kaiming_uniform_(rgcn2.weight, mode='fan_in').
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .conv import RGCNConv
from .data import Data


class _RGCNStack(nn.Module):
    """x -> rgcn1 -> relu -> rgcn2 -> activation, the tail all three reference models share
    (model/layers.py:21-25, 62-66, 108-112), plus the parameter re-binding contract."""

    fuse_activations = True   # False: F.relu / activation as separate torch kernels, exactly as the reference writes it

    def _build_convs(self, emb_dim: int, hidden_l: int, num_labels: int, num_relations: int) -> None:
        self.rgcn1 = RGCNConv(in_channels=emb_dim, out_channels=hidden_l, num_relations=num_relations, num_bases=None)
        self.rgcn2 = RGCNConv(hidden_l, num_labels, num_relations, num_bases=None)

    def _kaiming_convs(self) -> None:
        nn.init.kaiming_uniform_(self.rgcn1.weight, mode="fan_in")
        nn.init.kaiming_uniform_(self.rgcn2.weight, mode="fan_in")

    def _tail(self, x: Tensor, training_data: Data, activation: Callable) -> Tensor:
        """``activation(rgcn2(relu(rgcn1(x))))`` with both activations fused into the layer kernels: the ReLU in
        rgcn1's store and -- because ``h`` is consumed by rgcn2 alone -- its backward as the mask on rgcn2's dX
        store; ``torch.sigmoid`` in rgcn2's store.  Any other ``activation`` callable runs as given."""
        ei, et = training_data.edge_index, training_data.edge_type
        if not self.fuse_activations:
            h = F.relu(self.rgcn1(x, ei, et))
            return activation(self.rgcn2(h, ei, et))
        h = self.rgcn1(x, ei, et, _activation="relu", _grad_premasked=True)
        if activation is torch.sigmoid:
            return self.rgcn2(h, ei, et, _activation="sigmoid", _input_relu=True)
        return activation(self.rgcn2(h, ei, et, _input_relu=True))

    def override_params(self, weight_1: Tensor, bias_1: Tensor, root_1: Tensor, weight_2: Tensor,
                        bias_2: Tensor, root_2: Tensor, grad: bool = True) -> None:
        """Re-bind both convs' parameters to fresh ``nn.Parameter`` objects (weight transfer from the
        summary model, model/modelTrainer.py:26-39); ``grad=False`` freezes them."""
        for conv, (w, b, r) in ((self.rgcn1, (weight_1, bias_1, root_1)), (self.rgcn2, (weight_2, bias_2, root_2))):
            conv.weight = nn.Parameter(w, requires_grad=grad)
            conv.bias = nn.Parameter(b, requires_grad=grad)
            conv.root = nn.Parameter(r, requires_grad=grad)


class Emb_Layers(_RGCNStack):
    """Trainable node embedding -> 2 x RGCN (reference model/layers.py:11-46)."""

    def __init__(self, num_relations: int, hidden_l: int, num_labels: int, num_nodes: int, emb_dim: int, _=None) -> None:
        super().__init__()
        self.embedding = nn.Embedding(num_nodes, emb_dim)
        self._build_convs(emb_dim, hidden_l, num_labels, num_relations)
        self._kaiming_convs()

    def forward(self, training_data: Data, activation: Callable) -> Tensor:
        return self._tail(self.embedding.weight, training_data, activation)

    def reset_embedding(self, num_nodes: int, emb_dim: int) -> None:
        self.embedding = nn.Embedding(num_nodes, emb_dim)

    def load_embedding(self, embedding: Tensor, freeze: bool = True) -> None:
        self.embedding = nn.Embedding.from_pretrained(embedding, freeze=freeze)


class Emb_ATT_Layers(_RGCNStack):
    """Stacked summary embeddings ``[S, N, emb]`` -> multi-head attention over the S axis (heads = S,
    dropout 0.2), first output slice -> 2 x RGCN (reference model/layers.py:49-87)."""

    def __init__(self, num_relations: int, hidden_l: int, num_labels: int, _, emb_dim: int, num_embs: int) -> None:
        super().__init__()
        self.embedding = None
        self.att = nn.MultiheadAttention(embed_dim=emb_dim, num_heads=num_embs, dropout=0.2)
        self._build_convs(emb_dim, hidden_l, num_labels, num_relations)
        self._kaiming_convs()

    def forward(self, training_data: Data, activation: Callable) -> Tensor:
        attn_output, _ = self.att(self.embedding, self.embedding, self.embedding, average_attn_weights=True)
        return self._tail(attn_output[0], training_data, activation)

    def load_embedding(self, embedding: Tensor, freeze: bool = True) -> None:
        self.embedding = nn.Parameter(embedding, requires_grad=not freeze)


class Emb_MLP_Layers(_RGCNStack):
    """Concatenated summary embeddings ``[N, S*emb]`` -> Linear -> tanh -> Linear -> 2 x RGCN
    (reference model/layers.py:90-130)."""

    def __init__(self, num_relations: int, hidden_l: int, num_labels: int, num_nodes: int, emb_dim: int, num_sums: int):
        super().__init__()
        in_f = num_sums * emb_dim
        out_f = round((in_f * (2 / 3)) + num_labels)
        self.embedding = nn.Embedding(num_nodes, emb_dim)
        self.lin1 = nn.Linear(in_features=in_f, out_features=out_f)
        self.lin2 = nn.Linear(in_features=out_f, out_features=emb_dim)
        self._build_convs(emb_dim, hidden_l, num_labels, num_relations)
        nn.init.kaiming_uniform_(self.lin1.weight, mode="fan_in")
        nn.init.kaiming_uniform_(self.lin2.weight, mode="fan_in")
        self._kaiming_convs()

    def forward(self, training_data: Data, activation: Callable, save=False) -> Tensor:
        x = self.lin2(torch.tanh(self.lin1(self.embedding.weight)))
        return self._tail(x, training_data, activation)

    def load_embedding(self, embedding: Tensor, freeze: bool = True) -> None:
        self.embedding = nn.Embedding.from_pretrained(embedding, freeze=freeze)
