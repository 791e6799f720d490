"""CPU twins of the model wrappers (TEST INFRASTRUCTURE): the same ``Emb_*_Layers`` module with its two
``RGCNConv`` layers replaced by ``OracleConv`` -- the oracle's restatement of PyG's per-relation loop
(oracle/rgcn_oracle.py, the ops the reference executes on CPU) under autograd -- and identical parameters."""
import copy

import torch
from torch import nn

from oracle import rgcn_oracle as O


class OracleConv(nn.Module):
    def __init__(self, conv):
        super().__init__()
        self.weight = nn.Parameter(conv.weight.detach().cpu().clone(), requires_grad=conv.weight.requires_grad)
        self.root = nn.Parameter(conv.root.detach().cpu().clone(), requires_grad=conv.root.requires_grad)
        self.bias = nn.Parameter(conv.bias.detach().cpu().clone(), requires_grad=conv.bias.requires_grad)

    def forward(self, x, edge_index, edge_type):
        return O.rgcn_conv_loop(x, edge_index, edge_type, self.weight, self.root, self.bias)


def cpu_twin(model):
    """deep copy on the CPU with oracle convolutions and the unfused tail (F.relu / activation as torch ops)"""
    twin = copy.deepcopy(model).cpu()
    twin.rgcn1, twin.rgcn2 = OracleConv(model.rgcn1), OracleConv(model.rgcn2)
    twin.fuse_activations = False
    return twin


def oracleize_(model):
    """IN PLACE: swap the model's two RGCNConv layers for OracleConv (same parameters) and unfuse the tail, so that the
    reference's flow -- which builds its models itself (Trainer.train_summaries / train_original) and re-binds their
    parameters (override_params) -- can run as a CPU twin."""
    if not isinstance(model.rgcn1, OracleConv):
        model.rgcn1, model.rgcn2 = OracleConv(model.rgcn1), OracleConv(model.rgcn2)
        model.fuse_activations = False
    return model


def make_cpu_twin_trainer(trainer_cls):
    """A Trainer that runs the same flow on the CPU through the oracle (every model it trains is oracleize_d first)."""

    class CpuTwinTrainer(trainer_cls):
        device = torch.device("cpu")

        def train(self, model, graph, loss_f, activation, sum_graph=True):
            return super().train(oracleize_(model), graph, loss_f, activation, sum_graph)

    return CpuTwinTrainer
