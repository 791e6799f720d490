"""MI355X-native R-GCN layer (forward + backward) behind the reference's model/trainer API.

Importable name of the ``scaling-rgcn-training_amd`` package (a hyphen cannot be imported).
Sub-modules:

* ``plan``    -- graph plan (tile / chunk layout in HBM), pure tensor plumbing, runs anywhere
* ``_lib``    -- ctypes binding of ``librgcn_mi355x.so`` (the C ABI of include/rgcn_mi355x.h)
* ``conv``    -- ``RGCNConv``: PyG-2.3.1-compatible ``nn.Module`` whose forward/backward are the HIP kernels
* ``data``    -- ``Data`` attribute bag (stand-in for ``torch_geometric.data.Data``)
* ``layers``  -- ``Emb_Layers`` / ``Emb_ATT_Layers`` / ``Emb_MLP_Layers`` (reference model/layers.py API)
* ``trainer`` -- full-batch loop of reference model/modelTrainer.py (+ device-correct evaluation)
* ``graphs``  -- N-Triples ingest, dataset assembly, summary -> original embedding transfer (graphs/*.py, embeddingTricks.py)
* ``dist``    -- one-process-per-GPU edge partition + per-layer collective over RCCL
* ``summaries`` -- attribute-summary generation (murmur3 x64-128 of predicate sets; graphs/createAttributeSum.py)

There is no CPU compute path: the layer raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"

from .plan import CHUNK, TilePlan, GraphPlans, build_plan, build_graph_plans, edge_weights  # noqa: F401


def __getattr__(name):
    # lazy: these need torch.nn / the HIP library, keep `import scaling_rgcn_training_amd` light
    if name in ("RGCNConv", "rgcn_conv_function"):
        from . import conv
        return getattr(conv, name)
    if name == "Data":
        from .data import Data
        return Data
    if name in ("Emb_Layers", "Emb_ATT_Layers", "Emb_MLP_Layers"):
        from . import layers
        return getattr(layers, name)
    if name == "Trainer":
        from .trainer import Trainer
        return Trainer
    if name in ("create_sum_map", "hash128"):
        from . import summaries
        return getattr(summaries, name)
    raise AttributeError(name)
