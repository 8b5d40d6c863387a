"""MI355X-native R-GCN message-passing engine + DistMult head.

Drop-in for the one hot path of arnold117/PrimeKG-RGCN-LinkPrediction: the relational
graph convolution (``torch_geometric.nn.RGCNConv`` as used by ``src/models/rgcn.py``) and
the DistMult scoring head, forward and backward, as hand-written HIP kernels for gfx950
behind a C ABI (``include/rgcn_hip.h`` -> ``librgcn_hip.so``).  See DESIGN.md.
"""
from .conv import RGCNConv, rgcn_conv, rgcn_encoder2, rgcn_encoder2_step
from .head import LinkPredictor, distmult
from .model import DrugDiseaseModel, DrugDiseaseRGCN
from . import consumers, ops, synth

__all__ = ["RGCNConv", "rgcn_conv", "rgcn_encoder2", "rgcn_encoder2_step", "LinkPredictor", "distmult", "DrugDiseaseModel",
           "DrugDiseaseRGCN", "consumers", "ops", "synth"]
