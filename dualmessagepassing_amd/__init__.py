"""MI355X-native dual-message-passing hot path (gfx950).

Host side: PyTorch-ROCm ``nn.Module``s with the reference's names and signatures
(``DMPLayer``, ``DMPNNRep`` / ``DMPNN`` rep-net, graph object, collate) over the C ABI of
``csrc/libdmp_hip.so`` (``include/dmp_hip.h``).  No CPU fallback: ops raise off-GPU.
"""
from . import constants  # noqa: F401
from ._lib import DmpError, load as load_library  # noqa: F401
from .graph import BatchedGraph, GraphIndex, function  # noqa: F401
from .dmpnn import DMPLayer, DMPNNRep, DMPNNRepMixin  # noqa: F401
from .collate import batch, batchify, collate_device  # noqa: F401
from .dp import FlatGradSync, shard_range  # noqa: F401

__all__ = ["BatchedGraph", "GraphIndex", "function", "DMPLayer", "DMPNNRep", "DMPNNRepMixin", "batch",
           "batchify", "collate_device", "FlatGradSync", "shard_range", "DmpError", "load_library"]
