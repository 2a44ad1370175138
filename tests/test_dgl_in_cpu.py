"""DGLGraph-in adapter: what it refuses (no GPU needed)."""
import pytest
import torch as th

from util_dglike import DGLike


def test_adapter_rejects_what_it_cannot_run():
    from dualmessagepassing_amd import DmpError
    from dualmessagepassing_amd.graph import BatchedGraph
    with pytest.raises(TypeError):
        BatchedGraph.from_graph(object())
    g = DGLike(th.tensor([0, 1]), th.tensor([1, 0]), 2)       # host tensors: there is no CPU path
    with pytest.raises(DmpError):
        BatchedGraph.from_graph(g)
