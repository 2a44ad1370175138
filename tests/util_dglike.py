"""A minimal graph class with the DGLGraph surface the reference's models touch -- written for the tests (NOT the
reference's code, not the product's BatchedGraph): eid-ordered edge list, frame objects that are mutable mappings
but not dicts, ``batch_num_nodes`` / ``batch_num_edges``.  Used to check the DGLGraph-in adapter."""
from collections.abc import MutableMapping

import torch as th


class Frame(MutableMapping):
    def __init__(self):
        self._cols = {}
        self.writes = []

    def __getitem__(self, k):
        return self._cols[k]

    def __setitem__(self, k, v):
        self.writes.append(k)
        self._cols[k] = v

    def __delitem__(self, k):
        del self._cols[k]

    def __iter__(self):
        return iter(self._cols)

    def __len__(self):
        return len(self._cols)


class DGLike:
    def __init__(self, u, v, n, batch_num_nodes=None, batch_num_edges=None):
        self._u, self._v, self._n = u, v, int(n)
        self.ndata, self.edata = Frame(), Frame()
        self._bnn, self._bne = batch_num_nodes, batch_num_edges

    @property
    def device(self):
        return self._u.device

    @property
    def batch_size(self):
        return 1 if self._bnn is None else len(self._bnn)

    def batch_num_nodes(self):
        return th.tensor([self._n]) if self._bnn is None else self._bnn

    def batch_num_edges(self):
        return th.tensor([self._u.numel()]) if self._bne is None else self._bne

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._u.numel())

    def all_edges(self, form="uv", order="eid"):
        assert form == "uv" and order == "eid"
        return self._u, self._v

    def in_degrees(self):
        return th.bincount(self._v, minlength=self._n)

    def out_degrees(self):
        return th.bincount(self._u, minlength=self._n)

    def to(self, device):
        g = DGLike(self._u.to(device), self._v.to(device), self._n,
                   None if self._bnn is None else self._bnn.to(device), None if self._bne is None else self._bne.to(device))
        for k, v in self.ndata.items():
            g.ndata[k] = v.to(device)
        for k, v in self.edata.items():
            g.edata[k] = v.to(device)
        return g
