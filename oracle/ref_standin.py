"""TEST INFRASTRUCTURE ONLY -- not part of the product path.

Stand-ins for the third-party modules the reference imports but which are not
installed in the development container (dgl, numba, igraph, tensorboardX,
torch._six).  With these in ``sys.modules`` the reference's *own* model code
(``/root/reference/SubgraphCountingMatching/models/dmpnn.py`` etc.) imports and
runs on CPU, which is how ``oracle/make_golden.py`` emits the golden fixtures
under ``tests/golden/``.  Nothing here is imported by the shipped package.

What DGL contributes to the hot path is only
  * gathers  ``edges.src[k]`` / ``edges.dst[k]``  (eid order),
  * ``fn.sum``  = segment-sum by destination,
  * ``in_degrees`` / ``out_degrees`` = bincount,
  * ``dgl.batch`` = node-offset concatenation,
  * graph mutation used by ``convert_to_dual_graph`` (utils/graph.py:74-169).
The stand-in implements exactly that surface with plain torch CPU ops; the
segment-sum is ``index_add_`` in eid order.
"""
import collections
import collections.abc
import sys
import types

import torch as th


# --------------------------------------------------------------------------- dgl.function
class _SumReducer:
    def __init__(self, msg, out):
        self.msg = msg
        self.out = out


class _CopyMessage:
    def __init__(self, target, in_key, out_key):
        self.target = target
        self.in_key = in_key
        self.out_key = out_key


class _TargetCode:
    SRC = 0
    DST = 1
    EDGE = 2


# --------------------------------------------------------------------------- batches handed to UDFs
class _EdgeBatch:
    def __init__(self, g):
        self._g = g
        self.src = _Gather(g.ndata, g._u)
        self.dst = _Gather(g.ndata, g._v)
        self.data = g.edata  # same dict: UDF side-effect writes persist (dmpnn.py:126)

    def __len__(self):
        return self._g.number_of_edges()


class _Gather:
    def __init__(self, frame, idx):
        self._frame = frame
        self._idx = idx

    def __getitem__(self, k):
        return self._frame[k][self._idx]

    def __contains__(self, k):
        return k in self._frame


class _NodeBatch:
    def __init__(self, g):
        self.data = g.ndata


# --------------------------------------------------------------------------- DGLGraph
class DGLGraph:
    """Minimal multigraph with eid-ordered edge list."""

    def __init__(self, *args, **kw):
        self._u = th.zeros((0,), dtype=th.long)
        self._v = th.zeros((0,), dtype=th.long)
        self._n = 0
        self.ndata = {}
        self.edata = {}
        self._batch_num_nodes = None
        self._batch_num_edges = None

    # construction helpers (stand-in API, not DGL's)
    @classmethod
    def from_edges(cls, u, v, n):
        g = cls()
        g._u = th.as_tensor(u, dtype=th.long).clone()
        g._v = th.as_tensor(v, dtype=th.long).clone()
        g._n = int(n)
        return g

    # ---- DGL surface
    def readonly(self, flag=True):
        pass

    @property
    def batch_size(self):
        return 1 if self._batch_num_nodes is None else len(self._batch_num_nodes)

    def batch_num_nodes(self, *a):
        if self._batch_num_nodes is None:
            return th.tensor([self._n])
        return th.as_tensor(self._batch_num_nodes)

    def batch_num_edges(self, *a):
        if self._batch_num_edges is None:
            return th.tensor([self._u.numel()])
        return th.as_tensor(self._batch_num_edges)

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return self._u.numel()

    num_edges = number_of_edges
    num_nodes = number_of_nodes

    def out_degrees(self):
        return th.bincount(self._u, minlength=self._n)

    def in_degrees(self):
        return th.bincount(self._v, minlength=self._n)

    def all_edges(self, form="uv", order="eid"):
        e = th.arange(self._u.numel())
        if order == "srcdst":  # DGL: sorted by (src, dst); equal pairs keep edge-id order
            key = self._u * max(self._n, 1) + self._v
            e = th.sort(key, stable=True)[1]
            if form == "eid":
                return e
            return (self._u[e], self._v[e]) if form == "uv" else (self._u[e], self._v[e], e)
        assert order == "eid"
        if form == "uv":
            return self._u, self._v
        if form == "eid":
            return e
        return self._u, self._v, e

    edges = all_edges

    def add_nodes(self, k, data=None):
        old = self._n
        self._n += int(k)
        for key, val in list(self.ndata.items()):
            pad = th.zeros((int(k),) + tuple(val.shape[1:]), dtype=val.dtype)
            self.ndata[key] = th.cat([val, pad], 0)
        if data:
            for key, val in data.items():
                if key not in self.ndata:
                    self.ndata[key] = th.zeros((old,) + tuple(val.shape[1:]), dtype=val.dtype)
                    self.ndata[key] = th.cat([self.ndata[key], val], 0)
                else:
                    self.ndata[key][old:] = val

    def add_edges(self, u, v, data=None):
        u = th.as_tensor(u, dtype=th.long)
        v = th.as_tensor(v, dtype=th.long)
        old = self._u.numel()
        k = u.numel()
        self._u = th.cat([self._u, u])
        self._v = th.cat([self._v, v])
        for key, val in list(self.edata.items()):
            pad = th.zeros((k,) + tuple(val.shape[1:]), dtype=val.dtype)  # DGL zero-fills
            self.edata[key] = th.cat([val, pad], 0)
        if data:
            for key, val in data.items():
                if key not in self.edata:
                    self.edata[key] = th.cat(
                        [th.zeros((old,) + tuple(val.shape[1:]), dtype=val.dtype), val], 0)
                else:
                    self.edata[key][old:] = val

    def remove_nodes(self, ids):
        ids = set(int(i) for i in th.as_tensor(ids).tolist())
        keep = th.tensor([i for i in range(self._n) if i not in ids], dtype=th.long)
        remap = th.full((self._n,), -1, dtype=th.long)
        remap[keep] = th.arange(keep.numel())
        emask = (remap[self._u] >= 0) & (remap[self._v] >= 0)
        self._u = remap[self._u[emask]]
        self._v = remap[self._v[emask]]
        for key in list(self.ndata):
            self.ndata[key] = self.ndata[key][keep]
        for key in list(self.edata):
            self.edata[key] = self.edata[key][emask]
        self._n = keep.numel()

    def incidence_matrix(self, typestr):
        assert typestr == "in"
        e = self._u.numel()
        idx = th.stack([self._v, th.arange(e)])
        return th.sparse_coo_tensor(idx, th.ones(e), size=(self._n, e))

    def update_all(self, mfn, rfn, ufn=None):
        m = mfn(_EdgeBatch(self))
        msg = m[rfn.msg]
        out = th.zeros((self._n,) + tuple(msg.shape[1:]), dtype=msg.dtype)
        out = out.index_add(0, self._v, msg)
        self.ndata[rfn.out] = out
        if ufn is not None:
            self.ndata.update(ufn(_NodeBatch(self)))

    def apply_edges(self, f):
        if isinstance(f, _CopyMessage):
            src = self._u if f.target == _TargetCode.SRC else self._v
            self.edata[f.out_key] = self.ndata[f.in_key][src]
            return
        self.edata.update(f(_EdgeBatch(self)))

    def local_var(self):
        return self

    def to(self, device):
        return self


def batch(graphs):
    """dgl.batch semantics: node-offset concatenation in list order."""
    out = graphs[0].__class__.__new__(graphs[0].__class__)
    DGLGraph.__init__(out)
    off = 0
    us, vs = [], []
    for g in graphs:
        us.append(g._u + off)
        vs.append(g._v + off)
        off += g._n
    out._u = th.cat(us)
    out._v = th.cat(vs)
    out._n = off
    out._batch_num_nodes = [g._n for g in graphs]
    out._batch_num_edges = [g._u.numel() for g in graphs]
    for key in graphs[0].ndata:
        out.ndata[key] = th.cat([g.ndata[key] for g in graphs], 0)
    for key in graphs[0].edata:
        out.edata[key] = th.cat([g.edata[key] for g in graphs], 0)
    return out


# --------------------------------------------------------------------------- numba
class _Sig:
    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return self


def _jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not isinstance(a[0], _Sig) and not k:
        return a[0]

    def deco(f):
        return f

    return deco


def install():
    """Put the stand-ins into sys.modules (idempotent)."""
    if "dgl" in sys.modules and getattr(sys.modules["dgl"], "_IS_STANDIN", False):
        return
    sys.dont_write_bytecode = True  # reference tree is read-only

    six = types.ModuleType("torch._six")
    six.container_abcs = collections.abc
    six.string_classes = (str, bytes)
    six.int_classes = (int,)
    sys.modules["torch._six"] = six

    ig = types.ModuleType("igraph")

    class _IGraph:  # never instantiated on the paths we exercise
        pass

    ig.Graph = _IGraph
    sys.modules["igraph"] = ig

    nb = types.ModuleType("numba")
    nb.jit = _jit
    nb.njit = _jit
    for name in ("int64", "int32", "float32", "float64", "boolean", "void"):
        setattr(nb, name, _Sig())
    sys.modules["numba"] = nb

    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    sys.modules["tensorboardX"] = tbx

    dgl = types.ModuleType("dgl")
    dgl._IS_STANDIN = True
    dgl.__version__ = "0.6.0"
    dgl.DGLGraph = DGLGraph
    dgl.batch = batch
    dgl.NID = "_ID"
    dgl.EID = "_ID"
    fn = types.ModuleType("dgl.function")
    fn.sum = _SumReducer
    fn.CopyMessageFunction = _CopyMessage
    fn.TargetCode = _TargetCode
    dgl.function = fn
    sys.modules["dgl"] = dgl
    sys.modules["dgl.function"] = fn
    dnn = types.ModuleType("dgl.nn")
    dnnp = types.ModuleType("dgl.nn.pytorch")

    class RelGraphConv(th.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    dnnp.RelGraphConv = RelGraphConv
    dnn.pytorch = dnnp
    dgl.nn = dnn
    sys.modules["dgl.nn"] = dnn
    sys.modules["dgl.nn.pytorch"] = dnnp


REF_SCM = "/root/reference/SubgraphCountingMatching"
REF_UNC = "/root/reference/UnsupervisedNodeClassification/Model/DMPNN/src"


def import_scm():
    """Import the reference SCM ``models`` / ``utils`` packages over the stand-ins."""
    install()
    if REF_SCM not in sys.path:
        sys.path.insert(0, REF_SCM)
    import models  # noqa: F401  (reference package)
    import utils.graph  # noqa: F401
    return sys.modules["models"]
