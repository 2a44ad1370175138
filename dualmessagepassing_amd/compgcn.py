"""CompGCNLayer on the MI355X kernels -- drop-in for
``SubgraphCountingMatching/models/compgcn.py:102-287`` (same constructor arguments,
parameter names/shapes, ``forward(graph, node_feat, edge_feat) -> (node_out, edge_out)``).

    msg[e]   = comp(X[src e], Z[e]) (r_e ? W_out : W_in) * norm[e]          (compgcn.py:226-238)
    node_out = drop(act(BN?((sum_{e->v} msg[e] + comp(X, loop_rel) W_loop) * 0.3333333 + b)))
    edge_out = Z W_rel                                                       (compgcn.py:260-263)

The composition and the per-edge norm are fused into the segment sum (one HIP kernel,
``dmp_compgcn_agg``), and the W_in / W_out products are applied to the N summed rows instead
of the E message rows.

``comp_opt="corr"`` -- the reference's DEFAULT composition (config.py:170-172) -- is the circular correlation
``irfft(conj(rfft(h)) * rfft(r))`` (compgcn.py:218-222).  The reference gathers ``X[src]`` to [E, H], runs two forward and
one inverse FFT over E rows and then the products.  The DFT is a fixed linear map, so it commutes with the sum over the
in-edges:

    sum_e norm_e irfft(conj(F x_src(e)) . F z_e) W = irfft(sum_e norm_e conj(F x_src(e)) . F z_e) W

    Fx = X D        [N, 2 bins]     D: the real DFT as an [H, 2 bins] matrix, (re, im) interleaved
    Fz = Z D        [E, 2 bins]     one E-row product (the only [E, .] tensor of the composition)
    S  = dmp_compgcn_agg(Fx, Fz, comp = conj-multiply)        [N, 2 x 2 bins]   the same fused gather + multiply + segment sum
    agg = S [D^-1 W_in ; D^-1 W_out]                          N rows; D^-1 W is an [2 bins, H] matrix built per call (tiny)

-- no per-edge FFT, no gathered [E, H] tensor, no library FFT call.
"""
import torch as th
import torch.nn as nn

from . import ops
from ._lib import on_input_device
from .act import init_weight, map_activation_str_to_layer
from .constants import (EDGEFEAT, INDEGREE, INNORM, NODEAGG, NODEFEAT, NORM, OUTDEGREE, OUTNORM, REVFLAG)
from .dmpnn import DMPNNRepMixin
from .graph import leave_detached, BatchedGraph, as_batched


_DFT = {}
USE_FREQ_DOMAIN_CORR = True    # False: the reference's own formulation (gather, library FFTs over E rows), kept for comparison


def dft_matrices(h, device):
    """``(D [h, w], Dinv [w, h])`` with ``x @ D`` = rfft(x) as interleaved (re, im) pairs (w = 2 (h // 2 + 1) rounded up to a
    multiple of 4, the padding columns zero) and ``F @ Dinv`` = irfft(F, n=h) (which ignores the imaginary parts of the
    DC and Nyquist bins, as the library's real inverse does).  Built in float64, cached per (h, device)."""
    key = (int(h), str(device))
    if key not in _DFT:
        bins = h // 2 + 1
        w = (2 * bins + 3) // 4 * 4
        j = th.arange(h, dtype=th.float64).view(-1, 1)
        k = th.arange(bins, dtype=th.float64).view(1, -1)
        ang = 2.0 * th.pi * j * k / h
        D = th.zeros(h, w, dtype=th.float64)
        D[:, 0:2 * bins:2] = th.cos(ang)
        D[:, 1:2 * bins:2] = -th.sin(ang)
        c = th.full((bins,), 2.0, dtype=th.float64)
        c[0] = 1.0
        if h % 2 == 0:
            c[-1] = 1.0
        Dinv = th.zeros(w, h, dtype=th.float64)
        Dinv[0:2 * bins:2] = (th.cos(ang) * c).t() / h
        Dinv[1:2 * bins:2] = (-th.sin(ang) * c).t() / h
        _DFT[key] = (D.float().to(device), Dinv.float().to(device))
    return _DFT[key]


def _cmul_conj(fx, fr):
    """``conj(fx) * fr`` over interleaved (re, im) rows (``fr`` broadcasts over the rows)."""
    xr, xi, rr, ri = fx[:, 0::2], fx[:, 1::2], fr[:, 0::2], fr[:, 1::2]
    return th.stack([xr * rr + xi * ri, xr * ri - xi * rr], dim=-1).reshape(fx.size(0), -1)


class CompGCNLayer(nn.Module):
    def __init__(
        self,
        input_dim,
        hidden_dim,
        self_loop=True,
        comp_opt="mult",
        edge_norm="both",
        bias=True,
        batch_norm=True,
        act_func="relu",
        dropout=0.0
    ):
        super(CompGCNLayer, self).__init__()
        assert edge_norm in ["none", "in", "out", "both"]
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.edge_norm = edge_norm
        self.num_rels = 3 if self_loop else 2
        self.comp_opt = comp_opt

        # compgcn.py:127-149 (registration order kept: it is the state_dict order)
        if self_loop:
            self.loop_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        else:
            self.register_parameter("loop_weight", None)
        if bias:
            self.bias = nn.Parameter(th.empty(hidden_dim))
        else:
            self.register_parameter("bias", None)
        self.bn = nn.BatchNorm1d(hidden_dim) if batch_norm else None
        self.in_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.out_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.rel_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        if self_loop:
            self.loop_rel = nn.Parameter(th.empty(1, input_dim))
        else:
            self.register_parameter("loop_rel", None)
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)

        # compgcn.py:152-160
        init_weight(self.in_weight, activation=act_func, init="uniform")
        init_weight(self.out_weight, activation=act_func, init="uniform")
        init_weight(self.rel_weight, activation=act_func, init="uniform")
        if self_loop:
            init_weight(self.loop_weight, activation=act_func, init="uniform")
            init_weight(self.loop_rel, activation=act_func, init="uniform")
        if bias:
            nn.init.zeros_(self.bias)

    @property
    def self_loop(self):
        return hasattr(self, "loop_weight") and self.loop_weight is not None

    # composition operators of compgcn.py:213-224 (corr: circular correlation along the feature axis via the real FFT)
    _COMP = {
        "sub": lambda h, r: h - r,
        "mult": lambda h, r: h * r,
        "corr": lambda h, r: th.fft.irfft(th.conj(th.fft.rfft(h, dim=-1)) * th.fft.rfft(r, dim=-1), n=h.size(-1), dim=-1),
    }

    def _comp_func(self, head, relation):
        try:
            return self._COMP[self.comp_opt](head, relation)
        except KeyError:
            raise NotImplementedError(self.comp_opt)

    def _node_norm(self, deg):
        """1 / (degree + 1) with a self loop; else 1 / degree with isolated nodes at 1 (compgcn.py:177-196)."""
        if self.self_loop:
            return (deg + 1).reciprocal().unsqueeze(-1)
        return deg.reciprocal().masked_fill_(deg == 0, 1.0).unsqueeze(-1)

    def _norms(self, g):
        """Per-edge normaliser [E] (None for edge_norm "none"): the destination's in-norm, the source's out-norm, or the
        square root of their product (compgcn.py:204-209); the node norms are cached in ``ndata`` like the reference's."""
        want_in, want_out = self.edge_norm in ("in", "both"), self.edge_norm in ("out", "both")
        if want_in and INNORM not in g.ndata:
            g.ndata[INNORM] = self._node_norm(g.in_degrees())
        if want_out and OUTNORM not in g.ndata:
            g.ndata[OUTNORM] = self._node_norm(g.out_degrees())
        if not (want_in or want_out):
            return None
        u, v = g.all_edges(form="uv", order="eid")
        parts = ([g.ndata[OUTNORM][u]] if want_out else []) + ([g.ndata[INNORM][v]] if want_in else [])
        g.edata[NORM] = parts[0] if len(parts) == 1 else (parts[0] * parts[1]) ** 0.5
        return g.edata[NORM].reshape(-1)

    @on_input_device
    def forward(self, graph, node_feat, edge_feat):
        g = as_batched(graph)   # DGLGraph-in (compgcn.py:265): frames shared with the caller's graph
        if node_feat is not None:
            g.ndata[NODEFEAT] = node_feat
        if edge_feat is not None:
            g.edata[EDGEFEAT] = edge_feat
        x, z = g.ndata[NODEFEAT], g.edata[EDGEFEAT]
        norm = self._norms(g)
        ix = g.index()
        has_rev = REVFLAG in g.edata
        h = self.input_dim

        # _node_message_func + fn.sum (compgcn.py:226-238,271)
        fx = None
        if self.comp_opt in ("sub", "mult"):
            s = ops.compgcn_agg(x, z, norm, ix, ops.COMP_SUB if self.comp_opt == "sub" else ops.COMP_MULT)
            w_in, w_out = self.in_weight, self.out_weight
        elif self.comp_opt == "corr" and x.is_cuda and USE_FREQ_DOMAIN_CORR:
            # the correlation in the frequency domain (module docstring): two products with the DFT matrix, the fused
            # conj-multiply segment sum, the inverse transform folded into the weights of the N-row product
            D, Dinv = dft_matrices(h, x.device)
            fx = ops.matmul_xw(x, D)
            s = ops.compgcn_agg(fx, ops.matmul_xw(z, D), norm, ix, ops.COMP_CMUL)
            w_in, w_out = Dinv @ self.in_weight, Dinv @ self.out_weight
            h = D.size(1)
        else:
            comp = self._comp_func(ops.gather_src(x, ix), z)
            s = ops.seg_sum2(comp, ix, norm, 1.0, 1.0)
            w_in, w_out = self.in_weight, self.out_weight
        if has_rev:
            agg = ops.matmul_xw(s, th.cat([w_in, w_out], dim=0))
        else:
            agg = ops.matmul_xw(s[:, :h], w_in)
        g.ndata[NODEAGG] = agg

        # _node_update_func (compgcn.py:240-258)
        if self.self_loop:
            if fx is not None:      # corr(x, loop_rel) from the transform of x that is already there
                D, Dinv = dft_matrices(self.input_dim, x.device)
                loop_msg = th.matmul(_cmul_conj(fx, th.matmul(self.loop_rel, D)), Dinv @ self.loop_weight)
            else:
                loop_msg = th.matmul(self._comp_func(x, self.loop_rel), self.loop_weight)
            out = (agg + loop_msg) * 0.3333333
        else:
            out = agg * 0.5
        if self.bias is not None:
            out = out + self.bias
        if self.bn is not None:
            out = self.bn(out)
        node_out = self.drop(self.act(out))

        # _edge_update_func (compgcn.py:260-263)
        edge_out = ops.matmul_xw(z, self.rel_weight)
        leave_detached(g.ndata, NODEFEAT, NODEAGG)
        leave_detached(g.edata, EDGEFEAT)
        return node_out, edge_out

    def extra_repr(self):
        return "\n".join([
            "in=%s, out=%s," % (self.input_dim, self.hidden_dim),
            "comp_opt=%s," % (self.comp_opt),
            "edge_norm=%s, self_loop=%s, bias=%s," % (self.edge_norm, self.self_loop, self.bias is not None),
        ])

    def get_output_dim(self):
        return self.hidden_dim


class CompGCNRepMixin(DMPNNRepMixin):
    """``create_rep_net`` of the reference's ``CompGCN`` model (compgcn.py:289-321); its
    ``get_pattern_rep`` / ``get_graph_rep`` (compgcn.py:323-385) are the DMPNN loops over
    ``rep_net["compgcn"]``."""

    rep_key = "compgcn"

    def _make_layer(self, **kw):
        return CompGCNLayer(self.hid_dim, self.hid_dim, comp_opt=kw.get("rep_compgcn_comp_opt", "mult"),
                            edge_norm=kw.get("rep_compgcn_edge_norm", "none"), batch_norm=kw.get("rep_compgcn_batch_norm", False),
                            act_func=kw.get("rep_act_func", "relu"), dropout=kw.get("rep_dropout", 0.0))
