"""Pooling prediction heads of the model skeleton -- ``SubgraphCountingMatching/models/pred.py:17-233``
(``PredictNet`` and its Mean / Sum / Max variants; the attention / memory heads are out of scope).
Dense ``[B, *]`` work after the representations have been pooled per graph."""
import torch as th
import torch.nn as nn

from .act import init_module, map_activation_str_to_layer


class _PooledHead(th.autograd.Function):
    """The pooled ``PredictNet`` tail (pred.py:93-156 on per-graph sums) as ONE autograd node: forward and backward
    written out in plain tensor ops -- about 10 launches forward and 20 backward per head instead of the ~55 that
    recording the same algebra op by op produces (concatenation backward as zero-padded adds, bias scalings as
    separate nodes, ...).  ReLU / LeakyReLU heads (``slope``); ``scale_p`` / ``scale_g``: the factor on the Linear's
    bias (padded length for sum pooling, 1 for mean pooling with pre-divided inputs)."""

    @staticmethod
    def forward(ctx, ps, gs, pl, gl, scale_p, scale_g, Wp, bp, Wg, bg, W1, b1, W2, b2, slope=0.0):
        h = Wp.size(0)
        p = th.addmm(bp, ps, Wp.t(), beta=scale_p)
        g = th.addmm(bg, gs, Wg.t(), beta=scale_g)
        s = th.cat([pl, gl, 1.0 / pl, 1.0 / gl], dim=1)                     # [B, 4]
        f = th.cat([p, g, g - p, g * p, s], dim=1)                          # [B, 4h + 4]
        y1s = th.empty((ps.size(0), h + 4), dtype=ps.dtype, device=ps.device)
        y1s[:, h:] = s
        if slope == 0.0:
            th.clamp_min(th.addmm(b1, f, W1.t()), 0.0, out=y1s[:, :h])
        else:
            y1s[:, :h] = th.nn.functional.leaky_relu_(th.addmm(b1, f, W1.t()), slope)
        y = th.addmm(b2, y1s, W2.t())
        ctx.save_for_backward(ps, gs, f, y1s, Wp, Wg, W1, W2)
        ctx.scale_p, ctx.scale_g, ctx.slope = scale_p, scale_g, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        ps, gs, f, y1s, Wp, Wg, W1, W2 = ctx.saved_tensors
        h = Wp.size(0)
        dy = dy.contiguous()
        dW2 = dy.t() @ y1s                                                  # [1, h + 4]
        db2 = dy.sum(0)
        if ctx.slope == 0.0:
            dy1 = th.ops.aten.threshold_backward(dy * W2[:, :h], y1s[:, :h], 0.0)    # [B, h]
        else:   # on the saved output: its sign is the pre-activation's for a positive slope
            dy1 = th.ops.aten.leaky_relu_backward(dy * W2[:, :h], y1s[:, :h], ctx.slope, True)
        dW1 = dy1.t() @ f
        df = dy1 @ W1                                                       # [B, 4h + 4]
        p, g = f[:, :h], f[:, h:2 * h]
        dfp, dfg, dfd, dfm = df[:, :h], df[:, h:2 * h], df[:, 2 * h:3 * h], df[:, 3 * h:4 * h]
        dp = th.addcmul(dfp - dfd, dfm, g)
        dg = th.addcmul(dfg + dfd, dfm, p)
        dWp, dWg = dp.t() @ ps, dg.t() @ gs
        # the three bias gradients as ONE ones-row product (a dim-0 reduction kernel per [B, h] matrix costs more)
        sums = (dy.new_ones((1, dy.size(0))) @ th.cat([dy1, dp, dg], dim=1)).view(3, h)
        db1, dbp, dbg = sums[0], sums[1] * ctx.scale_p, sums[2] * ctx.scale_g
        dps = dp @ Wp if ctx.needs_input_grad[0] else None
        dgs = dg @ Wg if ctx.needs_input_grad[1] else None
        return dps, dgs, None, None, None, None, dWp, dbp, dWg, dbg, dW1, db1, dW2, db2, None


_HEAD_STRUCTS = None


def _head_structs():
    """ctypes mirrors of dmp_head_weights / dmp_head_io / dmp_head_grads (include/dmp_hip.h), built once."""
    global _HEAD_STRUCTS
    if _HEAD_STRUCTS is None:
        _HEAD_STRUCTS = _make_head_structs()
    return _HEAD_STRUCTS


def _make_head_structs():
    import ctypes
    P, I64, F32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
    W = type("dmp_head_weights", (ctypes.Structure,), {"_fields_": [(n, P) for n in ("Wp", "bp", "Wg", "bg", "W1", "b1", "W2", "b2")]})
    IO = type("dmp_head_io", (ctypes.Structure,), {"_fields_": [("ps", P), ("ld_ps", I64), ("gs", P), ("ld_gs", I64), ("pl", P), ("gl", P),
                                                                ("scale_p", F32), ("scale_g", F32), ("F", P), ("Y1S", P), ("y", P)]})
    G = type("dmp_head_grads", (ctypes.Structure,), {"_fields_": [("dy", P), ("dy_scale", P), ("dY1", P), ("dP", P), ("dG", P), ("dps", P),
                                                                  ("ld_dps", I64), ("dgs", P), ("ld_dgs", I64)]
                                                     + [(n, P) for n in ("dWp", "dbp", "dWg", "dbg", "dW1", "db1", "dW2", "db2")]})
    return W, IO, G


class _PooledHeadsHIP(th.autograd.Function):
    """All pooled heads of the model and their blend in three HIP launches (csrc/dmp_heads.hip):
    ``pred = sum_i blend_i * head_i(sums_i[:B], sums_i[B:])``.  Inputs per head: ``sums`` [2B, H] (pattern rows, then
    target rows: one gradient buffer comes back), ``pl`` / ``gl`` [B, 1], the bias factors, the blend weight [B, 1]
    (no gradient) and the eight parameters.  H = hidden = 128 or 64; activation ReLU (``slope`` 0) or LeakyReLU."""

    @staticmethod
    def forward(ctx, n_heads, slope, *args):
        from . import _lib
        lib = _lib.load()
        Wt, IOt, Gt = _head_structs()
        per = 6 + 8
        heads = [args[i * per:(i + 1) * per] for i in range(n_heads)]
        B = heads[0][0].size(0) // 2
        dev = heads[0][0].device
        W, IO = (Wt * n_heads)(), (IOt * n_heads)()
        keep, ys = [], []
        for i, (sums, pl, gl, sp, sg, blend, Wp, bp, Wg, bg, W1, b1, W2, b2) in enumerate(heads):
            if sums.stride(1) != 1 or sums.stride(0) % 4 or sums.data_ptr() % 16:   # the kernels take any 16-byte-aligned row stride
                sums = sums.contiguous()
            _lib.require_gpu(sums, Wp)
            h = Wp.size(0)
            prm = [t.detach().contiguous() for t in (Wp, bp, Wg, bg, W1, b1, W2, b2)]
            pl_, gl_ = pl.reshape(-1).contiguous().float(), gl.reshape(-1).contiguous().float()
            F = th.empty((B, 4 * h + 4), dtype=th.float32, device=dev)
            Y1S = th.empty((B, h + 4), dtype=th.float32, device=dev)
            y = th.empty((B, 1), dtype=th.float32, device=dev)
            for name, t in zip(("Wp", "bp", "Wg", "bg", "W1", "b1", "W2", "b2"), prm):
                setattr(W[i], name, t.data_ptr())
            IO[i].ps, IO[i].ld_ps, IO[i].gs, IO[i].ld_gs = sums.data_ptr(), sums.stride(0), sums[B:].data_ptr(), sums.stride(0)
            IO[i].pl, IO[i].gl, IO[i].scale_p, IO[i].scale_g = pl_.data_ptr(), gl_.data_ptr(), float(sp), float(sg)
            IO[i].F, IO[i].Y1S, IO[i].y = F.data_ptr(), Y1S.data_ptr(), y.data_ptr()
            keep.append((sums, pl_, gl_, F, Y1S,
                         blend.reshape(-1).contiguous().float() if th.is_tensor(blend) else None, prm))
            ys.append(y)
        _lib.check(lib.dmp_heads_forward(W, IO, n_heads, B, heads[0][6].size(0), float(slope), _lib.stream_ptr()), "dmp_heads_forward")
        ctx.keep, ctx.structs, ctx.n, ctx.B, ctx.per, ctx.slope = keep, (W, IO), n_heads, B, per, float(slope)
        if n_heads > 1 and all(isinstance(h[5], str) and h[5] == "len" for h in heads):
            # blend by the sizes of the target graphs (basemodel.py:1488-1494): weights and blended count in one launch
            import ctypes
            ws = [th.empty(B, dtype=th.float32, device=dev) for _ in range(n_heads)]
            out = th.empty((B, 1), dtype=th.float32, device=dev)
            Y = (ctypes.c_void_p * n_heads)(*[y.data_ptr() for y in ys])
            G = (ctypes.c_void_p * n_heads)(*[k[2].data_ptr() for k in keep])
            Wp_ = (ctypes.c_void_p * n_heads)(*[w_.data_ptr() for w_ in ws])
            _lib.check(lib.dmp_heads_blend(Y, G, Wp_, n_heads, B, out.data_ptr(), _lib.stream_ptr()), "dmp_heads_blend")
            ctx.keep = [k[:5] + (w_,) + k[6:] for k, w_ in zip(keep, ws)]
            return out
        out = None
        for (k, y) in zip(keep, ys):
            term = y if k[5] is None else y * k[5].view(-1, 1)
            out = term if out is None else out + term
        return out

    @staticmethod
    def backward(ctx, d):
        from . import _lib
        lib = _lib.load()
        _, _, Gt = _head_structs()
        W, IO = ctx.structs
        n, B = ctx.n, ctx.B
        d = d.contiguous().float()
        G = (Gt * n)()
        grads, hold = [None, None], []
        for i, (sums, pl_, gl_, F, Y1S, blend, prm) in enumerate(ctx.keep):
            h = prm[0].size(0)
            dev = sums.device
            scr = th.empty((3, B, h), dtype=th.float32, device=dev)
            dsums = th.empty(sums.shape, dtype=sums.dtype, device=sums.device)
            shapes = [(h, h), (h,), (h, h), (h,), (h, 4 * h + 4), (h,), (1, h + 4), (1,)]
            sizes = [a[0] * (a[1] if len(a) > 1 else 1) for a in shapes]
            offs, tot = [], 0
            for z in sizes:
                offs.append(tot)
                tot += (z + 3) // 4 * 4
            buf = th.empty(tot, dtype=th.float32, device=dev)
            outs = [buf[o:o + z].view(shp) for o, z, shp in zip(offs, sizes, shapes)]
            G[i].dy, G[i].dy_scale = d.data_ptr(), (blend.data_ptr() if blend is not None else None)
            G[i].dY1, G[i].dP, G[i].dG = scr[0].data_ptr(), scr[1].data_ptr(), scr[2].data_ptr()
            G[i].dps, G[i].ld_dps, G[i].dgs, G[i].ld_dgs = dsums.data_ptr(), dsums.stride(0), dsums[B:].data_ptr(), dsums.stride(0)
            for name, t in zip(("dWp", "dbp", "dWg", "dbg", "dW1", "db1", "dW2", "db2"), outs):
                setattr(G[i], name, t.data_ptr())
            hold.append((scr, buf))
            grads += [dsums, None, None, None, None, None] + outs
        _lib.check(lib.dmp_heads_backward(W, IO, G, n, B, ctx.keep[0][6][0].size(0), ctx.slope, _lib.stream_ptr()), "dmp_heads_backward")
        return tuple(grads)


class PredictNet(nn.Module):
    """Count head (and optional per-row matching head) over a pattern and a target representation.

    Sub-modules, their order and their initialisers are the reference's (pred.py:20-54: ``state_dict`` keys and the
    seeded initial values depend on them); the computation is organised differently: the per-row matching head never
    builds the ``[B, L, 4h+2]`` concatenation -- the first Linear is applied block by block, and everything that
    depends on the pair only (the pattern vector, the size features, the bias) is computed once per pair and
    broadcast over the rows."""

    # name, (in, out) as functions of (input_dim, hidden_dim), initialiser (pred.py:46-54); weight_* only with return_weights
    _SPEC = (("p_fc", lambda d, h: (d, h), "normal", False), ("g_fc", lambda d, h: (d, h), "normal", False),
             ("pred_fc1", lambda d, h: (4 * h + 4, h), "normal", False), ("pred_fc2", lambda d, h: (h + 4, 1), "zero", False),
             ("weight_fc1", lambda d, h: (4 * h + 2, h), "normal", True), ("weight_fc2", lambda d, h: (h + 2, 1), "zero", True))

    def __init__(self, input_dim, hidden_dim, act_func="relu", dropout=0.0, return_weights=False):
        super(PredictNet, self).__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        made = []
        for name, dims, init, optional in self._SPEC:          # construction first, initialisation after (RNG order)
            lin = nn.Linear(*dims(input_dim, hidden_dim)) if (return_weights or not optional) else None
            setattr(self, name, lin)
            made.append((lin, init))
        for lin, init in made:
            if lin is not None:
                init_module(lin, activation=act_func, init=init)

    # ---- hooks of the reference's class (subclasses override agg_graph)
    def init_pattern(self, p_rep, p_mask=None):
        return self.p_fc(p_rep)

    def init_graph(self, g_rep, g_mask=None):
        return self.g_fc(g_rep)

    def agg_graph(self, g_rep, g_mask=None):
        raise NotImplementedError

    def agg_pattern(self, p_rep, p_mask=None):
        return self.agg_graph(p_rep, p_mask)

    # ---- pieces of forward
    @staticmethod
    def _size_features(p_mask, g_mask):
        """[B, 4] = (pl, gl, 1/pl, 1/gl): number of unmasked pattern / target positions of every pair (pred.py:93-96)."""
        pl = p_mask.sum(dim=1, dtype=th.float32).unsqueeze(1)
        gl = g_mask.sum(dim=1, dtype=th.float32).unsqueeze(1)
        return th.cat([pl, gl, pl.reciprocal(), gl.reciprocal()], dim=1)

    def _pattern_vector(self, p_rep, p_mask):
        if p_rep.dim() == 2:                                   # already one vector per pattern (pred.py:99-100)
            return p_rep
        if p_rep.dim() != 3:
            raise ValueError("p_rep must be [B, dim] or [B, p_len, dim]")
        return self.agg_pattern(self.drop(self.init_pattern(p_rep, p_mask)), p_mask)

    def _row_weights(self, p, g_rows, sizes):
        """The matching head (pred.py:113-131) on rows ``g_rows`` [B, L, h] against the pair vectors ``p`` [B, h]:
        ``weight_fc2([act(weight_fc1([p, g, g - p, g * p, pl, 1/pl])), pl, 1/pl])`` with weight_fc1 split into its
        column blocks: ``g (Wg + Wd)^T + (g * p) Wm^T`` per row, ``p (Wp - Wd)^T + [pl, 1/pl] Ws^T + b`` per pair."""
        h = self.hidden_dim
        W, b = self.weight_fc1.weight, self.weight_fc1.bias
        Wp, Wg, Wd, Wm, Ws = W[:, :h], W[:, h:2 * h], W[:, 2 * h:3 * h], W[:, 3 * h:4 * h], W[:, 4 * h:]
        ps = sizes[:, (0, 2)]                                  # (pl, 1/pl)
        per_pair = th.addmm(b, p, (Wp - Wd).t()) + ps @ Ws.t()                       # [B, h]
        hidden = self.act(g_rows @ (Wg + Wd).t() + (g_rows * p.unsqueeze(1)) @ Wm.t() + per_pair.unsqueeze(1))
        W2, b2 = self.weight_fc2.weight, self.weight_fc2.bias
        out = hidden @ W2[0, :h] + (ps @ W2[0, h:] + b2).unsqueeze(1)                 # [B, L]
        return out

    def _count(self, p, g, sizes):
        """pred.py:137-154: the count from the pair vectors."""
        feats = th.cat([p, g, g - p, g * p, sizes], dim=1)
        return self.pred_fc2(th.cat([self.act(self.pred_fc1(feats)), sizes], dim=1))

    def forward(self, p_rep, p_mask, g_rep, g_mask):
        """p_rep [B, p_len, D] (or [B, h]), g_rep [B, g_len, D], masks [B, len] -> (count [B, 1], row weights [B, g_len]
        or None)."""
        sizes = self._size_features(p_mask, g_mask)
        p = self._pattern_vector(p_rep, p_mask)
        g_rows = self.drop(self.init_graph(g_rep, g_mask))
        w = self._row_weights(p, g_rows, sizes) if self.weight_fc1 is not None else None
        return self._count(p, self.agg_graph(g_rows), sizes), w

    # ---- pool-then-project: sum/mean pooling commutes with the affine p_fc / g_fc, so the per-row
    # Linear over [B, L, D] (an E-row GEMM and its [B, L, hid] intermediate) collapses to a
    # [B, D] x [D, hid] product on the per-graph sums:  sum_j (W x_j + b) = W sum_j x_j + L b.
    pool_kind = None  # "sum" | "mean" for the poolable heads

    def poolable(self):
        return self.pool_kind is not None and self.weight_fc1 is None and not (self.drop.p > 0.0 and self.training)

    def act_slope(self):
        """0.0 (ReLU) / the negative slope (LeakyReLU: the reference's default pred_act_func) / None (anything else)."""
        from .fused import activation_slope
        return activation_slope(self.act)

    def hip_head_ok(self, sums):
        """The three-launch HIP heads (``_PooledHeadsHIP``) compute this head: ReLU / LeakyReLU, input = hidden width 128 or
        64 (the reference's shipped pred_hid_dim), fp32 on the GPU."""
        return (self.act_slope() is not None and self.pool_kind == "sum" and self.input_dim in (128, 64)
                and self.hidden_dim == self.input_dim and sums is not None and sums.is_cuda and sums.dtype == th.float32
                and sums.dim() == 2 and sums.size(1) == self.input_dim)

    def head_params(self):
        return (self.p_fc.weight, self.p_fc.bias, self.g_fc.weight, self.g_fc.bias, self.pred_fc1.weight, self.pred_fc1.bias,
                self.pred_fc2.weight, self.pred_fc2.bias)

    def forward_pooled(self, p_sum, p_pad_len, pl, g_sum, g_pad_len, gl):
        """p_sum / g_sum [B, D]: sums of the (masked) pattern / graph rows; *_pad_len: padded length L
        of the reference's [B, L, D] tensors (every padded or masked position contributes the bias);
        pl / gl [B, 1]: mask counts (pred.py:93-96)."""
        if self.act_slope() is not None and p_sum.is_cuda and p_sum.dim() == 2:
            sp, sg = (float(p_pad_len), float(g_pad_len)) if self.pool_kind == "sum" else (1.0, 1.0)
            if self.pool_kind != "sum":
                p_sum, g_sum = p_sum / float(p_pad_len), g_sum / float(g_pad_len)
            y = _PooledHead.apply(p_sum, g_sum, pl, gl, sp, sg, self.p_fc.weight, self.p_fc.bias, self.g_fc.weight,
                                  self.g_fc.bias, self.pred_fc1.weight, self.pred_fc1.bias, self.pred_fc2.weight,
                                  self.pred_fc2.bias, self.act_slope())
            return y, None
        pl_inv, gl_inv = 1.0 / pl, 1.0 / gl
        if self.pool_kind == "sum":
            # W sum_j x_j + L b as one addmm each (beta = L scales the bias)
            p = th.addmm(self.p_fc.bias, p_sum, self.p_fc.weight.t(), beta=float(p_pad_len))
            g = th.addmm(self.g_fc.bias, g_sum, self.g_fc.weight.t(), beta=float(g_pad_len))
        else:
            p = self.p_fc(p_sum / float(p_pad_len))
            g = self.g_fc(g_sum / float(g_pad_len))
        y = th.cat([p, g, g - p, g * p, pl, gl, pl_inv, gl_inv], dim=1)
        y = self.act(self.pred_fc1(y))
        y = self.pred_fc2(th.cat([y, pl, gl, pl_inv, gl_inv], dim=1))
        return y, None


class MeanPredictNet(PredictNet):
    pool_kind = "mean"

    def agg_graph(self, g_rep, g_mask=None):
        return th.mean(g_rep, dim=1)


class SumPredictNet(PredictNet):
    pool_kind = "sum"

    def agg_graph(self, g_rep, g_mask=None):
        return th.sum(g_rep, dim=1)


class MaxPredictNet(PredictNet):
    def agg_graph(self, g_rep, g_mask=None):
        return th.max(g_rep, dim=1)[0]


PRED_NETS = {"MeanPredictNet": MeanPredictNet, "SumPredictNet": SumPredictNet, "MaxPredictNet": MaxPredictNet}
