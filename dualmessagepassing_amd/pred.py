"""Pooling prediction heads of the model skeleton -- ``SubgraphCountingMatching/models/pred.py:17-233``
(``PredictNet`` and its Mean / Sum / Max variants; the attention / memory heads are out of scope).
Dense ``[B, *]`` work after the representations have been pooled per graph."""
import torch as th
import torch.nn as nn

from .act import init_module, map_activation_str_to_layer


class PredictNet(nn.Module):
    def __init__(self, input_dim, hidden_dim, act_func="relu", dropout=0.0, return_weights=False):
        super(PredictNet, self).__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        self.p_fc = nn.Linear(input_dim, hidden_dim)
        self.g_fc = nn.Linear(input_dim, hidden_dim)
        self.pred_fc1 = nn.Linear(hidden_dim * 4 + 4, hidden_dim)
        self.pred_fc2 = nn.Linear(hidden_dim + 4, 1)
        if return_weights:
            self.weight_fc1 = nn.Linear(hidden_dim * 4 + 2, hidden_dim)
            self.weight_fc2 = nn.Linear(hidden_dim + 2, 1)
        else:
            self.weight_fc1 = None
            self.weight_fc2 = None
        # pred.py:46-54
        init_module(self.p_fc, activation=act_func, init="normal")
        init_module(self.g_fc, activation=act_func, init="normal")
        init_module(self.pred_fc1, activation=act_func, init="normal")
        init_module(self.pred_fc2, activation=act_func, init="zero")
        if return_weights:
            init_module(self.weight_fc1, activation=act_func, init="normal")
            init_module(self.weight_fc2, activation=act_func, init="zero")

    def init_pattern(self, p_rep, p_mask=None):
        return self.p_fc(p_rep)

    def agg_pattern(self, p_rep, p_mask=None):
        return self.agg_graph(p_rep, p_mask)

    def init_graph(self, g_rep, g_mask=None):
        return self.g_fc(g_rep)

    def agg_graph(self, g_rep, g_mask=None):
        raise NotImplementedError

    def forward(self, p_rep, p_mask, g_rep, g_mask):
        # pred.py:87-156
        bsz = p_mask.size(0)
        g_len = g_mask.size(1)
        pl = p_mask.float().sum(dim=1).view(bsz, 1)
        pl_inv = 1.0 / pl
        gl = g_mask.float().sum(dim=1).view(bsz, 1)
        gl_inv = 1.0 / gl
        if p_rep.dim() == 2:
            p = p_rep.unsqueeze(1).expand(bsz, g_len, -1)
        elif p_rep.dim() == 3:
            p = self.init_pattern(p_rep, p_mask)
            p = self.drop(p)
            p = self.agg_pattern(p, p_mask)
            p = p.unsqueeze(1).expand(bsz, g_len, -1)
        else:
            raise ValueError
        g = self.init_graph(g_rep, g_mask)
        g = self.drop(g)
        if self.weight_fc1 is not None:
            w = th.cat([p, g, g - p, g * p, pl.expand(bsz, g_len).unsqueeze(-1),
                        pl_inv.expand(bsz, g_len).unsqueeze(-1)], dim=2)
            w = self.act(self.weight_fc1(w))
            w = self.weight_fc2(th.cat([w, pl.expand(bsz, g_len).unsqueeze(-1),
                                        pl_inv.expand(bsz, g_len).unsqueeze(-1)], dim=2))
            w = w.squeeze_(-1)
        else:
            w = None
        p = p[:, 0, :]
        g = self.agg_graph(g)
        y = th.cat([p, g, g - p, g * p, pl, gl, pl_inv, gl_inv], dim=1)
        y = self.act(self.pred_fc1(y))
        y = self.pred_fc2(th.cat([y, pl, gl, pl_inv, gl_inv], dim=1))
        return y, w


    # ---- pool-then-project: sum/mean pooling commutes with the affine p_fc / g_fc, so the per-row
    # Linear over [B, L, D] (an E-row GEMM and its [B, L, hid] intermediate) collapses to a
    # [B, D] x [D, hid] product on the per-graph sums:  sum_j (W x_j + b) = W sum_j x_j + L b.
    pool_kind = None  # "sum" | "mean" for the poolable heads

    def poolable(self):
        return self.pool_kind is not None and self.weight_fc1 is None and not (self.drop.p > 0.0 and self.training)

    def forward_pooled(self, p_sum, p_pad_len, pl, g_sum, g_pad_len, gl):
        """p_sum / g_sum [B, D]: sums of the (masked) pattern / graph rows; *_pad_len: padded length L
        of the reference's [B, L, D] tensors (every padded or masked position contributes the bias);
        pl / gl [B, 1]: mask counts (pred.py:93-96)."""
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
