"""Flip-aware gradient comparison with a PROOF obligation.

The derivative of ReLU / LeakyReLU is discontinuous at 0.  The product associates its sums differently from the oracle
(project-then-gather, folded first Linear), so a pre-activation that lies within fp32 rounding of zero may take the other
derivative branch -- about one activation in 10^5 at tau = 1e-5.  Such a flip changes ONE hidden unit's derivative and
spreads from there along the graph.  A gradient comparison may therefore exceed the tight tolerance, but ONLY in rows
that such an activation can reach:

  1. the oracle records its pre-activations (``dmp_oracle.PROBE``);
  2. ``ambiguous_rows`` finds the hidden rows with a pre-activation within ``tau * scale`` of zero;
  3. ``taint`` pushes them backward through the layers (a perturbed hidden row of the node MLP reaches that node's input
     gradient and the input gradients of its in-edges; a perturbed hidden row of the edge MLP reaches that edge's input
     gradient and both endpoints'), residual connections keep what is tainted tainted;
  4. ``close_or_traced`` fails on ANY out-of-tolerance element outside the tainted rows.

An indexing error (a wrong ``dst`` in one graph of a batch) puts errors into rows no ambiguous activation reaches and is
rejected: ``tests/test_gpu_dmplayer.py::test_flip_comparator_rejects_an_indexing_error``."""
import numpy as np
import torch as th

TAU = 1e-5          # |pre-activation| <= TAU * max(1, |pre|max): "within rounding of the kink"
LOOSE = 0.25       # bound on a traced element's error, relative to the reference's largest value (a flipped ReLU derivative moves one term of the sum by its full size)


def ambiguous_rows(pre, tau=TAU):
    """bool [rows]: rows of a hidden pre-activation tensor with at least one element within rounding of zero."""
    pre = pre.detach().double()
    if pre.numel() == 0:
        return th.zeros(pre.shape[0], dtype=th.bool, device=pre.device)
    scale = max(1.0, float(pre.abs().max()))
    # exact zeros are not ambiguous: they are structural (a gated-out row: zero inputs, zero bias), exact in the product
    # too, and both sides take the same branch at 0 (the negative-side slope)
    return ((pre.abs() <= tau * scale) & (pre != 0)).any(dim=1)


def count_ambiguous(probes, tau=TAU):
    n = 0
    for _, pre in probes:
        if pre.numel():
            n += int(((pre.detach().double().abs() <= tau * max(1.0, float(pre.abs().max()))) & (pre != 0)).sum())
    return n


def taint(src, dst, probes, num_nodes, tau=TAU, node_site="nmlp", edge_site="emlp"):
    """Rows of the input gradients (dX [N], dZ [E]) that an ambiguous activation of an L-layer rep-net can reach.
    ``probes``: ``dmp_oracle.PROBE`` after ONE forward of the stack (per layer: the node MLP's hidden pre-activation,
    then the edge MLP's).  Returns (node_rows, edge_rows) bool tensors on the CPU."""
    src, dst = th.as_tensor(src).cpu().long(), th.as_tensor(dst).cpu().long()
    E = src.numel()
    node_pre = [p for s, p in probes if s == node_site]
    edge_pre = [p for s, p in probes if s == edge_site]
    assert len(node_pre) == len(edge_pre) and len(node_pre) >= 1, [s for s, _ in probes]
    tn = th.zeros(num_nodes, dtype=th.bool)
    te = th.zeros(E, dtype=th.bool)
    for pn_, pe_ in zip(reversed(node_pre), reversed(edge_pre)):
        pn = tn | ambiguous_rows(pn_, tau).cpu()
        pe = te | ambiguous_rows(pe_, tau).cpu()
        new_n = pn.clone()
        new_n[src[pe]] = True
        new_n[dst[pe]] = True
        new_e = pe | pn[dst]
        tn, te = new_n, new_e
    return tn, te


def close_or_traced(got, ref, tol, tainted_rows, what, loose=LOOSE):
    """``|got - ref| <= tol * max(1, |ref|max)`` everywhere, except in ``tainted_rows`` (bool [rows] or None), where
    ``loose`` applies.  Returns True when some element needed the exception."""
    g = got.detach().double().cpu()
    r = (ref if isinstance(ref, th.Tensor) else th.from_numpy(np.asarray(ref))).detach().double().cpu()
    assert g.shape == r.shape, (what, g.shape, r.shape)
    if r.numel() == 0:
        return False
    scale = max(1.0, float(r.abs().max()))
    err = (g - r).abs()
    bad = err > tol * scale
    if not bool(bad.any()):
        return False
    bad_rows = bad.view(bad.shape[0], -1).any(dim=1)
    assert tainted_rows is not None, "%s: max err %g (scale %g) and no activation near its kink to explain it" % (what, float(err.max()), scale)
    stray = bad_rows & ~tainted_rows.cpu()
    assert not bool(stray.any()), \
        "%s: %d rows outside the tolerance (max err %g, scale %g) that no ambiguous activation reaches, e.g. row %d" % (
            what, int(stray.sum()), float(err[stray].max()), scale, int(stray.nonzero()[0]))
    assert float(err.max()) <= loose * scale, "%s: max err %g (scale %g) even for a flipped activation" % (what, float(err.max()), scale)
    return True
