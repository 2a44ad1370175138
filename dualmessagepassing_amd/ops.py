"""torch.autograd wrappers around the C ABI of libdmp_hip.so.

Each Function is one HIP kernel forward and one (or two) HIP kernels backward;
torch is only used for memory, streams and the autograd tape.  There is no CPU
implementation here -- CPU tensors raise (``_lib.require_gpu``).

Math (reference SubgraphCountingMatching/models/dmpnn.py:111-156):
  seg_sum        A[v]      = sum_{e: dst(e)=v} w_e M[e]                  (fn.sum, dmpnn.py:92)
  seg_sum2       S[v]      = [s0 * sum_{flag=0} w_e M[e] | s1 * sum_{flag=1} w_e M[e]]
  gather_rows    out[e]    = w_e X[idx[e]]                               (edges.src/dst[...])
  edge_combine   Y[e]      = G[e,:H] + coef[dst e] G[e,H:] + b + (+-) gathered P rows
"""
import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import check, ptr, stream_ptr


def _mat(t, min_cols=None):
    """fp32 2-D tensor with unit inner stride; returns (tensor, leading dim)."""
    if t.dtype != torch.float32:
        raise _lib.DmpError("feature matrices must be float32, got %s" % t.dtype)
    if t.dim() != 2:
        raise _lib.DmpError("feature matrices must be 2-D, got shape %s" % (tuple(t.shape),))
    if t.size(0) <= 1:
        t = t.contiguous()
        return t, max(t.size(1), 1)
    if t.stride(1) != 1 or t.stride(0) < t.size(1):
        t = t.contiguous()
    return t, t.stride(0)


def _vec(t, dtype):
    if t is None:
        return None
    if t.dtype != dtype:
        raise _lib.DmpError("expected %s tensor, got %s" % (dtype, t.dtype))
    return t.contiguous()


# ----------------------------------------------------------------------------- raw launches
TILE_MAX_ROWS = 512     # kTileRows of csrc/dmp_agg.hip::seg_sum_tiled
# Experiment switch, off: the LDS-staged incidence scatter-add reads every edge row once (no re-reads) and gives the plain
# kernel's bits, but measured 91.5 us against the plain kernel's 79-80 us at bench.py's shape (two 72 KB workgroups per
# CU: the load -> barrier -> sums structure is exposed where the plain kernel keeps 32 waves per CU in flight); DESIGN.md §8.
USE_TILED_SEG_SUM = False
import os as _os
USE_HIP_BATCHNORM = True   # training-mode BatchNorm1d of the MLPs on csrc/dmp_bn.hip


def graph_tiling(node_off, edge_off, Ba, max_edges_a, Bb=0, max_edges_b=None):
    """Tiling of a block-diagonal batch for ``dmp_seg_sum2_tiled``: ``(node_off, edge_off, Ba, Bb, ka, kb)`` with ka / kb
    whole graphs per tile such that no tile exceeds TILE_MAX_ROWS edge rows, or None when a graph alone is larger (or the
    per-graph maxima are unknown): the plain kernel then."""
    if max_edges_a is None or (Bb and max_edges_b is None):
        return None
    if max(int(max_edges_a), int(max_edges_b or 0)) > TILE_MAX_ROWS:
        return None
    ka = max(1, TILE_MAX_ROWS // max(int(max_edges_a), 1))
    kb = max(1, TILE_MAX_ROWS // max(int(max_edges_b or 1), 1))
    return (node_off, edge_off, int(Ba), int(Bb), ka, kb)


# The layer backward's scatter-add of dPre into both endpoint rows as ONE pass over the edge rows with the sums of a graph
# tile in registers (csrc/dmp_segacc.hip): on by default where a block-diagonal batch says how its graphs tile.  Off (a module
# attribute, for tests): dmp_seg_sum2 over the incidence CSR instead (same bits: both sum in ascending eid).
USE_GRAPH_SEG_SUM = True
GRAPH_ACC_NODES = 64       # dmp_seg_sum2_graphs_max_nodes(): node rows of a tile


def graph_node_tiling(node_off, edge_off, Ba, max_nodes_a, Bb=0, max_nodes_b=None):
    """Tiling of a block-diagonal batch for ``dmp_seg_sum2_graphs``: ``(node_off, edge_off, Ba, Bb, ka, kb)`` with ka / kb
    whole graphs per tile such that no tile has more than GRAPH_ACC_NODES nodes, or None when a graph
    alone is larger (or the per-graph maxima are unknown): ``dmp_seg_sum2`` over the incidence CSR then."""
    if max_nodes_a is None or (Bb and max_nodes_b is None):
        return None
    if max(int(max_nodes_a), int(max_nodes_b or 0)) > GRAPH_ACC_NODES:
        return None
    ka = max(1, GRAPH_ACC_NODES // max(int(max_nodes_a), 1))
    kb = max(1, GRAPH_ACC_NODES // max(int(max_nodes_b or 1), 1))
    return (node_off, edge_off, int(Ba), int(Bb), ka, kb)


def graph_seg_ok(index, M, H, out=None):
    """``endpoint_sums`` can take the one-pass kernel for this index / operand."""
    return (USE_GRAPH_SEG_SUM and getattr(index, "node_tiling", None) is not None and H in (64, 128) and M.is_cuda
            and M.dtype == torch.float32 and M.dim() == 2 and M.stride(1) == 1 and M.stride(0) % 4 == 0 and M.data_ptr() % 16 == 0
            and M.size(0) == index.num_edges and index.num_edges > 0
            and (out is None or (out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0)))


def endpoint_sums(M, index, out=None, mask=None, gate=None, nodes=None):
    """``[sum_{e: a_e = v} M[e] | -sum_{e: b_e = v} M[e]]`` ([N, 2H]; a_e = is_reversed ? src : dst, b_e the other endpoint):
    the gradient of the gathered node projections of the layer's edge pre-activation (dmpnn.py:111-127), every sum in
    ascending eid.  One pass over the edge rows where the batch tiles by graphs (``dmp_seg_sum2_graphs``), else
    ``dmp_seg_sum2`` over the incidence CSR (every row read twice).  ``mask`` (``fused.gate_row_mask``): zero bits mark rows
    of ``M`` that are all zeros (the rows a 0 / 1 edge gate wiped): the one-pass kernel does not fetch them; ``gate``: the same
    as [E] floats, for the segment sum over the incidence CSR (a row of weight 0 is not fetched there).
    ``nodes`` = ``(node mask words, (sel_a, sel_b) of GraphIndex.edge_select_nodes)`` (with ``mask``, one-pass kernel only):
    the rows of the result for nodes whose mask bit is clear are DEAD -- not summed, not stored."""
    H, N = M.size(1), index.num_nodes
    if nodes is not None and not (mask is not None and graph_seg_ok(index, M, H, out)):
        raise _lib.DmpError("endpoint_sums: a node mask needs the one-pass kernel and a row mask")
    if graph_seg_ok(index, M, H, out):
        lib = _lib.load()
        node_off, edge_off, Ba, Bb, ka, kb = index.node_tiling
        sel_a, sel_b = index.endpoint_select() if nodes is None else nodes[1]
        _lib.require_gpu(M, sel_a, sel_b, node_off, edge_off)
        if out is None:
            out = torch.empty((N, 2 * H), dtype=torch.float32, device=M.device)
        elif out.shape != (N, 2 * H) or out.dtype != torch.float32:
            raise _lib.DmpError("endpoint_sums: bad out tensor")
        E = M.size(0)
        nbytes = 4 * H * E + 8 * H * N + 8 * E + 16 * (Ba + Bb + 1)
        with _lib.timed("seg_sum2_graphs[H=%d,rows=%d,E=%d]", (H, N, E), nbytes):
            if mask is not None:
                check(lib.dmp_seg_sum2_graphs_masked(ptr(M), M.stride(0), ptr(sel_a), ptr(sel_b), ptr(node_off), ptr(edge_off), Ba, Bb,
                                                     ka, kb, H, 1.0, -1.0, ptr(out), out.stride(0) if N > 1 else 2 * H, ptr(mask), E,
                                                     None if nodes is None else ptr(nodes[0]), N, stream_ptr()),
                      "dmp_seg_sum2_graphs_masked")
            else:
                check(lib.dmp_seg_sum2_graphs(ptr(M), M.stride(0), ptr(sel_a), ptr(sel_b), ptr(node_off), ptr(edge_off), Ba, Bb, ka, kb,
                                              H, 1.0, -1.0, ptr(out), out.stride(0) if N > 1 else 2 * H, stream_ptr()),
                      "dmp_seg_sum2_graphs")
        return out
    inc_ptr, inc_ent = index.incidence()
    if gate is not None:
        return seg_sum_raw(M, inc_ptr, inc_ent, N, gate.reshape(-1), True, 1.0, -1.0, rows_shared=2, out=out)
    return seg_sum_raw(M, inc_ptr, inc_ent, N, None, True, 1.0, -1.0, rows_shared=2, out=out, tiling=index.tiling)


def seg_sum_raw(M, rowptr, ent, num_nodes, edge_w=None, split=False, s0=1.0, s1=1.0, rows_shared=True, out=None, tiling=None,
                rows=None, ptr_by_pos=False, tag=None, incidence=False):
    """``out``: optional destination (e.g. a column slice of a wider matrix: unit inner stride, any row stride).
    ``tiling`` (``graph_tiling``): the split sum over a CSR whose rows share source rows runs per graph tile from LDS.
    ``rows`` = ``(list, count)`` (``fused.kept_rows``; split sums without weights, with ``out``): only the destination rows of
    the list are summed and written (``dmp_seg_sum2_rows``); ``ptr_by_pos``: the CSR has one row per list POSITION
    (``dmp_incidence_keep``).  ``tag``: the HIP-event timer's record name instead of ``seg_sum2``."""
    lib = _lib.load()
    _lib.require_gpu(M, rowptr, ent, edge_w)
    M, ldm = _mat(M)
    H = M.size(1)
    width = 2 * H if split else H
    if out is None:
        out = torch.empty((num_nodes, width), dtype=torch.float32, device=M.device)
    elif out.shape != (num_nodes, width) or out.dtype != torch.float32 or (num_nodes > 1 and out.stride(1) != 1):
        raise _lib.DmpError("seg_sum: bad out tensor")
    ldo = out.stride(0) if num_nodes > 1 else max(out.size(1), 1)
    ew = _vec(edge_w, torch.float32)
    nent = ent.numel()
    # algorithmic bytes: every source row once, every output row once, the CSR arrays once
    src_rows = min(nent, M.size(0))  # incidence CSRs list every source row twice: distinct rows count once
    nbytes = 4 * H * src_rows + 4 * out.size(1) * num_nodes + 4 * nent + 4 * (num_nodes + 1) + (4 * nent if ew is not None else 0)
    if split and tiling is not None and USE_TILED_SEG_SUM and ew is None and H % 32 == 0 and ldm % 4 == 0 and ldo % 4 == 0 \
            and M.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and num_nodes > 0:
        node_off, edge_off, Ba, Bb, ka, kb = tiling
        if True:
            with _lib.timed("seg_sum2_tiled[H=%d,rows=%d,ent=%d]", (H, num_nodes, nent), nbytes):
                check(lib.dmp_seg_sum2_tiled(ptr(M), ldm, ptr(rowptr), ptr(ent), ptr(node_off), ptr(edge_off), Ba, Bb, ka, kb, H,
                                             s0, s1, ptr(out), ldo, stream_ptr()), "dmp_seg_sum2_tiled")
            return out
    if rows is not None:
        if not split or ew is not None:
            raise _lib.DmpError("seg_sum: a row list goes with the split sum without weights")
        with _lib.timed((tag or "seg_sum2") + "[H=%d,rows=%d,ent=%d]", (H, num_nodes, nent), nbytes):
            check(lib.dmp_seg_sum2_rows(ptr(M), ldm, ptr(rowptr), ptr(ent), ptr(rows[0]), ptr(rows[1]), int(bool(ptr_by_pos)), int(bool(incidence)),
                                        num_nodes, H, s0, s1, ptr(out), ldo, stream_ptr()), "dmp_seg_sum2_rows")
        return out
    if split:
        with _lib.timed("seg_sum2[H=%d,rows=%d,ent=%d]", (H, num_nodes, nent), nbytes):
            check(lib.dmp_seg_sum2(ptr(M), ldm, ptr(rowptr), ptr(ent), ptr(ew), num_nodes, H, s0, s1,
                                   ptr(out), ldo, int(rows_shared), stream_ptr()), "dmp_seg_sum2")
    else:
        with _lib.timed("seg_sum[H=%d,rows=%d,ent=%d]", (H, num_nodes, nent), nbytes):
            check(lib.dmp_seg_sum(ptr(M), ldm, ptr(rowptr), ptr(ent), ptr(ew), num_nodes, H,
                                  ptr(out), ldo, int(rows_shared), stream_ptr()), "dmp_seg_sum")
    return out


def gather_rows_raw(X, idx32, edge_w=None):
    lib = _lib.load()
    _lib.require_gpu(X, idx32, edge_w)
    X, ldx = _mat(X)
    E, H = idx32.numel(), X.size(1)
    out = torch.empty((E, H), dtype=torch.float32, device=X.device)
    ew = _vec(edge_w, torch.float32)
    with _lib.timed("gather_rows[H=%d,E=%d]", (H, E), 4 * H * (E + X.size(0)) + 4 * E):
        check(lib.dmp_gather_rows(ptr(X), ldx, ptr(idx32), ptr(ew), E, H, ptr(out), H, stream_ptr()),
              "dmp_gather_rows")
    return out


def gather_select_raw(D, dst32, rev8, H, edge_w=None, s0=1.0, s1=1.0, base=None):
    """``out[e] = base[e] + w_e * (rev[e] ? s1 D[dst e, H:] : s0 D[dst e, :H])`` (base optional)."""
    lib = _lib.load()
    _lib.require_gpu(D, dst32, rev8, edge_w, base)
    D, ldd = _mat(D)
    E = dst32.numel()
    out = torch.empty((E, H), dtype=torch.float32, device=D.device)
    ew = _vec(edge_w, torch.float32)
    ldb = 0
    if base is not None:
        base, ldb = _mat(base)
    nbytes = 4 * H * (E + 2 * D.size(0)) + 5 * E + (4 * H * E if base is not None else 0)
    with _lib.timed("gather_select%s[H=%d,E=%d]", ("+base" if base is not None else "", H, E), nbytes):
        check(lib.dmp_gather_select(ptr(D), ldd, ptr(dst32), ptr(rev8), ptr(ew), ptr(base), ldb, E, H, s0, s1,
                                    ptr(out), H, stream_ptr()), "dmp_gather_select")
    return out


# ----------------------------------------------------------------------------- autograd ops
class _SegSum(torch.autograd.Function):
    """fn.sum by destination; backward = row gather (dmpnn.py:92,163)."""

    @staticmethod
    def forward(ctx, M, index, edge_w):
        ctx.index = index
        ctx.edge_w = edge_w
        return seg_sum_raw(M, index.in_ptr, index.in_ent, index.num_nodes, edge_w)

    @staticmethod
    @once_differentiable
    def backward(ctx, dA):
        return gather_rows_raw(dA, ctx.index.dst32, ctx.edge_w), None, None


class _SegSumBySrc(torch.autograd.Function):
    """Sum of per-edge rows by *source* node; backward = gather by src."""

    @staticmethod
    def forward(ctx, M, index):
        ctx.index = index
        return seg_sum_raw(M, index.out_ptr, index.out_ent, index.num_nodes)

    @staticmethod
    @once_differentiable
    def backward(ctx, dA):
        return gather_rows_raw(dA, ctx.index.src32), None


class _SegSum2(torch.autograd.Function):
    """Flag-split segment sum over the in-CSR; backward = gather_select."""

    @staticmethod
    def forward(ctx, M, index, edge_w, s0, s1):
        ctx.index, ctx.edge_w, ctx.s0, ctx.s1, ctx.H = index, edge_w, s0, s1, M.size(1)
        return seg_sum_raw(M, index.in_ptr, index.in_ent, index.num_nodes, edge_w, True, s0, s1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dS):
        ix = ctx.index
        dM = gather_select_raw(dS, ix.dst32, ix.rev8, ctx.H, ctx.edge_w, ctx.s0, ctx.s1)
        return dM, None, None, None, None


class _GatherRows(torch.autograd.Function):
    """edges.src[k] (by_src=True) or edges.dst[k]; backward = segment sum."""

    @staticmethod
    def forward(ctx, X, index, by_src):
        ctx.index, ctx.by_src = index, by_src
        return gather_rows_raw(X, index.src32 if by_src else index.dst32)

    @staticmethod
    @once_differentiable
    def backward(ctx, dO):
        ix = ctx.index
        if ctx.by_src:
            return seg_sum_raw(dO, ix.out_ptr, ix.out_ent, ix.num_nodes), None, None
        return seg_sum_raw(dO, ix.in_ptr, ix.in_ent, ix.num_nodes), None, None


_IMMUTABLE = {}     # storage address -> a tensor of that storage (kept alive: the address stays unique)


def mark_immutable(*tensors):
    """Declare index tensors (node ids, relation types, triplets of a FIXED graph) as never refilled in place.  The index
    structures derived from such a tensor (``take_rows``'s CSR, ``PoolIndex.from_keys``) are memoised on the tensor's identity
    and version; while a step is being RECORDED (``dp.StepGraph``) a memo hit leaves the index build out of the recording,
    which is only right if the tensor's contents cannot change between replays -- a replay runs no Python, so an in-place
    refill of a closed-over buffer would go unnoticed.  Inside a recording the memos are therefore consulted for marked
    tensors only; everything else is rebuilt (and recorded).  The mark is on the STORAGE: views of a marked tensor
    (``ids.squeeze()``) are marked too.  Returns its argument(s)."""
    for t in tensors:
        if t is not None:
            _IMMUTABLE[t.untyped_storage().data_ptr()] = t
    return tensors[0] if len(tensors) == 1 else tensors


def is_immutable(t):
    return t.untyped_storage().data_ptr() in _IMMUTABLE


def _memo_usable(key_tensor):
    """A memo entry keyed on ``key_tensor`` may be used: always outside a stream capture, inside one only for tensors the
    caller has marked immutable."""
    return not (key_tensor.is_cuda and torch.cuda.is_current_stream_capturing()) or is_immutable(key_tensor)


class _TakeRows(torch.autograd.Function):
    """``X[idx]`` for an arbitrary int64 index vector: forward = row gather, backward = fixed-order
    segment sum over a CSR of the index values (torch's advanced-indexing backward serialises on
    repeated indices: 1.9 ms for 21,716 lookups into 2,708 rows in the UNC score head)."""

    _memo = []     # (ident of the key tensor, (rowptr, ent, idx32), key tensor): see PoolIndex.from_keys

    @staticmethod
    def forward(ctx, X, idx, key=None, tag=None):
        lib = _lib.load()
        _lib.require_gpu(X, idx)
        M, N = idx.numel(), X.size(0)
        # The CSR of the index values only depends on ``idx``: a caller that looks up the SAME index tensor step after step
        # (UNC: the node ids of its one graph, the triplets of a full-graph step) gets it from a small memo keyed on the
        # tensor's identity and version (``key``: the tensor ``idx`` was derived from, when ``idx`` itself is a temporary).
        k = key if key is not None else idx
        # (``tag``: how ``idx`` was derived from ``key`` -- two derivations of equal length from one key must not share an entry)
        ident = (k.data_ptr(), k._version, int(k.numel()), tuple(k.stride()), str(k.device), str(k.dtype), int(M), int(N), tag)
        memo, hit = _TakeRows._memo, None
        for i, m in enumerate(memo if _memo_usable(k) else ()):
            if m[0] == ident:
                if i:
                    memo.insert(0, memo.pop(i))
                hit = m[1]
                _lib.pin_for_capture(*hit)
                break
        if hit is None:
            idx = idx.view(-1).to(torch.int64).contiguous()
            i32 = dict(dtype=torch.int32, device=X.device)
            rowptr, ent = torch.empty(N + 1, **i32), torch.empty(M, **i32)
            idx32, deg = torch.empty(M, **i32), torch.empty(N, dtype=torch.int64, device=X.device)
            status = torch.empty(1, **i32)
            ws = torch.empty(lib.dmp_csr_workspace_words(N, M), **i32)
            check(lib.dmp_csr_build(ptr(idx), None, M, N, ptr(rowptr), ptr(ent), ptr(idx32), ptr(deg), ptr(status),
                                    ptr(ws), stream_ptr()), "dmp_csr_build(index)")
            if _lib.VALIDATE and int(status.item()) != 0:
                raise _lib.DmpError("take_rows: index outside [0, %d)" % N)
            hit = (rowptr, ent, idx32)
            if not torch.cuda.is_current_stream_capturing():    # arrays made inside a recording belong to its memory pool
                memo.insert(0, (ident, hit, k))                 # the key tensor is kept alive: its address stays unique
                del memo[4:]
        ctx.rowptr, ctx.ent = hit[0], hit[1]
        ctx.N = N
        return gather_rows_raw(X.contiguous(), hit[2])

    @staticmethod
    @once_differentiable
    def backward(ctx, dO):
        return seg_sum_raw(dO.contiguous(), ctx.rowptr, ctx.ent, ctx.N, rows_shared=False), None, None, None


def take_rows(X, idx, key=None, tag=None):
    """Differentiable ``X[idx]`` (rows) on the gather / segment-sum kernels.  ``key``: the tensor ``idx`` was computed
    from, if ``idx`` is a temporary (the index's CSR is memoised on the key's identity and version); ``tag`` (hashable): which
    derivation of ``key`` this is, when a caller derives more than one index of the same length from it."""
    return _TakeRows.apply(X, idx, key, tag)


class _TakeRowsSmallTable(torch.autograd.Function):
    """``W[idx]`` for a table with few rows and very many lookups per row (relation embeddings):
    backward = keyed two-level segment sum (``PoolIndex.from_keys``), fixed order."""

    @staticmethod
    def forward(ctx, W, idx):
        _lib.require_gpu(W, idx)
        idx = idx.view(-1).to(torch.int64)
        ctx.pool = PoolIndex.from_keys(idx, W.size(0))
        return gather_rows_raw(W.contiguous(), idx.to(torch.int32))

    @staticmethod
    @once_differentiable
    def backward(ctx, dO):
        p = ctx.pool
        part = seg_sum_raw(dO.contiguous(), p.vptr, p.vent, p.num_chunks, None, False, rows_shared=False)
        return seg_sum_raw(part, p.gptr, p.gent, p.num_graphs, None, False, rows_shared=False), None


def take_rows_small_table(W, idx):
    return _TakeRowsSmallTable.apply(W, idx)


class _EdgeCombine(torch.autograd.Function):
    """DMPLayer edge pre-activation, one kernel (dmpnn.py:112,120,124,142-151)."""

    @staticmethod
    def forward(ctx, G, P, bias, coef, index):
        lib = _lib.load()
        _lib.require_gpu(G, P, bias, coef)
        G, ldg = _mat(G)
        P, ldp = _mat(P)
        E, H = G.size(0), G.size(1) // 2
        if G.size(1) != 2 * H or P.size(1) != 2 * H or E != index.num_edges:
            raise _lib.DmpError("edge_combine: G must be [E,2H], P [N,2H]")
        Y = torch.empty((E, H), dtype=torch.float32, device=G.device)
        b = _vec(bias, torch.float32)
        with _lib.timed("edge_combine[H=%d,E=%d]", (H, E), 4 * H * (3 * E + 2 * P.size(0)) + 9 * E + 4 * P.size(0)):
            check(lib.dmp_edge_combine(ptr(G), ldg, ptr(P), ldp, ptr(coef), ptr(b), ptr(index.src32),
                                       ptr(index.dst32), ptr(index.rev8), E, H, 0, 0.0, ptr(Y), H, stream_ptr()),
                  "dmp_edge_combine")
        ctx.index, ctx.coef, ctx.H = index, coef, H
        ctx.has_bias = bias is not None
        return Y

    @staticmethod
    @once_differentiable
    def backward(ctx, dY):
        lib = _lib.load()
        ix, H = ctx.index, ctx.H
        dY, ldy = _mat(dY)
        E = dY.size(0)
        dG = dP = db = None
        if ctx.needs_input_grad[0]:
            dG = torch.empty((E, 2 * H), dtype=torch.float32, device=dY.device)
            with _lib.timed("edge_combine_bwd_g[H=%d,E=%d]", (H, E), 4 * H * 3 * E + 4 * E + 4 * ix.num_nodes):
                check(lib.dmp_edge_combine_bwd_g(ptr(dY), ldy, ptr(ctx.coef), ptr(ix.dst32), E, H, ptr(dG),
                                                 2 * H, stream_ptr()), "dmp_edge_combine_bwd_g")
        if ctx.needs_input_grad[1]:
            dP = endpoint_sums(dY, ix)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dY.sum(0)
        return dG, dP, db, None, None


class _CompGCNAgg(torch.autograd.Function):
    """CompGCN message + fn.sum with the W_in/W_out products moved behind the sum
    (compgcn.py:213-238,271)."""

    @staticmethod
    def forward(ctx, X, Z, norm, index, comp):
        lib = _lib.load()
        _lib.require_gpu(X, Z, norm)
        X, ldx = _mat(X)
        Z, ldz = _mat(Z)
        H = X.size(1)
        out = torch.empty((index.num_nodes, 2 * H), dtype=torch.float32, device=X.device)
        nrm = _vec(norm, torch.float32)
        check(lib.dmp_compgcn_agg(ptr(X), ldx, ptr(Z), ldz, ptr(index.in_ptr), ptr(index.in_ent),
                                  ptr(index.src32), ptr(nrm), index.num_nodes, H, comp, ptr(out), 2 * H,
                                  stream_ptr()), "dmp_compgcn_agg")
        ctx.save_for_backward(X, Z)
        ctx.index, ctx.norm, ctx.comp = index, nrm, comp
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dS):
        lib = _lib.load()
        X, Z = ctx.saved_tensors
        ix = ctx.index
        X, ldx = _mat(X)
        Z, ldz = _mat(Z)
        dS, ldd = _mat(dS)
        E, H = ix.num_edges, X.size(1)
        dZ = torch.empty((E, H), dtype=torch.float32, device=X.device)
        dXe = torch.empty((E, H), dtype=torch.float32, device=X.device)
        check(lib.dmp_compgcn_agg_bwd(ptr(dS), ldd, ptr(X), ldx, ptr(Z), ldz, ptr(ix.src32), ptr(ix.dst32),
                                      ptr(ix.rev8), ptr(ctx.norm), E, H, ctx.comp, ptr(dZ), H, ptr(dXe), H,
                                      stream_ptr()), "dmp_compgcn_agg_bwd")
        dX = seg_sum_raw(dXe, ix.out_ptr, ix.out_ent, ix.num_nodes)
        return dX, dZ, None, None, None


# ----------------------------------------------------------------------------- public functional API
def seg_sum(M, index, edge_w=None):
    """``out[v] = sum_{e: dst(e)=v} w_e M[e]`` -- DGL ``fn.sum`` by destination."""
    return _SegSum.apply(M, index, edge_w)


def seg_sum_by_src(M, index):
    return _SegSumBySrc.apply(M, index)


def seg_sum2(M, index, edge_w=None, s0=-1.0, s1=1.0):
    """``[s0 * sum_{non-reversed in-edges} | s1 * sum_{reversed in-edges}]`` -> [N, 2H]."""
    return _SegSum2.apply(M, index, edge_w, float(s0), float(s1))


def gather_src(X, index):
    return _GatherRows.apply(X, index, True)


def gather_dst(X, index):
    return _GatherRows.apply(X, index, False)


def edge_combine(G, P, bias, coef, index):
    return _EdgeCombine.apply(G, P, bias, coef, index)


COMP_SUB, COMP_MULT, COMP_CMUL = 0, 1, 2      # CMUL: conj(x) * z over interleaved (re, im) pairs (the corr composition in the frequency domain)


def compgcn_agg(X, Z, norm, index, comp):
    return _CompGCNAgg.apply(X, Z, norm, index, int(comp))


# ----------------------------------------------------------------------------- dense projections
# The [rows,H] x [H,H'] products stay on the MFMA units through torch (hipBLASLt).  What is
# ours here is the *shape* of the weight-gradient product: dW = A^T B has K = rows (5e5 edge
# rows) and a 128 x 256 output, which a single GEMM call runs on ~32 workgroups (16-26 TF/s
# measured); cutting K into 4096-row slices as one batched GEMM plus a tiny reduction fills
# the chip (115-126 TF/s measured on MI355X, scripts/mb_gemm.py).
_SPLITK_ROWS = 4096


def atb_splitk(a, b):
    """``a.T @ b`` for tall-skinny a [R,K], b [R,N]."""
    R = a.size(0)
    if R < 4 * _SPLITK_ROWS:
        return a.t() @ b
    a, b = a.contiguous(), b.contiguous()
    S = R // _SPLITK_ROWS
    main = S * _SPLITK_ROWS
    out = torch.bmm(a[:main].view(S, _SPLITK_ROWS, a.size(1)).transpose(1, 2),
                    b[:main].view(S, _SPLITK_ROWS, b.size(1))).sum(0)
    if main < R:
        out = out + a[main:].t() @ b[main:]
    return out


class _MatmulXW(torch.autograd.Function):
    """``x @ W`` with W in the reference's [in, out] layout (dmpnn.py:33-38)."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        from . import fused
        if (x.is_cuda and x.dtype == torch.float32 and W.dtype == torch.float32 and x.dim() == 2 and W.dim() == 2
                and 1 <= x.size(1) <= fused.SMALLK_MAX and W.size(1) in fused.MFMA_WIDTHS and x.stride(1) == 1 and x.size(0) > 0
                and W.is_contiguous()):
            # narrow inputs (label encodings @ embedding table): the small-K kernel, W in registers -- a library GEMM
            # pays its solution lookup again for every new row count (ragged batches: ~80 us of host time per call)
            return fused.smallk_embed(x, W)
        return x @ W

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = dy @ W.t() if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            from . import fused
            if (x.is_cuda and x.dtype == torch.float32 and dy.dtype == torch.float32 and x.dim() == 2 and dy.dim() == 2
                    and 1 <= x.size(1) <= fused.SMALLK_MAX and dy.size(1) in fused.MFMA_WIDTHS and x.stride(1) == 1 and x.size(0) > 0):
                dW = fused.smallk_atb(x, dy.contiguous(), None)   # narrow inputs (label encodings): one pass over dy
            else:
                dW = atb_splitk(x, dy)
        return dx, dW


class _LinearNN(torch.autograd.Function):
    """``F.linear(x, weight, bias)`` (nn.Linear layout [out, in]) with an optional fused ReLU
    epilogue (hipBLASLt) or an in-place LeakyReLU (``slope`` > 0) and the split-K weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, slope=0.0):
        if relu and bias is not None and slope == 0.0:
            y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        else:
            y = torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()
            if relu:
                y = torch.relu_(y) if slope == 0.0 else torch.nn.functional.leaky_relu_(y, slope)
        ctx.relu, ctx.slope = relu, slope
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        if ctx.relu and ctx.slope == 0.0:
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)
        elif ctx.relu:   # on the saved output: sign(y) == sign(pre-activation) for a positive slope
            dy = torch.ops.aten.leaky_relu_backward(dy, y, ctx.slope, True)
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dw = atb_splitk(dy, x) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if dy.is_cuda and dy.dtype == torch.float32 and dy.dim() == 2 and dy.size(1) % 4 == 0 and dy.is_contiguous() and dy.size(0) > 0:
                from . import fused
                db = fused.colsum(dy)           # fixed-order partial sums: two small launches (the library reduction takes 17 us at [1e4, 256])
            else:
                db = dy.sum(0)
        return dx, dw, db, None, None


def matmul_xw(x, W):
    return _MatmulXW.apply(x, W)


def linear_nn(x, weight, bias=None, relu=False, slope=0.0):
    return _LinearNN.apply(x, weight, bias, bool(relu), float(slope))


class _BatchNormActTrain(torch.autograd.Function):
    """``act(BatchNorm1d(x))`` in training mode (csrc/dmp_bn.hip): statistics, running-average update, normalisation and
    the activation that follows in three small launches; the backward likewise."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, slope, rows_dev=None):
        """``rows_dev`` (int64 [1] on the device, or None): only the first ``rows_dev[0]`` rows exist -- a batch padded to a capacity
        (``unc_harness.SampledStep``): the others stay out of the statistics and come out as zeros, forward and backward."""
        lib = _lib.load()
        _lib.require_gpu(x, rows_dev)
        x = x.contiguous()
        R, C = x.shape
        nb = int(lib.dmp_bn_partial_rows(R, C))
        partial = torch.empty((nb, 2 * C), dtype=torch.float32, device=x.device)
        stats = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        act = slope is not None
        check(lib.dmp_bn_train_fwd_rows(ptr(x), x.stride(0), R, ptr(rows_dev), C, ptr(gamma), ptr(beta), float(eps), float(momentum),
                                        ptr(running_mean), ptr(running_var), int(act), float(slope or 0.0), ptr(partial), ptr(stats), ptr(out),
                                        out.stride(0), stream_ptr()), "dmp_bn_train_fwd")
        ctx.save_for_backward(x, out if act else None, gamma)
        ctx.stats, ctx.partial, ctx.slope, ctx.rows_dev = stats, partial, slope, rows_dev
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var) if t is not None])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        lib = _lib.load()
        x, y, gamma = ctx.saved_tensors
        R, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        act = ctx.slope is not None
        check(lib.dmp_bn_train_bwd_rows(ptr(x), x.stride(0), ptr(y), y.stride(0) if act else 0, ptr(dy), dy.stride(0), R, ptr(ctx.rows_dev), C,
                                        ptr(gamma), int(act), float(ctx.slope or 0.0), ptr(ctx.partial), ptr(ctx.stats), ptr(dx), dx.stride(0),
                                        stream_ptr()), "dmp_bn_train_bwd")
        dgamma = ctx.stats[3 * C:].clone() if gamma is not None else None      # copies: the statistics buffer stays the node's own
        dbeta = ctx.stats[2 * C:3 * C].clone() if gamma is not None else None
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm_act_ok(bn, x):
    """``batch_norm_act`` runs this module on this input: a training-mode ``BatchNorm1d`` with running statistics and a fixed
    momentum over fp32 rows on the GPU, a width the kernel covers (csrc/dmp_bn.hip)."""
    C = bn.num_features
    return (type(bn) is torch.nn.BatchNorm1d and bn.training and bn.track_running_stats and bn.momentum is not None
            and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.size(0) > 1 and x.size(1) == C
            and C % 4 == 0 and C <= 1024 and 256 % (C // 4) == 0 and (bn.weight is None) == (bn.bias is None))


def batch_norm_act(bn, x, slope=None, rows_dev=None):
    """``LeakyReLU(slope)(bn(x))`` (``slope`` None: no activation) for a module / input that ``batch_norm_act_ok`` accepts:
    same values and the same side effects on the module's buffers as the module call.  ``rows_dev``: the number of rows that
    exist, on the device (the rest is padding: out of the statistics, zeros in the result)."""
    with torch.no_grad():
        bn.num_batches_tracked.add_(1)
    return _BatchNormActTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, slope, rows_dev)


def _act_slope(m):
    if type(m) is torch.nn.ReLU:
        return 0.0
    if type(m) is torch.nn.LeakyReLU and 0.0 < m.negative_slope <= 1.0:
        return float(m.negative_slope)
    return None


def apply_mlp(seq, x, rows_dev=None):
    """Run an ``nn.Sequential`` of Linear / BatchNorm / activation modules (nmlp / emlp,
    dmpnn.py:45-60) with the Linear layers on ``linear_nn``, Linear+ReLU pairs fused and training-mode
    BatchNorm1d (+ the activation after it) on ``batch_norm_act``.  ``rows_dev``: x's rows past that device-side count
    are padding (a BatchNorm the kernels cannot take then runs as ``masked_batch_norm``: tensor ops over the real rows)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if isinstance(m, torch.nn.Linear):
            slope = _act_slope(nxt)
            fuse = slope is not None and m.bias is not None
            x = linear_nn(x, m.weight, m.bias, relu=fuse, slope=slope if fuse else 0.0)
            i += 2 if fuse else 1
        elif USE_HIP_BATCHNORM and isinstance(m, torch.nn.BatchNorm1d) and batch_norm_act_ok(m, x):
            slope = _act_slope(nxt)
            x = batch_norm_act(m, x, slope, rows_dev)
            i += 2 if slope is not None else 1
        else:
            if rows_dev is not None and isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training:
                x = masked_batch_norm(m, x, rows_dev)         # a width the kernels do not take (the reference's hid 50): tensor ops
            else:
                x = m(x)
            i += 1
    return x


def masked_batch_norm(bn, x, rows_dev):
    """Training-mode ``BatchNorm1d`` over the first ``rows_dev[0]`` rows of ``x`` with tensor ops (no host sync: it records):
    statistics and running averages from those rows only, zeros in the others -- what ``batch_norm_act(rows_dev=...)`` does
    for the widths the HIP kernels take."""
    R = x.size(0)
    n = rows_dev.to(x.dtype).clamp(min=1.0)
    m = (torch.arange(R, device=x.device) < rows_dev).to(x.dtype).unsqueeze(1)
    mean = (x * m).sum(0) / n
    d = (x - mean) * m
    var = (d * d).sum(0) / n
    y = d / torch.sqrt(var + bn.eps)
    if bn.weight is not None:
        y = y * bn.weight + bn.bias
    with torch.no_grad():
        if bn.track_running_stats and bn.running_mean is not None:
            mom = bn.momentum if bn.momentum is not None else 0.1
            bn.num_batches_tracked.add_(1)
            bn.running_mean.mul_(1 - mom).add_(mom * mean.detach())
            bn.running_var.mul_(1 - mom).add_(mom * (var.detach() * n / (n - 1.0).clamp(min=1.0)))
    return y * m


# ----------------------------------------------------------------------------- per-graph pooling
class PoolIndex:
    """Index for summing the rows (nodes or edges) of every graph of a batch with the segment-sum
    kernel: each graph's contiguous row range is cut into chunks of ``CHUNK`` rows ("virtual rows",
    enough of them to fill the chip), chunk sums are then summed per graph.  ``flag`` (edges:
    is_reversed) splits the sum into [non-flagged | flagged] halves."""

    CHUNK = 64

    @classmethod
    def from_keys(cls, keys, num_keys):
        """Index for summing rows by an arbitrary integer key in [0, num_keys) (e.g. the relation
        type of an edge): rows are visited key by key in ascending row order (stable sort), so the
        sums are bit-stable; same two-level chunking as for the contiguous per-graph ranges."""
        # one-entry memo: a fixed graph's relation types (UNC trains on ONE graph) meet this with the same tensor every step,
        # and the construction is ~25 small launches (sort, scans, searches)
        memo = cls.__dict__.get("_keys_memo")
        if memo is None:
            memo = cls._keys_memo = []
        ident = (keys.data_ptr(), keys._version, int(keys.numel()), tuple(keys.stride()), str(keys.device), str(keys.dtype), int(num_keys))
        for i, m in enumerate(memo if _memo_usable(keys) else ()):
            if m[0] == ident:
                if i:
                    memo.insert(0, memo.pop(i))             # most recently used first
                _lib.pin_for_capture(m[1])                  # a recording that reads the index keeps it alive
                return m[1]
        keys_in = keys
        keys = keys.view(-1).to(torch.int64)
        if int(num_keys) == 1 and keys.is_cuda:
            # one key (a single relation type): every row belongs to it, in row order -- no sort, the contiguous-range build
            out = cls(torch.full((1,), int(keys.numel()), dtype=torch.int64, device=keys.device), num_rows=int(keys.numel()))
            if not torch.cuda.is_current_stream_capturing():
                memo.insert(0, (ident, out, keys_in))
                del memo[4:]
            return out
        skeys, order = torch.sort(keys, stable=True)
        # segment sizes from the sorted keys' boundaries (a bincount serialises its atomics on hot keys)
        marks = torch.searchsorted(skeys, torch.arange(num_keys + 1, device=keys.device))
        out = cls(marks[1:] - marks[:-1], order=order, seg=keys, num_rows=int(keys.numel()))
        if not (keys_in.is_cuda and torch.cuda.is_current_stream_capturing()):   # an index made inside a recording lives in its pool
            memo.insert(0, (ident, out, keys_in))         # the tensor is kept alive: its address stays unique
            del memo[4:]                                    # a few entries: a step meets two or three different key tensors
        return out

    def __init__(self, sizes, flag=None, order=None, seg=None, num_rows=None):
        """``num_rows`` (host int: the total number of rows) makes the construction free of host syncs:
        the chunk table is then sized by the bound ``num_rows // CHUNK + len(sizes)`` and its unused tail
        consists of empty chunks.  ``sizes`` / ``flag`` may be PAIRS of tensors (the pattern graphs, then the target
        graphs of a union pass): on the GPU the whole index is then built by two launches (``dmp_pool_index``)."""
        pieces = sizes if isinstance(sizes, (tuple, list)) else (sizes,)
        if num_rows is not None and order is None and pieces[0].is_cuda:
            self._build_device(pieces, flag, seg, int(num_rows))
            return
        if len(pieces) > 1:
            sizes = torch.cat([p.view(-1) for p in pieces])
            if isinstance(flag, (tuple, list)):
                flag = None if any(f is None for f in flag) else torch.cat([f.view(-1) for f in flag])
        dev = sizes.device
        sizes = sizes.to(torch.int64)
        B = int(sizes.numel())
        off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(sizes, 0, out=off[1:])
        C = self.CHUNK
        nchunk = (sizes + C - 1) // C
        coff = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nchunk, 0, out=coff[1:])
        if num_rows is None:
            R, V = int(off[-1].item()), int(coff[-1].item())
            cgraph = torch.repeat_interleave(torch.arange(B, device=dev), nchunk, output_size=V)
            cstart = off[cgraph] + (torch.arange(V, device=dev) - coff[cgraph]) * C
        else:
            R, V = int(num_rows), int(num_rows) // C + B
            v = torch.arange(V, device=dev)
            cgraph = torch.searchsorted(coff[1:], v, right=True)              # == B for the unused tail
            real = cgraph < B
            cg = cgraph.clamp(max=max(B - 1, 0))
            cstart = torch.where(real, off[cg] + (v - coff[cg]) * C, torch.full_like(v, R))
        vptr = torch.cat([cstart, torch.full((1,), R, dtype=torch.int64, device=dev)])
        self.num_graphs, self.num_rows, self.num_chunks = B, R, V
        self.vptr = vptr.to(torch.int32)
        rows = torch.arange(R, device=dev, dtype=torch.int64) if order is None else order.to(torch.int64)
        ent = rows << 1
        if flag is not None:
            ent = ent | flag.view(-1).to(torch.int64)[rows]
        self.vent = ent.to(torch.int32)
        self.gptr = coff.to(torch.int32)
        self.gent = (torch.arange(V, device=dev, dtype=torch.int64) << 1).to(torch.int32)
        self.sizes = sizes
        self.offsets = off
        self.rows_in_order = order is None                # graph i's rows are rows offsets[i] .. offsets[i + 1]
        self.seg32 = (torch.repeat_interleave(torch.arange(B, device=dev, dtype=torch.int32), sizes, output_size=R)
                      if seg is None else seg.to(torch.int32).contiguous())
        self.flag8 = None if flag is None else flag.view(-1).to(torch.uint8).contiguous()


_POOL_JOB = None


def _pool_job_type():
    global _POOL_JOB
    if _POOL_JOB is None:
        import ctypes
        P, I = ctypes.c_void_p, ctypes.c_int64

        class _Job(ctypes.Structure):
            _fields_ = [("sizes_a", P), ("sizes_b", P), ("Ba", I), ("Bb", I), ("flag_a", P), ("flag_b", P), ("rows_a", I), ("R", I),
                        ("chunk", ctypes.c_int), ("off", P), ("gptr", P), ("vptr", P), ("vent", P), ("gent", P), ("seg", P),
                        ("flag8", P), ("rowmap", P), ("sizes", P)]
        _POOL_JOB = _Job
    return _POOL_JOB


def _pool_prepare(self, pieces, flag, seg, R, job):
    """Allocate this index's arrays and describe its build as one ``dmp_pool_job``."""
    sizes = [p.view(-1).to(torch.int64).contiguous() for p in pieces]
    sa, sb = sizes[0], (sizes[1] if len(sizes) > 1 else None)
    flags = flag if isinstance(flag, (tuple, list)) else (flag,)
    if any(f is None for f in flags):
        flags = (None, None)
    f8 = [None if f is None else (f.contiguous().view(-1).view(torch.uint8) if f.element_size() == 1
                                  else (f.view(-1) != 0).view(torch.uint8)) for f in flags]
    fa, fb = f8[0], (f8[1] if len(f8) > 1 else None)
    _lib.require_gpu(sa, sb, fa, fb)
    dev = sa.device
    Ba, Bb = int(sa.numel()), (int(sb.numel()) if sb is not None else 0)
    B, C = Ba + Bb, self.CHUNK
    V = R // C + B
    rows_a = R if fb is None and sb is None else (int(fa.numel()) if fa is not None else None)
    if rows_a is None:      # two size pieces without flags: the split point is irrelevant
        rows_a = 0
    i32 = dict(dtype=torch.int32, device=dev)
    off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    self.gptr, self.vptr = torch.empty(B + 1, **i32), torch.empty(V + 1, **i32)
    self.vent, self.gent = torch.empty(R, **i32), torch.empty(V, **i32)
    self.seg32 = torch.empty(R, **i32) if seg is None else seg.to(torch.int32).contiguous()
    self.num_graphs, self.num_rows, self.num_chunks = B, R, V
    self.sizes = sa if sb is None else torch.empty(B, dtype=torch.int64, device=dev)       # both pieces back to back
    self.offsets = off
    self.rows_in_order = True
    self.flag8, self._rowmap = None, None
    if fa is not None:
        self.flag8 = fa if fb is None else torch.empty(R, dtype=torch.uint8, device=dev)
        self._rowmap = torch.empty(R, **i32)              # the graph of every row, -1 for flagged rows (fused.pool_rowmap)
    job.sizes_a, job.sizes_b, job.Ba, job.Bb = sa.data_ptr(), ptr(sb), Ba, Bb
    job.flag_a, job.flag_b, job.rows_a, job.R, job.chunk = ptr(fa), ptr(fb), rows_a, R, C
    job.off, job.gptr, job.vptr, job.vent, job.gent = off.data_ptr(), self.gptr.data_ptr(), self.vptr.data_ptr(), ptr(self.vent), ptr(self.gent)
    job.seg = self.seg32.data_ptr() if (seg is None and R > 0) else None
    job.flag8 = self.flag8.data_ptr() if (fb is not None and R > 0) else None
    job.rowmap = self._rowmap.data_ptr() if (self._rowmap is not None and R > 0) else None
    job.sizes = self.sizes.data_ptr() if (sb is not None and B > 0) else None
    self._keep = (sizes, f8)


def _pool_build_device(self, pieces, flag, seg, R):
    J = (_pool_job_type() * 1)()
    _pool_prepare(self, pieces, flag, seg, R, J[0])
    check(_lib.load().dmp_pool_index_jobs(J, 1, stream_ptr()), "dmp_pool_index_jobs")


def pool_indexes(specs):
    """Several ``PoolIndex`` objects from ONE pair of launches (``dmp_pool_index_jobs``): ``specs`` = list of
    ``(sizes, flag, num_rows)`` as ``PoolIndex(sizes, flag, num_rows=...)`` takes them (device tensors)."""
    out = [PoolIndex.__new__(PoolIndex) for _ in specs]
    lib = _lib.load()
    for i in range(0, len(specs), 4):                    # DMP_POOL_MAX_JOBS
        part = specs[i:i + 4]
        J = (_pool_job_type() * len(part))()
        for k, (sizes, flag, rows) in enumerate(part):
            pieces = sizes if isinstance(sizes, (tuple, list)) else (sizes,)
            _pool_prepare(out[i + k], pieces, flag, None, int(rows), J[k])
        check(lib.dmp_pool_index_jobs(J, len(part), stream_ptr()), "dmp_pool_index_jobs")
    return out


PoolIndex._build_device = _pool_build_device


class _SegPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pool):
        ctx.pool, ctx.H = pool, x.size(1)
        split = pool.flag8 is not None
        part = seg_sum_raw(x, pool.vptr, pool.vent, pool.num_chunks, None, split, 1.0, 1.0, rows_shared=False)
        return seg_sum_raw(part, pool.gptr, pool.gent, pool.num_graphs, None, False, rows_shared=False)

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        p = ctx.pool
        if p.flag8 is not None:
            return gather_select_raw(d, p.seg32, p.flag8, ctx.H, None, 1.0, 1.0), None
        return gather_rows_raw(d, p.seg32), None


def seg_pool(x, pool):
    """Per-graph sums of the rows of ``x``: [B, H], or [B, 2H] = [sum over non-flagged | flagged]
    when the PoolIndex carries a flag.  Fixed summation order (bit-stable)."""
    return _SegPool.apply(x, pool)
