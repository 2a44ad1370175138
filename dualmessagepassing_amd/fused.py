"""One DMPLayer (+ the rep-net's gate and residual) as a single autograd node.

Same math as ``dmpnn.dual_message_passing`` + the two 2-layer ReLU MLPs + ``v*gate`` +
residual add (SubgraphCountingMatching/models/dmpnn.py:111-166,245-277), but forward and
backward are orchestrated by hand so that every elementwise step between the GEMMs is one
HIP streaming pass (``csrc/dmp_fused.hip``) and gradient accumulation happens inside the
kernels / GEMM epilogues instead of as separate ``add`` launches:

The first Linear of each MLP follows the message sum with no non-linearity in between
(dmpnn.py:131-136,147-152), so its weight is folded into the projections:
``(Z Wel + c Z Wsd + P_d[v] - P_s[u] + be) W0^T + b0 = Z (Wel W0^T) + c Z (Wsd W0^T) + P'_d[v] - P'_s[u] + b'``
-- one [E,H]x[H,H] product less per layer in forward, input-gradient and weight-gradient
(9 instead of 12 E-row GEMM units), and the pre-activation tensor never exists.

  forward   S=seg_sum2(Z) | XP=X[Wnl W0n^T | Wdst W0e^T | Wsrc W0e^T] | H1n=relu(XP0+S Bn+b'n) | On | Xn=X+gv*On
            G=Z[Wel W0e^T | (Wsrc-Wdst) W0e^T] | H1e=edge_combine_relu(G,XP12,b'e) | Oe | Zn=Z+ge*Oe
  backward  gate*dOut (+colsum) -> GEMMs -> relu_bwd fused with edge_combine's backward (+colsum),
            seg_sum2 over the incidence CSR, dZ = dZn + gather_select(dS), dZ += dG W'^T,
            then the small [H,H] chain back to the individual parameters
  saved     X, Z, S, H1n, H1e  (ONE new [E,H] tensor per layer)

Eligibility (``DMPLayer.fused_ok``): 2-layer MLPs, ReLU, no BatchNorm, bias, no active dropout,
square weights, H % 4 == 0, ``is_reversed`` present, gates without gradient.  Anything else
takes the modular path (same kernels, torch autograd in between).
"""
import threading

import torch
from torch.autograd.function import once_differentiable

from . import _lib, ops
from ._lib import check, ptr, stream_ptr


def _partials(rows, H, dev):
    lib = _lib.load()
    return torch.empty((int(lib.dmp_colsum_partial_rows(rows, H)), H), dtype=torch.float32, device=dev)


MAX_REDUCE_SEGMENTS = 16   # DMP_REDUCE_MAX_SEGMENTS


class _Deferred(threading.local):
    """Per-thread stack of active ``deferred_reductions`` collectors (autograd runs backward on one worker thread
    per device: a job must never land in another thread's collector, whose flush is on another stream)."""

    def __init__(self):
        self.stack = []


_deferred = _Deferred()


class deferred_reductions:
    """Inside the block ``reduce_partials`` only records its job and returns the (not yet written) result
    tensor; the jobs run as ONE launch (``dmp_reduce_partials_multi``) when the block is left.  For results
    that are consumed later anyway -- a layer backward's parameter gradients: 9 reductions -> 1."""

    def __enter__(self):
        self.jobs = []
        _deferred.stack.append(self)
        return self

    def flush(self):
        import ctypes
        lib = _lib.load()
        jobs, self.jobs = self.jobs, []
        for i in range(0, len(jobs), MAX_REDUCE_SEGMENTS):
            chunk = jobs[i:i + MAX_REDUCE_SEGMENTS]
            n = len(chunk)
            P = (ctypes.c_void_p * n)(*[ptr(j[0]) for j in chunk])
            S = (ctypes.c_int64 * n)(*[j[1] for j in chunk])
            L = (ctypes.c_int64 * n)(*[j[2] for j in chunk])
            O = (ctypes.c_void_p * n)(*[ptr(j[3]) for j in chunk])
            check(lib.dmp_reduce_partials_multi(P, S, L, O, n, stream_ptr()), "dmp_reduce_partials_multi")

    def __exit__(self, exc_type, exc, tb):
        _deferred.stack.remove(self)
        if exc_type is None:
            self.flush()
        return False


def reduce_partials(partial, out=None, accumulate=False):
    """``out[l] (+)= sum_s partial[s, l]`` in a fixed order (HIP)."""
    lib = _lib.load()
    S = partial.size(0)
    L = partial.numel() // max(S, 1)
    if out is None:
        out = torch.empty(L, dtype=torch.float32, device=partial.device)
    if _deferred.stack and not accumulate and partial.is_contiguous():
        _deferred.stack[-1].jobs.append((partial, S, L, out))      # keeps ``partial`` alive until the launch
        return out
    check(lib.dmp_reduce_partials(ptr(partial), S, L, ptr(out), int(accumulate), stream_ptr()), "dmp_reduce_partials")
    return out


def gate_residual(prev, upd, gate):
    lib = _lib.load()
    R, H = upd.shape
    out = torch.empty_like(upd)
    with _lib.timed("gate_residual[H=%d,R=%d]", (H, R), 4 * H * R * (3 if prev is not None else 2) + (4 * R if gate is not None else 0)):
        check(lib.dmp_gate_residual(ptr(prev), H, ptr(upd), H, ptr(gate), R, H, ptr(out), H, stream_ptr()),
              "dmp_gate_residual")
    return out


def add_bias_relu_(a, b, bias, slope=0.0):
    """a <- act(a + b + bias) in one pass (``b`` may be a column slice of a wider matrix); act = ReLU, or LeakyReLU
    with negative slope ``slope``."""
    lib = _lib.load()
    R, H = a.shape
    with _lib.timed("add_bias_relu[H=%d,R=%d]", (H, R), 12 * H * R):
        check(lib.dmp_add_bias_relu(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(bias), R, H, slope, ptr(a), a.stride(0),
                                    stream_ptr()), "dmp_add_bias_relu")
    return a


def scale_rows_colsum(d_out, gate):
    """-> (gate (.) d_out  [aliases d_out when gate is None], column sums [H])."""
    lib = _lib.load()
    R, H = d_out.shape
    part = _partials(R, H, d_out.device)
    d_upd = torch.empty_like(d_out) if gate is not None else None
    with _lib.timed("scale_rows_colsum[H=%d,R=%d]", (H, R), 4 * H * R * (2 if gate is not None else 1)):
        check(lib.dmp_scale_rows_colsum(ptr(d_out), H, ptr(gate), R, H, ptr(d_upd), H, ptr(part), stream_ptr()),
              "dmp_scale_rows_colsum")
    return (d_upd if gate is not None else d_out), reduce_partials(part)


def relu_bwd_colsum_(d_h, act, out=None, slope=0.0):
    """d_h <- act > 0 ? d_h : slope * d_h (in place, or into ``out``: e.g. a column slice of a wider matrix);
    returns (result, column sums [H])."""
    lib = _lib.load()
    R, H = d_h.shape
    part = _partials(R, H, d_h.device)
    dst = d_h if out is None else out
    ldo = dst.stride(0) if R > 1 else H
    with _lib.timed("relu_bwd_colsum[H=%d,R=%d]", (H, R), 12 * H * R):
        check(lib.dmp_relu_bwd_colsum(ptr(d_h), H, ptr(act), H, R, H, slope, ptr(dst), ldo, ptr(part), stream_ptr()),
              "dmp_relu_bwd_colsum")
    return dst, reduce_partials(part)


def relu_bwd_gathered_colsum(table, rowmap, gate, act, slope=0.0):
    """``dPre[r] = act[r] > 0 ? u : slope * u`` with ``u = gate[r] * table[rowmap[r]]`` (a zero row where ``rowmap[r] < 0``), and
    the column sums of dPre: the activation backward of a layer whose upstream gradient is one [H] vector per graph."""
    lib = _lib.load()
    R, H = act.shape
    part = _partials(R, H, act.device)
    out = torch.empty((R, H), dtype=torch.float32, device=act.device)
    with _lib.timed("relu_bwd_gathered[H=%d,R=%d]", (H, R), 8 * H * R + 8 * R + 4 * H * table.size(0)):
        check(lib.dmp_relu_bwd_gathered_colsum(ptr(table), table.stride(0), ptr(rowmap), ptr(gate), ptr(act), act.stride(0), R, H,
                                               slope, ptr(out), H, ptr(part), stream_ptr()), "dmp_relu_bwd_gathered_colsum")
    return out, reduce_partials(part)


def pool_relu_bwd(table, rowmap, gate, act, pool, slope=0.0, skip_dead=False):
    """``relu_bwd_gathered_colsum`` and ``pool_rows(act, pool, gate)`` in ONE pass over ``act`` (``dmp_pool_relu_bwd``):
    -> (dPre [R, H], column sums of dPre [H], Q [graphs, 2H] = gated per-graph sums of ``act`` by flag).
    ``skip_dead`` (every consumer of dPre leaves out the rows under a zero of the 0 / 1 ``gate``): the pass walks the kept
    rows' chunk table (``keep_pool_csr``) -- dPre's rows under a zero gate are not stored, nor are their entries walked."""
    lib = _lib.load()
    R, H = act.shape
    V = pool.num_chunks
    nb = int(lib.dmp_pool_relu_bwd_blocks(V, H))
    part = torch.empty((nb, H), dtype=torch.float32, device=act.device)
    kc = keep_pool_csr(pool, gate) if skip_dead else None
    vptr, vent = (pool.vptr, pool.vent) if kc is None else kc
    out = dead_rows_buffer((R, H), act.device) if kc is not None else torch.empty((R, H), dtype=torch.float32, device=act.device)
    qc = torch.empty((V, 2 * H), dtype=torch.float32, device=act.device)
    with _lib.timed("pool_relu_bwd[H=%d,R=%d]", (H, R), 8 * H * R + 12 * R + 8 * H * V):
        check(lib.dmp_pool_relu_bwd(ptr(act), act.stride(0), ptr(table), table.stride(0), ptr(rowmap), ptr(gate), ptr(vptr),
                                    ptr(vent), V, R, H, slope, ptr(out), H, ptr(qc), ptr(part), stream_ptr()), "dmp_pool_relu_bwd")
    Q = ops.seg_sum_raw(qc, pool.gptr, pool.gent, pool.num_graphs, None, False, rows_shared=False)
    return out, reduce_partials(part), Q


USE_KEEP_POOL = True


def keep_pool_csr(pool, gate):
    """``(ptr, ent)``: the chunk table of ``pool`` over the rows a 0 / 1 ``gate`` keeps (``dmp_csr_keep`` on the pooling index's
    chunk CSR), memoised on the gate per index, or None.  A pooled pass with a 0 / 1 row weight is then the PLAIN segment sum
    over 42 % of the entries (full load slots) instead of the weighted one that walks every entry: 46 -> 28 us at bench.py's
    shape; same sums, same order (the skipped rows added zeros)."""
    if gate is None or not USE_KEEP_POOL or not USE_ROW_MASKS or pool.num_chunks == 0 or gate.numel() == 0:
        return None
    owner = _gate_owner(gate)
    if getattr(owner, "_dmp_dense_gate", False) or not getattr(owner, "_dmp_binary", False):
        return None
    memo = getattr(owner, "_dmp_keep_pool", None)
    if memo is None or memo[0] != owner._version:
        memo = (owner._version, {})
        try:
            owner._dmp_keep_pool = memo
        except Exception:
            return None
    hit = memo[1].get(id(pool))
    if hit is not None and hit[0] is pool:
        return hit[1]
    lib = _lib.load()
    V, dev = pool.num_chunks, pool.vptr.device
    nscr = int(lib.dmp_csr_keep_scratch_words(V))
    ws = torch.empty(nscr + V + 1 + pool.vent.numel(), dtype=torch.int32, device=dev)
    row_cnt, kptr, kent = ws[:nscr], ws[nscr:nscr + V + 1], ws[nscr + V + 1:]
    check(lib.dmp_csr_keep(ptr(pool.vptr), ptr(pool.vent), ptr(gate.reshape(-1)), V, pool.vent.numel(), ptr(row_cnt), ptr(kptr), ptr(kent), stream_ptr()),
          "dmp_csr_keep")
    memo[1][id(pool)] = (pool, (kptr, kent))
    return kptr, kent


def pool_rows(x, pool, weight=None):
    """Per-graph sums of the rows of ``x`` over ``pool`` (``ops.PoolIndex``): [G, H], or [G, 2H] = [non-flagged | flagged] when
    the index carries a flag; ``weight`` [rows]: a row scale (a 0 / 1 gate: the pass runs over the kept rows' chunk table,
    ``keep_pool_csr``).  Raw (no autograd): two launches of the segment-sum kernel."""
    split = pool.flag8 is not None
    kc = keep_pool_csr(pool, weight) if (weight is not None and x.size(1) % 4 == 0 and x.is_contiguous()) else None
    if kc is not None:
        part = ops.seg_sum_raw(x, kc[0], kc[1], pool.num_chunks, None, split, 1.0, 1.0, rows_shared=False)
    else:
        part = ops.seg_sum_raw(x, pool.vptr, pool.vent, pool.num_chunks, weight, split, 1.0, 1.0, rows_shared=False)
    return ops.seg_sum_raw(part, pool.gptr, pool.gent, pool.num_graphs, None, False, rows_shared=False)


def pool_weight_sums(pool, weight=None):
    """[G, 1] per-graph sums of a row weight over ``pool`` (``ops.PoolIndex``), [G, 2] = [non-flagged | flagged] when the
    index carries a flag; ``weight`` None: row counts (``dmp_pool_weight_sums``)."""
    lib = _lib.load()
    if not getattr(pool, "rows_in_order", False):
        raise _lib.DmpError("pool_weight_sums: an index over rows in their own order")
    # (a gate's sums are memoised on the gate per index: dmpnn.prefetch_joint_indexes makes them ahead, on the side stream)
    owner = None if weight is None else _gate_owner(weight)
    if owner is not None:
        memo = getattr(owner, "_dmp_pool_wsums", None)
        if memo is not None and memo[0] == owner._version and memo[1] is pool:
            return memo[2]
    halves = 2 if pool.flag8 is not None else 1
    out = torch.empty((pool.num_graphs, halves), dtype=torch.float32, device=pool.offsets.device)
    w = None if weight is None else weight.reshape(-1).contiguous()
    check(lib.dmp_pool_weight_sums(ptr(w), ptr(pool.flag8), ptr(pool.offsets), pool.num_graphs, ptr(out), stream_ptr()),
          "dmp_pool_weight_sums")
    if owner is not None and not weight.requires_grad:
        try:
            owner._dmp_pool_wsums = (owner._version, pool, out)
        except Exception:
            pass
    return out


def pool_rowmap(pool):
    """int32 [rows]: the graph of every row, -1 for flagged rows (they are masked out of the pooled sum)."""
    m = getattr(pool, "_rowmap", None)                  # written by the index build itself (dmp_pool_index_jobs) where it ran
    if m is None:
        m = pool.seg32 if pool.flag8 is None else torch.where(pool.flag8 != 0, torch.full_like(pool.seg32, -1), pool.seg32)
        pool._rowmap = m
    return m


def bwd_g_colsum(d_y, coef, dst32):
    lib = _lib.load()
    E, H = d_y.shape
    part = _partials(E, H, d_y.device)
    d_g = torch.empty((E, 2 * H), dtype=torch.float32, device=d_y.device)
    with _lib.timed("edge_combine_bwd_g[H=%d,E=%d]", (H, E), 12 * H * E + 4 * E + 4 * coef.numel()):
        check(lib.dmp_edge_combine_bwd_g_colsum(ptr(d_y), H, ptr(coef), ptr(dst32), E, H, ptr(d_g), 2 * H, ptr(part),
                                                stream_ptr()), "dmp_edge_combine_bwd_g_colsum")
    return d_g, reduce_partials(part)


def colsum(a):
    lib = _lib.load()
    R, H = a.shape
    part = _partials(R, H, a.device)
    check(lib.dmp_colsum_partials(ptr(a), H, R, H, ptr(part), stream_ptr()), "dmp_colsum_partials")
    return reduce_partials(part)


def atb(a, b):
    """``a.T @ b`` for tall-skinny operands: batched GEMM over 4096-row slices (MFMA through
    hipBLASLt) + the fixed-order HIP reduction of the slice products."""
    R = a.size(0)
    rows = ops._SPLITK_ROWS
    if R < 4 * rows:
        return a.t() @ b
    S = R // rows
    main = S * rows
    tail = 1 if main < R else 0
    part = torch.empty((S + tail, a.size(1), b.size(1)), dtype=a.dtype, device=a.device)
    torch.bmm(a[:main].view(S, rows, a.size(1)).transpose(1, 2), b[:main].view(S, rows, b.size(1)), out=part[:S])
    if tail:
        torch.mm(a[main:].t(), b[main:], out=part[S])        # the ragged last slice: one more partial
    return reduce_partials(part.view(S + tail, -1)).view(a.size(1), b.size(1))


def edge_combine_raw(G, ldg, P, ldp, bias, coef, index, H, relu=False, slope=0.0):
    lib = _lib.load()
    E = index.num_edges
    Y = torch.empty((E, H), dtype=torch.float32, device=G.device)
    with _lib.timed("edge_combine[H=%d,E=%d]", (H, E), 4 * H * (3 * E + 2 * index.num_nodes) + 9 * E + 4 * index.num_nodes):
        check(lib.dmp_edge_combine(ptr(G), ldg, ptr(P), ldp, ptr(coef), ptr(bias), ptr(index.src32), ptr(index.dst32),
                                   ptr(index.rev8), E, H, int(relu), slope, ptr(Y), H, stream_ptr()), "dmp_edge_combine")
    return Y


def relu_bwd_g_colsum(d_h, act, coef, dst32, slope=0.0):
    """-> (dG = [dPre | coef[dst] dPre] with dPre = act>0 ? d_h : slope d_h,  column sums of dPre)."""
    lib = _lib.load()
    E, H = d_h.shape
    part = _partials(E, H, d_h.device)
    d_g = torch.empty((E, 2 * H), dtype=torch.float32, device=d_h.device)
    with _lib.timed("relu_bwd_g_colsum[H=%d,E=%d]", (H, E), 16 * H * E + 4 * E + 4 * coef.numel()):
        check(lib.dmp_relu_bwd_g_colsum(ptr(d_h), H, ptr(act), H, ptr(coef), ptr(dst32), E, H, slope, ptr(d_g), 2 * H,
                                        ptr(part), stream_ptr()), "dmp_relu_bwd_g_colsum")
    return d_g, reduce_partials(part)


USE_MFMA_KERNELS = True  # H == 128 / 64: fused MFMA kernels for the edge chain (csrc/dmp_mfma.hip)
MFMA_WIDTHS = (128, 64)  # one-panel kernels (out_fwd, bwd_h1), class-typed kernels and the weight-gradient kernels


def mfma_ok(index, H):
    """The fused MFMA kernels take H = 128 and address the gathered tables / per-edge arrays with
    32-bit byte offsets: [N, 3H] fp32 projections and E-float arrays must stay below 4 GiB."""
    return USE_MFMA_KERNELS and H == 128 and index.num_nodes * 3 * H * 4 < 2 ** 32 and index.num_edges * 4 < 2 ** 32


def onepanel_ok(H):
    """out_fwd_mfma / bwd_h1_mfma (dPre alone) take this width (128, or the reference's shipped 64)."""
    return USE_MFMA_KERNELS and H in MFMA_WIDTHS


def edge_fwd_mfma(z, Wes, P, ldp, bias, coef, index, slope=0.0):
    """act(z Wes[:, :H] + coef[dst] z Wes[:, H:] + gathers(P) + bias): one fused MFMA kernel (H=128; the class-typed variant also H=64)."""
    lib = _lib.load()
    E, H = z.shape
    out = torch.empty((E, H), dtype=torch.float32, device=z.device)
    Wes = Wes.contiguous()
    sel_a, sel_b, coef_e = index.edge_select(coef)
    with _lib.timed("edge_fwd_mfma[H=%d,E=%d]", (H, E), 4 * H * (2 * E + 2 * index.num_nodes) + 12 * E):
        check(lib.dmp_edge_fwd_fused(ptr(z), H, ptr(Wes), Wes.size(1), ptr(P), ldp, index.num_nodes, ptr(bias),
                                     ptr(sel_a), ptr(sel_b), ptr(coef_e), E, H, slope, ptr(out), H, stream_ptr()),
              "dmp_edge_fwd_fused")
    return out


USE_TYPED_KERNELS = True  # degree-class tiles: one weight panel instead of two in edge_fwd / bwd_z (csrc/dmp_typed.hip)


def typed_ok(index, H):
    """Class-typed kernels: rows are addressed by index (structured buffer descriptors), so the [E, H] arrays may be of
    any size (BASELINE config 4's 1024-pair shard: 8.4 M edge rows, 4.3 GB per array, one pass); what is 32-bit are the
    packed (row << 1 | flag) index entries and the per-edge int32 arrays."""
    return USE_TYPED_KERNELS and onepanel_ok(H) and index.num_nodes < 2 ** 31 and index.num_edges < 2 ** 30


def edge_fwd_typed(z, Wes, P, ldp, bias, coef, index, slope=0.0, dead_gate=None, sel=None):
    """edge_fwd_mfma with W_g = Wes[:, :H] + c_g Wes[:, H:] per degree class: one product instead of two.
    ``dead_gate``: an edge gate all of whose zeros mark DEAD output rows (every consumer multiplies them by that zero and
    skips them): those edges are padding slots of the tile list -- nothing is read, computed or stored for them.
    ``sel`` = ``NodeRows.sel``: the selectors with the nodes whose rows of ``P`` were never written replaced by -1 (read as zeros)."""
    lib = _lib.load()
    E, H = z.shape
    out = dead_rows_buffer((E, H), z.device)
    Wes = Wes.contiguous()
    sel_a, sel_b = (sel[0], sel[1]) if sel is not None else index.edge_select(coef)[:2]
    lt = live_tiles(index, coef, dead_gate) if dead_gate is not None else None
    if lt is not None:
        slot_edge, tile_scale, num_tiles, bound = lt
    else:
        slot_edge, tile_scale, num_tiles, bound = index.class_tiles(coef)
        ms = masked_slots(index, coef, dead_gate) if dead_gate is not None else None
        if ms is not None:
            slot_edge = ms
    with _lib.timed("edge_fwd_typed[H=%d,E=%d]", (H, E), 4 * H * (2 * E + 2 * index.num_nodes) + 12 * E):
        check(lib.dmp_edge_fwd_typed(ptr(z), H, ptr(Wes), Wes.size(1), ptr(P), ldp, index.num_nodes, ptr(bias),
                                     ptr(sel_a), ptr(sel_b), ptr(slot_edge), ptr(tile_scale), ptr(num_tiles), bound,
                                     E, H, slope, ptr(out), H, stream_ptr()), "dmp_edge_fwd_typed")
    return out


import ctypes as _ctypes
import os as _os


def masked_slots(index, coef, gate):
    """The class-tile slot list of ``index`` with the edges under a zero ``gate`` turned into padding (``dmp_mask_slots``), or
    None (no gate, masks off, a gate known to be dense).  For kernels whose streamed rows are ZERO for such edges -- ``dPre``
    of a gated layer: ``atb_typed`` skips the edge, ``bwd_z_typed`` skips the fetch of its ``dPre`` row.  Memoised on the gate
    (shared by the layers of a rep-net) per slot list."""
    if gate is None or not USE_ROW_MASKS:
        return None
    owner = gate._base if gate._base is not None else gate
    if owner.data_ptr() != gate.data_ptr() or owner.numel() != gate.numel():
        owner = gate
    if getattr(owner, "_dmp_dense_gate", False):
        return None
    slot_edge = index.class_tiles(coef)[0]
    hit = getattr(owner, "_dmp_masked_slots", None)
    if hit is not None and hit[0] == owner._version and hit[1] is slot_edge:
        return hit[2]
    lib = _lib.load()
    out = torch.empty_like(slot_edge)
    check(lib.dmp_mask_slots(ptr(slot_edge), slot_edge.numel(), ptr(gate), gate.numel(), ptr(out), stream_ptr()), "dmp_mask_slots")
    try:
        owner._dmp_masked_slots = (owner._version, slot_edge, out)
    except Exception:
        pass
    return out


USE_LIVE_TILES = True      # (module attributes, not environment switches: tests flip them; the documented DMP_* switches are listed in DESIGN.md 2)


def live_tiles(index, coef, gate):
    """The class-tile list of ``index`` over the edges a 0 / 1 ``gate`` keeps (``GraphIndex.class_tiles_gated``: an edge under
    a zero gate has no slot at all, so the kernels walk 46 % of the tiles of a ScalarFilter batch instead of all of them with
    padding slots inside), or None -- a gate not flagged 0 / 1, masks off, a coefficient of unknown origin.  For the launches
    that LEAVE OUT such edges (``masked_slots`` turns them into padding where every edge needs its slot).  Memoised on the gate."""
    if gate is None or not USE_ROW_MASKS or not USE_LIVE_TILES:
        return None
    owner = _gate_owner(gate)
    if getattr(owner, "_dmp_dense_gate", False) or not getattr(owner, "_dmp_binary", False):
        return None
    hit = getattr(owner, "_dmp_live_tiles", None)
    if hit is not None and hit[0] == owner._version and hit[1] is index and hit[2] is coef:
        return hit[3]
    res = index.class_tiles_gated(coef, gate.reshape(-1))
    try:
        owner._dmp_live_tiles = (owner._version, index, coef, res)
    except Exception:
        pass
    return res


USE_KEEP_CSR = True


def keep_in_csr(index, gate):
    """``(keep_ptr, keep_ent)``: ``index``'s CSR by destination over the edges a 0 / 1 ``gate`` keeps (``dmp_csr_keep``), memoised
    on the gate (the layers of a rep-net share it), or None."""
    if gate is None or not USE_KEEP_CSR or not USE_ROW_MASKS or index.num_edges == 0:
        return None
    owner = _gate_owner(gate)
    if getattr(owner, "_dmp_dense_gate", False) or not getattr(owner, "_dmp_binary", False):
        return None
    hit = getattr(owner, "_dmp_keep_csr", None)
    if hit is not None and hit[0] == owner._version and hit[1] is index:
        return hit[2]
    lib = _lib.load()
    N, dev = index.num_nodes, index.in_ptr.device
    nscr = int(lib.dmp_csr_keep_scratch_words(N))
    ws = torch.empty(nscr + N + 1 + index.in_ent.numel(), dtype=torch.int32, device=dev)
    row_cnt, keep_ptr, keep_ent = ws[:nscr], ws[nscr:nscr + N + 1], ws[nscr + N + 1:]
    check(lib.dmp_csr_keep(ptr(index.in_ptr), ptr(index.in_ent), ptr(gate.reshape(-1)), N, index.in_ent.numel(), ptr(row_cnt), ptr(keep_ptr),
                           ptr(keep_ent), stream_ptr()), "dmp_csr_keep")
    res = (keep_ptr, keep_ent)
    try:
        owner._dmp_keep_csr = (owner._version, index, res)
    except Exception:
        pass
    return res


# the launches over the kept edges' tiles whose panel does not depend on the degree class (second Linear forward / backward, its
# weight gradient) walk the kept rows in ASCENDING order instead of class by class: the 512-byte row gathers then move through
# memory front to back (measured on one box: atb_typed plain 81 -> 71 us, bwd_h1_typed 77 -> 72 us; -0.03 ms per step)
PLAIN_ATB_ASCENDING = True
PLAIN_ROWS_ASCENDING = True


_ZEROS = {}


def const_zeros(n, device):
    """A read-only float32 zero vector of ``n`` entries (a view of one cached buffer per device): the ``tile_scale`` of tile lists
    whose kernels use a plain panel -- nobody writes it, so no launch fills it step after step.  Inside a stream capture a
    buffer made NOW would belong to the recording's pool: a fresh tensor there, as before."""
    key = (device.type, device.index)
    buf = _ZEROS.get(key)
    if buf is None or buf.numel() < n:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return torch.zeros(n, dtype=torch.float32, device=device)
        buf = _ZEROS[key] = torch.zeros(max(n, 1 << 16), dtype=torch.float32, device=device)
    return buf[:n]


_IDENTITY_TILES = {}


def identity_tiles(R, device):
    """The tile slot list of ALL rows 0 .. R-1 in order (padding -1 up to a whole tile), for the tile kernels on launches without
    a gate: ``(slot, tile_scale (zeros), num_tiles, bound)``.  A constant of (R, device): built once outside any recording."""
    key = (int(R), device.type, device.index)
    hit = _IDENTITY_TILES.get(key)
    if hit is not None:
        return hit
    if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        return None                      # (never made inside a recording: the first, eager step of a shape makes it)
    bound = (R + 31) // 32
    slot = torch.arange(bound * 32, dtype=torch.int32, device=device)
    slot[R:] = -1
    res = (slot, const_zeros(bound, device), torch.tensor([bound], dtype=torch.int32, device=device), bound)
    if len(_IDENTITY_TILES) < 64:
        _IDENTITY_TILES[key] = res
    return res


def ascending_tiles(gate):
    """The rows a 0 / 1 ``gate`` keeps as a tile slot list in ASCENDING row order (``dmp_kept_rows(tiles = 1)``; the class tiles list
    them class by class): for products that need no class -- the plain weight gradient ``dO^T H1`` -- the row gathers then walk
    memory front to back.  ``(slot, tile_scale (zeros), num_tiles, bound)`` or None; memoised on the gate."""
    mask = binary_gate_mask(gate)
    if mask is None:
        return None
    owner = _gate_owner(gate)
    hit = getattr(owner, "_dmp_asc_tiles", None)
    if hit is not None and hit[0] == owner._version:
        return hit[1]
    R = gate.numel()
    lst, cnt = kept_rows(mask, 0, R, tiles=True)
    bound = (R + 31) // 32
    res = (lst, const_zeros(bound, mask.device), cnt[1:2], bound)
    try:
        owner._dmp_asc_tiles = (owner._version, res)
    except Exception:
        pass
    return res


def bwd_z_typed(d_pre, ld_pre, Wes, d_s, base, coef, index, WesT=None, base_map=None, gate=None, dead_rows=None, dst=None):
    """bwd_z_mfma with the per-class matrix: base + gather_select(d_s) + dPre W_g^T  (dPre [E, H], leading dim ld_pre).
    ``WesT``: ``[A'^T | B'^T]`` if the caller has it already (``fold_layers`` makes it in its launch).
    ``base_map`` (int32 [E]): ``base`` is a small table and edge e adds its row ``base_map[e]`` (< 0: nothing).
    ``gate``: the layer's edge gate when ``d_pre`` is the gated layer's (its rows under a zero gate ARE zero): those rows are
    not fetched (``masked_slots``).  ``dead_rows`` (with ``gate``; ``zero_rows_gate`` holds): the input gradient of an edge
    under a zero gate is never used as a number (whoever made the rows multiplied them by the gate) -- ``"leave"``: those edges
    are padding slots, nothing is fetched, computed or stored for them (every reader of the result leaves them out);
    ``"zero"``: likewise, their output rows are zeros.
    ``dst`` (``NodeRows.sel[2]``): the destinations with the nodes whose rows of ``d_s`` were never written replaced by -1."""
    lib = _lib.load()
    E, H = d_pre.size(0), Wes.size(0)
    lt = live_tiles(index, coef, gate) if dead_rows is not None else None
    ms = lt[0] if lt is not None else masked_slots(index, coef, gate)
    if ms is None:
        dead_rows = None
    out = (torch.zeros((E, H), dtype=torch.float32, device=d_pre.device) if dead_rows == "zero" else
           dead_rows_buffer((E, H), d_pre.device) if dead_rows == "leave" else torch.empty((E, H), dtype=torch.float32, device=d_pre.device))
    # [A'^T | B'^T]: the kernel's per-class panel reads become coalesced (instead of a strided 128-instruction
    # panel read per class segment in each of its 768 workgroups)
    if WesT is None:
        WesT = torch.cat([Wes[:, :H].t(), Wes[:, H:].t()], dim=1)
    d_s = d_s.contiguous()
    if lt is not None:
        slot_edge, tile_scale, num_tiles, bound = lt
    else:
        slot_edge, tile_scale, num_tiles, bound = index.class_tiles(coef)
        if dead_rows is not None:
            slot_edge = ms
    with _lib.timed("bwd_z_typed[H=%d,E=%d]", (H, E), 4 * H * E * (3 if base is not None else 2) + 5 * E):
        check(lib.dmp_bwd_z_typed_arow(ptr(d_pre), ld_pre, ptr(WesT), WesT.size(1), ptr(d_s), d_s.size(1), index.num_nodes,
                                       ptr(base), base.stride(0) if base is not None else H, ptr(index.dst32 if dst is None else dst), ptr(index.rev8), -1.0, 1.0,
                                       ptr(slot_edge), ptr(ms), ptr(tile_scale), ptr(num_tiles), bound, E, H, 1,
                                       ptr(base_map), base.size(0) if base_map is not None else 0, ptr(out), H, stream_ptr()),
              "dmp_bwd_z_typed")
    return out


USE_DZW = _os.environ.get("DMP_DEV_DZW", "1") == "1"     # the edge chain's input gradient and its class-typed weight gradient in ONE launch (``dmp_bwd_z_w``)


def bwd_z_w(d_pre, z, Wes, d_s, base, coef, index, WesT=None, base_map=None, gate=None, dead_rows=None, dst=None):
    """``(dz, dWes)`` = ``(bwd_z_typed(...), atb_typed(z, d_pre, ...))`` from one pass over ``d_pre`` (``dmp_bwd_z_w``: the two-role kernel
    of csrc/dmp_h1w.hip with the class-typed panel and the class emission), or None where it does not apply: it needs the kept edges'
    class tiles of a 0 / 1 gate with ``dead_rows`` set (both launches then walk the SAME list), H = 128, bf16x6, arrays below 4 GiB."""
    lib = _lib.load()
    E, H = d_pre.size(0), Wes.size(0)
    lim = (1 << 32) - 65536
    if (not USE_DZW or (dead_rows is None) != (gate is None) or H != 128 or lib.dmp_dev_get_exact_fp32() or z is None or z.size(0) != E or E * H * 4 >= lim
            or E * d_pre.stride(0) * 4 >= lim or E * z.stride(0) * 4 >= lim or d_pre.data_ptr() % 16 or z.data_ptr() % 16
            or d_pre.stride(0) % 4 or z.stride(0) % 4 or index.num_nodes * d_s.size(1) * 4 >= lim):
        return None
    if gate is None:          # no gate at all (the all-rows step): the class tiles over every edge, every output row written
        if getattr(index, "_coef_deg", None) is None or index._coef_deg[0] is not coef or index.num_edges == 0:
            return None
        lt = index.class_tiles(coef)
    else:
        lt = live_tiles(index, coef, gate)
    if lt is None:
        return None
    if base is not None and (base.stride(0) % 4 or base.data_ptr() % 16 or base.size(0) * base.stride(0) * 4 >= lim):
        return None
    slot_edge, tile_scale, num_tiles, bound = lt
    out = (torch.zeros((E, H), dtype=torch.float32, device=d_pre.device) if dead_rows == "zero" else
           dead_rows_buffer((E, H), d_pre.device) if dead_rows == "leave" else torch.empty((E, H), dtype=torch.float32, device=d_pre.device))
    if WesT is None:
        WesT = torch.cat([Wes[:, :H].t(), Wes[:, H:].t()], dim=1)
    d_s = d_s.contiguous()
    G = int(lib.dmp_bwd_h1_w_blocks(bound))
    part = torch.empty((G, H * 2 * H), dtype=torch.float32, device=d_pre.device)
    with _lib.timed("bwd_z_w[H=%d,E=%d]", (H, E), 4 * H * E * (4 if base is not None else 3) + 5 * E):
        check(lib.dmp_bwd_z_w(ptr(d_pre), d_pre.stride(0), ptr(z), z.stride(0), ptr(WesT), WesT.size(1), ptr(d_s), d_s.size(1), index.num_nodes,
                              ptr(base), base.stride(0) if base is not None else H, ptr(index.dst32 if dst is None else dst), ptr(index.rev8), -1.0, 1.0,
                              ptr(slot_edge), ptr(tile_scale), ptr(num_tiles), bound, E, H, ptr(base_map),
                              base.size(0) if base_map is not None else 0, ptr(out), H, ptr(part), stream_ptr()), "dmp_bwd_z_w")
    return out, reduce_partials(part).view(H, 2 * H)


USE_ATB2 = _os.environ.get("DMP_DEV_ATB2", "1") == "1"     # the tile-list weight gradients on the bf16-piece LDS image (``dmp_atb2_jobs``)


class _Atb2Job(_ctypes.Structure):
    _fields_ = [("Z", _ctypes.c_void_p), ("ldz", _ctypes.c_int64), ("D", _ctypes.c_void_p), ("ldd", _ctypes.c_int64),
                ("partial_T", _ctypes.c_void_p), ("partial_B", _ctypes.c_void_p), ("partial_stride", _ctypes.c_int64), ("ldp", _ctypes.c_int)]


def atb2_ok(a, b, H):
    lib = _lib.load()
    lim = (1 << 32) - 65536
    return bool(USE_ATB2 and H == 128 and not lib.dmp_dev_get_exact_fp32() and a.size(0) == b.size(0) and a.size(0) * a.stride(0) * 4 < lim
                and b.size(0) * b.stride(0) * 4 < lim and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0
                and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def atb2(pairs, tiles, wide=False):
    """``[a^T b for (a, b) in pairs]`` over the rows of ``tiles`` (a, b: [R, 128] operands, column slices allowed): the products of
    ONE launch (``dmp_atb2_jobs``: at most 6).  ``wide``: each result is ``[a^T b | a^T (c (.) b)]`` ([128, 256]; c = the class
    coefficient of a row's tile: ``atb_typed``'s layout of dWes)."""
    lib = _lib.load()
    slot, tile_scale, num_tiles, bound = tiles
    n, H = len(pairs), 128
    R = pairs[0][0].size(0)
    G = int(lib.dmp_atb2_blocks(bound, n))
    W = 2 * H if wide else H
    jobs = (_Atb2Job * n)()
    parts = []
    for j, (a, b) in enumerate(pairs):
        part = torch.empty((G, H * W), dtype=torch.float32, device=a.device)
        parts.append(part)
        jobs[j].Z, jobs[j].ldz, jobs[j].D, jobs[j].ldd = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0)
        jobs[j].partial_T, jobs[j].partial_B = part.data_ptr(), (part.data_ptr() + 4 * H if wide else None)
        jobs[j].partial_stride, jobs[j].ldp = H * W, W
    with _lib.timed("atb2[jobs=%d,R=%d]", (n, R), 8 * H * R * n):
        check(lib.dmp_atb2_jobs(jobs, n, ptr(slot), ptr(tile_scale), ptr(num_tiles), bound, R, H, stream_ptr()), "dmp_atb2_jobs")
    return [reduce_partials(part).view(H, W) for part in parts]


def atb_typed(z, d_pre, coef, index, gate=None, plain=False):
    """``[z^T d_pre | z^T (coef[dst] (.) d_pre)]``  ([H, 2H]: the gradient of ``Wes`` in its layout) over the
    class-sorted tiles: one product's worth of MFMAs for both halves (csrc/dmp_atb.hip), one fixed-order
    reduction of the workgroup partials.  ``gate``: the layer's edge gate when ``d_pre`` is the gated layer's (zero rows under
    a zero gate): those edges are skipped, neither of their rows is fetched (``masked_slots``)."""
    lib = _lib.load()
    E, H = z.shape
    lt = live_tiles(index, coef, gate)
    if plain and lt is not None and PLAIN_ATB_ASCENDING:
        at = ascending_tiles(gate)
        if at is not None:
            lt = at
    if lt is not None:
        slot_edge, tile_scale, num_tiles, bound = lt
    else:
        slot_edge, tile_scale, num_tiles, bound = index.class_tiles(coef)
        ms = masked_slots(index, coef, gate)
        if ms is not None:
            slot_edge = ms
    if atb2_ok(z, d_pre, H):
        # every fetched element split into bf16 pieces once, fragments through transposed LDS reads (csrc/dmp_h1w.hip::atb2_k)
        return atb2([(z, d_pre)], (slot_edge, tile_scale, num_tiles, bound), wide=not plain)[0]
    G = int(lib.dmp_atb_typed_blocks_h(bound, H))
    if plain:       # ``z^T d_pre`` alone ([H, H]): half the partials
        part = torch.empty((G, H, H), dtype=torch.float32, device=z.device)
        with _lib.timed("atb_typed[H=%d,E=%d]", (H, E), 8 * H * E):
            check(lib.dmp_atb_typed(ptr(z), z.stride(0), ptr(d_pre), d_pre.stride(0), ptr(slot_edge), ptr(tile_scale),
                                    ptr(num_tiles), bound, E, H, ptr(part), None, stream_ptr()), "dmp_atb_typed")
        return reduce_partials(part.view(G, -1)).view(H, H)
    part = torch.empty((G, H, 2 * H), dtype=torch.float32, device=z.device)
    with _lib.timed("atb_typed[H=%d,E=%d]", (H, E), 8 * H * E):
        check(lib.dmp_atb_typed(ptr(z), z.stride(0), ptr(d_pre), d_pre.stride(0), ptr(slot_edge), ptr(tile_scale),
                                ptr(num_tiles), bound, E, H, ptr(part), ptr(part[0, 0, H:]), stream_ptr()), "dmp_atb_typed")
    return reduce_partials(part.view(G, -1)).view(H, 2 * H)


USE_ROW_MASKS = _os.environ.get("DMP_ROW_MASKS", "1") == "1"   # gated E-row kernels do not fetch the rows a zero gate annihilates
# ... and the two kernels that PRODUCE the first MLP's activation H1 leave out the rows all of whose consumers skip them
# (dead values: out_fwd, bwd_h1, atb_rows, the pooled passes multiply them by the zero gate).  Only with USE_ROW_MASKS.
SKIP_DEAD_ROWS = True
POISON_DEAD_ROWS = _os.environ.get("DMP_POISON_DEAD_ROWS", "0") == "1"   # testing aid: buffers with dead rows start as NaN


def dead_rows_buffer(shape, device):
    """An [E, H] buffer whose dead rows nobody writes: uninitialised -- or NaN throughout under ``POISON_DEAD_ROWS``, so that a
    consumer that does fetch a dead row shows up as NaN in its results."""
    if POISON_DEAD_ROWS:
        return torch.full(shape, float("nan"), dtype=torch.float32, device=device)
    return torch.empty(shape, dtype=torch.float32, device=device)


def gate_row_mask(gate):
    """uint32 [(R + 31) // 32]: bit r of word t = (gate[32 t + r] != 0) (``dmp_row_mask_bits``), for the ``_masked`` kernels;
    memoised on the gate tensor (a layer's forward and backward, and every layer of a rep-net, share the gate)."""
    if gate is None or not USE_ROW_MASKS:
        return None
    # the layers hand their kernels reshaped VIEWS of the gate they were given (a new tensor object per layer): the mask
    # hangs on the tensor the views share (same memory, same version counter), so a rep-net builds it once per step
    owner = gate._base if gate._base is not None else gate
    if owner.data_ptr() != gate.data_ptr() or owner.numel() != gate.numel():
        owner = gate
    if getattr(owner, "_dmp_dense_gate", False):            # a gate its maker knows to be (almost) all ones: nothing to skip
        return None
    hit = getattr(owner, "_dmp_row_mask", None)
    if hit is not None and hit[0] == owner._version:
        return hit[1]
    lib = _lib.load()
    R = gate.numel()
    # (a miss on the caller's stream while index builds are in flight on the side stream: the gate itself may have been
    # written THERE (dmpnn.prefetch_joint_indexes makes the union's gates inside its fork) -- order this stream behind that
    # stage before reading it.  The prefetch normally builds this mask too and the call above returns its memo: no wait.)
    from . import side
    side.wait("erows")
    mask = torch.empty(((R + 31) // 32,), dtype=torch.int32, device=gate.device)
    check(lib.dmp_row_mask_bits(ptr(gate), R, ptr(mask), stream_ptr()), "dmp_row_mask_bits")
    try:
        owner._dmp_row_mask = (owner._version, mask)
    except Exception:
        pass
    return mask


def code_row_mask(enc, K):
    """uint32 [(R + 31) // 32]: bit r of word t = (row 32 t + r of ``enc`` has a non-zero among its first K entries)
    (``dmp_row_mask_rows``): the packed label codes' live rows -- a gated-out row is all zeros, and so is every product with it."""
    if not USE_ROW_MASKS or enc is None:
        return None
    lib = _lib.load()
    R = enc.size(0)
    mask = torch.empty(((R + 31) // 32,), dtype=torch.int32, device=enc.device)
    check(lib.dmp_row_mask_rows(ptr(enc), enc.stride(0), int(K), R, ptr(mask), stream_ptr()), "dmp_row_mask_rows")
    return mask


def out_fwd_mfma(h1, W2, b2, gate, prev, W2t=None, dead_rows=0):
    """prev + gate * (h1 W2^T + b2): Linear + gate + residual in one fused MFMA kernel (H = 128 or 64).
    ``W2t``: ``W2.t()`` contiguous if the caller has it already (``fold_layers`` makes it in its launch).
    With a gate the rows of ``h1`` under a zero gate are not fetched (``gate_row_mask``): their term is multiplied by 0.
    ``dead_rows`` (``zero_rows_gate`` holds for the gate): 1 = the rows of ``prev`` under a zero gate are zeros and not fetched;
    3 = the output rows under a zero gate (zeros) are not stored either -- every reader leaves them out."""
    lib = _lib.load()
    R, H = h1.shape
    mask = gate_row_mask(gate)
    if mask is None:
        dead_rows = 0
    out = dead_rows_buffer((R, H), h1.device) if dead_rows & 2 else torch.empty((R, H), dtype=torch.float32, device=h1.device)
    if W2t is None:
        W2t = W2.t().contiguous() # [in, out]: coalesced weight-panel reads in each of the kernel's workgroups
    with _lib.timed("out_fwd_mfma[H=%d,R=%d]", (H, R), 4 * H * R * (3 if prev is not None else 2)):
        check(lib.dmp_out_fwd_fused_rows(ptr(h1), H, ptr(W2t), W2t.size(1), ptr(b2), ptr(gate), ptr(mask), int(dead_rows), ptr(prev), H,
                                         R, H, 1, ptr(out), H, stream_ptr()), "dmp_out_fwd_fused")
    return out


USE_TYPED_ATB_ROWS = True
USE_TYPED_ROWS = True   # out_fwd / bwd_h1 over the kept edges' tiles where dead rows need no store


_TYPED_JOB = None


def typed_jobs(jobs, tiles):
    """Up to six products ``out = prev + prev2 + (a W + bias)`` (optionally through LeakyReLU) for the rows of ONE tile list in
    ONE launch (``dmp_out_fwd_typed``: grid.y = the job).  ``jobs``: dicts with ``a`` [R, H] (any row stride), ``W`` (``[in, out]``,
    or with ``w_in_out=False`` a Linear's ``[out, in]``; any row stride: a block of a wider matrix), ``out`` [R, H] view, and
    optionally ``bias``, ``prev``, ``prev2``, ``slope``.  Rows outside the tiles are not written."""
    global _TYPED_JOB
    import ctypes
    lib = _lib.load()
    if _TYPED_JOB is None:
        P, I64, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
        _TYPED_JOB = type("dmp_typed_job", (ctypes.Structure,), {"_fields_": [
            ("Hin", P), ("ldh", I64), ("W2", P), ("ldw", I64), ("w_in_out", I), ("bias", P), ("R", P), ("ldr", I64), ("R2", P), ("ldr2", I64),
            ("act", I), ("slope", F), ("out", P), ("ldo", I64)]})
    slot_edge, tile_scale, num_tiles, bound = tiles
    R, H = jobs[0]["a"].shape
    J = (_TYPED_JOB * len(jobs))()
    for j, q in zip(J, jobs):
        a, W, out = q["a"], q["W"], q["out"]
        prev, prev2, slope = q.get("prev"), q.get("prev2"), q.get("slope")
        _lib.require_gpu(a, W, out, prev, prev2, q.get("bias"))
        j.Hin, j.ldh, j.W2, j.ldw, j.w_in_out = a.data_ptr(), a.stride(0), W.data_ptr(), W.stride(0), int(bool(q.get("w_in_out", True)))
        j.bias = ptr(q.get("bias"))
        j.R, j.ldr = ptr(prev), (prev.stride(0) if prev is not None else H)
        j.R2, j.ldr2 = ptr(prev2), (prev2.stride(0) if prev2 is not None else H)
        j.act, j.slope = int(slope is not None), float(slope or 0.0)
        j.out, j.ldo = out.data_ptr(), out.stride(0)
    with _lib.timed("out_fwd_typed[H=%d,R=%d,jobs=%d]", (H, R, len(jobs)), 4 * H * R * 3 * len(jobs)):
        check(lib.dmp_out_fwd_typed(J, len(jobs), ptr(slot_edge), ptr(tile_scale), ptr(num_tiles), bound, R, H, stream_ptr()),
              "dmp_out_fwd_typed")


def l0_rows(l0, W0, z):
    """The first layer's edge rows ``z``: as given, or -- when ``dmpnn.joint_rep`` left them out (``l0.z_from_codes``) and the layer
    did not take the launch that forms them in registers -- made here: the pattern's embedded rows (``l0.z_head``) over the target's
    from their label codes."""
    if l0 is None or not getattr(l0, "z_from_codes", False):
        return z
    n, K, H = l0.z_head.size(0), l0.K, W0.size(1)
    out = torch.empty((l0.enc.size(0), H), dtype=torch.float32, device=l0.enc.device)
    out[:n].copy_(l0.z_head)
    if out.size(0) > n:
        smallk_embed(l0.enc[n:, :K], W0.detach()[-K:], None, out[n:], H)
    return out


USE_OUT_CODES_DENSE = _os.environ.get("DMP_DEV_OUT_CODES_DENSE", "1") == "1"   # ... also without a gate (over the identity tile list)
USE_OUT_CODES = _os.environ.get("DMP_DEV_OUT_CODES", "1") == "1"   # the first layer's residual rows z0 = codes W_e as a K-extension of its second Linear


def out_codes_ok(h1, enc, K, Wc):
    """``out_fwd_typed_codes`` applies: H = 128, bf16x6, at most 16 code columns in rows of whole 16-byte pieces, arrays below 4 GiB."""
    lib = _lib.load()
    lim = (1 << 32) - 65536
    return bool(USE_OUT_CODES and h1.size(1) == 128 and not lib.dmp_dev_get_exact_fp32() and 1 <= K <= 16 and Wc.size(0) == K and Wc.size(1) == 128
                and enc.dtype == torch.float32 and enc.stride(1) == 1 and enc.stride(0) % 4 == 0 and enc.data_ptr() % 16 == 0
                and enc.size(0) == h1.size(0) and h1.size(0) * 128 * 4 < lim and Wc.stride(1) == 1)


def out_fwd_typed_codes(h1, W2t, b2, enc, K, Wc, tiles, prev=None, out=None, w_in_out=True, row0=0):
    """``prev + (h1 W2^T + enc[:, :K] Wc + b2)`` for the rows of ``tiles`` (``dmp_out_fwd_typed_codes``): the second Linear of a FIRST
    layer whose residual rows are the label embedding ``enc Wc`` -- added as one more 16-deep k-group of the product, so the [E, H] rows
    ``z0`` are neither written (``dmp_smallk_embed_live``) nor read.  ``row0``: the rows below it take no codes term (the pattern's
    edges, embedded by another table); ``prev`` may then hold just those rows ([row0 or more, H]: missing rows count as zeros)."""
    global _TYPED_JOB
    lib = _lib.load()
    if _TYPED_JOB is None:
        P, I64, I, F = _ctypes.c_void_p, _ctypes.c_int64, _ctypes.c_int, _ctypes.c_float
        _TYPED_JOB = type("dmp_typed_job", (_ctypes.Structure,), {"_fields_": [
            ("Hin", P), ("ldh", I64), ("W2", P), ("ldw", I64), ("w_in_out", I), ("bias", P), ("R", P), ("ldr", I64), ("R2", P), ("ldr2", I64),
            ("act", I), ("slope", F), ("out", P), ("ldo", I64)]})
    slot_edge, tile_scale, num_tiles, bound = tiles
    R, H = h1.shape
    if out is None:
        out = dead_rows_buffer((R, H), h1.device)
    Wc = Wc.detach()
    _lib.require_gpu(h1, W2t, out, prev, b2, enc, Wc)
    j = _TYPED_JOB()
    j.Hin, j.ldh, j.W2, j.ldw, j.w_in_out = h1.data_ptr(), h1.stride(0), W2t.data_ptr(), W2t.stride(0), int(bool(w_in_out))
    j.bias = ptr(b2)
    j.R, j.ldr = ptr(prev), (prev.stride(0) if prev is not None else H)
    j.R2, j.ldr2 = None, H
    j.act, j.slope = 0, 0.0
    j.out, j.ldo = out.data_ptr(), out.stride(0)
    with _lib.timed("out_fwd_typed_codes[H=%d,R=%d]", (H, R), 4 * H * R * 2 + 4 * enc.size(1) * R):
        check(lib.dmp_out_fwd_typed_codes(_ctypes.byref(j), ptr(enc), enc.stride(0), int(K), ptr(Wc), Wc.stride(0), int(row0),
                                          0 if prev is None else min(R, prev.size(0)), ptr(slot_edge), ptr(tile_scale),
                                          ptr(num_tiles), bound, R, H, stream_ptr()), "dmp_out_fwd_typed_codes")
    return out


def out_fwd_typed(h1, W2t, b2, prev, tiles, out=None, w_in_out=True, slope=None, prev2=None):
    """``prev + (h1 W2^T + b2)`` for the rows of ``tiles`` (``live_tiles``: the edges a 0 / 1 gate keeps, gate 1 there; or
    ``NodeRows.tiles``: the kept nodes); the other rows of the result are not written (``dead_rows_buffer``).  ``W2t`` [in, out]
    (any row stride: a column block of a wider matrix), or with ``w_in_out=False`` the Linear's own [out, in].  ``out``: the
    destination (may be ``prev``: a product accumulates onto its own output); ``prev2``: a second addend; ``slope``:
    LeakyReLU(slope) on the result."""
    R, H = h1.shape
    if out is None:
        out = dead_rows_buffer((R, H), h1.device)
    typed_jobs([dict(a=h1, W=W2t, bias=b2, prev=prev, prev2=prev2, out=out, w_in_out=w_in_out, slope=slope)], tiles)
    return out


def bwd_h1_typed(d_o, W2, h1, tiles, slope=0.0, out=None):
    """``(dPre, column sums of dPre, column sums of the kept rows of d_o)`` with ``dPre[e] = act'(h1[e]) (.) (d_o[e] W2)`` for the
    rows of ``tiles`` (``live_tiles`` / ``NodeRows.tiles``); the other rows of ``dPre`` are not written (``dead_rows_buffer``, or
    ``out``: e.g. a column block of a wider matrix)."""
    lib = _lib.load()
    E, H = d_o.shape
    slot_edge, tile_scale, num_tiles, bound = tiles
    d_g = out if out is not None else dead_rows_buffer((E, H), d_o.device)
    G = int(lib.dmp_typed_partial_rows(bound, H))
    part = torch.empty((G, H), dtype=torch.float32, device=d_o.device)
    part_rows = torch.empty_like(part)
    W2 = W2.contiguous()
    with _lib.timed("bwd_h1_typed[H=%d,E=%d]", (H, E), 12 * H * E + 4 * E):
        check(lib.dmp_bwd_h1_typed(ptr(d_o), d_o.stride(0), ptr(W2), W2.size(1), ptr(h1), h1.stride(0), ptr(slot_edge), ptr(tile_scale),
                                   ptr(num_tiles), bound, E, H, slope, ptr(d_g), d_g.stride(0), ptr(part), ptr(part_rows), stream_ptr()),
              "dmp_bwd_h1_typed")
    return d_g, reduce_partials(part), reduce_partials(part_rows)


USE_H1W_DENSE = True    # ... also where no gate applies (every row live): over the identity tile list
USE_H1W = True    # the second edge Linear's backward as ONE launch (dPre and dO^T H1 from the same fetched rows, csrc/dmp_h1w.hip)


def h1w_ok(d_o, h1, H):
    """The one-launch form of ``bwd_h1_typed`` + ``atb_typed(plain=True)`` applies: H = 128, the bf16x6 arithmetic, arrays below 4 GiB."""
    lib = _lib.load()
    return (USE_H1W and H == 128 and not lib.dmp_dev_get_exact_fp32() and d_o.size(0) * d_o.stride(0) * 4 < (1 << 32) - 65536
            and h1.size(0) * h1.stride(0) * 4 < (1 << 32) - 65536 and d_o.size(0) * H * 4 < (1 << 32) - 65536)


def bwd_h1_w(d_o, W2, h1, tiles, slope=0.0, out=None):
    """``(dPre, column sums of dPre, column sums of the kept rows of d_o, d_o^T h1)`` over the rows of ``tiles``: what
    ``bwd_h1_typed`` and ``atb_typed(d_o, h1, plain=True)`` return, from one pass over the two operands (``dmp_bwd_h1_w``)."""
    lib = _lib.load()
    E, H = d_o.shape
    slot_edge, _, num_tiles, bound = tiles
    d_g = out if out is not None else dead_rows_buffer((E, H), d_o.device)
    G = int(lib.dmp_bwd_h1_w_blocks(bound))
    part = torch.empty((G, H), dtype=torch.float32, device=d_o.device)
    part_rows = torch.empty_like(part)
    part_w = torch.empty((G, H * H), dtype=torch.float32, device=d_o.device)
    W2 = W2.contiguous()
    with _lib.timed("bwd_h1_w[H=%d,E=%d]", (H, E), 12 * H * E + 4 * E):
        check(lib.dmp_bwd_h1_w(ptr(d_o), d_o.stride(0), ptr(W2), W2.size(1), ptr(h1), h1.stride(0), ptr(slot_edge), ptr(num_tiles), bound,
                               E, H, slope, ptr(d_g), d_g.stride(0), ptr(part), ptr(part_rows), ptr(part_w), stream_ptr()), "dmp_bwd_h1_w")
    return d_g, reduce_partials(part), reduce_partials(part_rows), reduce_partials(part_w).view(H, H)


USE_MASKED_SUMS = True   # the scatter-adds skip the rows a 0 / 1 edge gate wiped


def zero_rows_gate(gate):
    """``gate`` is 0 / 1 (``_dmp_binary``) AND its maker multiplied the rep-net's input rows by it (``_dmp_zero_rows``, set by
    ``dmpnn.joint_rep``: the union's edge rows are ``[pattern rows | gate * target rows]``): every layer's input rows under a
    zero of the gate are zeros -- ``zn = z + gate (...)`` (dmpnn.py:215-277) keeps them so -- and sums over rows may leave
    them out."""
    if gate is None or not USE_MASKED_SUMS or not USE_ROW_MASKS:
        return False
    owner = _gate_owner(gate)
    return bool(getattr(owner, "_dmp_binary", False) and getattr(owner, "_dmp_zero_rows", False)
                and not getattr(owner, "_dmp_dense_gate", False))


def binary_gate_mask(gate):
    """The row mask of a gate flagged 0 / 1 (``_dmp_binary``, e.g. a ScalarFilter gate), else None: for such a gate the mask says
    everything the gate says."""
    if gate is None or not USE_PLAIN_ATB or not getattr(_gate_owner(gate), "_dmp_binary", False):
        return None
    return gate_row_mask(gate)


def bwd_h1_mfma(d_o, W2, h1, coef=None, index=None, both_halves=True, gate=None, out=None, slope=0.0, rows_colsum=False,
                skip_dead_stores=False):
    """-> (dG = [dPre | coef[dst] dPre] (or dPre alone) with dPre = h1>0 ? d_o W2 : slope (d_o W2), column sums of dPre); H=128.
    ``gate`` (dPre alone only): ``d_o`` is the ungated output gradient, its rows are scaled by the gate here.
    ``out`` (dPre alone only): destination [R, H], e.g. a column slice of a wider matrix.
    ``rows_colsum``: a third result, the column sums of the rows of ``d_o`` the kernel fetched -- with the row mask of a 0 / 1
    gate (``binary_gate_mask``) that is ``sum_e gate_e d_o[e]``, the bias gradient of the Linear behind the gate.
    ``skip_dead_stores`` (dPre alone, gated): dPre's rows under a zero gate (zeros) are not stored: every reader leaves them out."""
    lib = _lib.load()
    E, H = d_o.shape
    mask = gate_row_mask(gate)
    skip_dead_stores = bool(skip_dead_stores and mask is not None and not both_halves and out is None)
    if both_halves:
        d_g = torch.empty((E, 2 * H), dtype=torch.float32, device=d_o.device)
        coef_e = index.edge_select(coef)[2]
    else:
        d_g = out if out is not None else (dead_rows_buffer((E, H), d_o.device) if skip_dead_stores
                                           else torch.empty((E, H), dtype=torch.float32, device=d_o.device))
        coef_e = None
    part = torch.empty((int(lib.dmp_mfma_partial_rows_h(E, H)), H), dtype=torch.float32, device=d_o.device)
    W2 = W2.contiguous()
    with _lib.timed("bwd_h1_mfma[H=%d,E=%d]", (H, E), (16 if both_halves else 12) * H * E + 4 * E):
        part_rows = torch.empty_like(part) if rows_colsum else None
        check(lib.dmp_bwd_h1_fused_rows(ptr(d_o), d_o.stride(0), ptr(W2), W2.size(1), ptr(h1), h1.stride(0), ptr(coef_e),
                                        ptr(gate), ptr(mask), int(skip_dead_stores), E, H, slope, ptr(d_g),
                                        d_g.stride(0) if E > 1 else d_g.size(1), ptr(part), ptr(part_rows), stream_ptr()),
              "dmp_bwd_h1_fused")
    if rows_colsum:
        return d_g, reduce_partials(part), reduce_partials(part_rows)
    return d_g, reduce_partials(part)


def _gate_owner(gate):
    """The tensor a gate's views share (``gate_row_mask``): where its memoised masks and flags hang."""
    owner = gate._base if gate._base is not None else gate
    if owner.data_ptr() != gate.data_ptr() or owner.numel() != gate.numel():
        owner = gate
    return owner


USE_PLAIN_ATB = True


def atb_rows(a, b, gate=None, colsum=True):
    """``((gate (.) a)^T b  [M,N],  column sums of gate (.) a  [M] (or None))`` in one MFMA pass over the rows
    (csrc/dmp_atb.hip); M, N multiples of 128 (or of 64).  A Linear's weight and bias gradient with a row gate fused in.
    A gate flagged as 0 / 1 (``_dmp_binary``: a ScalarFilter gate) is fully expressed by its row mask: the product then runs
    ungated over the masked-in rows on the bf16 pipe (``dmp_atb_rows_plain``), the column sums as a one-column ``smallk_atb``."""
    lib = _lib.load()
    R, M = a.shape
    N = b.size(1)
    blk = atb_block(M, N)
    G = int(lib.dmp_atb_rows_blocks_h(R, M, N, blk))
    part = torch.empty((G, M * N), dtype=torch.float32, device=a.device)
    mask = binary_gate_mask(gate)
    if mask is not None and M % blk == 0 and blk in (64, 128):
        with _lib.timed("atb_rows_plain[M=%d,N=%d,R=%d]", (M, N, R), 4 * (M + N) * R):
            check(lib.dmp_atb_rows_plain(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(mask), R, M, N, blk, ptr(part), stream_ptr()),
                  "dmp_atb_rows_plain")
        cs = None
        if colsum:   # sum over the kept rows of a = gate^T a: the K = 1 case of the narrow weight-gradient kernel, same mask
            cs = smallk_atb_cols(gate.reshape(-1, 1), a, H=blk, mask=mask).reshape(M)
        return reduce_partials(part).view(M, N), cs
    part_cs = torch.empty((G, M), dtype=torch.float32, device=a.device) if colsum else None
    with _lib.timed("atb_rows[M=%d,N=%d,R=%d]", (M, N, R), 4 * (M + N) * R + (4 * R if gate is not None else 0)):
        check(lib.dmp_atb_rows_masked(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(gate), ptr(gate_row_mask(gate)), ATB_ROWS_X6,
                                      R, M, N, blk, ptr(part), ptr(part_cs), stream_ptr()), "dmp_atb_rows_masked")
    return reduce_partials(part).view(M, N), (reduce_partials(part_cs) if colsum else None)


ATB_ROWS_X6 = 0   # the gated rows weight gradient on the bf16 pipe (csrc/dmp_atb.hip)
SMALLK_MAX = 16


def smallk_atb(x, d, gate=None, out=None):
    """``x^T (gate (.) d)``  ([K, H], K <= 16, H = 128 or 64) in one pass over ``d``: the weight gradient of a narrow
    input layer (label encodings @ W) whose output was gated row-wise (csrc/dmp_fused.hip::smallk_atb_k)."""
    lib = _lib.load()
    R, K = x.shape
    H = d.size(1)
    G = int(lib.dmp_smallk_atb_blocks(R))
    part = torch.empty((G, K * H), dtype=torch.float32, device=d.device)
    with _lib.timed("smallk_atb[K=%d,R=%d]", (K, R), 4 * (H + K + 1) * R):
        check(lib.dmp_smallk_atb(ptr(x), x.stride(0), K, ptr(d), d.stride(0), ptr(gate), R, H, ptr(part), stream_ptr()),
              "dmp_smallk_atb")
    return reduce_partials(part, None if out is None else out.view(-1)).view(K, H)


def smallk_embed(x, W, gate=None, out=None, H=None):
    """``gate (.) (x @ W)``  for a narrow ``x`` ([R, K <= 16]: label encodings) and ``W`` [K, C]: one pass, the K rows of W
    in registers (csrc/dmp_fused.hip::smallk_embed_k).  C = 128 or 64, or a multiple of 128 (one launch, a grid row per
    block of 128 columns); ``W`` / ``out`` may be column blocks of wider matrices (unit inner stride)."""
    lib = _lib.load()
    R, K = x.shape
    C = W.size(1)
    H = H if H is not None else (C if C in (64, 128) else 128)
    ncols = C // H
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32, device=x.device)
    with _lib.timed("smallk_embed[K=%d,R=%d]", (K, R), 4 * (C + K + 1) * R):
        check(lib.dmp_smallk_embed_cols(ptr(x), x.stride(0), K, ptr(W), W.stride(0), ptr(gate), R, H, ncols, ptr(out), out.stride(0),
                                        stream_ptr()), "dmp_smallk_embed_cols")
    return out


def smallk_atb_cols(x, d, d2=None, out=None, H=None, mask=None, rows=None):
    """``x^T [d | d2]`` per block of 128 (or 64: one block) columns -> [blocks, K, H]: ``smallk_atb`` over the column blocks
    of ``d`` [R, ncols H] and one more matrix ``d2`` [R, H] in ONE launch.  ``mask`` (uint32 words, ``gate_row_mask``, aligned
    to x's first row): the rows of ``d`` / ``d2`` whose row of ``x`` is known to be all zeros are not fetched.  ``rows`` =
    ``(list, count)`` (``kept_rows`` relative to x's first row) instead of ``mask``: the launch walks the list -- every batch of
    rows is live rows; a row outside the list must be a zero row of ``x``."""
    lib = _lib.load()
    R, K = x.shape
    C = d.size(1)
    H = H if H is not None else (C if C in (64, 128) else 128)
    ncols = C // H
    nblk = ncols + (1 if d2 is not None else 0)
    G = int(lib.dmp_smallk_atb_blocks(R))
    part = torch.empty((nblk, G, K * H), dtype=torch.float32, device=d.device)
    if out is None:
        out = torch.empty((nblk, K, H), dtype=torch.float32, device=d.device)
    with _lib.timed("smallk_atb[K=%d,R=%d]", (K, R), 4 * (nblk * H + K + 1) * R):
        if rows is not None:
            check(lib.dmp_smallk_atb_cols_rows(ptr(x), x.stride(0), K, ptr(d), d.stride(0), ncols, ptr(d2),
                                               d2.stride(0) if d2 is not None else 0, ptr(rows[0]), ptr(rows[1]), R, H, ptr(part), stream_ptr()),
                  "dmp_smallk_atb_cols_rows")
        else:
            check(lib.dmp_smallk_atb_cols_masked(ptr(x), x.stride(0), K, ptr(d), d.stride(0), ncols, ptr(d2),
                                                 d2.stride(0) if d2 is not None else 0, None, ptr(mask), R, H, ptr(part), stream_ptr()),
                  "dmp_smallk_atb_cols")
    for j in range(nblk):
        reduce_partials(part[j], out[j].view(-1))
    return out


_SMALLK_JOB = None
USE_SMALLK_JOBS = _os.environ.get("DMP_DEV_SMALLK_JOBS", "1") == "1"


def smallk_atb_jobs(jobs, H):
    """Several ``smallk_atb`` products of the same K and H in ONE launch (``dmp_smallk_atb_jobs``): ``jobs`` = up to four
    ``(x [R, K], d [R, H], out [K, H], mask or None)`` (``mask``: ``gate_row_mask`` words aligned to the job's first row).  The
    sums land in each job's ``out``."""
    global _SMALLK_JOB
    lib = _lib.load()
    if _SMALLK_JOB is None:
        P, I64 = _ctypes.c_void_p, _ctypes.c_int64
        _SMALLK_JOB = type("dmp_smallk_job", (_ctypes.Structure,), {"_fields_": [
            ("X", P), ("ldx", I64), ("D", P), ("ldd", I64), ("gate", P), ("rowmask", P), ("R", I64), ("partial", P)]})
    K = jobs[0][0].size(1)
    J = (_SMALLK_JOB * len(jobs))()
    parts = []
    for n, (x, d, out, mask) in enumerate(jobs):
        R = x.size(0)
        _lib.require_gpu(x, d, out, mask)
        if x.size(1) != K or d.size(0) != R or d.size(1) != H or tuple(out.shape) != (K, H) or not out.is_contiguous():
            raise _lib.DmpError("smallk_atb_jobs: shapes of job %d" % n)
        part = torch.empty((int(lib.dmp_smallk_atb_blocks(R)), K * H), dtype=torch.float32, device=d.device)
        J[n].X, J[n].ldx, J[n].D, J[n].ldd = (x.data_ptr() if R else None), x.stride(0) if R > 1 else max(K, x.stride(0)), (d.data_ptr() if R else None), d.stride(0) if R > 1 else max(H, d.stride(0))
        J[n].gate, J[n].rowmask, J[n].R, J[n].partial = None, ptr(mask), R, part.data_ptr()
        parts.append(part)
    with _lib.timed("smallk_atb_jobs[K=%d,R=%d]", (K, sum(j[0].size(0) for j in jobs)), sum(4 * (H + K + 1) * j[0].size(0) for j in jobs)):
        check(lib.dmp_smallk_atb_jobs(J, len(jobs), K, H, stream_ptr()), "dmp_smallk_atb_jobs")
    for part, (_, _, out, _) in zip(parts, jobs):
        reduce_partials(part, out.view(-1))


# ---- the first layer of a rep-net straight from the label codes (csrc/dmp_layer0.hip)
L0_KMAX = 16
# the joint rep-net pass hands the first layer the packed label codes instead of differentiable [E, H] rows
# (dmpnn.joint_rep); DMP_LAYER0=0 keeps the general path
USE_LAYER0 = _os.environ.get("DMP_LAYER0", "1") == "1"
USE_LAYER0_NODES = True    # ... and the node rows' codes as well


def l0_pack(enc_p, enc_g, gate=None, stacked=False):
    """``[enc_p ; gate * enc_g]`` zero-padded to a multiple of 4 columns: the gated label codes of the union's rows.
    ``stacked``: the second kind of rows keep their codes in columns K..2K-1, so that two embedding tables stacked to
    [2K, H] act as ONE table on the packed codes (the node rows: 2K <= 16)."""
    lib = _lib.load()
    _lib.require_gpu(enc_p, enc_g)
    K = enc_g.size(1)
    goff = K if stacked else 0
    Kpad = (goff + K + 3) // 4 * 4
    n, rows_g = enc_p.size(0), enc_g.size(0)
    out = torch.empty((n + rows_g, Kpad), dtype=torch.float32, device=enc_g.device)
    gt = None if gate is None else gate.reshape(-1).contiguous()
    check(lib.dmp_l0_pack(ptr(enc_p), enc_p.stride(0) if n else K, n, ptr(enc_g), enc_g.stride(0) if rows_g else K, ptr(gt), rows_g,
                          K, Kpad, goff, ptr(out), stream_ptr()), "dmp_l0_pack")
    return out


_L0_PACK_JOB = None


def l0_pack_many(specs):
    """``[l0_pack(*spec) for spec in specs]`` (``spec`` = ``(enc_p, enc_g, gate, stacked)``; at most two) in ONE launch
    (``dmp_l0_pack_jobs``): a step's edge codes and node codes."""
    global _L0_PACK_JOB
    lib = _lib.load()
    if _L0_PACK_JOB is None:
        P, I64, I = _ctypes.c_void_p, _ctypes.c_int64, _ctypes.c_int
        _L0_PACK_JOB = type("dmp_l0_pack_job", (_ctypes.Structure,), {"_fields_": [
            ("enc_p", P), ("ldp", I64), ("rows_p", I64), ("enc_g", P), ("ldg", I64), ("gate", P), ("rows_g", I64),
            ("K", I), ("Kpad", I), ("goff", I), ("out", P)]})
    J = (_L0_PACK_JOB * len(specs))()
    outs, keep = [], []
    for i, (enc_p, enc_g, gate, stacked) in enumerate(specs):
        _lib.require_gpu(enc_p, enc_g)
        K = enc_g.size(1)
        goff = K if stacked else 0
        Kpad = (goff + K + 3) // 4 * 4
        n, rows_g = enc_p.size(0), enc_g.size(0)
        out = torch.empty((n + rows_g, Kpad), dtype=torch.float32, device=enc_g.device)
        gt = None if gate is None else gate.reshape(-1).contiguous()
        J[i].enc_p, J[i].ldp, J[i].rows_p = ptr(enc_p), (enc_p.stride(0) if n else K), n
        J[i].enc_g, J[i].ldg, J[i].gate, J[i].rows_g = ptr(enc_g), (enc_g.stride(0) if rows_g else K), ptr(gt), rows_g
        J[i].K, J[i].Kpad, J[i].goff, J[i].out = K, Kpad, goff, out.data_ptr()
        outs.append(out)
        keep.append(gt)
    check(lib.dmp_l0_pack_jobs(J, len(specs), stream_ptr()), "dmp_l0_pack_jobs")
    return outs


USE_L0_ROW_LISTS = True
L0_LIST_MIN_ROWS = 32768       # shorter row ranges (the pattern side) keep the masked form: the list costs two launches


def kept_rows(mask, r0, r1, tiles=False):
    """``(list int32 [r1 - r0], count int32 [1])``: the rows of ``[r0, r1)`` (r0 a multiple of 32) whose bit of ``mask`` is set, as
    ids relative to r0, ascending (``dmp_kept_rows``); memoised on the mask tensor (the forward and the backward of a layer
    share it).  ``tiles``: the list padded with -1 to whole 32-row tiles and ``count`` int32 [2] = (rows, tiles): a slot list
    for the tile kernels (``node_tiles``)."""
    if tiles:
        return _kept_row_tiles(mask, r0, r1)
    memo = getattr(mask, "_dmp_kept_rows", None)
    if memo is None:
        memo = {}
        try:
            mask._dmp_kept_rows = memo
        except Exception:
            pass
    hit = memo.get((r0, r1))
    if hit is not None:
        return hit
    lib = _lib.load()
    R = r1 - r0
    out = torch.empty(R + 1 + int(lib.dmp_kept_rows_scratch_words(R)), dtype=torch.int32, device=mask.device)
    lst, cnt, scratch = out[:R], out[R:R + 1], out[R + 1:]
    check(lib.dmp_kept_rows(ptr(mask[r0 // 32:]), R, 0, ptr(scratch), ptr(lst), ptr(cnt), stream_ptr()), "dmp_kept_rows")
    memo[(r0, r1)] = (lst, cnt)
    return lst, cnt


def _kept_row_tiles(mask, r0, r1):
    memo = getattr(mask, "_dmp_kept_tiles", None)
    if memo is None:
        memo = {}
        try:
            mask._dmp_kept_tiles = memo
        except Exception:
            pass
    hit = memo.get((r0, r1))
    if hit is not None:
        return hit
    lib = _lib.load()
    R = r1 - r0
    cap = (R + 31) // 32 * 32
    out = torch.empty(cap + 2 + int(lib.dmp_kept_rows_scratch_words(R)), dtype=torch.int32, device=mask.device)
    lst, cnt, scratch = out[:cap], out[cap:cap + 2], out[cap + 2:]
    check(lib.dmp_kept_rows(ptr(mask[r0 // 32:]), R, 1, ptr(scratch), ptr(lst), ptr(cnt), stream_ptr()), "dmp_kept_rows")
    memo[(r0, r1)] = (lst, cnt)
    return lst, cnt


USE_GATE_BUNDLE = True      # the index arrays a step derives from its two 0 / 1 gates in three launches instead of ten (``gate_bundle``)


class _RowMaskJob(_ctypes.Structure):
    _fields_ = [("gate", _ctypes.c_void_p), ("R", _ctypes.c_int64), ("mask", _ctypes.c_void_p)]


class _KeptJob(_ctypes.Structure):
    _fields_ = [("mask", _ctypes.c_void_p), ("R", _ctypes.c_int64), ("tiles", _ctypes.c_int), ("scratch", _ctypes.c_void_p),
                ("list", _ctypes.c_void_p), ("count", _ctypes.c_void_p)]


def _mask_owner(gate):
    owner = gate._base if gate._base is not None else gate
    if owner.data_ptr() != gate.data_ptr() or owner.numel() != gate.numel():
        owner = gate
    return owner


def gate_bundle(index, v_gate, e_gate, want_nodes=False, l0_range=None, want_ascending=False):
    """What ``gate_row_mask`` (both gates), ``kept_rows`` (the kept nodes' tiles, the first layer's kept target rows, the kept
    edges' ascending tiles) and ``GraphIndex.edge_select_nodes`` build in ten launches of ~5 us -- links of the step's index
    chain, which the first layer waits for -- in THREE: both masks (``dmp_row_mask_bits_jobs``), then every list's block counts
    and the selectors, then every list's fill (``dmp_kept_rows_jobs``).  Each product lands in the memo its own function looks
    it up in (same bits: the multi-job kernels are the single ones over ``blockIdx.y``); anything not asked for, or already
    there, is left to those functions."""
    if not USE_GATE_BUNDLE or not USE_ROW_MASKS:
        return
    lib = _lib.load()
    todo = []
    for g in (v_gate, e_gate):
        if g is None or not g.is_cuda or g.dtype != torch.float32:
            return
        owner = _mask_owner(g)
        if getattr(owner, "_dmp_dense_gate", False):
            return
        hit = getattr(owner, "_dmp_row_mask", None)
        if hit is None or hit[0] != owner._version:
            todo.append((g, owner))
    if todo:
        jobs = (_RowMaskJob * len(todo))()
        masks = []
        for j, (g, owner) in enumerate(todo):
            R = g.numel()
            m = torch.empty(((R + 31) // 32,), dtype=torch.int32, device=g.device)
            masks.append(m)
            jobs[j].gate, jobs[j].R, jobs[j].mask = ptr(g), R, ptr(m)
        check(lib.dmp_row_mask_bits_jobs(jobs, len(todo), stream_ptr()), "dmp_row_mask_bits_jobs")
        for (g, owner), m in zip(todo, masks):
            try:
                owner._dmp_row_mask = (owner._version, m)
            except Exception:
                return
    vmask, emask = gate_row_mask(v_gate), gate_row_mask(e_gate)
    if vmask is None or emask is None:
        return
    N, E = index.num_nodes, index.num_edges
    lists = []          # (mask view, R, tiles, memo dict, key)
    if want_nodes and N > 0:
        memo = vmask.__dict__.setdefault("_dmp_kept_tiles", {})
        if (0, N) not in memo:
            lists.append((vmask, N, 1, memo, (0, N)))
    if l0_range is not None and l0_range[1] > l0_range[0] and l0_range[0] % 32 == 0:
        memo = emask.__dict__.setdefault("_dmp_kept_rows", {})
        if tuple(l0_range) not in memo:
            lists.append((emask[l0_range[0] // 32:], l0_range[1] - l0_range[0], 0, memo, tuple(l0_range)))
    if want_ascending and E > 0:
        memo = emask.__dict__.setdefault("_dmp_kept_tiles", {})
        if (0, E) not in memo:
            lists.append((emask, E, 1, memo, (0, E)))
    sel = want_nodes and E > 0 and (getattr(index, "_esel_nodes", None) is None or index._esel_nodes[0] is not vmask)
    if not lists and not sel:
        return
    jobs = (_KeptJob * max(len(lists), 1))()
    keep = []
    for j, (m, R, tiles, memo, key) in enumerate(lists):
        cap = (R + 31) // 32 * 32 if tiles else R
        ncnt = 2 if tiles else 1
        out = torch.empty(cap + ncnt + int(lib.dmp_kept_rows_scratch_words(R)), dtype=torch.int32, device=m.device)
        lst, cnt, scratch = out[:cap], out[cap:cap + ncnt], out[cap + ncnt:]
        jobs[j].mask, jobs[j].R, jobs[j].tiles, jobs[j].scratch, jobs[j].list, jobs[j].count = ptr(m), R, tiles, ptr(scratch), ptr(lst), ptr(cnt)
        keep.append((memo, key, (lst, cnt)))
    so = torch.empty((3, E), dtype=torch.int32, device=index.device) if sel else None
    check(lib.dmp_kept_rows_jobs(jobs, len(lists), ptr(index.src32) if sel else None, ptr(index.dst32) if sel else None,
                                 ptr(index.rev8) if sel else None, ptr(vmask) if sel else None, E if sel else 0,
                                 ptr(so[0]) if sel else None, ptr(so[1]) if sel else None, ptr(so[2]) if sel else None, stream_ptr()),
          "dmp_kept_rows_jobs")
    for memo, key, val in keep:
        memo[key] = val
    if sel:
        index._esel_nodes = (vmask, (so[0], so[1], so[2]))


USE_NODE_ROWS = _os.environ.get("DMP_NODE_ROWS", "1") == "1"   # the node side of a layer over the nodes a 0 / 1 node gate keeps
USE_KEPT_INCIDENCE = True     # ... and its backward's endpoint sums as a segment sum over the kept edges' incidence CSR
USE_NODE_TILE_ATB = True      # ... and the node side's weight gradients over the kept nodes' tiles


class NodeRows:
    """The nodes a 0 / 1 node gate keeps, as the kernels of a layer's node side want them (``node_rows``): ``mask`` (uint32
    words, bit = kept), ``rows`` = ``(list, count)`` ascending ids (``dmp_kept_rows``), ``tiles`` = the same list as a tile slot
    list ``(slot, tile_scale, num_tiles, bound)`` for ``dmp_out_fwd_typed`` / ``dmp_bwd_h1_typed``, ``sel`` = the per-edge
    selectors / destinations with the other nodes replaced by -1 (``GraphIndex.edge_select_nodes``)."""

    def __init__(self, mask, rows, tiles, sel, prefix=0):
        self.mask, self.rows, self.tiles, self.sel = mask, rows, tiles, sel
        self.prefix = int(prefix)     # the first ``prefix`` nodes are all kept (the pattern's nodes of a joint pass): list[q] == q there
        self._kinc = None

    def kept_incidence(self, index, e_gate, in_only=False):
        """``(ptr, ent)``: the incidence CSR over the edges the 0 / 1 ``e_gate`` keeps, one row per POSITION of ``rows``
        (``dmp_incidence_keep``): what the backward's endpoint sums walk -- an edge row without a kept endpoint is never
        fetched.  ``in_only``: the in-entries alone = the kept edges' CSR by destination, a row per position (the forward
        aggregation).  Memoised per gate (the layers of a rep-net share it)."""
        owner = _gate_owner(e_gate)
        memo = self._kinc if self._kinc is not None and self._kinc[0] is owner and self._kinc[1] == owner._version else (owner, owner._version, {})
        self._kinc = memo
        hit = memo[2].get(bool(in_only))
        if hit is not None:
            return hit
        lib = _lib.load()
        N, dev = index.num_nodes, index.in_ptr.device
        nscr = int(lib.dmp_csr_keep_scratch_words(N))
        ws = torch.empty(nscr + N + 1 + (1 if in_only else 2) * index.num_edges, dtype=torch.int32, device=dev)
        row_cnt, kptr, kent = ws[:nscr], ws[nscr:nscr + N + 1], ws[nscr + N + 1:]
        check(lib.dmp_incidence_keep(ptr(index.in_ptr), ptr(index.in_ent), None if in_only else ptr(index.out_ptr),
                                     None if in_only else ptr(index.out_ent), ptr(e_gate.reshape(-1)),
                                     ptr(self.rows[0]), ptr(self.rows[1]), N, ptr(row_cnt), ptr(kptr), ptr(kent), stream_ptr()),
              "dmp_incidence_keep")
        memo[2][bool(in_only)] = (kptr, kent)
        return kptr, kent


def node_rows(index, v_gate, H):
    """``NodeRows`` when a layer's node side can run on the kept nodes only: ``v_gate`` is 0 / 1 and its maker multiplied the
    rep-net's input node rows by it (``zero_rows_gate``: the rows under its zeros are zeros in EVERY layer, dmpnn.py:245-277,
    so for such a node v the aggregate A[v], the projections P[v], node_out[v] and every gradient row of v are dead), the
    tile kernels take the shape.  Memoised on the gate (the layers of a rep-net share it).  Else None."""
    if (not USE_NODE_ROWS or not USE_PLAIN_ATB or v_gate is None or not zero_rows_gate(v_gate) or not typed_ok(index, H)
            or index.num_edges == 0):
        return None
    mask = gate_row_mask(v_gate)
    if mask is None:
        return None
    owner = _gate_owner(v_gate)
    hit = getattr(owner, "_dmp_node_rows", None)
    if hit is not None and hit[0] == owner._version and hit[1] is index:
        return hit[2]
    N = index.num_nodes
    lst, cnt = kept_rows(mask, 0, N, tiles=True)
    bound = (N + 31) // 32
    scale = const_zeros(bound, mask.device)
    res = NodeRows(mask, (lst, cnt[0:1]), (lst, scale, cnt[1:2], bound), index.edge_select_nodes(mask),
                   prefix=min(int(getattr(owner, "_dmp_ones_prefix", 0)), N))
    try:
        owner._dmp_node_rows = (owner._version, index, res)
    except Exception:
        pass
    return res


# (measured, round 5: LARGE kernels do not gain from the side stream -- the weight-gradient launches of a layer's backward beside its
# data-gradient chain +2.6 %, the second half of a layer's node side beside its edge side +1.8 % per step: two bandwidth-bound
# launches in flight slow each other down by more than the tails they fill; the partial reductions of a layer's weight gradients
# beside the rest of its backward +1 %, the first layer's pass over dPre / dZn beside the node side's short launches +0.7 %, the first
# layer's input rows (a 119 MB store) on a second side lane beside the layer's first launches +0.6 %.  The side stream carries the small
# index builds only.)
USE_SMALLK_LIST = True      # the first layer's node-code weight gradient over the kept nodes' list (40 -> 30 us)
S0_KEPT_ROWS = True         # the first layer's code sums over the kept nodes' rows only (15 -> 10 us)
USE_L0_NODE_FWD = True     # the first layer's node side from the label codes as one pass (csrc/dmp_layer0.hip::l0_node_fwd_k)
L0_NODE_MAX_COLS = 40      # code columns per node row that kernel holds in registers: VK + 2 K0


def l0_node_pack(VK, K0, H, Mv, Ma, Mb):
    """The packed matrix ``dmp_l0_node_fwd`` reads, [L0_NODE_MAX_COLS, 3H], from its three parts (``Mv`` [VK, 3H] = WV0 Wx, ``Ma`` /
    ``Mb`` [K0, H] = W0 Bn_in / W0 Bn_out); the layer writes the parts in place instead (one small-product launch)."""
    W = torch.zeros((L0_NODE_MAX_COLS, 3 * H), dtype=torch.float32, device=Mv.device)
    W[:VK] = Mv
    W[VK:VK + K0, :H] = Ma
    W[VK + K0:VK + 2 * K0, :H] = Mb
    return W


def l0_node_fwd(venc, VK, S0, K0, Kp, W, bias, slope, mask, n0, n1, H, h1, P, rows=None, q_begin=0):
    """``h1[n0:n1] = act([venc | S0_in | S0_out] W[:, :H] + bias)`` and ``P[n0:n1] = [venc | ..] W[:, H:3H]`` from the node rows'
    label codes and their edges' code sums (``dmp_l0_node_fwd``; ``W``: the packed matrix, ``l0_node_pack``); ``mask``: the
    kept nodes (``NodeRows.mask``) -- a dead node's ``h1`` row is left unwritten, its ``P`` row is zeros; ``rows`` =
    ``NodeRows.rows`` (the kept nodes' list over ALL nodes, with ``mask``): the launch walks the list instead of the masked rows,
    from position ``q_begin`` on (<= n0: e.g. ``NodeRows.prefix``, the leading nodes known to be all kept)."""
    lib = _lib.load()
    lst, cnt = rows if (rows is not None and mask is not None) else (None, None)
    with _lib.timed("l0_node_fwd[K=%d,R=%d]", (VK + 2 * K0, n1 - n0), 4 * (3 * H + VK + 2 * K0) * (n1 - n0)):
        check(lib.dmp_l0_node_fwd(ptr(venc), venc.stride(0), VK, ptr(S0), S0.stride(0), K0, Kp, ptr(W), W.stride(0), ptr(bias), float(slope),
                                  ptr(mask), ptr(lst), ptr(cnt), 0 if lst is None else lst.numel(), min(int(q_begin), n0), n0, n1, H,
                                  ptr(h1), h1.stride(0), ptr(P), P.stride(0), stream_ptr()), "dmp_l0_node_fwd")


def l0_edge_fwd(enc, K, M, P, ldp, bias, coef, index, slope=0.0, rows=None, out=None, mask=None):
    """``act(enc M[:, :H] + coef[dst] enc M[:, H:] + P[a, 0:H] - P[b, H:2H] + bias)`` -- ``edge_fwd_typed`` for input rows
    ``enc W`` of rank K with ``M = W Wes``: no class tiles, rows in their own order.  ``rows = (r0, r1)``: only that range
    of the edge rows (written into ``out[r0:r1]``): rows of another embedding table take another ``M``.  ``mask`` (a
    ``gate_row_mask`` over all E rows): its zero bits mark DEAD output rows, left out entirely."""
    lib = _lib.load()
    E, H = enc.size(0), M.size(1) // 2
    if out is None:
        out = torch.empty((E, H), dtype=torch.float32, device=enc.device)
    r0, r1 = (0, E) if rows is None else rows
    sel_a, sel_b, coef_e = index.edge_select(coef)
    if mask is not None and r0 % 32 != 0:
        mask = None
    if mask is not None and USE_L0_ROW_LISTS and r1 - r0 >= L0_LIST_MIN_ROWS:
        lst, cnt = kept_rows(mask, r0, r1)
        with _lib.timed("l0_edge_fwd[K=%d,E=%d]", (K, r1 - r0), 4 * H * (r1 - r0 + min(2 * index.num_nodes, 2 * (r1 - r0))) + (4 * enc.size(1) + 12) * (r1 - r0)):
            check(lib.dmp_l0_edge_fwd_rows(ptr(enc[r0:]), enc.stride(0), K, ptr(M), M.stride(0), ptr(P), ldp, ptr(bias), ptr(coef_e[r0:]),
                                           ptr(sel_a[r0:]), ptr(sel_b[r0:]), ptr(lst), ptr(cnt), r1 - r0, H, slope, ptr(out[r0:]),
                                           out.stride(0), stream_ptr()), "dmp_l0_edge_fwd_rows")
        return out
    with _lib.timed("l0_edge_fwd[K=%d,E=%d]", (K, r1 - r0), 4 * H * (r1 - r0 + min(2 * index.num_nodes, 2 * (r1 - r0))) + (4 * enc.size(1) + 12) * (r1 - r0)):
        check(lib.dmp_l0_edge_fwd_masked(ptr(enc[r0:]), enc.stride(0), K, ptr(M), M.stride(0), ptr(P), ldp, ptr(bias), ptr(coef_e[r0:]),
                                         ptr(sel_a[r0:]), ptr(sel_b[r0:]), None if mask is None else ptr(mask[r0 // 32:]), r1 - r0, H,
                                         slope, ptr(out[r0:]), out.stride(0), stream_ptr()),
              "dmp_l0_edge_fwd")
    return out


def l0_bwd_w(enc, K, coef_e, d_pre, d_zn=None, rows=None, out=None, mask=None):
    """-> ``[enc^T dPre | (coef_e enc)^T dPre (| enc^T dZn)]`` as one [K, 2H or 3H] matrix: one pass over the gradients
    (over the rows ``rows = (r0, r1)`` only; ``out``: where the reduced sums go).  ``mask`` (``code_row_mask(enc, K)``): the
    gradient rows of all-zero code rows are not fetched."""
    lib = _lib.load()
    E, H = d_pre.shape
    r0, r1 = (0, E) if rows is None else rows
    own = False                          # ``mask`` covers the range alone (bit 0 = row r0), not all E rows
    if mask is not None and r0 % 32 != 0:
        # a row range that starts inside a mask word (a ragged pattern batch: the target's rows begin anywhere): the range's own
        # mask from its code rows.  NOT optional: where the layers leave dead rows unwritten (``dead_rows_buffer``) the rows
        # of ``d_pre`` / ``d_zn`` under a zero gate hold garbage -- their code rows are zeros, so this mask leaves them out
        mask, own = code_row_mask(enc[r0:r1], K), True
    nacc = (3 if d_zn is not None else 2) * K
    G = int(lib.dmp_l0_bwd_w_blocks(r1 - r0))
    part = torch.empty((G, nacc * H), dtype=torch.float32, device=d_pre.device)
    if mask is not None and USE_L0_ROW_LISTS and r1 - r0 >= L0_LIST_MIN_ROWS:
        lst, cnt = kept_rows(mask, 0, r1 - r0) if own else kept_rows(mask, r0, r1)
        with _lib.timed("l0_bwd_w[K=%d,E=%d]", (K, r1 - r0), 4 * (H * (2 if d_zn is not None else 1) + enc.size(1) + 1) * (r1 - r0)):
            check(lib.dmp_l0_bwd_w_rows(ptr(enc[r0:]), enc.stride(0), K, ptr(coef_e[r0:]), ptr(d_pre[r0:]), d_pre.stride(0),
                                        ptr(d_zn[r0:]) if d_zn is not None else None, d_zn.stride(0) if d_zn is not None else 0,
                                        ptr(lst), ptr(cnt), r1 - r0, H, ptr(part), stream_ptr()), "dmp_l0_bwd_w_rows")
        return reduce_partials(part, None if out is None else out.view(-1)).view(K, (nacc // K) * H)
    words = None if mask is None else (mask if own else mask[r0 // 32:])
    with _lib.timed("l0_bwd_w[K=%d,E=%d]", (K, r1 - r0), 4 * (H * (2 if d_zn is not None else 1) + enc.size(1) + 1) * (r1 - r0)):
        check(lib.dmp_l0_bwd_w_masked(ptr(enc[r0:]), enc.stride(0), K, ptr(coef_e[r0:]), ptr(d_pre[r0:]), d_pre.stride(0),
                                      ptr(d_zn[r0:]) if d_zn is not None else None, d_zn.stride(0) if d_zn is not None else 0,
                                      ptr(words), r1 - r0, H, ptr(part), stream_ptr()),
              "dmp_l0_bwd_w")
    return reduce_partials(part, None if out is None else out.view(-1)).view(K, (nacc // K) * H)


class Layer0Codes:
    """What the first layer of a rep-net needs INSTEAD of differentiable input rows, when those rows are label embeddings.
    ``enc`` [E, Kpad]: the packed gated label codes of the union's edge rows (``l0_pack``), ``K`` of them per row;
    ``W`` [K, H]: the embedding table (edge rows = ``enc W``), or [2K, H] = the pattern's table over the target's, the
    pattern's rows being the first ``esplit`` edges / ``nsplit`` nodes.  Optionally the same for the node rows:
    ``venc`` [N, Kvpad], ``VK``, ``WV`` [VK or 2 VK, H]."""

    def __init__(self, enc, K, W, esplit=0, nsplit=0, venc=None, VK=0, WV=None):
        self.enc, self.K, self.W, self.esplit, self.nsplit = enc, K, W, esplit, nsplit
        self.venc, self.VK, self.WV = venc, VK, WV

    def vtables(self, N):
        """-> one ``(table, node rows)`` per table of the node rows' embedding."""
        if self.WV.size(0) == self.VK:
            return [(0, (0, N))]
        return [t for t in ((0, (0, self.nsplit)), (1, (self.nsplit, N))) if t[1][1] > t[1][0]]

    def tables(self, E, N):
        """-> one ``(table, edge rows, node rows)`` per embedding table."""
        if self.W.size(0) == self.K:
            return [(0, (0, E), (0, N))]
        # a table stays in the list while it has edge rows OR node rows (a pattern batch without edges still has nodes, whose
        # aggregates are zero rows): the loops over the list skip the empty range themselves
        return [t for t in ((0, (0, self.esplit), (0, self.nsplit)), (1, (self.esplit, E), (self.nsplit, N)))
                if t[1][1] > t[1][0] or t[2][1] > t[2][0]]


_ATB_JOB = None
MAX_ATB_JOBS = 8


def atb_rows_multi(products, tiles=None):
    """Several ``atb_rows`` products over the SAME rows in ONE launch: ``products`` = list of ``(a, b, gate, colsum[, mask])``
    (a [R, 128 ma], b [R, 128 nb]; ``mask``: a ``gate_row_mask`` whose zero bits mark rows where ``gate (.) a`` or ``b`` is
    known to be all zeros: not fetched); returns a list of ``(a^T b  [128 ma, 128 nb], column sums or None)``.  The launch's
    workgroups are shared by all 128 x 128 output blocks, so short inputs (the node side: R = nodes) get long
    tile ranges per workgroup instead of paying every workgroup's fixed costs once per product.
    ``tiles`` (``NodeRows.tiles``): the products run over that tile list (rows gathered by slot: the kept nodes of a 0 / 1 node
    gate) instead of all rows -- jobs then carry neither gates, column sums nor masks: the list says which rows take part."""
    global _ATB_JOB
    import ctypes
    lib = _lib.load()
    if _ATB_JOB is None:
        P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        _ATB_JOB = type("dmp_atb_job", (ctypes.Structure,), {"_fields_": [("A", P), ("lda", I64), ("B", P), ("ldb", I64), ("gate", P), ("partial", P),
                                                                          ("partial_stride", I64), ("ldp", I), ("partial_colsum", P), ("cs_ld", I),
                                                                          ("rowmask", P)]})
    R = products[0][0].size(0)
    products = [tuple(pr) + (None,) * (5 - len(pr)) for pr in products]
    if (tiles is not None and all(gate is None and not colsum and mask is None for _, _, gate, colsum, mask in products)
            and all(a.size(1) % 128 == 0 and b.size(1) % 128 == 0 and atb2_ok(a, b, 128) for a, b, _, _, _ in products)):
        # over a tile list: 128 x 128 output blocks as jobs of ``dmp_atb2_jobs`` (at most 6 per launch), every product's blocks
        # written into ONE partial of its own shape, so that one fixed-order reduction per product gives the [M, N] result
        blocks = [(pi, ia, ib) for pi, (a, b, _, _, _) in enumerate(products) for ia in range(a.size(1) // 128) for ib in range(b.size(1) // 128)]
        slot, tile_scale, num_tiles, bound = tiles
        res = []
        for i0 in range(0, len(blocks), 6):
            chunk = blocks[i0:i0 + 6]
            G = int(lib.dmp_atb2_blocks(bound, len(chunk)))
            jobs2 = (_Atb2Job * len(chunk))()
            parts2 = {}
            for j, (pi, ia, ib) in enumerate(chunk):
                a, b = products[pi][0], products[pi][1]
                M, N = a.size(1), b.size(1)
                if pi not in parts2:
                    parts2[pi] = torch.empty((G, M * N), dtype=torch.float32, device=a.device)
                jobs2[j].Z, jobs2[j].ldz = a.data_ptr() + 4 * 128 * ia, a.stride(0)
                jobs2[j].D, jobs2[j].ldd = b.data_ptr() + 4 * 128 * ib, b.stride(0)
                jobs2[j].partial_T, jobs2[j].partial_B = parts2[pi].data_ptr() + 4 * (ia * 128 * N + ib * 128), None
                jobs2[j].partial_stride, jobs2[j].ldp = M * N, N
            with _lib.timed("atb2[jobs=%d,R=%d]", (len(chunk), R), 8 * 128 * R * len(chunk)):
                check(lib.dmp_atb2_jobs(jobs2, len(chunk), ptr(slot), ptr(tile_scale), ptr(num_tiles), bound, R, 128, stream_ptr()), "dmp_atb2_jobs")
            res.append(parts2)
        out = []
        for pi, (a, b, _, _, _) in enumerate(products):
            hold = [r_[pi] for r_ in res if pi in r_]
            if len(hold) != 1:      # (a product whose blocks fell into two launches: not with <= 6 blocks; kept simple)
                raise _lib.DmpError("atb_rows_multi(tiles=...): a product's blocks must share one launch")
            out.append((reduce_partials(hold[0]).view(a.size(1), b.size(1)), None))
        return out
    blk = min(atb_block(a.size(1), b.size(1)) for a, b, _, _, _ in products)     # one block size per launch
    nblk = sum((a.size(1) // blk) * (b.size(1) // blk) for a, b, _, _, _ in products)
    if nblk > MAX_ATB_JOBS:
        raise ValueError("atb_rows_multi: more than %d output blocks" % MAX_ATB_JOBS)
    if tiles is not None and any(gate is not None or colsum or mask is not None for _, _, gate, colsum, mask in products):
        raise _lib.DmpError("atb_rows_multi(tiles=...): jobs without gate, column sums or mask")
    G = int(lib.dmp_atb_tile_jobs_blocks(tiles[3], nblk, blk)) if tiles is not None else int(lib.dmp_atb_jobs_blocks_h(R, nblk, blk))
    jobs = (_ATB_JOB * nblk)()
    parts, k = [], 0
    for a, b, gate, colsum, mask in products:
        M, N = a.size(1), b.size(1)
        part = torch.empty((G, M * N), dtype=torch.float32, device=a.device)
        part_cs = torch.empty((G, M), dtype=torch.float32, device=a.device) if colsum else None
        parts.append((part, part_cs, M, N))
        for ia in range(M // blk):
            for ib in range(N // blk):
                j = jobs[k]
                j.A, j.lda, j.B, j.ldb = a.data_ptr() + 4 * blk * ia, a.stride(0), b.data_ptr() + 4 * blk * ib, b.stride(0)
                j.gate = ptr(gate)
                j.rowmask = ptr(mask)
                j.partial, j.partial_stride, j.ldp = part.data_ptr() + 4 * (ia * blk * N + ib * blk), M * N, N
                j.partial_colsum = (part_cs.data_ptr() + 4 * blk * ia) if (colsum and ib == 0) else None
                j.cs_ld = M
                k += 1
    with _lib.timed("atb_rows_multi[blocks=%d,R=%d]", (nblk, R), 0):
        if tiles is not None:
            check(lib.dmp_atb_rows_jobs_h(jobs, nblk, R, blk, ptr(tiles[0]), ptr(tiles[1]), ptr(tiles[2]), tiles[3], stream_ptr()),
                  "dmp_atb_rows_jobs_h")
        else:
            check(lib.dmp_atb_rows_jobs_h(jobs, nblk, R, blk, None, None, None, 0, stream_ptr()), "dmp_atb_rows_jobs_h")
    return [(reduce_partials(part).view(M, N), (reduce_partials(part_cs) if part_cs is not None else None))
            for part, part_cs, M, N in parts]


def atb_block(M, N):
    """Output block size of the MFMA weight-gradient kernel for an [M, N] product: 128, 64 or 0 (not taken)."""
    if M % 128 == 0 and N % 128 == 0:
        return 128
    return 64 if (M % 64 == 0 and N % 64 == 0 and 64 in MFMA_WIDTHS) else 0


def atb_ok(a, b):
    """The MFMA weight-gradient kernel takes this product (else: ``atb``, batched library GEMMs)."""
    return (a.is_cuda and a.dtype == torch.float32 and atb_block(a.size(1), b.size(1)) > 0 and a.size(0) >= 4096
            and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and a.size(0) < 2 ** 31 - 64 and max(a.stride(0), b.stride(0)) * 4 <= 16383)   # rows by index: any array size


def bwd_z_mfma(d_g, Wes, d_s, base, coef, index):
    """base + gather_select(d_s) + dPre Wes[:, :H]^T + coef[dst] dPre Wes[:, H:]^T  (dPre = d_g[:, :H]); H=128."""
    lib = _lib.load()
    E, H = d_g.size(0), d_g.size(1) // 2
    out = torch.empty((E, H), dtype=torch.float32, device=d_g.device)
    Wes = Wes.contiguous()
    d_s = d_s.contiguous()
    with _lib.timed("bwd_z_mfma[H=%d,E=%d]", (H, E), 4 * H * E * (3 if base is not None else 2) + 5 * E):
        check(lib.dmp_bwd_z_fused(ptr(d_g), 2 * H, ptr(Wes), Wes.size(1), ptr(d_s), d_s.size(1), index.num_nodes,
                                  ptr(base), H, ptr(index.edge_select(coef)[2]), ptr(index.dst32), ptr(index.rev8),
                                  -1.0, 1.0, E, H, ptr(out), H, stream_ptr()), "dmp_bwd_z_fused")
    return out


# ----------------------------------------------------------------------------- parameter algebra (fold / unfold)
_FOLD_IN = ("nloop_w", "in_w", "out_w", "nbias", "eloop_w", "src_w", "dst_w", "ebias", "nW0", "nb0", "eW0", "eb0")
_FOLD_AUX_IN = ("nW2", "eW2", "eye")          # forward only: no gradient flows through the transposed copies
_FOLD_OUT = ("Bn", "bn", "Wx", "Wes", "be")
_FOLD_AUX_OUT = ("WesT", "nW2t", "eW2t")
_FOLD_GRAD = ("nloop_w", "in_w", "out_w", "nbias", "eloop_w", "src_w", "dst_w", "ebias", "nW0", "eW0")


def _struct(name, fields):
    import ctypes
    return type(name, (ctypes.Structure,), {"_fields_": [(f, ctypes.c_void_p) for f in fields]})


_LayerWeights = _struct("dmp_layer_weights", _FOLD_IN + _FOLD_AUX_IN)
_LayerFolded = _struct("dmp_layer_folded", _FOLD_OUT + _FOLD_AUX_OUT)
_LayerFoldedGrads = _struct("dmp_layer_folded_grads", ["d" + f for f in _FOLD_OUT])
_LayerWeightGrads = _struct("dmp_layer_weight_grads", _FOLD_GRAD)


def _carve(buf, shapes):
    out, off = [], 0
    for shp in shapes:
        n = 1
        for d in shp:
            n *= d
        out.append(buf[off:off + n].view(shp))
        off += n
    return out


_EYE = {}


class _FoldLayers(torch.autograd.Function):
    """The first MLP Linear of every layer folded into its projections (see include/dmp_hip.h,
    ``dmp_fold_layers``): one launch forward, two backward, for all layers -- instead of a dozen small
    library products, concatenations and additions per layer and direction."""

    @staticmethod
    def forward(ctx, L, *params):
        # per layer: the 12 folded inputs + the two second Linears (nW2, eW2: only their transposes are made here)
        lib = _lib.load()
        params = [p.detach().contiguous() for p in params]
        _lib.require_gpu(*params)
        H = params[0].size(1)
        dev = params[0].device
        eye = _EYE.get((dev.index, H))
        if eye is None:
            eye = _EYE[(dev.index, H)] = torch.eye(H, dtype=torch.float32, device=dev)
        shapes = [(2 * H, H), (H,), (H, 3 * H), (H, 2 * H), (H,), (H, 2 * H), (H, H), (H, H)]
        per = sum(a[0] * (a[1] if len(a) > 1 else 1) for a in shapes)
        buf = torch.empty(L * per, dtype=torch.float32, device=dev)
        W, F, outs, aux = (_LayerWeights * L)(), (_LayerFolded * L)(), [], []
        for l in range(L):
            pl = params[14 * l:14 * l + 14]
            for name, t in zip(_FOLD_IN + _FOLD_AUX_IN, pl + [eye]):
                setattr(W[l], name, ptr(t))
            views = _carve(buf[l * per:(l + 1) * per], shapes)
            for name, t in zip(_FOLD_OUT + _FOLD_AUX_OUT, views):
                setattr(F[l], name, ptr(t))
            outs += views
            aux += views[5:]
        check(lib.dmp_fold_layers(W, F, L, H, stream_ptr()), "dmp_fold_layers")
        ctx.save_for_backward(*params)
        ctx.L, ctx.H, ctx.weights = L, H, W
        ctx.mark_non_differentiable(*aux)
        ctx.set_materialize_grads(False)   # no zero tensors for the (non-differentiable) transposed copies
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        lib = _lib.load()
        params = ctx.saved_tensors
        L, H = ctx.L, ctx.H
        shapes = [(H, H), (H, H), (H, H), (H,), (H, H), (H, H), (H, H), (H,), (H, H), (H, H)]
        per = 8 * H * H + 2 * H
        buf = torch.empty(L * per, dtype=torch.float32, device=params[0].device)
        G, D, out, keep = (_LayerFoldedGrads * L)(), (_LayerWeightGrads * L)(), [None], []
        fshapes = [(2 * H, H), (H,), (H, 3 * H), (H, 2 * H), (H,)]
        for l in range(L):
            gl = []
            for g, shp, name in zip(grads[8 * l:8 * l + 5], fshapes, _FOLD_OUT):
                g = torch.zeros(shp, dtype=torch.float32, device=buf.device) if g is None else g.contiguous()
                setattr(G[l], "d" + name, ptr(g))
                gl.append(g)
            keep.append(gl)
            d = _carve(buf[l * per:(l + 1) * per], shapes)
            for name, t in zip(_FOLD_GRAD, d):
                setattr(D[l], name, ptr(t))
            # order of the inputs: nloop, in, out, nbias, eloop, src, dst, ebias, nW0, nb0 (= dbn), eW0, eb0 (= dbe)
            out += d[:9] + [gl[1], d[9], gl[4], None, None]          # + nW2, eW2: nothing flows back through the transposes
        check(lib.dmp_unfold_layers(ctx.weights, G, D, L, H, stream_ptr()), "dmp_unfold_layers")
        return tuple(out)


def _layer_params(layer):
    n0, e0 = layer.nmlp[0], layer.emlp[0]
    return (layer.nloop_weight, layer.in_weight, layer.out_weight, layer.nbias, layer.eloop_weight, layer.src_weight,
            layer.dst_weight, layer.ebias, n0.weight, n0.bias, e0.weight, e0.bias)


def _layer_aux_params(layer):
    return (layer.nmlp[2].weight, layer.emlp[2].weight)


def fold_layers(layers):
    """-> one ``(Bn, bn, Wx, Wes, be, WesT, nW2t, eW2t)`` per layer (the last three: transposed copies without
    gradient, the layouts the kernels read coalesced).  H = 128 / 64 on the GPU: one HIP launch for all layers (and two for
    their backward); otherwise the same algebra in differentiable torch ops."""
    layers = list(layers)
    H = layers[0].nloop_weight.size(1)
    if H in MFMA_WIDTHS and layers[0].nloop_weight.is_cuda and all(l.nloop_weight.size(1) == H for l in layers):
        flat = _FoldLayers.apply(len(layers), *[p for l in layers for p in _layer_params(l) + _layer_aux_params(l)])
        return [tuple(flat[8 * i:8 * i + 8]) for i in range(len(layers))]
    out = []
    for l in layers:
        nloop, in_w, out_w, nbias, eloop, src_w, dst_w, ebias, nW0, nb0, eW0, eb0 = _layer_params(l)
        Cn = torch.cat([nloop, in_w, out_w, nbias.unsqueeze(0)], dim=0) @ nW0.t()               # [3H+1, H]
        Ce = torch.cat([eloop, src_w - dst_w, dst_w, src_w, ebias.unsqueeze(0)], dim=0) @ eW0.t()  # [4H+1, H]
        Wes = torch.cat([Ce[:H], Ce[H:2 * H]], dim=1)
        nW2, eW2 = _layer_aux_params(l)
        with torch.no_grad():
            aux = (torch.cat([Wes[:, :H].t(), Wes[:, H:].t()], dim=1), nW2.t().contiguous(), eW2.t().contiguous())
        out.append((Cn[H:3 * H], Cn[3 * H] + nb0, torch.cat([Cn[:H], Ce[2 * H:3 * H], Ce[3 * H:4 * H]], dim=1),
                    Wes, Ce[4 * H] + eb0) + aux)
    return out


_GEMM_TYPES = None
USE_SMALL_GEMM_JOBS = True
SMALL_GEMM_MAX_ROWS = 1024     # output rows up to which a product goes into the small-product launch (latency-oriented: a column per thread)


def small_gemm_jobs(jobs):
    """Several small fp32 products in ONE launch (``dmp_small_gemm_jobs``, csrc/dmp_fold.hip).  ``jobs``: list of
    ``(out, terms, addend)`` with ``terms`` a list of ``(A, transA, B, transB)`` -- ``out = sum_t op(A_t) op(B_t) (+ addend)``,
    ``op(X) = X^T`` when the flag is set; operands are 2-D fp32 tensors with unit inner stride (any row stride: column slices
    and ``out`` views of wider tensors are fine), ``out`` must not alias an operand.  At most 12 jobs of at most 3 terms."""
    global _GEMM_TYPES
    import ctypes
    lib = _lib.load()
    if _GEMM_TYPES is None:
        class _Term(ctypes.Structure):
            _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("B", ctypes.c_void_p), ("ldb", ctypes.c_int64),
                        ("transA", ctypes.c_int32), ("transB", ctypes.c_int32), ("K", ctypes.c_int32), ("pad", ctypes.c_int32)]

        class _Job(ctypes.Structure):
            _fields_ = [("term", _Term * 3), ("C0", ctypes.c_void_p), ("ldc0", ctypes.c_int64), ("C", ctypes.c_void_p),
                        ("ldc", ctypes.c_int64), ("num_terms", ctypes.c_int32), ("M", ctypes.c_int32), ("N", ctypes.c_int32),
                        ("pad", ctypes.c_int32)]
        _GEMM_TYPES = (_Term, _Job)
    _, Job = _GEMM_TYPES
    if not 0 < len(jobs) <= 12:
        raise _lib.DmpError("small_gemm_jobs: 1..12 jobs")
    J = (Job * len(jobs))()

    def ld(t):
        if t.dim() != 2 or t.dtype != torch.float32 or (t.size(1) > 1 and t.stride(1) != 1):
            raise _lib.DmpError("small_gemm_jobs: 2-D fp32 operands with unit inner stride")
        return t.stride(0) if t.size(0) > 1 else max(t.size(1), 1)

    keep = []
    for n, (out, terms, addend) in enumerate(jobs):
        M, N = out.shape
        if not 0 < len(terms) <= 3:
            raise _lib.DmpError("small_gemm_jobs: 1..3 terms per job")
        _lib.require_gpu(out, addend, *[x for tm in terms for x in (tm[0], tm[2])])
        for q, (A, ta, B, tb) in enumerate(terms):
            K = A.size(0) if ta else A.size(1)
            if (A.size(1) if ta else A.size(0)) != M or (B.size(1) if tb else B.size(0)) != K or (B.size(0) if tb else B.size(1)) != N:
                raise _lib.DmpError("small_gemm_jobs: shapes of job %d, term %d" % (n, q))
            T = J[n].term[q]
            T.A, T.lda, T.B, T.ldb, T.transA, T.transB, T.K = A.data_ptr(), ld(A), B.data_ptr(), ld(B), int(bool(ta)), int(bool(tb)), K
            keep.append((A, B))
        J[n].num_terms, J[n].M, J[n].N, J[n].C, J[n].ldc = len(terms), M, N, out.data_ptr(), ld(out)
        if addend is not None:
            if addend.shape != out.shape:
                raise _lib.DmpError("small_gemm_jobs: addend shape")
            J[n].C0, J[n].ldc0 = addend.data_ptr(), ld(addend)
    check(lib.dmp_small_gemm_jobs(J, len(jobs), stream_ptr()), "dmp_small_gemm_jobs")


class _FusedDMPLayer(torch.autograd.Function):
    """One DMPNN layer + gate + residual over the folded weights of ``fold_layers``."""

    @staticmethod
    def forward(ctx, index, coef, residual, x, z, v_gate, e_gate, Bn, bn, Wx, Wes, be, nW2, nb2, eW2, eb2,
                WesT=None, nW2t=None, eW2t=None, slope=0.0, vpool=None, epool=None, l0=None, W0=None, WV0=None, edge_rows=True,
                inner=0):
        """``inner``: this layer's place among the layers of a rep-net that share its gates -- bit 0: another layer follows (the
        only reader of the edge rows returned here), bit 1: another layer precedes (the only reader of the edge rows' gradient).
        ``l0`` (``Layer0Codes``) with ``W0 = l0.W`` / ``WV0 = l0.WV`` as differentiable inputs: the FIRST layer of a rep-net
        whose edge rows are a label embedding, ``z = enc W0``.  Every product with ``z`` runs on its K-column factor
        (csrc/dmp_layer0.hip); ``z`` itself is only read as the residual and takes no gradient: the embedding's gradient comes
        back as ``dW0``.  With ``l0.venc`` the same for the node rows ``x = venc WV0`` (products with ``x`` -> ``dWV0``).
        ``edge_rows=False`` (with ``epool``): ``zn`` is not formed (None is returned in its place); its per-graph sums are
        ``pool(zn) = pool(z) + pool(g (.) H1) W2^T + (sum g) b2`` -- two pooled passes over E rows instead of the Linear +
        gate + residual kernel (two reads, one write) and a pooled pass over its output.
        ``vpool`` / ``epool`` (``ops.PoolIndex`` over the node / edge rows; the LAST layer of a rep-net whose outputs
        feed sum / mean pooling heads): two more outputs, the per-graph sums of ``xn`` / ``zn`` ([G, H]; edges with a flag
        [G, 2H] = [non-reversed | reversed]).  A gradient that arrives ONLY through the edge sums is never expanded to
        [E, H]: every row of a graph has the same gradient vector, see ``backward``."""
        _lib.require_gpu(x, z)
        H = Bn.size(1)
        x, z = x.contiguous(), z.contiguous()
        Bn, Wx, Wes = Bn.contiguous(), Wx.contiguous(), Wes.contiguous()
        N = index.num_nodes
        # ---- the nodes a 0 / 1 node gate keeps (``node_rows``): a node under a zero of a gate whose maker wiped the input rows
        # is a zero row in every layer, so its aggregate, its projections, its update and all of its gradient rows are dead --
        # the node side runs on the kept nodes' tiles, the edge kernels read a dead node's (unwritten) rows as zeros
        from . import side      # (index builds issued ahead on the side stream, dmpnn.prefetch_joint_indexes: wait for a stage at its first use)
        side.wait("nodes")
        nd = node_rows(index, v_gate, H) if (N >= 4096 and onepanel_ok(H)) else None
        if nd is not None and l0 is not None and any(n0 % 32 for _, _, (n0, _) in l0.tables(z.size(0), N)):
            nd = None          # (the backward's row masks start at a table's first node)
        # ---- node side (dmpnn.py:113,121,125 + fn.sum + 129-140)
        S0 = M0 = tables = None
        if l0 is not None:     # sum of z over a node's edges = (sum of the label codes) W0
            enc0, K0 = l0.enc, l0.K
            tables = l0.tables(z.size(0), N)
            if nd is not None and S0_KEPT_ROWS and enc0.size(1) % 4 == 0:
                # (the code sums of the kept nodes only: every reader -- the node pass over the kept list, the backward's masked
                # K-column products -- leaves the dead nodes' rows out)
                S0 = ops.seg_sum_raw(enc0, index.in_ptr, index.in_ent, N, None, True, -1.0, 1.0,
                                     out=dead_rows_buffer((N, 2 * enc0.size(1)), z.device), rows=nd.rows, tag="seg_sum2_codes")
            else:
                S0 = ops.seg_sum_raw(enc0, index.in_ptr, index.in_ent, N, None, True, -1.0, 1.0)    # [N, 2 Kpad] = [in | out]
            # S = [S0_in W0 | S0_out W0] is never built:  S Bn = S0_in (W0 Bn_in) + S0_out (W0 Bn_out), two K-column products
            Kp = enc0.size(1)
            vcodes_f = l0.venc is not None
            node_pass = (USE_L0_NODE_FWD and vcodes_f and USE_SMALL_GEMM_JOBS and H in MFMA_WIDTHS and l0.VK + 2 * K0 <= L0_NODE_MAX_COLS
                         and W0.size(0) // K0 <= 2)
            if USE_SMALL_GEMM_JOBS:   # the parameter-only products of this layer in ONE launch: W0 Bn_in, W0 Bn_out, W0 [A | B], WV0 Wx
                TK = W0.size(0)
                M0 = torch.empty((TK, 2 * H), dtype=torch.float32, device=z.device)
                jobs = [(M0, [(W0, False, Wes, False)], None)]
                if node_pass:
                    # ... written straight into the node pass's packed matrix, one per embedding table: rows 0 .. VK-1 = WV0 Wx (three
                    # column blocks), then K0 rows W0 Bn_in and K0 rows W0 Bn_out in the first block; zeros elsewhere
                    VK = l0.VK
                    WP = torch.zeros((TK // K0, L0_NODE_MAX_COLS, 3 * H), dtype=torch.float32, device=z.device)
                    for t in range(TK // K0):
                        WVt = WV0 if WV0.size(0) == VK else WV0[t * VK:(t + 1) * VK]
                        jobs += [(WP[t, :VK], [(WVt, False, Wx, False)], None),
                                 (WP[t, VK:VK + K0, :H], [(W0[t * K0:(t + 1) * K0], False, Bn[:H], False)], None),
                                 (WP[t, VK + K0:VK + 2 * K0, :H], [(W0[t * K0:(t + 1) * K0], False, Bn[H:], False)], None)]
                else:
                    WB = torch.empty((2, TK, H), dtype=torch.float32, device=z.device)
                    jobs += [(WB[0], [(W0, False, Bn[:H], False)], None), (WB[1], [(W0, False, Bn[H:], False)], None)]
                    if vcodes_f:
                        MV = torch.empty((WV0.size(0), 3 * H), dtype=torch.float32, device=z.device)
                        jobs.append((MV, [(WV0, False, Wx, False)], None))
                small_gemm_jobs(jobs)
            else:
                WB = torch.matmul(W0, Bn.view(2, H, H))                                              # [2, T K, H]
            if node_pass:
                # the whole node side before the second Linear in ONE pass over the codes: H1n and the two projection blocks
                # (the kept nodes' H1n rows only; a dead node's projection rows are zeros)
                H1n = dead_rows_buffer((N, H), z.device) if nd is not None else torch.empty((N, H), dtype=torch.float32, device=z.device)
                PX = torch.empty((N, 2 * H), dtype=torch.float32, device=z.device)
                for t, _, (n0, n1) in tables:
                    if n1 > n0 and (nd is None or n1 <= nd.prefix):      # (no gate, or the pattern's nodes -- all kept: every row)
                        l0_node_fwd(l0.venc, VK, S0, K0, Kp, WP[t], bn, slope, None, n0, n1, H, H1n, PX)
                    elif n1 > n0:
                        l0_node_fwd(l0.venc, VK, S0, K0, Kp, WP[t], bn, slope, nd.mask, n0, n1, H, H1n, PX, rows=nd.rows,
                                    q_begin=min(nd.prefix, n0))
            else:
                SB = torch.empty((N, H), dtype=torch.float32, device=z.device)
                for t, _, (n0, n1) in tables:      # (N-row products: the small-product kernel is built for a few hundred rows)
                    if n1 > n0:
                        torch.mm(S0[n0:n1, :K0], WB[0, t * K0:(t + 1) * K0], out=SB[n0:n1])
                        SB[n0:n1].addmm_(S0[n0:n1, Kp:Kp + K0], WB[1, t * K0:(t + 1) * K0])
            if not USE_SMALL_GEMM_JOBS:
                M0 = W0 @ Wes                                                                        # [T K, 2H] = W0 [A | B]
            S = None
        else:
            # (input rows under a zero of a gate whose maker wiped them -- ``_dmp_zero_rows`` -- are zeros: not fetched)
            side.wait("keepcsr")
            kc = keep_in_csr(index, e_gate) if zero_rows_gate(e_gate) else None
            if kc is not None and nd is not None:   # ... and over the kept NODES' rows only: the others' aggregates are dead
                # (measured twice -- before and after the remap fix of finding (u): row pointers by list position,
                # ``NodeRows.kept_incidence(in_only=True)``, one dependent load less per row group, leave the launch where it is)
                S = ops.seg_sum_raw(z, kc[0], kc[1], N, None, True, -1.0, 1.0, out=dead_rows_buffer((N, 2 * H), z.device), rows=nd.rows)
            elif kc is not None:      # the CSR over the kept edges: the plain kernel, no entries of skipped rows in its stream
                S = ops.seg_sum_raw(z, kc[0], kc[1], N, None, True, -1.0, 1.0)
            else:
                S = ops.seg_sum_raw(z, index.in_ptr, index.in_ent, N, e_gate.reshape(-1) if zero_rows_gate(e_gate) else None, True, -1.0, 1.0)
            if nd is None:
                SB = S @ Bn
        if l0 is None and nd is not None:
            # the node side on the kept nodes' tiles (the tile kernel with a plain panel, 128-wide block products; independent
            # products share a launch): H1n = act(x Wx0 + S_in Bn_in + S_out Bn_out + bn) as two partial products + a third
            # that adds them up, the two gathered projection blocks, xn = x + (H1n W2^T + b2) -- nothing is read or written
            # for the other nodes
            T = nd.tiles
            XP = dead_rows_buffer((N, 3 * H), z.device)          # [x Wx0 + bn (a partial) | P_dst | P_src]
            T1 = dead_rows_buffer((N, H), z.device)
            H1n = dead_rows_buffer((N, H), z.device)
            typed_jobs([dict(a=x, W=Wx[:, :H], bias=bn, out=XP[:, :H]), dict(a=S[:, :H], W=Bn[:H], out=T1),
                        dict(a=x, W=Wx[:, H:2 * H], out=XP[:, H:2 * H]), dict(a=x, W=Wx[:, 2 * H:], out=XP[:, 2 * H:])], T)
            out_fwd_typed(S[:, H:], Bn[H:], None, XP[:, :H], T, out=H1n, slope=slope, prev2=T1)
        elif l0 is not None and node_pass:
            pass
        elif l0 is not None and l0.venc is not None:     # x Wx = venc (WV0 Wx): three column blocks of K-column products
            VK = l0.VK
            if not USE_SMALL_GEMM_JOBS:
                MV = WV0 @ Wx                                                                        # [T VK, 3H]
            XP = torch.empty((N, 3 * H), dtype=torch.float32, device=z.device)
            for t, (n0, n1) in l0.vtables(N):
                smallk_embed(l0.venc[n0:n1, :VK], MV[t * VK:(t + 1) * VK], None, XP[n0:n1], H)
        else:
            XP = x @ Wx
        if l0 is not None and node_pass:
            PXr, ldpx = PX, 2 * H                # the gathered projections [P_dst | P_src]
        else:
            PXr, ldpx = XP[:, H:], 3 * H
            if l0 is not None or nd is None:
                H1n = add_bias_relu_(SB, XP[:, :H], bn, slope)
        if nd is not None:
            # (an INNER layer's xn is read by the next layer of this rep-net only, under the same gate: the dead rows stay
            # unwritten; the last layer's rows go to the caller: zeros there)
            xn = dead_rows_buffer((N, H), z.device) if (inner & 1) else torch.zeros((N, H), dtype=torch.float32, device=z.device)
            if nW2t is not None:
                out_fwd_typed(H1n, nW2t, nb2, x if residual else None, nd.tiles, out=xn)
            else:
                out_fwd_typed(H1n, nW2, nb2, x if residual else None, nd.tiles, out=xn, w_in_out=False)
        elif onepanel_ok(H):   # Linear + gate + residual in one fused MFMA kernel, as on the edge side
            xn = out_fwd_mfma(H1n, nW2, nb2, v_gate, x if residual else None, nW2t)
        else:
            xn = gate_residual(x if residual else None, torch.addmm(nb2, H1n, nW2.t()), v_gate)
        # ---- edge side (dmpnn.py:112,120,124 + 142-156)
        sums_only = epool is not None and not edge_rows
        zn = None
        # H1e rows under a zero edge gate are DEAD values when the layer runs on the masked kernels: out_fwd / the pooled
        # passes forward and bwd_h1 / atb_rows / pool_relu_bwd backward all multiply them by that zero and do not fetch them
        # (typed_ok: the backward takes the same branch).  The two producers then leave those rows out.
        side.wait("erows" if l0 is not None else "tiles")      # (a layer on the label codes needs the tiles only for its second Linear)
        dead_gate = e_gate if (SKIP_DEAD_ROWS and USE_ROW_MASKS and e_gate is not None and typed_ok(index, H)
                               and gate_row_mask(e_gate) is not None) else None
        if l0 is not None:
            H1e = dead_rows_buffer((z.size(0), H), z.device) if dead_gate is not None else torch.empty((z.size(0), H), dtype=torch.float32, device=z.device)
            for t, rows, _ in tables:
                if rows[1] > rows[0]:
                    l0_edge_fwd(enc0, K0, M0[t * K0:(t + 1) * K0], PXr, ldpx, be, coef, index, slope, rows, H1e,
                                mask=gate_row_mask(dead_gate) if dead_gate is not None else None)
        elif typed_ok(index, H):
            # (under ``nd`` the projection rows of the dead nodes were never written: their selectors read as zeros)
            H1e = edge_fwd_typed(z, Wes, PXr, ldpx, be, coef, index, slope, dead_gate=dead_gate, sel=None if nd is None else nd.sel)
        elif mfma_ok(index, H):
            H1e = edge_fwd_mfma(z, Wes, PXr, ldpx, be, coef, index, slope)
        else:
            G = z @ Wes
            H1e = edge_combine_raw(G, 2 * H, PXr, ldpx, be, coef, index, H, relu=True, slope=slope)
            del G
            if not sums_only:
                Oe = torch.addmm(eb2, H1e, eW2.t())
                zn = gate_residual(z if residual else None, Oe, e_gate)
        if zn is None and not sums_only:
            # a gate whose maker wiped the rep-net's input rows (zero_rows_gate): z's rows under its zeros are zeros -- not
            # fetched; as an INNER layer of a rep-net (``inner``: the next layer of the same rep-net, under the same gate,
            # is the only reader of zn) the output's zero rows are not stored either
            dead = (3 if (inner & 1) else 1) if (dead_gate is not None and zero_rows_gate(e_gate)) else 0
            side.wait("tiles")
            lt = live_tiles(index, coef, e_gate) if (dead == 3 and USE_TYPED_ROWS and eW2t is not None) else None
            if lt is not None:      # nothing to write for the rows under a zero gate: the kept edges' tiles only
                # (a plain panel needs no class: the kept edges' tiles in ascending row order -- ``ascending_tiles``)
                tl_ = (ascending_tiles(e_gate) if PLAIN_ROWS_ASCENDING else None) or lt
                if l0 is not None and residual and getattr(l0, "z_from_codes", False) and out_codes_ok(H1e, l0.enc, l0.K, W0[-l0.K:]):
                    # the target's residual rows z = enc W0 are not in memory at all (``dmpnn.joint_rep`` did not make them): one more
                    # k-group over their codes; the pattern's few rows (their own table) come in as the residual operand
                    zn = out_fwd_typed_codes(H1e, eW2t, eb2, l0.enc, l0.K, W0[-l0.K:], tl_, prev=l0.z_head, row0=l0.z_head.size(0))
                else:
                    zn = out_fwd_typed(H1e, eW2t, eb2, l0_rows(l0, W0, z) if residual else None, tl_)
            elif (l0 is not None and residual and e_gate is None and getattr(l0, "z_from_codes", False) and eW2t is not None and USE_TYPED_ROWS
                  and out_codes_ok(H1e, l0.enc, l0.K, W0[-l0.K:]) and identity_tiles(H1e.size(0), H1e.device) is not None):
                # no gate at all (the all-rows step): the same launch over the identity tile list
                zn = out_fwd_typed_codes(H1e, eW2t, eb2, l0.enc, l0.K, W0[-l0.K:], identity_tiles(H1e.size(0), H1e.device), prev=l0.z_head,
                                         row0=l0.z_head.size(0))
            else:
                zn = out_fwd_mfma(H1e, eW2, eb2, e_gate, l0_rows(l0, W0, z) if residual else None, eW2t, dead_rows=dead)
        ctx.index, ctx.coef, ctx.residual, ctx.H = index, coef, residual, H
        ctx.v_gate, ctx.e_gate, ctx.WesT, ctx.slope = v_gate, e_gate, WesT, slope
        ctx.vpool, ctx.epool = vpool, epool
        ctx.inner = int(inner)
        ctx.nd = nd
        ctx.l0, ctx.l0_S0, ctx.l0_tables = l0, S0, tables
        ctx.l0_W = None if l0 is None else (W0.detach(), None if WV0 is None else WV0.detach())
        ctx.save_for_backward(x, z if l0 is None else None, S, H1n, H1e, Bn, Wx, Wes, nW2, eW2)
        if vpool is None and epool is None:
            return xn, zn
        ctx.set_materialize_grads(False)    # a missing gradient stays None (the common case: only the sums are used)
        vs = pool_rows(xn, vpool) if vpool is not None else None
        if sums_only:
            halves = 2 if epool.flag8 is not None else 1
            G_ = epool.num_graphs
            Q = pool_rows(H1e, epool, e_gate).view(G_ * halves, H)                      # sum of g H1 per graph (and flag)
            cnt = pool_weight_sums(epool, e_gate)                                        # sum of g per graph (and flag)
            ctx.gcnt = cnt                                                               # the backward's db2 needs them again
            cnt = cnt.view(G_ * halves, 1)
            # (z's rows under a zero of a gate whose maker wiped the input are zeros, and as an inner layer's output not even
            # written: weight 0, not fetched)
            zs = pool_rows(z, epool, e_gate if zero_rows_gate(e_gate) else None).view(G_ * halves, H) if residual else None
            if USE_SMALL_GEMM_JOBS and Q.is_contiguous() and eW2.is_contiguous() and Q.size(0) <= SMALL_GEMM_MAX_ROWS:
                es = torch.empty((G_ * halves, H), dtype=torch.float32, device=Q.device)  # Q W2^T + cnt b2 (+ sum of z): one launch
                small_gemm_jobs([(es, [(Q, False, eW2, True), (cnt, False, eb2.view(1, H), False)], zs)])
            else:
                # (thousands of graphs: the small-product kernel is built for a few hundred rows -- 43 us at 4096 rows against
                # 13 for a library product + one row pass)
                es = torch.addmm(zs, Q, eW2.t()) if residual else Q @ eW2.t()
                es.addcmul_(cnt, eb2.view(1, H))
            # ... of which only the NON-FLAGGED half is returned ([G, H]): without the rows the backward propagates a gradient
            # through that half only (the row map is -1 for flagged rows), so the flagged sums are not handed out as if they
            # were differentiable (the heads mask reversed edges out and read [:, :H] anyway, basemodel.py:1545-1631)
            es = es.view(G_, halves * H)[:, :H]
        else:
            es = pool_rows(zn, epool) if epool is not None else None
        return xn, zn, vs, es

    @staticmethod
    @once_differentiable
    def backward(ctx, dxn, dzn, dvs=None, des=None):
        x, z, S, H1n, H1e, Bn, Wx, Wes, nW2, eW2 = ctx.saved_tensors
        ix, coef, H, slope = ctx.index, ctx.coef, ctx.H, ctx.slope
        N = ix.num_nodes
        nd = ctx.nd       # the kept nodes of a 0 / 1 node gate (``node_rows``): every gradient row of another node is dead
        # ---- gradients through the pooled sums (last layer).  Node side: expanded (N rows are cheap).  Edge side: if the
        # sums are the ONLY consumer of zn, dzn[e] = T[graph of e] (zero for reversed edges) with T = des[:, :H] and
        #   dH1 = (g (.) dzn) W2 = g[e] (T W2)[graph of e]             -> no E-row product, no [E, H] gradient tensor
        #   dW2 = (g (.) dzn)^T H1 = T^T Q,  Q[b] = sum_{e in b} g[e] H1[e]   -> one pooled pass over H1
        #   db2 = T^T (sum_{e in b} g[e])
        # and the residual term of dz reads T through the row map.
        lazy = None
        if dvs is not None:
            vp = ctx.vpool
            exp = ops.gather_rows_raw(dvs.contiguous(), vp.seg32)
            dxn = exp if dxn is None else dxn + exp
        if des is not None:
            ep = ctx.epool
            if dzn is None and typed_ok(ix, H) and H1e.size(0) > 0 and des.size(1) == H:
                # des is [G, H]: the row-less forward hands out the non-flagged sums only, or no flag splits the sums.  A
                # [G, 2H] gradient (forward with rows, flag-split sums) may carry a flagged half: the expanded path below
                lazy = (des.contiguous(), pool_rowmap(ep), ep)
            else:
                if ep.flag8 is not None and des.size(1) == H:     # the row-less forward's [G, H]: nothing flows to the flagged rows
                    des = torch.cat([des, torch.zeros_like(des)], dim=1)
                exp = (ops.gather_select_raw(des.contiguous(), ep.seg32, ep.flag8, H, None, 1.0, 1.0) if ep.flag8 is not None
                       else ops.gather_rows_raw(des.contiguous(), ep.seg32))
                dzn = exp if dzn is None else dzn + exp
        if dxn is None:
            dxn = torch.zeros_like(x)
        if dzn is None and lazy is None:
            dzn = torch.zeros_like(z)
        dxn = dxn.contiguous()
        dzn = dzn.contiguous() if dzn is not None else None
        # the parameter-gradient partials (biases, split-K weight gradients) are consumed after the layer (by the
        # unfold of fold_layers / the optimizer): their reductions run as one launch when this block is left
        with deferred_reductions():
            # ---- edge side, down to the gathered node projections
            mfma, typed = mfma_ok(ix, H), typed_ok(ix, H)
            if lazy is not None:
                T, emap, ep = lazy
                if H % 4 == 0 and H <= 256:
                    # (dPre's rows under a zero of a 0 / 1 gate: every reader below leaves them out -- the kept incidence sums,
                    # the class-tile kernels over the kept edges' tiles)
                    skip = (typed and USE_MASKED_SUMS and SKIP_DEAD_ROWS and ctx.l0 is None and binary_gate_mask(ctx.e_gate) is not None
                            and zero_rows_gate(ctx.e_gate) and live_tiles(ix, coef, ctx.e_gate) is not None and nd is not None
                            and USE_KEPT_INCIDENCE)
                    dG, dbe, Q = pool_relu_bwd(T @ eW2, emap, ctx.e_gate, H1e, ep, slope, skip_dead=skip)       # dG is dPre; one pass over H1e
                    Q = Q[:, :H]
                else:
                    dG, dbe = relu_bwd_gathered_colsum(T @ eW2, emap, ctx.e_gate, H1e, slope)
                    Q = pool_rows(H1e, ep, ctx.e_gate)[:, :H]
                dW2e = T.t() @ Q
                # gated row counts per graph
                gcnt = getattr(ctx, "gcnt", None)
                if gcnt is None:
                    gcnt = pool_weight_sums(ep, ctx.e_gate)
                db2e = gcnt[:, 0] @ T
            elif typed:
                # the gate is applied inside the two consumers of dO = gate * dzn (no [E,H] pass of its own)
                if binary_gate_mask(ctx.e_gate) is not None:
                    # a 0 / 1 gate: bwd_h1 hands out sum_e g_e dzn[e] (db2) from the rows it fetches anyway, and the weight
                    # gradient runs ungated over the masked-in rows on the bf16 pipe
                    # (dPre's zero rows are not stored: the class-tile kernels, the scatter-add and the layer-0 products below
                    # leave them out)
                    skip = (USE_MASKED_SUMS and SKIP_DEAD_ROWS and gate_row_mask(ctx.e_gate) is not None
                            and (ctx.l0 is None or getattr(ctx.l0, "enc_mask", None) is not None))
                    lt = live_tiles(ix, coef, ctx.e_gate) if (skip and USE_TYPED_ROWS) else None
                    dW2e = None
                    if lt is not None and USE_TYPED_ATB_ROWS and h1w_ok(dzn, H1e, H):
                        # both products of the second Linear's backward from ONE pass over dO and H1 (csrc/dmp_h1w.hip)
                        dG, dbe, db2e, dW2e = bwd_h1_w(dzn, eW2, H1e, (ascending_tiles(ctx.e_gate) if PLAIN_ROWS_ASCENDING else None) or lt, slope)
                    elif lt is not None:      # over the kept edges' tiles only
                        dG, dbe, db2e = bwd_h1_typed(dzn, eW2, H1e, (ascending_tiles(ctx.e_gate) if PLAIN_ROWS_ASCENDING else None) or lt, slope)
                    else:
                        dG, dbe, db2e = bwd_h1_mfma(dzn, eW2, H1e, coef, ix, both_halves=False, gate=ctx.e_gate, slope=slope, rows_colsum=True,
                                                    skip_dead_stores=skip)
                    if dW2e is not None:
                        pass
                    elif lt is not None and USE_TYPED_ATB_ROWS:
                        # dO^T H1 over the kept edges' tiles: the class-tile weight-gradient kernel's first half (its class-scaled
                        # second half costs no further products) -- no tiles' worth of zero rows in between
                        dW2e = atb_typed(dzn, H1e, coef, ix, gate=ctx.e_gate, plain=True)
                    else:
                        dW2e = atb_rows(dzn, H1e, ctx.e_gate, colsum=False)[0]
                else:
                    it = identity_tiles(dzn.size(0), dzn.device) if (ctx.e_gate is None and USE_H1W_DENSE and h1w_ok(dzn, H1e, H)) else None
                    if it is not None:
                        # no gate at all (a model without a filter net: the reference's one-label datasets): every row is live --
                        # both products of the second Linear's backward in one launch over the identity tile list
                        dG, dbe, db2e, dW2e = bwd_h1_w(dzn, eW2, H1e, it, slope)
                    else:
                        dW2e, db2e = atb_rows(dzn, H1e, ctx.e_gate)
                        dG, dbe = bwd_h1_mfma(dzn, eW2, H1e, coef, ix, both_halves=False, gate=ctx.e_gate, slope=slope)  # dG is dPre
            else:
                dOe, db2e = scale_rows_colsum(dzn, ctx.e_gate)
                dW2e = atb(dOe, H1e)
                if mfma:
                    dG, dbe = bwd_h1_mfma(dOe, eW2, H1e, coef, ix, both_halves=True, slope=slope)     # dG[:, :H] is dPre
                else:
                    dH1e = dOe @ eW2
                    dG, dbe = relu_bwd_g_colsum(dH1e, H1e, coef, ix.dst32, slope)
                    del dH1e
            # [dPn | dP]: written in place, no concatenation (under ``nd``: the kept nodes' rows only)
            dXP = dead_rows_buffer((N, 3 * H), x.device) if nd is not None else torch.empty((N, 3 * H), dtype=torch.float32, device=x.device)
            # dPre into both endpoints' rows: the backward scatter-add (dPre's rows under a zero edge gate are zeros: not fetched)
            sums_masked = typed and USE_MASKED_SUMS and ctx.e_gate is not None and gate_row_mask(ctx.e_gate) is not None
            # (... and the rows of the dead nodes are neither summed nor stored)
            if nd is not None and sums_masked and USE_KEPT_INCIDENCE and zero_rows_gate(ctx.e_gate) and H % 4 == 0:
                # every kept node's two sums over its KEPT edges (the incidence CSR over the kept edges, a row per kept node):
                # the plain segment-sum kernel, ascending edge id -- an edge row without a kept endpoint is never fetched
                kp, ke = nd.kept_incidence(ix, ctx.e_gate)
                ops.seg_sum_raw(dG[:, :H], kp, ke, N, None, True, 1.0, -1.0, out=dXP[:, H:], rows=nd.rows, ptr_by_pos=True, tag="seg_sum2_kept_inc",
                                incidence=True)
            else:
                nodes = (nd.mask, nd.sel[:2]) if (nd is not None and sums_masked and ops.graph_seg_ok(ix, dG[:, :H], H, dXP[:, H:])) else None
                ops.endpoint_sums(dG[:, :H], ix, out=dXP[:, H:], mask=gate_row_mask(ctx.e_gate) if sums_masked else None,
                                  gate=ctx.e_gate if sums_masked else None, nodes=nodes)
            l0, tables = ctx.l0, ctx.l0_tables
            vcodes = l0 is not None and l0.venc is not None
            if l0 is not None:   # z = enc W0: one pass over dPre (and the residual gradient) on the K-column factor
                W0, WV0 = ctx.l0_W
                K0, TK = l0.K, W0.size(0)
                full = all(r[1] > r[0] and n[1] > n[0] for _, r, n in tables) and len(tables) * K0 == TK   # every table has edge rows and node rows
                XX = (torch.empty if full else torch.zeros)((TK, (3 if ctx.residual else 2) * H), dtype=torch.float32, device=dG.device)
                for t, rows, _ in tables:
                    if rows[1] > rows[0]:
                        # (the rows to add: those with a non-zero code row -- or, where dPre's rows under a zero gate were not
                        # stored, the rows the gate keeps: the same list the forward walked; a kept row with a zero code adds zeros)
                        l0_bwd_w(l0.enc, K0, ix.edge_select(coef)[2], dG, dzn if ctx.residual else None, rows, XX[t * K0:(t + 1) * K0],
                                 mask=(gate_row_mask(ctx.e_gate) if (getattr(l0, "enc_mask", None) is not None and binary_gate_mask(ctx.e_gate) is not None)
                                       else getattr(l0, "enc_mask", None)))
                dWes = None
            else:
                # (dG = dPre has zero rows under a zero edge gate, whichever kernel made it: the typed kernels skip those edges)
                # (with the input gradient still to come -- ``dz`` below, over the same tile list -- both products of dPre run in ONE
                # launch there, ``bwd_z_w``: decided here, made there)
                fuse_dzw = bool(typed and USE_DZW and ctx.needs_input_grad[4] and H == 128 and
                                (ctx.e_gate is None or (SKIP_DEAD_ROWS and zero_rows_gate(ctx.e_gate) and live_tiles(ix, coef, ctx.e_gate) is not None)))
                dWes = None if fuse_dzw else (atb_typed(z, dG, coef, ix, gate=ctx.e_gate) if typed else atb(z, dG))   # [H,2H] = [dA_e | dB_e]
            # ---- node side
            wg = (lambda a, b: atb_rows(a, b, colsum=False)[0]) if atb_ok(x, dXP) else atb   # MFMA kernel or library GEMMs
            one_launch = onepanel_ok(H) and atb_ok(dxn, H1n) and atb_ok(x, dXP) and (l0 is not None or atb_ok(S, dXP))
            if nd is not None and not (one_launch and binary_gate_mask(ctx.v_gate) is not None):
                raise _lib.DmpError("fused layer backward: the kept-node path needs the masked weight-gradient launch")
            if nd is not None:
                # dPn = act'(H1n) (.) (dxn W2) on the kept nodes' tiles; db2n = the column sums of the dxn rows it fetches
                dPn, dbn, db2n = bwd_h1_typed(dxn, nW2, H1n, nd.tiles, slope, out=dXP[:, :H])
                if vcodes:
                    dW2n = atb_rows(dxn, H1n, ctx.v_gate, colsum=False)[0]
            elif onepanel_ok(H) and atb_ok(dxn, H1n):
                # as on the edge side: the node gate lives inside the two consumers of dO = v_gate * dxn
                if binary_gate_mask(ctx.v_gate) is not None:
                    # a 0 / 1 node gate: db2n from the rows bwd_h1 fetches, the weight gradient(s) ungated over the masked-in rows
                    dPn, dbn, db2n = bwd_h1_mfma(dxn, nW2, H1n, both_halves=False, gate=ctx.v_gate, out=dXP[:, :H], slope=slope,
                                                 rows_colsum=True)
                    if not one_launch or vcodes:
                        dW2n = atb_rows(dxn, H1n, ctx.v_gate, colsum=False)[0]
                else:
                    if not one_launch or vcodes:
                        dW2n, db2n = atb_rows(dxn, H1n, ctx.v_gate)
                    dPn, dbn = bwd_h1_mfma(dxn, nW2, H1n, both_halves=False, gate=ctx.v_gate, out=dXP[:, :H], slope=slope)
            else:
                dOn, db2n = scale_rows_colsum(dxn, ctx.v_gate)
                dW2n = wg(dOn, H1n)
                dH1n = dOn @ nW2
                dPn, dbn = relu_bwd_colsum_(dH1n, H1n, out=dXP[:, :H], slope=slope)
            if l0 is not None:
                # S = [S0_in W0 | S0_out W0] was never built: X[h] = S0_h^T dPn per table (N rows, K columns) carries both
                # dBn_h = W0^T X[h] and the segment sums' part of the embedding gradient, sum_h X[h] Bn_h^T (no dS either)
                Kp, S0 = l0.enc.size(1), ctx.l0_S0
                Xn = (torch.empty if full else torch.zeros)((2, TK, H), dtype=torch.float32, device=dPn.device)
                xjobs = [(S0[n0:n1, h * Kp:h * Kp + K0], dPn[n0:n1], Xn[h, t * K0:(t + 1) * K0],
                          nd.mask[n0 // 32:] if nd is not None else None) for t, _, (n0, n1) in tables if n1 > n0 for h in (0, 1)]
                if USE_SMALLK_JOBS and 0 < len(xjobs) <= 4 and dPn.stride(1) == 1 and dPn.stride(0) % 2 == 0 and dPn.data_ptr() % 8 == 0:
                    smallk_atb_jobs(xjobs, H)          # both tables x both halves: one launch
                    xjobs = []
                for t, _, (n0, n1) in (tables if xjobs else []):
                    if n1 > n0:
                        for h in (0, 1):
                            if nd is not None:     # (dPn's rows of the dead nodes were not written: masked out)
                                smallk_atb_cols(S0[n0:n1, h * Kp:h * Kp + K0], dPn[n0:n1], None, Xn[h, t * K0:(t + 1) * K0].unsqueeze(0), H,
                                                mask=nd.mask[n0 // 32:])
                            else:
                                smallk_atb(S0[n0:n1, h * Kp:h * Kp + K0], dPn[n0:n1], out=Xn[h, t * K0:(t + 1) * K0])
                dS = dBn = dWx = None
                if vcodes:
                    # x = venc WV0:  x^T dXP = WV0^T (venc^T dXP)  and  venc^T dx = venc^T dxn + (venc^T dXP) Wx^T -- N-row
                    # passes on K columns instead of two [N,H] x [H,3H] products and the three-block weight gradient
                    VK, TVK = l0.VK, WV0.size(0)
                    vfull = all(n1 > n0 for _, (n0, n1) in l0.vtables(N)) and len(l0.vtables(N)) * VK == TVK
                    Yn = (torch.empty if vfull else torch.zeros)((4 if ctx.residual else 3, TVK, H), dtype=torch.float32, device=dPn.device)
                    vmask = getattr(l0, "venc_mask", None)
                    for t, (n0, n1) in l0.vtables(N):
                        # (the packed node codes carry the node gate: a gated-out node's code row is all zeros)
                        # (one table over all nodes under the node gate: the kept nodes' list -- a node outside it has a zero code row)
                        klist = nd.rows if (nd is not None and n0 == 0 and n1 == N and USE_SMALLK_LIST) else None
                        smallk_atb_cols(l0.venc[n0:n1, :VK], dXP[n0:n1], dxn[n0:n1] if ctx.residual else None,
                                        Yn[:, t * VK:(t + 1) * VK], H,
                                        mask=vmask[n0 // 32:] if (klist is None and vmask is not None and n0 % 32 == 0) else None, rows=klist)
                elif one_launch:
                    vm = binary_gate_mask(ctx.v_gate)
                    if nd is not None and USE_NODE_TILE_ATB:     # over the kept nodes' tiles
                        (dW2n, _), (dWx, _) = atb_rows_multi([(dxn, H1n, None, False), (x, dXP, None, False)], tiles=nd.tiles)
                    elif vm is not None:    # (db2n came from bwd_h1; no job carries a gate: the launch runs on the bf16 pipe)
                        (dW2n, _), (dWx, _) = atb_rows_multi([(dxn, H1n, None, False, vm), (x, dXP, None, False, vm if nd is not None else None)])
                    else:
                        (dW2n, db2n), (dWx, _) = atb_rows_multi([(dxn, H1n, ctx.v_gate, True, gate_row_mask(ctx.v_gate)), (x, dXP, None, False)])
                else:
                    dWx = wg(x, dXP)
            else:
                if nd is not None:     # dS = dPn Bn^T on the kept nodes' tiles; the edge kernel reads a dead node's rows as zeros
                    dS = dead_rows_buffer((N, 2 * H), x.device)
                    jobs = [dict(a=dPn, W=Bn[:H], w_in_out=False, out=dS[:, :H]), dict(a=dPn, W=Bn[H:], w_in_out=False, out=dS[:, H:])]
                    if ctx.needs_input_grad[3]:    # ... and in the same launch two of the three partial products of dx = dxn + dXP Wx^T
                        dxU = dead_rows_buffer((N, 2 * H), x.device)
                        jobs += [dict(a=dXP[:, :H], W=Wx[:, :H], w_in_out=False, prev=dxn if ctx.residual else None, out=dxU[:, :H]),
                                 dict(a=dXP[:, H:2 * H], W=Wx[:, H:2 * H], w_in_out=False, out=dxU[:, H:])]
                    typed_jobs(jobs, nd.tiles)
                else:
                    dS = dPn @ Bn.t()
                if one_launch:   # the three node-side weight gradients (1 + 2 + 3 output blocks) share one launch
                    # (dPn = act'(H1n) ((v_gate dxn) W2): zero rows under a zero node gate -- the first two products skip them)
                    vm = gate_row_mask(ctx.v_gate)
                    if nd is not None and USE_NODE_TILE_ATB:
                        # the three products over the kept nodes' tiles (rows gathered by slot): 41 % of the row tiles at
                        # bench.py's labels, and nothing of a dead node's (unwritten) rows is touched
                        (dW2n, _), (dBn, _), (dWx, _) = atb_rows_multi([(dxn, H1n, None, False), (S, dPn, None, False), (x, dXP, None, False)],
                                                                        tiles=nd.tiles)
                    elif binary_gate_mask(ctx.v_gate) is not None:   # (db2n came from bwd_h1; no job carries a gate: bf16 pipe)
                        # (under ``nd`` x's and dXP's rows of the dead nodes were never written: the third product skips them too)
                        (dW2n, _), (dBn, _), (dWx, _) = atb_rows_multi([(dxn, H1n, None, False, vm), (S, dPn, None, False, vm),
                                                                         (x, dXP, None, False, vm if nd is not None else None)])
                    else:
                        (dW2n, db2n), (dBn, _), (dWx, _) = atb_rows_multi([(dxn, H1n, ctx.v_gate, True, vm), (S, dPn, None, False, vm),
                                                                            (x, dXP, None, False)])
                else:
                    dBn = wg(S, dPn)                                         # [2H,H]
                    dWx = wg(x, dXP)                                         # [H,3H] = [dA_n | dPd | dPs]
            dx = None
            if ctx.needs_input_grad[3] and nd is not None:
                # dx = dxn + dXP Wx^T on the kept nodes' tiles: the third 128-deep block of the contraction + the two partial
                # products made beside dS.  A dead node's row is multiplied by the gate's zero further down: left unwritten
                # where the layer before is the only reader (``inner`` bit 1), zeros for whoever made the first layer's rows
                dx = dead_rows_buffer((N, H), x.device) if (ctx.inner & 2) else torch.zeros((N, H), dtype=torch.float32, device=x.device)
                if l0 is None:
                    out_fwd_typed(dXP[:, 2 * H:], Wx[:, 2 * H:], None, dxU[:, :H], nd.tiles, out=dx, w_in_out=False, prev2=dxU[:, H:])
                else:      # (a first layer on the label codes of its edge rows only: no dS launch to share)
                    for b3 in range(3):
                        out_fwd_typed(dXP[:, b3 * H:(b3 + 1) * H], Wx[:, b3 * H:(b3 + 1) * H], None,
                                      (dxn if ctx.residual else None) if b3 == 0 else dx, nd.tiles, out=dx, w_in_out=False)
            elif ctx.needs_input_grad[3]:
                dx = torch.addmm(dxn, dXP, Wx.t()) if ctx.residual else dXP @ Wx.t()
            # ---- edge side, input gradient: residual + seg_sum2 backward + GEMM, accumulated in place
            dz = None
            if l0 is None and ctx.needs_input_grad[4]:
                # the input gradient of an edge under a zero of a gate that wiped the rep-net's input rows is multiplied by that
                # zero further down: not computed.  After the first layer (``inner`` bit 1) its reader is the layer before,
                # which leaves those rows out; the first layer hands zeros to whoever made the rows.
                dead_dz = (("leave" if (ctx.inner & 2) else "zero") if (typed and SKIP_DEAD_ROWS and zero_rows_gate(ctx.e_gate)) else None)
                dst_m = None if nd is None else nd.sel[2]      # (dS's rows of the dead nodes were not written: read as zeros)
                both = None
                if typed and dWes is None:
                    base_ = (lazy[0] if lazy is not None else dzn) if ctx.residual else None
                    both = bwd_z_w(dG, z, Wes, dS, base_, coef, ix, ctx.WesT, base_map=lazy[1] if (lazy is not None and ctx.residual) else None,
                                   gate=ctx.e_gate, dead_rows=dead_dz, dst=dst_m)
                    if both is None:
                        dWes = atb_typed(z, dG, coef, ix, gate=ctx.e_gate)
                if both is not None:
                    dz, dWes = both
                elif lazy is not None:
                    dz = bwd_z_typed(dG, dG.stride(0), Wes, dS, lazy[0] if ctx.residual else None, coef, ix, ctx.WesT,
                                     base_map=lazy[1] if ctx.residual else None, gate=ctx.e_gate, dead_rows=dead_dz, dst=dst_m)
                elif typed:
                    dz = bwd_z_typed(dG, dG.stride(0), Wes, dS, dzn if ctx.residual else None, coef, ix, ctx.WesT, gate=ctx.e_gate,
                                     dead_rows=dead_dz, dst=dst_m)
                elif mfma:
                    dz = bwd_z_mfma(dG, Wes, dS, dzn if ctx.residual else None, coef, ix)
                else:
                    dz = ops.gather_select_raw(dS, ix.dst32, ix.rev8, H, None, -1.0, 1.0,
                                               base=dzn if ctx.residual else None)
                    dz.addmm_(dG, Wes.t())
        dW0 = dWV0 = None
        if l0 is not None and USE_SMALL_GEMM_JOBS:       # after the reductions of the block above have run: ONE launch
            # dWes = W0^T (enc^T [dPre | c dPre]);  dBn_h = W0^T X_h;  enc^T dz = (enc^T [dPre | c dPre]) [A | B]^T + sum_h X_h Bn_h^T
            # (+ enc^T dzn);  dWx = WV0^T (venc^T dXP) block by block;  venc^T dx = sum_b Y_b Wx_b^T (+ venc^T dxn)
            dev = XX.device
            dWes = torch.empty((H, 2 * H), dtype=torch.float32, device=dev)
            dBn = torch.empty((2 * H, H), dtype=torch.float32, device=dev)
            dW0 = torch.empty((W0.size(0), H), dtype=torch.float32, device=dev)
            jobs = [(dWes, [(W0, True, XX[:, :2 * H], False)], None),
                    (dBn[:H], [(W0, True, Xn[0], False)], None), (dBn[H:], [(W0, True, Xn[1], False)], None),
                    (dW0, [(XX[:, :2 * H], False, Wes, True), (Xn[0], False, Bn[:H], True), (Xn[1], False, Bn[H:], True)],
                     XX[:, 2 * H:] if ctx.residual else None)]
            if vcodes:
                dWx = torch.empty((H, 3 * H), dtype=torch.float32, device=dev)
                dWV0 = torch.empty((WV0.size(0), H), dtype=torch.float32, device=dev)
                for b3 in range(3):
                    jobs.append((dWx[:, b3 * H:(b3 + 1) * H], [(WV0, True, Yn[b3], False)], None))
                jobs.append((dWV0, [(Yn[b3], False, Wx[:, b3 * H:(b3 + 1) * H], True) for b3 in range(3)], Yn[3] if ctx.residual else None))
            small_gemm_jobs(jobs)
        elif l0 is not None:
            dWes = W0.t() @ XX[:, :2 * H]                                  # z^T [dPre | c dPre] = W0^T (enc^T [dPre | c dPre])
            dBn = torch.matmul(W0.t(), Xn).view(2 * H, H)
            T3 = torch.bmm(Xn, Bn.view(2, H, H).transpose(1, 2)).sum(0)
            # enc^T dz = enc^T dzn + (enc^T [dPre | c dPre]) [A | B]^T + (sum of enc)^T dS
            dW0 = torch.addmm(T3 + XX[:, 2 * H:] if ctx.residual else T3, XX[:, :2 * H], Wes.t())
            if vcodes:
                Y = Yn[:3].transpose(0, 1).reshape(WV0.size(0), 3 * H)     # venc^T dXP  [T VK, 3H]
                dWx = WV0.t() @ Y
                dWV0 = torch.addmm(Yn[3], Y, Wx.t()) if ctx.residual else Y @ Wx.t()
        return (None, None, None, dx, dz, None, None, dBn, dbn, dWx, dWes, dbe, dW2n, db2n, dW2e, db2e, None, None, None, None,
                None, None, None, dW0, dWV0, None, None)


def l0_z_from_codes(index, H, eg, l0, live_edges, residual):
    """May ``dmpnn.joint_rep`` leave the target's part of the first layer's edge rows ``z = enc W`` out of memory altogether?  With
    a residual layer and the conditions under which its second Linear runs over the kept edges' tiles (``l0_dead_inputs``' edge
    half) -- or without any edge gate, over all rows -- that launch adds ``enc W`` as one more k-group (``out_fwd_typed_codes``).
    (A layer that ends up elsewhere makes the rows itself: ``l0_rows``.)"""
    return bool(USE_OUT_CODES and (live_edges or (eg is None and USE_OUT_CODES_DENSE and USE_TYPED_ROWS)) and residual and l0 is not None and H == 128
                and not _lib.load().dmp_dev_get_exact_fp32() and l0.enc.stride(0) % 4 == 0 and l0.enc.data_ptr() % 16 == 0
                and index.num_edges * 128 * 4 < (1 << 32) - 65536)


def l0_dead_inputs(index, H, vg, eg, l0):
    """``(edges, nodes)``: may the input rows of a first layer on the label codes (``l0``) stay UNWRITTEN under the zeros of the
    union's 0 / 1 gates?  The layer reads its edge rows ``z`` only as the residual term of its second Linear -- over the kept
    edges' tiles when the conditions below hold (the ones ``_FusedDMPLayer.forward`` tests) -- and its node rows ``x`` (with
    node codes) only as the residual term over the kept nodes' tiles."""
    if l0 is None or not SKIP_DEAD_ROWS or not USE_ROW_MASKS or not USE_TYPED_ROWS or not typed_ok(index, H):
        return False, False
    edges = bool(eg is not None and zero_rows_gate(eg) and gate_row_mask(eg) is not None and USE_LIVE_TILES
                 and getattr(index, "_coef_deg", None) is not None)
    N = index.num_nodes
    nodes = bool(l0.venc is not None and vg is not None and USE_NODE_ROWS and USE_PLAIN_ATB and N >= 4096 and onepanel_ok(H) and zero_rows_gate(vg)
                 and gate_row_mask(vg) is not None and index.num_edges > 0
                 and not any(n0 % 32 for _, _, (n0, _) in l0.tables(index.num_edges, N)))
    return edges, nodes


def activation_slope(act):
    """Negative slope of an MLP activation module the fused path can run: 0.0 for ``nn.ReLU``, ``negative_slope``
    for ``nn.LeakyReLU`` (the reference's default ``leaky_relu``: 1/5.5, utils/act.py:27,466); None otherwise."""
    if type(act) is torch.nn.ReLU:
        return 0.0
    if type(act) is torch.nn.LeakyReLU and 0.0 <= float(act.negative_slope) <= 1.0:
        return float(act.negative_slope)
    return None


def l0_ok(index, H, enc_p, enc_g, W_p, W_g):
    """The first layer can run on the label codes: embedding tables of the same (at most L0_KMAX) number of rows on both
    sides, the class-typed kernels' shape limits (the backward's dPre comes from them)."""
    return (USE_LAYER0 and H in MFMA_WIDTHS and enc_p.size(1) == enc_g.size(1) == W_g.size(0) == W_p.size(0) <= L0_KMAX
            and W_g.size(1) == W_p.size(1) == H and W_g.dtype == W_p.dtype == enc_p.dtype == enc_g.dtype == torch.float32
            and enc_p.stride(1) == 1 and enc_g.stride(1) == 1 and enc_g.is_cuda and index.num_edges > 0 and typed_ok(index, H)
            and onepanel_ok(H))


def l0_nodes_ok(H, enc_p, enc_g, W_p, W_g):
    """The node rows of the first layer can run on their label codes as well (same conditions as the edge rows)."""
    return (enc_p.size(1) == enc_g.size(1) == W_g.size(0) == W_p.size(0) <= SMALLK_MAX and W_g.size(1) == W_p.size(1) == H
            and W_g.dtype == W_p.dtype == enc_p.dtype == enc_g.dtype == torch.float32 and enc_p.stride(1) == 1
            and enc_g.stride(1) == 1 and enc_g.is_cuda and H in MFMA_WIDTHS)


def fused_dmp_layer(index, coef, residual, x, z, v_gate, e_gate, layer, folded=None, pools=None, l0=None, inner=0):
    """``folded``: this layer's entry of ``fold_layers`` (rep-nets fold all their layers in one launch).
    ``pools`` = ``(node PoolIndex or None, edge PoolIndex or None[, edge rows wanted])``: also returns the per-graph sums of
    both outputs; with the third entry False the edge rows themselves are not formed (None in their place).
    ``l0`` (``Layer0Codes``): ``z`` is ``l0.enc @ l0.W`` (and ``x`` is ``l0.venc @ l0.WV``), see ``_FusedDMPLayer.forward``."""
    n2, e2 = layer.nmlp[2], layer.emlp[2]
    Bn, bn, Wx, Wes, be, WesT, nW2t, eW2t = folded if folded is not None else fold_layers([layer])[0]
    vpool, epool = pools[:2] if pools is not None else (None, None)
    edge_rows = pools[2] if pools is not None and len(pools) > 2 else True
    return _FusedDMPLayer.apply(index, coef, bool(residual), x, z, v_gate, e_gate, Bn, bn, Wx, Wes, be,
                                n2.weight, n2.bias, e2.weight, e2.bias, WesT, nW2t, eW2t, activation_slope(layer.nmlp[1]),
                                vpool, epool, l0, None if l0 is None else l0.W, None if l0 is None else l0.WV, edge_rows, int(inner))
