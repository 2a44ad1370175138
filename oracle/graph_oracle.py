"""ORACLE -- TEST INFRASTRUCTURE.  numpy/ctypes front end of oracle/graph_oracle.c
(sequential C restatement of the reference's integer graph transforms) plus the
frame handling of ``convert_to_dual_graph`` (utils/graph.py:74-169) in numpy."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgraph_oracle.so")
        src = os.path.join(_HERE, "graph_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        _LIB = ctypes.CDLL(path)
        _LIB.orc_line_graph_plain.restype = ctypes.c_int64
        _LIB.orc_line_graph_id.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def csr_build(key, flag, n):
    key = _i64(key)
    e = len(key)
    f = None if flag is None else np.ascontiguousarray(flag, dtype=np.uint8)
    ptr = np.zeros(n + 1, np.int32)
    ent = np.zeros(max(e, 1), np.int32)
    lib().orc_csr_build(_p(key), _p(f), ctypes.c_int64(e), ctypes.c_int64(n), _p(ptr), _p(ent))
    return ptr, ent[:e]


def add_reversed_edges(src, dst, eid, el, max_ne, max_nel):
    src, dst, eid, el = map(_i64, (src, dst, eid, el))
    e = len(src)
    o = [np.zeros(2 * e, np.int64) for _ in range(4)]
    rev = np.zeros(2 * e, np.uint8)
    lib().orc_add_reversed_edges(_p(src), _p(dst), _p(eid), _p(el), ctypes.c_int64(e), ctypes.c_int64(max_ne),
                                 ctypes.c_int64(max_nel), _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(rev))
    return o[0], o[1], o[2], o[3], rev.astype(bool)


def collate(local_src, local_dst, num_nodes, num_edges):
    ls, ld, nn, ne = map(_i64, (local_src, local_dst, num_nodes, num_edges))
    b, e, n = len(nn), len(ls), int(nn.sum())
    node_off, edge_off = np.zeros(b + 1, np.int64), np.zeros(b + 1, np.int64)
    src, dst = np.zeros(e, np.int64), np.zeros(e, np.int64)
    eg, ng = np.zeros(max(e, 1), np.int32), np.zeros(max(n, 1), np.int32)
    lib().orc_collate(_p(ls), _p(ld), _p(nn), _p(ne), ctypes.c_int64(b), _p(node_off), _p(edge_off), _p(src),
                      _p(dst), _p(eg), _p(ng))
    return src, dst, node_off, edge_off, eg[:e], ng[:n]


def eigen_bounds(src, dst, n):
    src, dst = _i64(src), _i64(dst)
    a, b = ctypes.c_float(), ctypes.c_float()
    lib().orc_eigen_bounds(_p(src), _p(dst), ctypes.c_int64(len(src)), ctypes.c_int64(n), ctypes.byref(a),
                           ctypes.byref(b))
    return a.value, b.value


def convert_to_dual_graph(src, dst, n, ndata, edata):
    """One graph.  ndata / edata: dict name -> numpy array.  Returns
    (dual_src, dual_dst, dual_num_nodes, dual_ndata, dual_edata) like the reference's
    DGL branch (utils/graph.py:75-169)."""
    src, dst = _i64(src), _i64(dst)
    e = len(src)
    cap = int((np.bincount(dst, minlength=n)[src]).sum()) if e else 0
    dsrc, ddst, pay = (np.zeros(max(cap, 1), np.int64) for _ in range(3))
    if "id" in edata and e > 0:
        eid = _i64(edata["id"])
        k = int(eid.max()) + 1
        if "label" not in ndata:
            raise NotImplementedError("edge ids without node labels")
        first = np.zeros(k, np.int64)
        nk = ctypes.c_int64()
        m = lib().orc_line_graph_id(_p(src), _p(dst), _p(eid), _p(_i64(ndata["label"])), ctypes.c_int64(e),
                                    ctypes.c_int64(n), ctypes.c_int64(k), _p(first), _p(dsrc), _p(ddst), _p(pay),
                                    ctypes.byref(nk))
        rep = first[first >= 0]
        dual_ndata = {key: np.asarray(v)[rep] for key, v in edata.items()}   # graph.py:90-95 + :161-164
        dn = nk.value
    else:
        m = lib().orc_line_graph_plain(_p(src), _p(dst), ctypes.c_int64(e), ctypes.c_int64(n), _p(dsrc), _p(ddst),
                                       _p(pay))
        dual_ndata = {key: np.asarray(v) for key, v in edata.items()}        # graph.py:99-101
        if "id" not in edata:
            dual_ndata["id"] = np.arange(e, dtype=np.int64)                  # graph.py:102-103
        dn = e
    dsrc, ddst, pay = dsrc[:m], ddst[:m], pay[:m]
    dual_edata = {key: np.asarray(v)[pay] for key, v in ndata.items()}      # graph.py:139-140
    if "id" not in ndata:
        dual_edata["id"] = pay.copy()                                       # graph.py:141-142
    return dsrc, ddst, dn, dual_ndata, dual_edata


def subiso_node_weights(sub, num_nodes):
    sub = _i64(sub)
    rows, width = (sub.shape if sub.ndim == 2 else (0, 1))
    w = np.zeros(max(num_nodes, 1), np.int64)
    lib().orc_subiso_node_weights(_p(sub), ctypes.c_int64(rows), ctypes.c_int64(width), ctypes.c_int64(num_nodes), _p(w))
    return w[:num_nodes]


def subiso_edge_weights(p_u, p_v, p_el, g_u, g_v, g_el, sub):
    p_u, p_v, p_el, g_u, g_v, g_el, sub = map(_i64, (p_u, p_v, p_el, g_u, g_v, g_el, sub))
    rows, width = (sub.shape if sub.ndim == 2 else (0, 1))
    w = np.zeros(max(len(g_u), 1), np.int64)
    lib().orc_subiso_edge_weights(_p(p_u), _p(p_v), _p(p_el), ctypes.c_int64(len(p_u)), _p(g_u), _p(g_v), _p(g_el),
                                  ctypes.c_int64(len(g_u)), _p(sub), ctypes.c_int64(rows), ctypes.c_int64(width), _p(w))
    return w[:len(g_u)]


def pre_pad(rows, pad=0):
    """``batch_convert_tensor_to_tensor(rows, pre_pad=True)`` (utils/dl.py:89-110) for 1-D rows."""
    m = max(len(r) for r in rows)
    out = np.full((len(rows), m), pad, dtype=rows[0].dtype)
    for i, r in enumerate(rows):
        if len(r):
            out[i, m - len(r):] = r
    return out


def batch_subiso_weights(d):
    """Whole fixture (tests/golden/subiso_weights_*.npz layout) -> padded node / edge weights."""
    nw, ew = [], []
    po, go, pn, gn = 0, 0, d["p_num_nodes"], d["g_num_nodes"]
    for i in range(len(pn)):
        pe, ge = int(d["p_num_edges"][i]), int(d["g_num_edges"][i])
        sub = d["sub_flat"][d["sample_ptr"][i]:d["sample_ptr"][i + 1]].reshape(-1, int(pn[i]))
        nw.append(subiso_node_weights(sub, int(gn[i])))
        ew.append(subiso_edge_weights(d["p_src"][po:po + pe], d["p_dst"][po:po + pe], d["p_elabel"][po:po + pe],
                                      d["g_src"][go:go + ge], d["g_dst"][go:go + ge], d["g_elabel"][go:go + ge], sub))
        po, go = po + pe, go + ge
    return pre_pad(nw), pre_pad(ew)


def dual_subisomorphisms(p_u, p_v, p_el, g_u, g_v, g_el, sub):
    """``get_dual_subisomorphisms`` (utils/graph.py:277-316), plain Python (small cases).  ``g_u, g_v, g_el``: the graph's
    edges sorted by (src, dst), as ``all_edges(order="srcdst")`` hands them over (train.py:426-428).  Runs of consecutive
    pattern edges with an equal (u, v) key form one entry -- a later run of the same key REPLACES the earlier one but keeps
    its position (dict semantics, graph.py:293-300) --; entry k of a row receives the index (in the sorted order) of the
    LAST graph edge between the mapped endpoints whose label occurs in the run; entries beyond the number of keys stay 0."""
    p_u, p_v, p_el = _i64(p_u), _i64(p_v), _i64(p_el)
    g_u, g_v, g_el = _i64(g_u), _i64(g_v), _i64(g_el)
    sub = np.asarray(sub, dtype=np.int64).reshape(-1, int(max(p_u.max(), p_v.max())) + 1 if len(sub) == 0 else np.asarray(sub).shape[1])
    out = np.zeros((len(sub), len(p_el)), dtype=np.int64)
    runs = {}
    i = 0
    while i < len(p_el):
        j = i + 1
        while j < len(p_el) and (p_u[i], p_v[i]) == (p_u[j], p_v[j]):
            j += 1
        runs[(int(p_u[i]), int(p_v[i]))] = [int(x) for x in p_el[i:j]]
        i = j
    for r, row in enumerate(sub):
        for k, ((u, v), labels) in enumerate(runs.items()):
            mu, mv = int(row[u]), int(row[v])
            for idx in range(len(g_el)):
                if g_u[idx] == mu and g_v[idx] == mv and int(g_el[idx]) in labels:
                    out[r, k] = idx
    return out


# ----------------------------------------------------------------------------- UNC samplers (utils.py:279-349)
def rng_hash(seed, a, b):
    """The counter-based generator of csrc/dmp_graph.hip (``rng_hash``): a 32-bit mix of (seed, a, b), vectorised."""
    m = np.uint64(0xFFFFFFFF)
    seed, a, b = (np.asarray(v, dtype=np.uint64) & m for v in (seed, a, b))
    t = (b * np.uint64(0x85EBCA77)) & m
    x = seed ^ ((a * np.uint64(0x9E3779B1)) & m) ^ (((t << np.uint64(15)) | (t >> np.uint64(17))) & m)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x


def fold_seed(seed):
    seed = int(seed) & (2 ** 64 - 1)
    return (seed ^ (seed >> 32)) & 0xFFFFFFFF


def random_walks(src, dst, n, seeds, walks, depth, seed):
    """``dgl.sampling.random_walk`` semantics (uniform out-edge per step, -1 after a node without out-edges) with the
    product's generator: walk t takes, at step k, the out-edge number (hash(seed, t, k) * out_degree) >> 32 of the
    current node's out-edges in ascending edge id.  Returns traces [len(seeds) * walks, depth + 1]."""
    src, dst = _i64(src), _i64(dst)
    out = [[] for _ in range(n)]
    for e in range(len(src)):
        out[src[e]].append(e)
    s32 = fold_seed(seed)
    traces = np.full((len(seeds) * walks, depth + 1), -1, np.int64)
    for t in range(len(seeds) * walks):
        cur = int(seeds[t // walks])
        traces[t, 0] = cur
        for k in range(1, depth + 1):
            if cur < 0 or not out[cur]:
                cur = -1
            else:
                h = int(rng_hash(s32, t, k))
                cur = int(dst[out[cur][(h * len(out[cur])) >> 32]])
            traces[t, k] = cur
    return traces


def sample_in_edges(dst, n, wanted, width, seed):
    """``dgl.sampling.sample_neighbors(edge_dir="in")`` semantics (all in-edges of a wanted node if it has at most
    ``width``, else ``width`` of them uniformly without replacement) with the product's keys: the ``width`` smallest
    (hash(seed, edge id, 0), edge id).  Returns a bool mask over the edges."""
    dst = _i64(dst)
    s32 = fold_seed(seed)
    mask = np.zeros(len(dst), bool)
    keys = rng_hash(s32, np.arange(len(dst)), 0)
    for v in range(n):
        if not wanted[v]:
            continue
        es = np.nonzero(dst == v)[0]
        if len(es) > width:
            es = sorted(es, key=lambda e: (int(keys[e]), int(e)))[:width]
        mask[es] = True
    return mask
