"""The reference's on-disk dataset layout and run directory conventions, host side.

Layout (``SubgraphCountingMatching/utils/io.py:43-218``): ``patterns/*.gml`` and ``graphs/**/*.gml``
(igraph GML: vertices ``id, label``; edges ``source, target, label, key``), ``metadata/<pattern>.csv``
with the columns ``g_id, counts, subisomorphisms`` (the last one a Python list literal), optional
``train.txt / dev.txt / test.txt`` index files, otherwise the split by the graph's trailing number:
``% 10`` (> 1 train, 0 dev, 1 test) when every pattern has its own graph directory, ``% 3``
(> 1, 0, 1) when the patterns share the graphs.  Run directory (``train.py:1088,1334-1340``,
``utils/log.py:50-76``): ``config.json``, ``epoch%d.pt`` state dicts, ``log.txt`` with "best" lines.

igraph is not available offline, so the GML reader below parses the subset of GML that igraph
writes for such graphs; the format is pinned by round trips only (no reference reader to run).
"""
import csv
import json
import os
import re
from collections import OrderedDict

import numpy as np

csv.field_size_limit(500 * 1024 * 1024)


# ----------------------------------------------------------------------------- GML
def write_gml(path, num_nodes, src, dst, vlabel, elabel, key=None):
    """One directed multigraph in the GML dialect igraph writes (``Graph.write_gml``)."""
    key = np.zeros(len(src), np.int64) if key is None else key
    with open(path, "w") as f:
        f.write("Creator \"dualmessagepassing_amd\"\nVersion 1\ngraph\n[\n  directed 1\n")
        for i in range(num_nodes):
            f.write("  node\n  [\n    id %d\n    label \"%d\"\n  ]\n" % (i, int(vlabel[i])))
        for u, v, l, k in zip(src, dst, elabel, key):
            f.write("  edge\n  [\n    source %d\n    target %d\n    label \"%d\"\n    key %d\n  ]\n" % (int(u), int(v), int(l), int(k)))
        f.write("]\n")


_TOKEN = re.compile(r'"[^"]*"|\[|\]|[^\s\[\]]+')


def read_gml(path):
    """-> dict(num_nodes, src, dst, vlabel, elabel, key) as int64 arrays; node ids are mapped to
    0..n-1 in file order (``ig.read`` numbers vertices in file order, utils/io.py:50-54)."""
    with open(path) as f:
        tokens = _TOKEN.findall(f.read())
    pos = 0

    def parse_block():
        nonlocal pos
        items = []
        while pos < len(tokens) and tokens[pos] != "]":
            k = tokens[pos]
            pos += 1
            if tokens[pos] == "[":
                pos += 1
                items.append((k, parse_block()))
                pos += 1  # the closing bracket
            else:
                items.append((k, tokens[pos].strip('"')))
                pos += 1
        return items

    top = dict((k, v) for k, v in parse_block() if k == "graph")
    if "graph" not in top:
        raise ValueError("%s: no graph block" % path)
    nodes, edges = [], []
    for k, v in top["graph"]:
        if k == "node":
            nodes.append(dict(v))
        elif k == "edge":
            edges.append(dict(v))
    ids = {int(float(nd["id"])): i for i, nd in enumerate(nodes)}
    geti = lambda d, k, dflt=0: int(float(d.get(k, dflt)))
    return {"num_nodes": len(nodes),
            "vlabel": np.array([geti(nd, "label") for nd in nodes], np.int64),
            "src": np.array([ids[geti(e, "source")] for e in edges], np.int64),
            "dst": np.array([ids[geti(e, "target")] for e in edges], np.int64),
            "elabel": np.array([geti(e, "label") for e in edges], np.int64),
            "key": np.array([geti(e, "key") for e in edges], np.int64)}


# ----------------------------------------------------------------------------- metadata
def write_metadata_csv(path, rows):
    """rows: iterable of (g_id, counts, subisomorphisms [counts, pattern_nodes])."""
    with open(path, "w", newline="") as f:
        w = csv.writer(f, delimiter=",")
        w.writerow(["g_id", "counts", "subisomorphisms"])
        for g_id, counts, sub in rows:
            w.writerow([g_id, int(counts), str(np.asarray(sub, np.int64).tolist())])


def read_metadata_csv(path):
    """utils/io.py:99-115."""
    meta = {}
    with open(path, newline="") as f:
        r = csv.reader(f, delimiter=",")
        header = next(r)
        gi, ci, si = header.index("g_id"), header.index("counts"), header.index("subisomorphisms")
        for row in r:
            sub = np.asarray(json.loads(row[si]), dtype=np.int64)
            meta[row[gi]] = {"counts": int(row[ci]), "subisomorphisms": sub}
    return meta


# ----------------------------------------------------------------------------- dataset directory
def _read_dir(dirpath):
    out = {}
    for name in sorted(os.listdir(dirpath)):
        full = os.path.join(dirpath, name)
        if os.path.isfile(full) and name.endswith(".gml"):
            out[os.path.splitext(name)[0]] = read_gml(full)
    return out


def _indices(metadata_dir, name):
    path = os.path.join(metadata_dir, name + ".txt")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return set(int(x) for x in f if x.strip())


def load_data(pattern_dir, graph_dir, metadata_dir):
    """``utils/io.py:145-218``: -> (OrderedDict(train / dev / test lists of samples), shared_graph).
    A sample: ``{"id": "<pattern>-<graph>", "pattern": g, "graph": g, "counts", "subisomorphisms"}``."""
    patterns = _read_dir(pattern_dir)
    sub = [d for d in sorted(os.listdir(graph_dir)) if os.path.isdir(os.path.join(graph_dir, d))]
    graphs = {d: _read_dir(os.path.join(graph_dir, d)) for d in sub}
    graphs.update(_read_dir(graph_dir))
    meta = {os.path.splitext(n)[0]: read_metadata_csv(os.path.join(metadata_dir, n))
            for n in sorted(os.listdir(metadata_dir)) if n.endswith(".csv")}
    fixed = {k: _indices(metadata_dir, k) for k in ("train", "dev", "test")}
    data = OrderedDict((("train", []), ("dev", []), ("test", [])))
    shared_graph = True
    for p, pattern in patterns.items():
        own = p in graphs and isinstance(graphs[p], dict) and "num_nodes" not in graphs[p]
        if own:
            shared_graph = False
        pool = graphs[p] if own else {g: v for g, v in graphs.items() if isinstance(v, dict) and "num_nodes" in v}
        mod = 10 if own else 3
        for g, graph in pool.items():
            x = {"id": "%s-%s" % (p, g), "pattern": pattern, "graph": graph,
                 "subisomorphisms": meta[p][g]["subisomorphisms"], "counts": meta[p][g]["counts"]}
            g_idx = int(g.rsplit("_", 1)[-1])
            rule = {"train": g_idx % mod > 1, "dev": g_idx % mod == 0, "test": g_idx % mod == 1}
            for split in data:
                if (g_idx in fixed[split]) if fixed[split] is not None else rule[split]:
                    data[split].append(x)
    return data, shared_graph


def save_pairs(root, samples, shared_graph=False):
    """Write (pattern, graph) samples in the layout ``load_data`` reads: every distinct pattern once,
    its graphs under ``graphs/<pattern>/`` (or ``graphs/`` when shared), one csv per pattern.
    ``samples``: dicts with ``pattern`` / ``graph`` (num_nodes, src, dst, vlabel, elabel),
    ``pattern_id``, ``graph_id`` (ending in ``_<number>``), ``counts``, ``subisomorphisms``."""
    for d in ("patterns", "graphs", "metadata"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    rows = {}
    for s in samples:
        p, g = s["pattern_id"], s["graph_id"]
        ppath = os.path.join(root, "patterns", p + ".gml")
        if not os.path.exists(ppath):
            write_gml(ppath, **{k: s["pattern"][k] for k in ("num_nodes", "src", "dst", "vlabel", "elabel")})
        gdir = os.path.join(root, "graphs") if shared_graph else os.path.join(root, "graphs", p)
        os.makedirs(gdir, exist_ok=True)
        gpath = os.path.join(gdir, g + ".gml")
        if not os.path.exists(gpath):
            write_gml(gpath, **{k: s["graph"][k] for k in ("num_nodes", "src", "dst", "vlabel", "elabel")})
        rows.setdefault(p, []).append((g, s["counts"], s["subisomorphisms"]))
    for p, r in rows.items():
        write_metadata_csv(os.path.join(root, "metadata", p + ".csv"), r)


# ----------------------------------------------------------------------------- run directory
def save_config(config, path):
    with open(path, "w") as f:
        json.dump(dict(config), f)


def load_config(path):
    with open(path) as f:
        return json.load(f)


def checkpoint_path(save_dir, epoch):
    return os.path.join(save_dir, "epoch%d.pt" % epoch)


def best_line(data_type, epoch, total_epochs, **kw):
    """utils/log.py:50-57."""
    return "\t".join(["data_type: " + str(data_type)] + ["best %s: %s" % (k, v) for k, v in kw.items()]
                     + ["(epoch: %d/%d)" % (epoch, total_epochs)])


_BEST = re.compile(r"data_type:\s+(\w+)\s+best\s+([a-zA-Z0-9\.\-\+\_]+):\s+([a-zA-Z0-9\.\-\+\_]+)\s+\(epoch:\s+(\d+)/\d+\)")


def get_best_epochs(log_file):
    """utils/log.py:60-76: {metric: {data_type: (epoch, value)}} from the "best" lines of a log."""
    best = {}
    with open(log_file) as f:
        for line in f:
            for dt, name, val, ep in _BEST.findall(line):
                best.setdefault(name, {})[dt] = (int(ep), float(val))
    return best
