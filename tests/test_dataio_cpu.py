"""The reference's dataset directory layout and run-directory conventions (dataio.py): GML / csv
round trips, the split rule of utils/io.py:177-218, the "best" log lines of utils/log.py:50-76."""
import os

import numpy as np

from dualmessagepassing_amd import dataio
from dualmessagepassing_amd.harness import PairDataset, SyntheticPairs


def test_gml_and_csv_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    n, m = 7, 12
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)          # multigraph with self loops
    vl, el, key = rng.integers(0, 5, n), rng.integers(0, 3, m), rng.integers(0, 2, m)
    path = os.path.join(tmp_path, "g.gml")
    dataio.write_gml(path, n, src, dst, vl, el, key)
    g = dataio.read_gml(path)
    assert g["num_nodes"] == n
    for k, want in (("src", src), ("dst", dst), ("vlabel", vl), ("elabel", el), ("key", key)):
        assert np.array_equal(g[k], want), k
    sub = rng.integers(0, n, (4, 3))
    cpath = os.path.join(tmp_path, "P_0.csv")
    dataio.write_metadata_csv(cpath, [("G_3", 4, sub), ("G_4", 0, np.zeros((0, 3), np.int64))])
    meta = dataio.read_metadata_csv(cpath)
    assert meta["G_3"]["counts"] == 4 and np.array_equal(meta["G_3"]["subisomorphisms"], sub)
    assert meta["G_4"]["counts"] == 0 and meta["G_4"]["subisomorphisms"].size == 0


def test_dataset_directory_round_trip_and_split_rule(tmp_path):
    ds = SyntheticPairs(24, 3, 2, 7, 14, 2, 2, seed=9)
    root = str(tmp_path)
    ds.to_files(root)
    assert sorted(os.listdir(root)) == ["graphs", "metadata", "patterns"]
    data, shared = dataio.load_data(os.path.join(root, "patterns"), os.path.join(root, "graphs"), os.path.join(root, "metadata"))
    assert not shared
    ids = {k: sorted(int(x["id"].rsplit("_", 1)[-1]) for x in v) for k, v in data.items()}
    assert ids["dev"] == [0, 10, 20] and ids["test"] == [1, 11, 21]
    assert ids["train"] == [i for i in range(24) if i % 10 > 1]
    back = PairDataset.from_loaded(sorted(sum(data.values(), []), key=lambda x: int(x["id"].rsplit("_", 1)[-1])))
    assert back.shape == ds.shape
    for a, b in zip(ds.samples, back.samples):
        assert a["counts"] == b["counts"] and np.array_equal(a["subisomorphisms"], b["subisomorphisms"])
        for side in ("pattern", "graph"):
            for k in ("src", "dst", "vlabel", "elabel", "eid", "rev"):
                assert np.array_equal(a[side][k], b[side][k]), (side, k)
    # explicit index files override the modulo rule (utils/io.py:150-161)
    with open(os.path.join(root, "metadata", "train.txt"), "w") as f:
        f.write("0\n1\n2\n")
    data2, _ = dataio.load_data(os.path.join(root, "patterns"), os.path.join(root, "graphs"), os.path.join(root, "metadata"))
    assert sorted(int(x["id"].rsplit("_", 1)[-1]) for x in data2["train"]) == [0, 1, 2] and len(data2["dev"]) == 3


def test_shared_graph_layout_uses_modulo_three(tmp_path):
    ds = SyntheticPairs(6, 3, 2, 6, 10, 1, 1, seed=2)
    for i, s in enumerate(ds.samples):
        s["id"] = "P_0-G_%d" % i
        s["pattern"] = ds.samples[0]["pattern"]
    ds.to_files(str(tmp_path), shared_graph=True)
    data, shared = dataio.load_data(os.path.join(tmp_path, "patterns"), os.path.join(tmp_path, "graphs"), os.path.join(tmp_path, "metadata"))
    assert shared
    assert sorted(x["id"] for x in data["train"]) == ["P_0-G_2", "P_0-G_5"]
    assert sorted(x["id"] for x in data["dev"]) == ["P_0-G_0", "P_0-G_3"]


def test_best_lines_round_trip(tmp_path):
    path = os.path.join(tmp_path, "log.txt")
    with open(path, "w") as f:
        f.write(dataio.best_line("dev", 3, 10, **{"eval-MAE": "1.25000"}) + "\n")
        f.write("data_type: dev\tepoch: 004/010\teval-MAE: 1.1\n")
        f.write(dataio.best_line("dev", 4, 10, **{"eval-MAE": "1.10000"}) + "\n")
    assert dataio.get_best_epochs(path) == {"eval-MAE": {"dev": (4, 1.1)}}
    assert dataio.checkpoint_path("run", 4) == os.path.join("run", "epoch4.pt")
