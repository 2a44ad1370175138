"""The committed fixtures are what the reference's code produces TODAY.

``oracle/make_golden.py`` imports the reference from /root/reference (development container only; the GPU box does not
have it, so the test skips there), runs single-threaded with deterministic algorithms and must reproduce every array of
``tests/golden/*.npz``: integers and flags bit for bit, floats to 1e-6 of the array's largest value.  A generator edit
that silently changes a fixture -- or a fixture edited by hand -- fails here."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference/SubgraphCountingMatching"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (development container only)")
@pytest.mark.timeout(1200)
def test_committed_fixtures_equal_a_fresh_run_of_the_reference(tmp_path):
    env = dict(os.environ, DMP_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1100)
    assert r.returncode == 0, r.stdout[-3000:]
    fresh = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    have = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert fresh == have, (sorted(set(fresh) ^ set(have)))
    for f in have:
        with np.load(os.path.join(GOLDEN, f), allow_pickle=False) as a, np.load(os.path.join(tmp_path, f), allow_pickle=False) as b:
            assert sorted(a.files) == sorted(b.files), f
            for k in a.files:
                x, y = a[k], b[k]
                assert x.shape == y.shape and x.dtype == y.dtype, (f, k)
                if x.dtype.kind in "fc":
                    if x.size:
                        assert np.array_equal(np.isnan(x), np.isnan(y)), (f, k)
                        scale = max(1.0, float(np.nanmax(np.abs(x))))
                        err = float(np.nanmax(np.abs(x.astype(np.float64) - y.astype(np.float64)))) if not np.isnan(x).all() else 0.0
                        assert err <= 1e-6 * scale, (f, k, err, scale)
                else:
                    assert x.tobytes() == y.tobytes(), (f, k)
