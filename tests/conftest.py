import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: g[k] for k in g.files}
    for k in ("n", "nsuper", "ssize", "xsize", "maxcsize", "maxesize", "status", "minor", "xtype", "stype",
              "ordering"):
        d[k] = int(d[k])
    d["beta"] = float(d["beta"])
    return d


def sym_of(g):
    return dict(n=g["n"], nsuper=g["nsuper"], ssize=g["ssize"], xsize=g["xsize"], maxcsize=g["maxcsize"],
                maxesize=g["maxesize"], Super=g["Super"], Lpi=g["Lpi"], Lpx=g["Lpx"], Ls=g["Ls"])


def relerr(a, b):
    den = max(float(np.abs(b).max()) if b.size else 0.0, 1e-300)
    return float(np.abs(a - b).max() / den) if b.size else 0.0


@pytest.fixture(scope="session")
def have_ref():
    from oracle import refcholmod
    return refcholmod.available()
