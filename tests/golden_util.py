"""Loader for tests/golden/<case>.npz (data produced by tests/golden/make_golden.py from the real reference)."""
import os

import numpy as np

from tests.golden.make_golden import CASES, SEEDED_ONLY, case_inputs  # noqa: F401

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    n, fin, fout, seed = [int(v) for v in d["meta"]]
    if name in SEEDED_ONLY:
        inp = case_inputs(name)
        assert np.array_equal(inp["src"], d["src"]) and np.array_equal(inp["dst"], d["dst"])
        for k in ("X", "W", "bias", "G"):
            d[k] = inp[k]
    d.update(n=n, fin=fin, fout=fout)
    return d


def same(a, b):
    """Bit-exact numeric equality (treats -0 == +0, the only licence the oracle takes; NaN never appears)."""
    return a.shape == b.shape and bool(np.all(a == b))
