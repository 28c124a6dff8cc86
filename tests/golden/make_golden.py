"""Generates tests/golden/hotpath_golden.npz from the reference's OWN CPU code
(oracle/_ref/libconvnet_ref.so = eigenmat/*.cc + src/CPUMatrix.cc compiled unmodified by
oracle/Makefile).  Run in the build container (where /root/reference is mounted):

    python tests/golden/make_golden.py

The case list and seeds live in tests/golden_cases.py; only outputs are stored.  The reference
ships no golden vectors of its own — these are its pinned outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from golden_cases import compute_all  # noqa: E402

if __name__ == "__main__":
    assert oracle.ref is not None, "build oracle/_ref first (make -C oracle)"
    out = compute_all(oracle.ref)
    path = os.path.join(HERE, "hotpath_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
