"""CPU: the portable oracle (oracle/convnet_oracle.c) must reproduce the committed golden outputs
of the reference's own CPU code (tests/golden/hotpath_golden.npz, made by
tests/golden/make_golden.py).  Runs anywhere (no /root/reference, no GPU)."""
import os

import numpy as np
import pytest

import oracle
from golden_cases import compute_all, rel_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return dict(np.load(GOLDEN))


def test_port_matches_golden(golden):
    got = compute_all(oracle.port)
    assert set(got) == set(golden)
    for k in sorted(golden):
        if k.endswith("/max") or k in ("softmax/grad", "softmax/correct"):
            assert np.array_equal(got[k], golden[k]), k   # selection ops: bit-exact
        else:
            assert rel_err(got[k], golden[k]) < 2e-6, (k, rel_err(got[k], golden[k]))


@pytest.mark.skipif(oracle.ref is None, reason="oracle/_ref not built")
def test_reference_build_reproduces_golden(golden):
    got = compute_all(oracle.ref)
    for k in sorted(golden):
        assert rel_err(got[k], golden[k]) < 1e-6, k
