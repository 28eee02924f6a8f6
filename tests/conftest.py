import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def chain_or_threshold(post, device_probs, oracle_probs, device_dec_by_oracle, oracle_dec, got, want, tol=2e-4):
    """Whole-chain segment equality, made unconditional: either the oracle post-processor gives the same decisions on the device scores
    as on the oracle scores -- then the product's segments must equal the oracle driver's -- or the two score tracks (equal within the
    score tolerance) put some smoothed frame within `tol` of the decision threshold, which is the only way decisions may differ.
    Returns 1 when the full chain was compared, 0 when a frame on the threshold excused it (tests with several clips sum these)."""
    import numpy as np
    if np.array_equal(device_dec_by_oracle, oracle_dec):
        assert got == want
        return 1
    sm_o = np.asarray(post.smooth(np.asarray(oracle_probs, dtype=np.float32)), dtype=np.float64)
    sm_d = np.asarray(post.smooth(np.asarray(device_probs, dtype=np.float32)), dtype=np.float64)
    thr = float(post.thr) if hasattr(post, "thr") else float(post.prob_threshold)
    near = np.minimum(np.abs(sm_o - thr), np.abs(sm_d - thr)).min()
    assert near <= tol, f"decisions differ although no smoothed frame is within {tol} of the threshold (closest {near:.3g})"
    return 0
