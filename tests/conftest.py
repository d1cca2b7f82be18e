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


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    # -m gpu tests must never be silently skipped on the GPU box; off it they are deselected by -m "not gpu".
    if not _has_gpu():
        skip = pytest.mark.skip(reason="no GPU in this container")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


GOLDEN_CASES = {
    # name: (F, model kwargs for oracle ModelCfg / seeded_state_dict)
    "pyannet2_f80_T500": dict(F=80, num_layers=4, bidirectional=True),
    "pyannet2_f64_T1000": dict(F=64, num_layers=4, bidirectional=True),
    "pyannet2_f64_T3000": dict(F=64, num_layers=4, bidirectional=True),
    "pyannet2_f64_T7": dict(F=64, num_layers=4, bidirectional=True),
    "pyannet2_uni_f64_T200": dict(F=64, num_layers=4, bidirectional=False),
    "pyannet2_l1_f64_T100": dict(F=64, num_layers=1, bidirectional=True),
    "pyannet2_nonmono_f64_T50": dict(F=64, num_layers=4, bidirectional=True),
}


def load_golden(name):
    """Fixture written by tools/gen_golden.py from the REFERENCE's own PyanNet2 class, plus the
    seeded weights it was produced with (regenerated and checked against the stored digest)."""
    import hashlib
    from oracle.torch_ref import seeded_state_dict
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    c = GOLDEN_CASES[name]
    sd = seeded_state_dict(c["F"], 128, c["num_layers"], c["bidirectional"], seed=int(g["weights_seed"]))
    dig = hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).hexdigest()
    assert dig == str(g["weights_sha256"]), "seeded weights differ from the ones the fixture was generated with"
    return g, sd, c
