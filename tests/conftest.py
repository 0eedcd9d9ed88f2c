import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(autouse=True)
def _switchboard_back_to_the_environment():
    """The library's switchboard (include/cnf.h: cnf_tuning) is process-wide and creating a handle no longer re-reads it:
    monkeypatch restores the environment after a test, this restores the board (autouse fixtures are torn down last, i.e. after
    monkeypatch has put the variables back), so no switch leaks into the next test."""
    yield
    try:
        entry.load_package().reload_tuning()
    except Exception:
        pass   # no library built: nothing to restore


@pytest.fixture(scope="session")
def oracles():
    # the fp64 oracle issues thousands of tiny autograd calls; on a many-core host torch's default intra-op
    # thread count makes each of them crawl (measured on the 128-core GPU box: 0.25 s per call vs 2 ms at 8)
    import torch
    torch.set_num_threads(min(8, torch.get_num_threads()))
    return entry.load_oracle()


def golden_index():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


def load_golden(name):
    o64, _ = entry.load_oracle()
    meta = golden_index()[name]
    spec = o64.Spec(nvars=meta["nvars"], naug=meta["naug"], ncond=meta["ncond"],
                    autonomous=meta["autonomous"], widths=meta["widths"], acts=meta["acts"],
                    mode=meta["mode"], nprobes=meta["nprobes"], reg_z=meta["reg_z"],
                    reg_j=meta["reg_j"], reg_aug=meta["reg_aug"])
    spec.check()
    data = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    data.setdefault("ys", None)
    return spec, meta, data


GOLDEN_NAMES = sorted(golden_index().keys())
