import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
DATA = ROOT / "tests" / "data"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def unhex(s):
    return float("nan") if s == "nan" else float.fromhex(s)


def load_golden(name):
    with open(GOLDEN / name) as f:
        return json.load(f)


def same_f64(a, b) -> bool:
    """Bit-equality of two doubles, NaN == NaN."""
    a = np.float64(a)
    b = np.float64(b)
    if np.isnan(a) or np.isnan(b):
        return bool(np.isnan(a) and np.isnan(b))
    return a.tobytes() == b.tobytes()


def stat_case_inputs(c):
    """Decode one entry of stats_cases.json into numpy inputs."""
    return dict(
        ref_gts=np.array(c["ref_gts"], dtype=np.int64).reshape(len(c["ref_gts"]), -1),
        tgt_gts=np.array(c["tgt_gts"], dtype=np.int64).reshape(len(c["tgt_gts"]), -1),
        src_gts_list=[np.array(s, dtype=np.int64).reshape(len(s), -1) for s in c["src_gts_list"]],
        ploidy=list(c["ploidy"]),
        pos=np.array(c["pos"], dtype=np.int64),
        w=unhex(c["w"]),
        x=unhex(c["x"]),
        quantile=unhex(c["quantile"]),
        y_list=[(op, unhex(y)) for op, y in c["y_list"]],
        anc=bool(c["anc_allele_available"]),
    )


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture()
def in_repo_root(monkeypatch):
    """The reference's configs use paths relative to the repo root
    (tests/data/...); run those tests from there."""
    monkeypatch.chdir(ROOT)
    return ROOT
