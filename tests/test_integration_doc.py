"""INTEGRATION.md section 1 is executable documentation: the ctypes stub a sai maintainer would add
is extracted from the document and run as it stands against libsaihip.so, and the U / Q it returns
are compared with the oracle."""

import re
import types

import numpy as np
import pytest

from conftest import ROOT, same_f64

pytestmark = pytest.mark.gpu


def test_the_documented_stub_runs_and_agrees_with_the_oracle(monkeypatch):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from oracle import sai_oracle as O
    from sai_amd import _ffi

    text = (ROOT / "INTEGRATION.md").read_text()
    block = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    assert "sai_single_window" in block and "def u_compute" in block
    monkeypatch.setenv("SAIHIP_LIB", str(_ffi.LIB_PATH))
    mod = types.ModuleType("_saihip_stub")
    exec(compile(block, "INTEGRATION.md", "exec"), mod.__dict__)

    rng = np.random.default_rng(12)
    n = 700
    p = rng.random(n) ** 3
    ref = rng.binomial(2, p[:, None] * 0.3, size=(n, 30)).astype(np.int64)
    tgt = rng.binomial(2, np.clip(p[:, None] * 2, 0, 1), size=(n, 25)).astype(np.int64)
    tgt[rng.random(tgt.shape) < 0.02] = -2
    src = np.where(rng.random((n, 1)) < 0.4, 2, 0).astype(np.int64)
    pos = np.cumsum(rng.integers(1, 40, n))
    stat = types.SimpleNamespace(ref_gts=ref, tgt_gts=tgt, src_gts_list=[src], ref_ploidy=2, tgt_ploidy=2, src_ploidy_list=[2])
    kw = dict(ref_gts=ref, tgt_gts=tgt, src_gts_list=[src], ref_ploidy=2, tgt_ploidy=2, src_ploidy_list=[2])
    seen_u = seen_q = 0
    for anc in (True, False):
        for y_list in ([("=", 1.0)], [(">=", 0.5)]):
            u = mod.u_compute(stat, pos, 0.2, 0.3, y_list, anc)
            q = mod.q_compute(stat, pos, 0.2, y_list, 0.9, anc)
            eu = O.u_stat(pos=pos, w=0.2, x=0.3, y_list=y_list, anc_allele_available=anc, **kw)
            eq = O.q_stat(pos=pos, w=0.2, quantile=0.9, y_list=y_list, anc_allele_available=anc, **kw)
            assert u["value"] == eu["value"] and u["cdd_pos"].tolist() == eu["cdd_pos"].tolist()
            seen_u, seen_q = seen_u + eu["value"], seen_q + len(eq["cdd_pos"])
            assert same_f64(q["value"], eq["value"]) and np.asarray(q["cdd_pos"]).tolist() == np.asarray(eq["cdd_pos"]).tolist()
    assert seen_u > 0 and seen_q > 0  # the cases really exercise both statistics
