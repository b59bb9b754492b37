"""One tree, one evidence set: every kept figure of this round names the digest of the sources it was
measured on, and that digest is this tree's -- a source change after the measurements makes this test fail until the
set is measured again (tools/evidence.sh + tools/collect_evidence.py).  Files of earlier rounds live in profiles/history/."""

import json

import pytest

from conftest import ROOT

PROFILES = ROOT / "profiles"


@pytest.fixture(scope="module")
def digest():
    import bench

    return bench.source_digest()


def test_every_file_of_this_round_is_listed_with_this_trees_digest(digest):
    manifest = json.loads((PROFILES / "manifest.json").read_text())
    files = sorted(f.name for f in PROFILES.glob("r05_*"))
    measured = [f for f in files if f not in NOTES]
    assert measured, "no evidence of this round"
    missing = [f for f in measured if f not in manifest]
    assert not missing, f"not in profiles/manifest.json: {missing}"
    stale = {f: m["source_digest"] for f, m in manifest.items() if m["source_digest"] != digest}
    assert not stale, f"measured on other sources than this tree ({digest}): {stale}"
    assert not [f for f in manifest if not (PROFILES / f).exists()]


# experiment logs of the round that were measured on trees on the way to the final one (they say so in their headers)
NOTES = {"r05_c5_grid.txt", "r05_dd_pass.txt", "r05_score_parts.txt", "r05_placement.txt", "r05_placement_ab.txt", "r05_c5_product_trace.txt"}


def test_bench_lines_and_stored_figures_carry_the_digest(digest):
    lines = sorted(PROFILES.glob("r05_*bench*.json"))
    assert len(lines) >= 6
    for f in lines:
        line = json.loads(f.read_text())
        assert line["config"]["source_digest"] == digest, f.name
    traffic = json.loads((PROFILES / "traffic.json").read_text())
    assert {"c2", "c3", "c4", "c5"} <= set(traffic)
    assert all(v.get("source_digest") == digest for v in traffic.values()), {k: v.get("source_digest") for k, v in traffic.items()}
    assert json.loads((PROFILES / "one_gpu_base.json").read_text())["c4"]["source_digest"] == digest


def test_older_rounds_live_in_history():
    assert not [f.name for f in PROFILES.glob("r0[1-4]*")]
    assert len(list((PROFILES / "history").glob("r04_*"))) > 20
