"""sai_amd/placement.py: the decision between two placements (CPU), and -- on the GPU -- that a population moved
into another allocation holds the same bytes and scores to the same records."""

import numpy as np
import pytest


def test_worth_moving_needs_more_than_a_levels_own_spread():
    from sai_amd.placement import GAIN, worth_moving

    assert worth_moving(3.00, 2.85) and worth_moving(3.07, 2.96)  # the levels measured at C3
    assert not worth_moving(2.85, 3.00) and not worth_moving(2.85, 2.84) and not worth_moving(3.00, 2.99)
    edge = 3.0 * (1.0 - GAIN)
    assert worth_moving(3.0, edge - 1e-9) and not worth_moving(3.0, edge + 1e-9)


def test_settle_block_leaves_small_blocks_and_the_switch_alone(monkeypatch):
    from types import SimpleNamespace

    from sai_amd import placement

    class Tiles:
        def __init__(self, n):
            self.n = n

        def numel(self):
            return self.n

    small = [SimpleNamespace(tiles=Tiles(1 << 20), n_sites=64, n_ind=1) for _ in range(3)]
    report = {}
    assert placement.settle_block(None, small, report) == small and report == {"enabled": True}  # no engine is touched
    monkeypatch.setenv("SAI_AMD_PLACEMENT", "0")
    big = [SimpleNamespace(tiles=Tiles(2 << 30), n_sites=64, n_ind=1) for _ in range(2)]
    assert placement.settle_block(None, big) == big


@pytest.mark.gpu
@pytest.mark.parametrize("times,moved,chosen,left_away", [
    # pairs in order: as built, (anchor, F1), (F1, other), (anchor, F2), (F2, other), (F1, F2)
    ([3.00, 2.85, 3.01, 2.86, 3.00, 2.852], ["other"], 2.85, ["other"]),  # both fresh pieces are of the anchor's class: the first one met
    ([2.85, 3.00, 3.01, 2.99, 3.02, 2.86], [], 2.85, []),  # the pair as built was of one class
    ([3.00, 3.01, 2.85, 3.00, 3.02, 2.99], ["anchor"], 2.85, ["anchor"]),  # F1 is of the other's class: the anchor goes there
    ([3.00, 3.01, 2.99, 3.02, 3.00, 2.84], ["anchor", "other"], 2.84, ["anchor", "other"]),  # three classes met: the two fresh pieces are the pair
    ([3.07, 2.93, 3.05, 2.85, 3.06, 2.94], ["other"], 2.85, ["other"]),  # a piece partly in two classes (2.93) is not mistaken for the fast one
    ([2.85, 2.86, 2.84, 2.85, 2.86, 2.85], [], 2.85, []),  # all of one class: a hundredth is not worth a move
    # all of one class and a move for its per cent (profiles/r05 C5 line of the round's last set): the array left is of
    # the populations' OWN class -- no arena from it
    ([2.8554, 2.8259, 2.8451, 2.8276, 2.8412, 2.8357], ["other"], 2.8259, []),
])  # fmt: skip
def test_a_moved_population_is_the_same_population(monkeypatch, times, moved, chosen, left_away):
    import torch

    from sai_amd import _ffi, placement
    from sai_amd.engine import Engine
    from sai_amd.resident import ResidentScorer, synth_block

    eng = Engine.get()
    block = synth_block(eng, 77, 1, 20_000, 300, 280, [2], missing_per_million=2000)
    script = list(times)
    real_ms = placement._PairTimer.ms

    def scripted(self, a, b, passes=placement.PASSES):
        real_ms(self, a, b, 1)  # the pass runs over the pair it is asked about
        return script.pop(0)

    monkeypatch.setattr(placement._PairTimer, "ms", scripted)
    report = {}
    ref2, tgt2 = placement.settle_pair(eng, block.pops[0], block.pops[1], report=report)
    log = report["pairs"][0]
    assert not script and log["ms"] == times and log["moved"] == moved and log["ms_chosen"] == chosen
    assert log["left_away"] == left_away  # which of the arrays a population left may serve as the output arena
    assert (tgt2 is block.pops[1]) == ("other" not in moved) and (ref2 is block.pops[0]) == ("anchor" not in moved)
    for now, was in ((ref2, block.pops[0]), (tgt2, block.pops[1])):
        assert (now.n_ind, now.n_sites) == (was.n_ind, was.n_sites) and torch.equal(now.tiles, was.tiles)
        assert (now.tiles.data_ptr() == was.tiles.data_ptr()) == (now is was)

    pos = block.pos.cpu().numpy()
    windows = [(int(pos[a]), int(pos[min(a + 900, len(pos) - 1)])) for a in range(0, len(pos) - 1, 450)]
    sets = [_ffi.make_params(0.3, 0.5, 0.9, [("=", 1.0)], True)]
    got = []
    for pops in (block.pops, [ref2, tgt2, block.pops[2]]):
        import dataclasses

        sc = ResidentScorer(eng, dataclasses.replace(block, pops=list(pops)), windows, sets)
        sc.step()
        got.append(sc.results())
    assert got[0].records.tobytes() == got[1].records.tobytes()
    assert np.array_equal(got[0].cdd_u, got[1].cdd_u) and np.array_equal(got[0].cdd_q, got[1].cdd_q)


@pytest.mark.gpu
def test_score_writes_the_same_files_from_settled_blocks(monkeypatch, tmp_path):
    """What ``score`` runs on: WindowGenerator.device_blocks settles the big populations once per generator; a
    population that moved IS the generator's population from then on, and the files are the ones an unsettled run
    writes."""
    import torch

    from sai_amd import placement
    from sai_amd.configs import PloidyConfig, StatConfig
    from sai_amd.engine import Engine
    from sai_amd.generators import WindowGenerator
    from sai_amd.preprocessors import FeaturePreprocessor
    from sai_amd.sai import write_headers

    eng = Engine.get(0)
    n_sites, seed = 120_000, 4242
    sizes = {"ref": 150, "tgt": 90, "src": 2}
    stats = StatConfig({"U": {"ref": {"ref": 0.05}, "tgt": {"tgt": 0.3}, "src": {"src": "=1"}},
                        "Q": {"ref": {"ref": 0.05}, "tgt": {"tgt": 0.9}, "src": {"src": "=1"}}})  # fmt: skip
    ploidies = PloidyConfig({"ref": {"ref": 2}, "tgt": {"tgt": 2}, "src": {"src": 2}})
    monkeypatch.setattr(placement, "MIN_BYTES", 1 << 20)  # ref and tgt count as big, src does not
    calls = []
    real_ms = placement._PairTimer.ms

    def first_copy_is_faster(self, a, b, passes=placement.PASSES):
        real_ms(self, a, b, 1)
        calls.append((a.n_ind, b.n_ind))
        return 3.0 if len(calls) == 1 else 2.85

    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SAI_AMD_PLACEMENT", mode)
        monkeypatch.setattr(placement._PairTimer, "ms", first_copy_is_faster)
        eng.__dict__.pop("_settled_pairs", None)
        pops = {k: eng.synth_population(seed, 1, 0, n_sites, i, n, 2, 3000) for i, (k, n) in enumerate(sizes.items())}
        was = {k: p.tiles.data_ptr() for k, p in pops.items()}
        pos_dev = eng.synth_positions(seed, 1, n_sites)
        wg = WindowGenerator.from_resident("7", pos_dev.cpu().numpy(), pos_dev, {"ref": pops["ref"]}, {"tgt": pops["tgt"]},
                                           {"src": pops["src"]}, 5000, 2500, ploidies)  # fmt: skip
        out = tmp_path / f"placement{mode}.tsv"
        fp = FeaturePreprocessor(str(out), stats, anc_allele_available=True)
        write_headers(str(out), stats, ploidies)
        fp.score_and_write(wg)
        fp.score_and_write(wg)  # the second call finds the blocks settled
        torch.cuda.synchronize()
        outs[mode] = [out.read_bytes(), out.with_suffix(".U.log").read_bytes(), out.with_suffix(".Q.log").read_bytes()]
        now = {k[1]: p.tiles.data_ptr() for k, p in wg.device_blocks(eng).items()}
        if mode == "0":
            assert not calls and now == was
        else:
            assert calls == [(150, 90)] * 6  # the pair as built and the five pairs with the two fresh pieces
            assert now["ref"] == was["ref"] and now["src"] == was["src"] and now["tgt"] != was["tgt"]
            assert wg.tgt_data["tgt"].GT.tiles.data_ptr() == now["tgt"]  # the copy is the population now
            assert torch.equal(wg.tgt_data["tgt"].GT.tiles, pops["tgt"].tiles)
    assert outs["0"] == outs["1"] and outs["0"][1].count(b":") > 5


@pytest.mark.gpu
def test_no_room_for_a_copy_leaves_the_pair_as_it_is(monkeypatch):
    """A card shared with other processes (several ranks rehearsed on one GPU) can run out between the look at the
    free memory and the allocation: the search stops, the block is used as built."""
    import torch

    from sai_amd import placement
    from sai_amd.engine import Engine
    from sai_amd.resident import synth_block

    eng = Engine.get()
    block = synth_block(eng, 78, 1, 10_000, 100, 90, [2])
    need = block.pops[0].tiles.numel()
    real_empty = torch.empty

    def no_room_for_a_population(*a, **k):
        if k.get("dtype") == torch.int8 and a and tuple(a[0]) == (need,):
            raise torch.cuda.OutOfMemoryError("HIP out of memory (simulated)")
        return real_empty(*a, **k)

    monkeypatch.setattr(torch, "empty", no_room_for_a_population)
    report = {}
    ref2, tgt2 = placement.settle_pair(eng, block.pops[0], block.pops[1], report=report)
    assert ref2 is block.pops[0] and tgt2 is block.pops[1]
    log = report["pairs"][0]
    assert log["stopped"] == "no room for another copy" and log["moved"] == [] and len(log["ms"]) == 1
    monkeypatch.undo()
    monkeypatch.setattr(placement, "KEEP_FREE", 1 << 60)  # ... or the look at the free memory says so already
    report = {}
    assert placement.settle_pair(eng, block.pops[0], block.pops[1], report=report) == (block.pops[0], block.pops[1])
    assert report["pairs"][0]["stopped"] == "no room for another copy"


@pytest.mark.gpu
def test_a_fixed_anchor_only_gets_new_partners(monkeypatch):
    """Later populations of a block are settled next to an anchor that stays: fresh pieces for them alone."""
    from sai_amd import placement
    from sai_amd.engine import Engine
    from sai_amd.resident import synth_block

    eng = Engine.get()
    block = synth_block(eng, 79, 1, 10_000, 100, 90, [2])
    script = [3.00, 3.01, 2.85, 2.86]
    monkeypatch.setattr(placement._PairTimer, "ms", lambda self, a, b, passes=placement.PASSES: script.pop(0))
    report = {}
    ref2, tgt2 = placement.settle_pair(eng, block.pops[0], block.pops[1], report=report, move_anchor=False)
    log = report["pairs"][0]
    assert ref2 is block.pops[0] and tgt2 is not block.pops[1] and log["moved"] == ["other"]
    assert log["pairs"] == ["ao", "a1", "a2", "a3"] and log["ms_chosen"] == 2.85 and not script


def test_output_arena_hands_out_aligned_views_front_to_back():
    import torch

    from sai_amd.placement import OutputArena

    arena = OutputArena(torch.zeros(4096, dtype=torch.uint8))
    a, b = arena.take(100), arena.take(1000)
    assert a.numel() == 100 and b.numel() == 1000 and b.data_ptr() - a.data_ptr() == 256
    assert arena.take(4096) is None and arena.take(2000).numel() == 2000  # what does not fit is refused, the rest stays usable
    a.fill_(7)
    assert int(arena.tensor[:100].sum()) == 700 and int(arena.tensor[100:].sum()) == 0


@pytest.mark.gpu
def test_a_piece_of_another_class_becomes_the_output_arena_and_the_scorer_writes_there(monkeypatch):
    import dataclasses

    import torch

    from sai_amd import _ffi, placement
    from sai_amd.engine import Engine
    from sai_amd.resident import ResidentScorer, synth_block

    eng = Engine.get()
    torch.cuda.empty_cache()  # the arena is carved from the piece that was released last: nothing else of its size may wait in the cache
    block = synth_block(eng, 80, 1, 200_000, 120, 100, [2], missing_per_million=1000)
    # as built fast; F1 of another class (slow next to both); F2 of the pair's class
    script = [2.85, 3.00, 3.01, 2.86, 2.85, 3.00]
    real_ms = placement._PairTimer.ms
    monkeypatch.setattr(placement._PairTimer, "ms", lambda self, a, b, passes=placement.PASSES: (real_ms(self, a, b, 1), script.pop(0))[1])
    report, arena = {}, []
    need = placement.ARENA_BYTES_PER_SITE * block.n_sites
    ref2, tgt2 = placement.settle_pair(eng, block.pops[0], block.pops[1], report=report, arena=arena, arena_bytes=need)
    log = report["pairs"][0]
    assert (ref2, tgt2) == (block.pops[0], block.pops[1]) and log["moved"] == [] and log.get("arena") == "piece 1"
    assert len(arena) == 1 and arena[0].tensor.numel() == need and arena[0].used == 0

    pos = block.pos.cpu().numpy()
    windows = [(int(pos[a]), int(pos[min(a + 3000, len(pos) - 1)])) for a in range(0, len(pos) - 1, 1500)]
    sets = [_ffi.make_params(0.3, 0.5, 0.9, [("=", 1.0)], True), _ffi.make_params(0.5, 0.2, 0.5, [(">=", 0.5)], False)]
    got = []
    for extra in ({}, {"output_arena": arena[0]}):
        for overlap in (False, True):
            sc = ResidentScorer(eng, dataclasses.replace(block, extra=extra), windows, sets, overlap=overlap)
            sc.step()
            sc.step()
            sc.flush()
            got.append(sc.results())
            lo, hi = arena[0].tensor.data_ptr(), arena[0].tensor.data_ptr() + need
            inside = [lo <= t.data_ptr() < hi for t in sc._tgt_freq + sc._flags]
            assert all(inside) if extra else not any(inside)  # one plus three buffer sets of this size fit into the arena
    assert arena[0].used > 0
    for other in got[1:]:
        assert other.records.tobytes() == got[0].records.tobytes()
        assert np.array_equal(other.cdd_u, got[0].cdd_u) and np.array_equal(other.cdd_q, got[0].cdd_q)
