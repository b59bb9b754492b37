"""What the per-site decision costs (round 5): the stand-alone site_flags kernel -- counts from HBM, nothing
but calc_freq's division and the sets' evaluation -- and the fused pass, for 1 .. 18 parameter sets of C5's
grid, with the predicate table (site_eval.hpp) and with the set-by-set form (SAI_NO_PRED_TABLE=1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sai_amd import _ffi
from sai_amd.engine import Engine
from sai_amd.resident import synth_block

eng = Engine.get(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
block = synth_block(eng, 20260635, 1, n, 1000, 1000, [1, 1])
specs = [dict(w=0.01, x=0.5, quantile=0.95, y_list=[(op, y1), (op, y2)], anc=True)
         for op in ("=", ">=") for y1 in (0.0, 0.5, 1.0) for y2 in (0.0, 0.5, 1.0)]
all_sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]

def timeit(fn, n=7):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[0], ts[len(ts) // 2]

counts = eng.site_counts(block.pops)
for n_sets in (1, 4, 8, 18):
    sets = all_sets[:n_sets]
    row = [f"{n_sets:2d} sets:"]
    ref = None
    for table in (True, False):
        if table:
            os.environ.pop("SAI_NO_PRED_TABLE", None)
        else:
            os.environ["SAI_NO_PRED_TABLE"] = "1"
        out = eng.site_flags(counts, block.ploidies, sets)
        mn, md = timeit(lambda: eng.site_flags(counts, block.ploidies, sets, out=out[:2]))
        fused = eng.site_pass(block.pops, block.ploidies, sets, freq_mode="candidates")
        mn2, md2 = timeit(lambda: eng.site_pass(block.pops, block.ploidies, sets, out=fused, freq_mode="candidates"))
        row.append(f"{'table' if table else 'sets '}: site_flags {mn:.3f} ms, fused pass {mn2:.3f} (median {md2:.3f})")
        if ref is None:
            ref = (out[1].clone(), fused[1].clone())
        else:
            assert torch.equal(ref[0], out[1]) and torch.equal(ref[1], fused[1])
    print("  ".join(row), flush=True)
os.environ.pop("SAI_NO_PRED_TABLE", None)
