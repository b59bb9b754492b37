"""Scratch timing of the C5 sweep (BASELINE.json configs[4]): two sources, 18 parameter sets."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd import _ffi
from sai_amd.engine import Engine
from sai_amd.resident import ResidentScorer, default_windows, synth_block

eng = Engine.get(0)
n = 10_000_000
block = synth_block(eng, 20260635, 1, n, 1000, 1000, [1, 1])
windows = default_windows(int(block.pos[0]), int(block.pos[-1]), 50000, 25000)
specs = [dict(w=0.01, x=0.5, quantile=0.95, y_list=[(op, y1), (op, y2)], anc=True)
         for op in ("=", ">=") for y1 in (0.0, 0.5, 1.0) for y2 in (0.0, 0.5, 1.0)]
sets = [_ffi.make_params(s["w"], s["x"], s["quantile"], s["y_list"], s["anc"]) for s in specs]

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[0], ts[len(ts) // 2]

counts = eng.site_counts(block.pops)
print("site_counts %.3f ms" % timeit(lambda: eng.site_counts(block.pops, out=counts))[0])
out = eng.site_flags(counts, block.ploidies, sets)
print("site_flags (18 sets) %.3f ms" % timeit(lambda: eng.site_flags(counts, block.ploidies, sets, out=out[:2]))[0])
lo, hi = eng.window_bounds(block.pos, [w[0] for w in windows], [w[1] for w in windows])
t0 = time.perf_counter()
res = eng.window_stats(out[0], out[1], sets, lo, hi, pos=block.pos, cap_hint=1 << 24)
t1 = time.perf_counter()
res = eng.window_stats(out[0], out[1], sets, lo, hi, pos=block.pos, cap_hint=1 << 24)
t2 = time.perf_counter()
print(f"window_stats (18 sets x {len(windows)} windows, incl. copies to host): {1e3 * (t2 - t1):.2f} ms (first call {1e3 * (t1 - t0):.2f})")
print("candidate totals", res.cdd_u.size, res.cdd_q.size, "cond sites per set", [int(r['n_cond'].sum()) for r in res.records][:18])
for chunk in (sets[:16], sets[16:]):
    sc = ResidentScorer(eng, block, windows, chunk, cap_u=1 << 24, cap_q=1 << 24)
    mn, md = timeit(lambda: sc.step())
    print(f"ResidentScorer.step with {len(chunk)} sets: {mn:.3f} ms")
