"""Scratch timing of the individual kernels on a synthetic C3-shaped block (not the bench)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd.engine import Engine
from sai_amd import _ffi

n_sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
n_ref = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_tgt = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
eng = Engine.get(0)
seed = 20260633
t0 = time.time()
pops = [eng.synth_population(seed, 1, 0, n_sites, 0, n_ref), eng.synth_population(seed, 1, 0, n_sites, 1, n_tgt),
        eng.synth_population(seed, 1, 0, n_sites, 2, 2)]
pos = eng.synth_positions(seed, 1, n_sites)
torch.cuda.synchronize(); print("synth s", time.time() - t0, flush=True)
nbytes = n_sites * (n_ref + n_tgt + 2)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in ev), sum(a.elapsed_time(b) for a, b in ev) / n

counts = eng.site_counts(pops)
mn, av = timeit(lambda: eng.site_counts(pops, out=counts))
print(f"site_counts min {mn:.3f} ms avg {av:.3f} ms  -> {nbytes / mn / 1e6:.1f} GB/s (min) {nbytes / av / 1e6:.1f} GB/s (avg)")
sets = [_ffi.make_params(0.01, 0.5, 0.95, [("=", 1.0)], True)]
fused = eng.site_pass(pops, [2, 2, 2], sets)
for mode in ("dense", "candidates"):
    mn, av = timeit(lambda: eng.site_pass(pops, [2, 2, 2], sets, out=fused, freq_mode=mode), 8)
    print(f"site_pass/{mode} min {mn:.3f} ms avg {av:.3f} ms  -> {nbytes / mn / 1e6:.1f} GB/s (min)")
out = eng.site_flags(counts, [2, 2, 2], sets)
mn, av = timeit(lambda: eng.site_flags(counts, [2, 2, 2], sets, out=out[:2]))
print(f"site_flags  min {mn:.3f} ms avg {av:.3f}")
p0, p1 = int(pos[0]), int(pos[-1])
win, step = 50000, 25000
s0 = max((p0 + step) // step * step - win + 1, 1)
starts = np.arange(s0, p1 + 1, step, dtype=np.int64)
ends = starts + win - 1
lo, hi = eng.window_bounds(pos, starts, ends)
mn, av = timeit(lambda: eng.window_bounds(pos, starts, ends))
print(f"window_bounds (incl. H2D of {len(starts)} windows) min {mn:.3f} ms")
bufs = eng.alloc_window_bufs(1, len(starts), 1 << 22, 1 << 22)
mn, av = timeit(lambda: eng.window_stats_async(out[0], out[1], sets, lo, hi, pos, bufs))
print(f"window_stats min {mn:.3f} ms avg {av:.3f}; totals {bufs[4].cpu().tolist()} windows {len(starts)}")
print(f"stream-read probe over the ref block: {eng.probe_stream_read(pops[0].tiles):.1f} GB/s")
