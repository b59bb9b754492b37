"""Scratch timing of the widened statistics (fd/df/Danc/Dplus, DD) on a C3-shaped block with an
outgroup: 1e7 sites, 1000 ref / 1000 tgt / 2 src / 100 outgroup diploids, 50 kb / 25 kb windows."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd.engine import Engine

n_sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
eng = Engine.get(0)
seed = 20260633
sizes = [1000, 1000, 2, 100]
pops = [eng.synth_population(seed, 1, 0, n_sites, i, n) for i, n in enumerate(sizes)]
pos = eng.synth_positions(seed, 1, n_sites)
p0, p1 = int(pos[0]), int(pos[-1])
win, step = 50000, 25000
s0 = max((p0 + step) // step * step - win + 1, 1)
starts = np.arange(s0, p1 + 1, step, dtype=np.int64)
lo, hi = eng.window_bounds(pos, starts, starts + win - 1)
n_w = len(starts)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[0], ts[len(ts) // 2]

nbytes = n_sites * sum(sizes)
counts = eng.site_counts(pops)
mn, md = timeit(lambda: eng.site_counts(pops, out=counts))
print(f"site_counts (4 pops, counts out): {mn:.3f} ms  {nbytes / mn / 1e6:.0f} GB/s")
freqs = eng.site_freqs(counts, [2, 2, 2, 2])
mn, md = timeit(lambda: eng.site_freqs(counts, [2, 2, 2, 2]))
print(f"site_freqs: {mn:.3f} ms  ({n_sites * 4 * 16 / mn / 1e6:.0f} GB/s of counts in + freqs out)")
mn, md = timeit(lambda: eng.window_fourpop(freqs, 1, True, lo, hi))
print(f"window_fourpop ({n_w} windows): {mn:.3f} ms  ({n_sites * 2 * 4 * 8 / mn / 1e6:.0f} GB/s of freqs, each site in 2 windows)")
ad_ref = eng.site_absdiff(pops[0], pops[2])
mn, md = timeit(lambda: eng.site_absdiff(pops[0], pops[2]))
print(f"site_absdiff ref x 2 src individuals: {mn:.3f} ms  {n_sites * 1002 / mn / 1e6:.0f} GB/s")
ad_tgt = eng.site_absdiff(pops[1], pops[2])
mn, md = timeit(lambda: eng.window_dd(ad_ref, 1000, ad_tgt, 1000, lo, hi))
print(f"window_dd: {mn:.3f} ms")
# DD riding along the site pass (round 5): the counts of all four populations and DD's terms of the two source
# individuals from ONE read of ref and tgt; and the fused U / Q pass with them
import os
from sai_amd import _ffi
ad = torch.empty((2, 2, n_sites), dtype=torch.int32, device=eng.device)
mn, md = timeit(lambda: eng.site_pass_dd(pops, None, [], 2, 1, counts=counts, absdiff=ad))
print(f"site_pass_dd counts + DD terms (4 pops, 2 source individuals): {mn:.3f} ms  {nbytes / mn / 1e6:.0f} GB/s")
assert torch.equal(ad[0], ad_ref) and torch.equal(ad[1], ad_tgt)
sets = [_ffi.make_params(0.01, 0.5, 0.95, [("=", 1.0)], True)]
out = eng.site_pass(pops[:3], [2, 2, 2], sets, freq_mode="candidates")
nb3 = n_sites * sum(sizes[:3])
mn, md = timeit(lambda: eng.site_pass(pops[:3], [2, 2, 2], sets, out=out, freq_mode="candidates"))
print(f"site_pass U/Q (3 pops): {mn:.3f} ms  {nb3 / mn / 1e6:.0f} GB/s")
mn, md = timeit(lambda: eng.site_pass_dd(pops[:3], [2, 2, 2], sets, 2, 1, out=out, freq_mode="candidates", absdiff=ad))
print(f"site_pass_dd U/Q + DD terms (3 pops): {mn:.3f} ms  {nb3 / mn / 1e6:.0f} GB/s")
for ns in (1, 3, 4):
    src = eng.synth_population(seed, 1, 0, n_sites, 2, ns)
    adk = torch.empty((2, ns, n_sites), dtype=torch.int32, device=eng.device)
    pk = [pops[0], pops[1], src]
    mn, md = timeit(lambda: eng.site_pass_dd(pk, [2, 2, 2], sets, 2, 1, out=out, freq_mode="candidates", absdiff=adk))
    print(f"site_pass_dd U/Q + DD terms, {ns} source individual(s): {mn:.3f} ms  {n_sites * (2000 + ns) / mn / 1e6:.0f} GB/s")
