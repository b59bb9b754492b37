"""Scratch timing of the packed2 site pass on the C3-shaped block."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd.engine import Engine
from sai_amd import _ffi

n_sites = 10_000_000
eng = Engine.get(0)
seed = 20260633
pops = [eng.synth_population(seed, 1, 0, n_sites, 0, 1000), eng.synth_population(seed, 1, 0, n_sites, 1, 1000),
        eng.synth_population(seed, 1, 0, n_sites, 2, 2)]
packed = [eng.pack2(p) for p in pops]
del pops
torch.cuda.empty_cache()
alg = n_sites * 2002 / 4
actual = sum(p.data.numel() for p in packed)
sets = [_ffi.make_params(0.01, 0.5, 0.95, [("=", 1.0)], True)]

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[0], ts[len(ts) // 2]

out = eng.site_pass_packed2(packed, [2, 2, 2], sets)
for mode in ("dense", "candidates"):
    mn, med = timeit(lambda: eng.site_pass_packed2(packed, [2, 2, 2], sets, out=out, freq_mode=mode))
    print(f"packed2 site_pass/{mode}: min {mn:.4f} med {med:.4f} ms  algorithmic {alg / mn / 1e6:.0f} GB/s, layout bytes {actual / mn / 1e6:.0f} GB/s")
print("probe GB/s", eng.probe_stream_read(packed[0].data))
