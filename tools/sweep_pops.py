import os, sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd.engine import Engine
n_sites = 10_000_000
eng = Engine.get(0)
seed = 20260633
pops = [eng.synth_population(seed, 1, 0, n_sites, 0, 1000), eng.synth_population(seed, 1, 0, n_sites, 1, 1000),
        eng.synth_population(seed, 1, 0, n_sites, 2, 2)]
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[0], ts[len(ts)//2]
for name, sel in [("ref", [0]), ("ref+tgt", [0, 1]), ("ref+tgt+src", [0, 1, 2]), ("ref+src", [0, 2]), ("ref+ref", [0, 0])]:
    ps = [pops[i] for i in sel]
    nbytes = sum(p.n_ind for p in ps) * n_sites
    counts = eng.site_counts(ps)
    mn, med = timeit(lambda: eng.site_counts(ps, out=counts))
    print(f"{name:>12}: min {mn:.3f} med {med:.3f} ms  {nbytes/mn/1e6:.0f} GB/s", flush=True)
big = eng.synth_population(seed, 1, 0, n_sites, 3, 2000)
counts = eng.site_counts([big])
mn, med = timeit(lambda: eng.site_counts([big], out=counts))
print(f"one pop 2000: min {mn:.3f} med {med:.3f} ms  {2000*n_sites/mn/1e6:.0f} GB/s", flush=True)
print("probe GB/s", eng.probe_stream_read(big.tiles))
