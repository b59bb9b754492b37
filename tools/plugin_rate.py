"""PCIe-inclusive rate of the per-window plugin call (host numpy in, results out): what a caller
pays when it keeps the reference's one-window-at-a-time API.  Never the bench `value`."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sai_amd.stats
from sai_amd.stats import QStatistic, UStatistic
rng = np.random.default_rng(0)
n_sites = 2000
ref = rng.integers(0, 3, (n_sites, 1000)).astype(np.int8); tgt = rng.integers(0, 3, (n_sites, 1000)).astype(np.int8)
src = np.full((n_sites, 2), 2, dtype=np.int8); pos = np.arange(n_sites) * 25
kw = dict(ref_gts=ref, tgt_gts=tgt, src_gts_list=[src], ref_ploidy=2, tgt_ploidy=2, src_ploidy_list=[2])
a = dict(pos=pos, w=0.6, y_list=[("=", 1.0)], anc_allele_available=True)
for dt_name, conv in (("int8", lambda m: m), ("int64", lambda m: m.astype(np.int64))):
    k = {**kw, "ref_gts": conv(ref), "tgt_gts": conv(tgt), "src_gts_list": [conv(src)]}
    UStatistic(**k).compute(x=0.5, **a)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        UStatistic(**k).compute(x=0.5, **a); QStatistic(**k).compute(quantile=0.95, **a)
    dt = (time.perf_counter() - t0) / n
    print(f"{dt_name} matrices: {dt * 1e3:.2f} ms per window (U+Q) = {1 / dt:.0f} windows/s; {ref.nbytes * 2 * 2 / dt / 1e9:.2f} GB/s of int8 over PCIe")
