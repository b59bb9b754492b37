"""PCIe-inclusive rate of the per-window plugin calls (host numpy in, results out): what a caller
pays when it keeps the reference's one-window-at-a-time API.  Never the bench `value`.

  classes   UStatistic(...).compute + QStatistic(...).compute, each on its own (two uploads)
  run()     FeaturePreprocessor.run, the product's per-window driver: one upload scope per window,
            so U and Q (and any further statistic) share one narrowing + upload of the matrices
"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import sai_amd.stats
from sai_amd.configs import PloidyConfig, StatConfig
from sai_amd.preprocessors import FeaturePreprocessor
from sai_amd.stats import QStatistic, UStatistic
rng = np.random.default_rng(0)
n_sites = 2000
ref = rng.integers(0, 3, (n_sites, 1000)).astype(np.int8); tgt = rng.integers(0, 3, (n_sites, 1000)).astype(np.int8)
src = np.full((n_sites, 2), 2, dtype=np.int8); pos = np.arange(n_sites) * 25
kw = dict(ref_gts=ref, tgt_gts=tgt, src_gts_list=[src], ref_ploidy=2, tgt_ploidy=2, src_ploidy_list=[2])
a = dict(pos=pos, w=0.6, y_list=[("=", 1.0)], anc_allele_available=True)
sc = StatConfig({"U": {"ref": {"r": 0.6}, "tgt": {"t": 0.5}, "src": {"s": "=1"}},
                 "Q": {"ref": {"r": 0.6}, "tgt": {"t": 0.95}, "src": {"s": "=1"}}})
pc = PloidyConfig({"ref": {"r": 2}, "tgt": {"t": 2}, "src": {"s": 2}})
fp = FeaturePreprocessor("/dev/null", sc, anc_allele_available=True)
for dt_name, conv in (("int8", lambda m: m), ("int64", lambda m: m.astype(np.int64))):
    k = {**kw, "ref_gts": conv(ref), "tgt_gts": conv(tgt), "src_gts_list": [conv(src)]}
    u0 = UStatistic(**k).compute(x=0.5, **a)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        UStatistic(**k).compute(x=0.5, **a); QStatistic(**k).compute(quantile=0.95, **a)
    dt = (time.perf_counter() - t0) / n
    print(f"{dt_name} matrices, classes: {dt * 1e3:.2f} ms per window (U+Q) = {1 / dt:.0f} windows/s; {ref.nbytes * 2 * 2 / dt / 1e9:.2f} GB/s of int8 over PCIe")
    rk = dict(chr_name="1", ref_pop="r", tgt_pop="t", src_pop_list=["s"], out_pop=None, start=0, end=50000, pos=pos,
              ref_gts=k["ref_gts"], tgt_gts=k["tgt_gts"], src_gts_list=k["src_gts_list"], out_gts=None, ploidy_config=pc)
    it = fp.run(**rk)[0]
    assert it["U"] == u0["value"]
    t0 = time.perf_counter()
    for _ in range(n):
        fp.run(**rk)
    dt = (time.perf_counter() - t0) / n
    print(f"{dt_name} matrices, run():   {dt * 1e3:.2f} ms per window (U+Q) = {1 / dt:.0f} windows/s; {ref.nbytes * 2 / dt / 1e9:.2f} GB/s of int8 over PCIe")
