import json, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from golden.seeded import fuzz_scenario
from oracle import sai_oracle as O
from sai_amd.configs import PloidyConfig, StatConfig
from sai_amd.generators import WindowGenerator
from sai_amd.preprocessors import FeaturePreprocessor
from sai_amd.utils import ChromosomeData
from sai_amd.engine import Engine
import torch
sc = fuzz_scenario(103)
pos = sc["pos"]
sel = (pos >= sc["start"]) & (pos <= sc["end"])
mk = lambda g: ChromosomeData(pos[sel], None, None, g[sel].astype(np.int8))
data = {grp: {k: mk(v) for k, v in sc["gts"][grp].items()} for grp in sc["gts"]}
pc = PloidyConfig(sc["pl"])
wg = WindowGenerator.from_arrays("7", data["ref"], data["tgt"], data["src"], sc["win"], sc["step"], pc, start=sc["start"], end=sc["end"], out_data=None)
fp = FeaturePreprocessor("/tmp/o.tsv", StatConfig(json.loads(json.dumps(sc["stats"]))), anc_allele_available=sc["anc"])
batch = fp.score_windows(wg)
eng = Engine.get()
for cb in batch.combos:
    print(cb.ref_pop, cb.tgt_pop, cb.src_comb, "windows", cb.windows.tolist(), "nsnps", cb.nsnps.tolist())
    print(" dd", cb.dd[:, 0].tolist())
    # standalone
    ref = data["ref"][cb.ref_pop].GT; tgt = data["tgt"][cb.tgt_pop].GT; src = data["src"]["S0"].GT
    p = data["ref"][cb.ref_pop].POS
    for (s, e) in cb.windows.tolist():
        m = (p >= s) & (p <= e)
        want = O.dd_stat(ref[m].astype(np.int64), tgt[m].astype(np.int64), [src[m].astype(np.int64)]) if m.any() else None
        print("   oracle", s, e, int(m.sum()), want)
    rt, tt, st = eng.tile(ref), eng.tile(tgt), eng.tile(src)
    lo, hi = eng.window_bounds(torch.as_tensor(p.astype(np.int32)).cuda(), cb.windows[:, 0], cb.windows[:, 1])
    dd = eng.window_dd(eng.site_absdiff(rt, st), rt.n_ind, eng.site_absdiff(tt, st), tt.n_ind, lo, hi)
    print(" standalone", dd.cpu().tolist(), lo.tolist(), hi.tolist())
