"""Container-only calibration (test infrastructure, like make_golden.py; run from the repo root): oracle (oracle/sai_oracle.py) vs the real reference on the same
in-memory synthetic input, one process.  The reference never leaves this container; only the
measured ratio is recorded."""
import sys, time, types, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from itertools import combinations
allel = types.ModuleType("allel"); allel.GenotypeVector = allel.GenotypeArray = object
natsort = types.ModuleType("natsort"); natsort.natsorted = sorted
sys.modules.update(allel=allel, natsort=natsort, pysam=types.ModuleType("pysam"))
sys.path.insert(0, "/root/reference")
import sai.stats
from sai.configs import PloidyConfig, StatConfig
from sai.generators import WindowGenerator
from sai.preprocessors import FeaturePreprocessor
from sai.utils import split_genome
from sai.utils.genomic_dataclasses import ChromosomeData
from oracle import sai_oracle as O
from sai_amd import _ffi
import bench

n_sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40000
lib = _ffi.load()
def blk(stream, n):
    out = np.empty((n_sites, n), dtype=np.int8)
    _ffi.check(lib.sai_synth_fill_host(bench.SEED, 1, 0, n_sites, stream, n, 2, 0, out.ctypes.data_as(C.c_void_p)))
    return out.astype(np.int64)
gaps = np.empty(n_sites, dtype=np.int32); lib.sai_synth_gaps_host(bench.SEED, 1, 0, n_sites, gaps.ctypes.data_as(C.c_void_p))
pos = np.cumsum(gaps).astype(np.int32)
ref, tgt, src = blk(0, 1000), blk(1, 1000), blk(2, 2)
stats = {"U": {"ref": {"ref": 0.01}, "tgt": {"tgt": 0.5}, "src": {"src": "=1"}}, "Q": {"ref": {"ref": 0.01}, "tgt": {"tgt": 0.95}, "src": {"src": "=1"}}}
pl = {"ref": {"ref": 2}, "tgt": {"tgt": 2}, "src": {"src": 2}}
# reference
wg = object.__new__(WindowGenerator)
wg.win_len, wg.win_step, wg.chr_name, wg.ploidy_config = 50000, 25000, "1", PloidyConfig(pl)
wg.ref_data = {"ref": ChromosomeData(POS=pos, REF=None, ALT=None, GT=ref)}; wg.ref_samples = {"ref": []}
wg.tgt_data = {"tgt": ChromosomeData(POS=pos, REF=None, ALT=None, GT=tgt)}; wg.tgt_samples = {"tgt": []}
wg.src_data = {"src": ChromosomeData(POS=pos, REF=None, ALT=None, GT=src)}; wg.src_samples = {"src": []}
wg.out_data = wg.out_samples = None; wg.num_src = 1
wg.src_combinations = list(combinations(["src"], 1))
wg.tgt_windows = {"tgt": split_genome(pos, 50000, 25000)}
import json
fp = FeaturePreprocessor(output_file="/tmp/x.tsv", stat_config=StatConfig(json.loads(json.dumps(stats))), anc_allele_available=True)
t0 = time.perf_counter(); ritems = []
for w in wg.get(): ritems.extend(fp.run(**w))
t_ref = time.perf_counter() - t0
vstats = {k: {"ref": v["ref"], "tgt": v["tgt"], "src": {"src": ("=", 1.0)}} for k, v in stats.items()}
t0 = time.perf_counter()
oitems = O.run_chunk("1", {"ref": O.Chrom(pos, ref)}, {"tgt": O.Chrom(pos, tgt)}, {"src": O.Chrom(pos, src)}, 50000, 25000, vstats, pl, True)
t_or = time.perf_counter() - t0
assert len(ritems) == len(oitems)
for a, b in zip(ritems, oitems):
    assert a["U"] == b["U"] and (a["Q"] == b["Q"] or (a["Q"] != a["Q"] and b["Q"] != b["Q"])) and a["nsnps"] == b["nsnps"]
print(f"{len(ritems)} windows, {n_sites} sites: reference {t_ref:.2f} s ({len(ritems)/t_ref:.1f} w/s), oracle {t_or:.2f} s ({len(oitems)/t_or:.1f} w/s), ratio oracle/reference {t_or/t_ref:.3f}")
