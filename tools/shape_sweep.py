"""Pipelined step of a synthetic block of a given shape (the site pass's grid rule against forced grids):
python tools/shape_sweep.py n_ref n_tgt n_sites n_sets [steps] [sources: "1,1" or "2"]   (SAI_STREAM_WAVES_PER_CU forces a grid)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sai_amd import _ffi
from sai_amd.engine import Engine
from sai_amd.resident import ResidentScorer, default_windows, synth_block

n_ref, n_tgt, n_sites, n_sets = int(sys.argv[1]), int(sys.argv[2]), int(float(sys.argv[3])), int(sys.argv[4])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
eng = Engine.get(0)
src = [int(x) for x in (sys.argv[6] if len(sys.argv) > 6 else "1,1").split(",")]
block = synth_block(eng, 20260640, 1, n_sites, n_ref, n_tgt, src)
windows = default_windows(int(block.pos[0]), int(block.pos[-1]), 50000, 25000)
grid = [(op, y1, y2) for op in ("=", ">=") for y1 in (1.0, 0.5, 0.0) for y2 in (1.0, 0.5, 0.0)][:n_sets]
sets = [_ffi.make_params(0.01, 0.5, 0.95, [(op, y1), (op, y2)][: len(src)], True) for op, y1, y2 in grid]
sc = ResidentScorer(eng, block, windows, sets, cap_u=1 << 22, cap_q=1 << 22, overlap=True)
for _ in range(3):
    sc.step()
sc.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    sc.step(time_counts=True)
sc.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
ms = sc.site_pass_ms()
gb = block.genotype_bytes / 1e9
print(f"waves={os.environ.get('SAI_STREAM_WAVES_PER_CU', 'rule')} {n_ref}/{n_tgt}/{'+'.join(map(str, src))} x {n_sites} sites, {n_sets} sets, {len(windows)} windows: "
      f"step {dt:.4f} ms, site pass {sum(ms) / len(ms):.4f} ms = {gb / (sum(ms) / len(ms)) * 1e3 / 8000:.3f} of peak ({gb:.2f} GB)")
