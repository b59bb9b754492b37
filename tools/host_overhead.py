"""Host time of one ResidentScorer.step() (enqueue only): tiny block, so the GPU never limits."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd import _ffi
from sai_amd.engine import Engine
from sai_amd.resident import ResidentScorer, default_windows, synth_block

eng = Engine.get(0)
block = synth_block(eng, 1, 1, 20_000, 64, 64, [2])
pos = block.pos.cpu().numpy()
windows = default_windows(int(pos[0]), int(pos[-1]), 50_000, 25_000)
sets = [_ffi.make_params(0.01, 0.5, 0.95, [("=", 1.0)], True)]
for layout in ("int8", "packed2"):
    for overlap in (False, True):
        sc = ResidentScorer(eng, block, windows, sets, layout=layout, overlap=overlap)
        for _ in range(20):
            sc.step(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            sc.step(True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{layout:8s} overlap={overlap}: host {1e6 * (t1 - t0) / n:.1f} us/step enqueue, {1e6 * (t2 - t0) / n:.1f} us/step incl. drain")
