"""Per-step site-pass durations of the pipelined scorer (does the windows stage fall behind?)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd import _ffi
from sai_amd.engine import Engine
from sai_amd.resident import ResidentScorer, default_windows, synth_block

eng = Engine.get(0)
block = synth_block(eng, 20260633, 1, 10_000_000, 1000, 1000, [2])
windows = default_windows(int(block.pos[0]), int(block.pos[-1]), 50000, 25000)
prm = _ffi.make_params(0.01, 0.5, 0.95, [("=", 1.0)], True)
for overlap in (True, False):
    sc = ResidentScorer(eng, block, windows, [prm], cap_u=1 << 22, cap_q=1 << 22, overlap=overlap)
    stage_ev = []
    for plan in sc._stage_plans:  # the windows stage is one prepared launch sequence per buffer set
        def timed_run(orig=plan.run, stage_ev=stage_ev):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); orig(); b.record()
            stage_ev.append((a, b))
        plan.run = timed_run
    for _ in range(3):
        sc.step()
    torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    import time
    host = []
    for _ in range(n):
        h0 = time.perf_counter()
        sc.step(True)
        host.append((time.perf_counter() - h0) * 1e3)
    t1.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in sc.count_events]
    starts = [sc.count_events[0][0].elapsed_time(a) for a, _ in sc.count_events]
    print(f"overlap={overlap}: total {t0.elapsed_time(t1) / n:.3f} ms/step; site pass per step:")
    print("  " + " ".join(f"{m:.2f}" for m in ms))
    print("  host ms per step() call: " + " ".join(f"{h:.2f}" for h in host[:40]))
    print("  gaps between site-pass starts: " + " ".join(f"{b - a:.2f}" for a, b in zip(starts, starts[1:])))
    ref = sc.count_events[0][0]
    st = stage_ev[3:]
    print("  step: site[start,end] stage[start,end] (ms from the first site pass)")
    for k in range(min(40, n)):
        s0 = ref.elapsed_time(sc.count_events[k][0]); s1 = ref.elapsed_time(sc.count_events[k][1])
        w0 = ref.elapsed_time(st[k][0]); w1 = ref.elapsed_time(st[k][1])
        print(f"   {k:2d}: site {s0:8.2f} {s1:8.2f} ({s1 - s0:.2f})  stage {w0:8.2f} {w1:8.2f} ({w1 - w0:.2f})")
