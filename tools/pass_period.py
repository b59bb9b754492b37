"""Period of back-to-back site passes on ONE queue with nothing else on the chip, against the pass's own
duration (its dispatch packet's timestamps): what a queue needs between two kernels of this size.
Usage: python tools/pass_period.py [sites=1e6] [n_ref=200]"""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd import _ffi
from sai_amd.engine import Engine, LaunchEvent
from sai_amd.resident import synth_block

n_sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = Engine.get(0)
block = synth_block(eng, 20260632, 1, n_sites, n, n, [2])
sets = [_ffi.make_params(0.01, 0.5, 0.95, [("=", 1.0)], True)]
bufs = []
for _ in range(3):
    tf = torch.full((n_sites,), float("nan"), dtype=torch.float64, device=eng.device)
    bufs.append((tf, eng.alloc_planes(n_sites, 1)))
plans = []
for out in bufs:
    p = eng.plan()
    p.add_site_pass(block.pops, block.ploidies, sets, out, counts=None, freq_mode="candidates")
    plans.append(p)
for mode in ("plain", "carried"):
    pairs = []
    if mode == "carried":
        for p in plans:
            pair = (LaunchEvent(eng), LaunchEvent(eng))
            p.set_pass_events(*pair)
            pairs.append(pair)
    for k in range(30):
        plans[k % 3].run()
    torch.cuda.synchronize()
    for reps in (1, 300):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for k in range(reps):
            plans[k % 3].run()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        own = f", own duration of the last pass {1e3 * pairs[(reps - 1) % 3][0].elapsed_time(pairs[(reps - 1) % 3][1]):.1f} us" if pairs else ""
        print(f"{mode:8s} {reps:4d} passes back to back: {1e3 * e0.elapsed_time(e1) / reps:.1f} us per pass (events), {1e6 * (t1 - t0) / reps:.1f} us wall{own}")

# the scorer's loop without a windows stage: enqueue pass k, then wait on the host for pass k - 1
def loop(label, fresh, wait, reps=300):
    pairs = [None] * 3
    done = []
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        b = k % 3
        if fresh or pairs[b] is None:
            pairs[b] = (LaunchEvent(eng), LaunchEvent(eng))
            plans[b].set_pass_events(*pairs[b])
        plans[b].run()
        done.append(pairs[b])
        torch.cuda.current_stream().query()
        if wait and k:
            done[k - 1][1].synchronize()
    e1.record()
    torch.cuda.synchronize()
    own = sum(a.elapsed_time(b) for a, b in done[-30:]) / 30 if fresh else done[-1][0].elapsed_time(done[-1][1])
    print(f"{label}: {1e3 * e0.elapsed_time(e1) / reps:.1f} us per pass, own duration {1e3 * own:.1f} us")

loop("carried, same three pairs, nobody waits      ", False, False)
loop("carried, same three pairs, host waits for k-1", False, True)
loop("carried, fresh pair per pass, nobody waits   ", True, False)
loop("carried, fresh pair per pass, host waits k-1 ", True, True)
