"""How long does the host take to notice that a ~3 ms kernel has finished?  Event.synchronize(),
Stream.synchronize() and polling Event.query(), measured against the kernel's own HIP-event duration
(the pipelined ResidentScorer polls; DESIGN.md section 4 says why)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sai_amd.engine import Engine
from sai_amd.resident import synth_block

eng = Engine.get(0)
block = synth_block(eng, 1, 1, 10_000_000, 1000, 1000, [2])
counts = eng.site_counts(block.pops)
torch.cuda.synchronize()
for mode in ("event.synchronize", "stream.synchronize", "poll event.query", "blocking event"):
    over = []
    for _ in range(20):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True, blocking=(mode == "blocking event"))
        e0.record()
        t0 = time.perf_counter()
        eng.site_counts(block.pops, out=counts)
        e1.record()
        if mode == "stream.synchronize":
            torch.cuda.current_stream().synchronize()
        elif mode == "poll event.query":
            while not e1.query():
                pass
        else:
            e1.synchronize()
        host_ms = (time.perf_counter() - t0) * 1e3
        over.append(host_ms - e0.elapsed_time(e1))
    over.sort()
    print(f"{mode:22s}: host wait - kernel time: median {over[len(over) // 2] * 1e3:7.1f} us, max {over[-1] * 1e3:7.1f} us")
