"""What a kernel queued behind a running site pass costs the pass (C3 block, the plain scorer's own plans): nothing,
a tiny kernel, window_bounds, the copy, the whole stage on the pass's stream, the stage on a second stream.  GPU box only."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch
import bench
from sai_amd.engine import Engine
from sai_amd.resident import ResidentScorer
from sai_amd.sharding import build_synth_shard

wl = bench.make_workload("c3", 0, 0, "strong", 1)
eng = Engine.get(0)
block, lay, _ = build_synth_shard(eng, wl, 0, 1)
windows = [(s, e) for _, s, e in lay.windows]
sets = wl.params()
plain = ResidentScorer(eng, block, windows, sets, overlap=False, cap_u=1 << 16, cap_q=1 << 16, fetch_lists=1 << 16)
plain.step(); plain.results()
cp, sp = plain._count_plans[0], plain._stage_plans[0]
ch = plain.chunks[0]
bounds = eng.plan(); bounds.add_window_bounds(block.pos, plain.win_start, plain.win_end, None, None, plain.lo, plain.hi)
copy = eng.plan(); copy.add_copy_to_host(ch.host_whole if ch.joined else ch.host_head, ch.dev_whole if ch.joined else ch.bufs[5])
side = torch.cuda.Stream()
small = torch.zeros(1024, device=eng.device)

def run(label, after):
    ds, ws = [], []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(); cp.run(); e1.record(); after(e1); torch.cuda.synchronize()
        ws.append(1e3 * (time.perf_counter() - t0)); ds.append(e0.elapsed_time(e1))
    print(f"{label:60s} pass {sum(ds[2:])/6:.4f} ms (min {min(ds[2:]):.4f}), wall {sum(ws[2:])/6:.3f} ms")

def on_side(e1):
    e1.synchronize()
    with torch.cuda.stream(side):
        sp.run()
    side.synchronize()

def on_side_stream_wait(e1):
    side.wait_event(e1)
    with torch.cuda.stream(side):
        sp.run()

import torch as _t
other_out = (_t.full((block.n_sites,), float("nan"), dtype=_t.float64, device=eng.device), eng.alloc_planes(block.n_sites, len(sets)))
other = eng.plan(); other.add_site_pass(block.pops, block.ploidies, sets, other_out, counts=None, freq_mode="candidates")
stats_only = eng.plan()
stats_only.add_window_stats(plain._tgt_freq[0], plain._flags[0][:, : 3 * len(sets)], sets, plain.lo, plain.hi, plain.list_pos, ch.bufs)

big = _t.zeros(32 << 20, device=eng.device)
counts = _t.zeros((len(block.pops), block.n_sites, 2), dtype=_t.int32, device=eng.device)
flags_out = (_t.full((block.n_sites,), float("nan"), dtype=_t.float64, device=eng.device), eng.alloc_planes(block.n_sites, len(sets)))
flags = eng.plan(); flags.add_site_flags(counts, block.ploidies, sets, flags_out)
probe_buf = _t.zeros(1 << 20, dtype=_t.int32, device=eng.device)
probe_out = _t.zeros((1,), dtype=_t.int32, device=eng.device)
from sai_amd import _ffi as _f

for rnd in range(2):
    run("a tiny torch kernel, then window_bounds behind it", lambda e: (small.add_(1.0), bounds.run()))
    run("a tiny torch kernel, then the whole stage behind it", lambda e: (small.add_(1.0), sp.run()))
    run("an event record, then window_bounds behind it", lambda e: (_t.cuda.Event().record(), bounds.run()))
    run("a torch elementwise kernel over 128 MB behind it", lambda e: big.add_(1.0))
    run("site_flags (one wave per tile, 4 per workgroup) behind it", lambda e: flags.run())
    run("the stream-read probe over 4 MB behind it", lambda e: _f.check(eng.lib.sai_probe_stream_read(eng.ctx, eng._ptr(probe_buf), 4 << 20, eng._ptr(probe_out), eng._stream())))
    run("nothing behind the pass", lambda e: None)
    run("another site pass behind it (other output buffers)", lambda e: other.run())
    run("window_stats + scan + lists behind it (no bounds)", lambda e: stats_only.run())
    run("a tiny torch kernel behind it", lambda e: small.add_(1.0))
    run("window_bounds behind it", lambda e: bounds.run())
    run("the copy of the records behind it", lambda e: copy.run())
    run("the whole stage behind it (same stream)", lambda e: sp.run())
    run("the stage on a second stream after a host wait", on_side)
    run("the stage on a second stream behind a stream-side wait", on_side_stream_wait)
