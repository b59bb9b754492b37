"""What a run's 3-5 % come from: the C3 site pass over tens of seconds next to what the card reports about
itself (temperatures, power, clocks from the hwmon / pp_dpm files of its sysfs node), then two scorers on the
same block taking turns (is a slow scorer slow because of WHEN it ran or because of WHAT it is?).

    python tools/drift_probe.py [--seconds 25] [--workload c3] > gpurun_out/drift_probe.txt

Nothing here is product code; it only reads sysfs and times passes with the scorer's own events.
"""

from __future__ import annotations

import argparse
import glob
import os
import sys
import threading
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402


def card_node(bus_id: str):
    """sysfs device directory of the card with this PCI bus id (all cards of the host may be visible)."""
    want = bus_id.lower()
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        real = os.path.realpath(dev)
        if os.path.basename(real).lower() == want:
            return real
    nodes = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    return os.path.realpath(nodes[0]) if len(nodes) == 1 else None


def read(path: str):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Sampler(threading.Thread):
    def __init__(self, node: str, period: float = 0.25):
        super().__init__(daemon=True)
        self.node, self.period, self.rows, self.stop_flag = node, period, [], False
        self.files = {}
        for hw in glob.glob(f"{node}/hwmon/hwmon*"):
            for f in sorted(glob.glob(f"{hw}/temp*_input") + glob.glob(f"{hw}/power*_average") + glob.glob(f"{hw}/power*_input")
                            + glob.glob(f"{hw}/freq*_input")):  # fmt: skip
                label = read(f.rsplit("_", 1)[0] + "_label") or os.path.basename(f)
                self.files[f"{label}:{os.path.basename(f).split('_')[0]}"] = f
        self.dpm = {k: f"{node}/{k}" for k in ("pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_sclk", "pp_dpm_socclk") if os.path.exists(f"{node}/{k}")}

    def sample(self) -> dict:
        row = {"t": time.perf_counter()}
        for k, f in self.files.items():
            v = read(f)
            if v is not None and v.lstrip("-").isdigit():
                row[k] = int(v)
        for k, f in self.dpm.items():
            txt = read(f) or ""
            cur = [ln for ln in txt.splitlines() if ln.rstrip().endswith("*")]
            if cur:
                row[k] = cur[0].split(":")[1].strip().rstrip("*").strip()
        busy = read(f"{self.node}/mem_busy_percent")
        if busy is not None:
            row["mem_busy"] = busy
        return row

    def run(self) -> None:
        while not self.stop_flag:
            self.rows.append(self.sample())
            time.sleep(self.period)

    def mean_between(self, t0: float, t1: float) -> dict:
        rows = [r for r in self.rows if t0 <= r["t"] <= t1] or [self.sample()]
        out = {}
        for k in rows[0]:
            if k == "t":
                continue
            vals = [r[k] for r in rows if k in r]
            if vals and all(isinstance(v, int) for v in vals):
                out[k] = sum(vals) / len(vals)
            elif vals:
                out[k] = vals[-1]
        return out


def fmt(d: dict) -> str:
    parts = []
    for k, v in d.items():
        if isinstance(v, float):
            if k.endswith(":temp") or ":temp" in k:
                parts.append(f"{k}={v / 1000:.1f}C")
            elif ":power" in k:
                parts.append(f"{k}={v / 1e6:.0f}W")
            elif ":freq" in k:
                parts.append(f"{k}={v / 1e6:.0f}MHz")
            else:
                parts.append(f"{k}={v:.0f}")
        else:
            parts.append(f"{k}={v}")
    return " ".join(parts)


def timed_batch(scorer, n: int):
    import torch

    before = len(scorer.site_pass_ms())
    t0 = time.perf_counter()
    for _ in range(n):
        scorer.step(time_counts=True)
    scorer.flush()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = scorer.site_pass_ms()[before:]
    ms.sort()
    return t0, t1, ms


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=25.0)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--idle", type=float, default=10.0)
    a = ap.parse_args()
    import torch

    from sai_amd.resident import ResidentScorer

    dev = bench.HipDevice()
    dev.start(0)
    ident = dev.identity()
    node = card_node(ident["pci_bus_id"])
    print("device", ident, "sysfs", node, flush=True)
    if node is None:
        print("no sysfs node for this card: temperatures / clocks unavailable")
    sam = Sampler(node or "/nonexistent")
    print("sensors:", sorted(sam.files), sorted(sam.dpm), flush=True)
    sam.start()
    time.sleep(1.0)
    print("idle      ", fmt(sam.mean_between(0, time.perf_counter())), flush=True)

    wl = bench.make_workload(a.workload)
    args = SimpleNamespace(layout="int8", overlap="on")
    t_build = time.perf_counter()
    block, lay, _, scorer = dev.build(wl, 0, 1, args)
    torch.cuda.synchronize()
    print(f"block built in {time.perf_counter() - t_build:.2f} s", flush=True)
    gb = block.genotype_bytes / 1e9

    def line(tag, t0, t1, ms):
        med = ms[len(ms) // 2]
        print(f"{tag:10s} t={t0 - T0:6.1f}s n={len(ms):4d} pass min/med/max {ms[0]:.3f} {med:.3f} {ms[-1]:.3f} ms "
              f"= {gb / med * 1e3:.0f} GB/s | {fmt(sam.mean_between(t0, t1))}", flush=True)  # fmt: skip

    T0 = time.perf_counter()
    # the very first passes of the process, in small batches: what a cold card does
    for _ in range(5):
        line("first", *timed_batch(scorer, 20))
    while time.perf_counter() - T0 < a.seconds:
        line("sustained", *timed_batch(scorer, a.batch))
    time.sleep(a.idle)
    print("after idle", fmt(sam.mean_between(time.perf_counter() - 1.0, time.perf_counter())), flush=True)
    for _ in range(4):
        line("resumed", *timed_batch(scorer, 25))

    # a second scorer on the same block, taking turns with the first
    windows = [(s, e) for _, s, e in lay.windows]
    other = ResidentScorer(dev.eng, block, windows, wl.params(), cap_u=1 << 22, cap_q=1 << 22, layout="int8", overlap=True,
                           window_segment=lay.window_segment)  # fmt: skip
    other.step()
    other.flush()
    for turn in range(4):
        line(f"A turn {turn}", *timed_batch(scorer, a.batch))
        line(f"B turn {turn}", *timed_batch(other, a.batch))
    probe = dev.stream_read_probe(block)
    print(f"stream-read probe {probe:.0f} GB/s", flush=True)
    sam.stop_flag = True


if __name__ == "__main__":
    main()
