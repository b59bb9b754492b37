"""Does a big process that has just exited slow the next one down?  (Some bench processes ran with every pass 0.1 ms
slower and a lower plain-read rate: always a few seconds after a process holding 60-100 GB had ended.)  A child
allocates and fills G GB of HBM and exits; right after it this process reads a 4-GB array with the library's plain-read
kernel every 0.25 s for a while and prints the rate over time.

    python tools/after_free_probe.py [--gb 150] [--seconds 20]
"""

from __future__ import annotations

import argparse
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

CHILD = """
import sys, torch
gb = int(sys.argv[1])
bufs = [torch.empty((1 << 30,), dtype=torch.int8, device="cuda").fill_(1) for _ in range(gb)]
torch.cuda.synchronize()
print("child held", gb, "GB", flush=True)
"""


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=int, default=150)
    ap.add_argument("--seconds", type=float, default=20.0)
    a = ap.parse_args()
    for gb in (0, a.gb, a.gb):
        if gb:
            subprocess.run([sys.executable, "-c", CHILD, str(gb)], check=True)
        t_exit = time.perf_counter()
        code = (
            "import sys, time; sys.path.insert(0, %r)\n"
            "import torch\n"
            "from sai_amd.engine import Engine\n"
            "t0 = time.perf_counter(); eng = Engine.get(0)\n"
            "buf = torch.zeros((4 << 30,), dtype=torch.int8, device=eng.device); torch.cuda.synchronize()\n"
            "print('  first touch of the GPU %%.2f s after start' %% (time.perf_counter() - t0), flush=True)\n"
            "end = time.perf_counter() + %f\n"
            "while time.perf_counter() < end:\n"
            "    r = eng.probe_stream_read(buf, repeats=1, launches=8)\n"
            "    print('  t=%%5.2f s  %%6.0f GB/s' %% (time.perf_counter() - t0, r), flush=True)\n"
            "    time.sleep(0.25)\n"
        ) % (str(ROOT), a.seconds if gb else 3.0)
        print(f"after a child that held {gb} GB (exited {time.perf_counter() - t_exit:.2f} s ago):", flush=True)
        subprocess.run([sys.executable, "-c", code], check=True)


if __name__ == "__main__":
    main()
