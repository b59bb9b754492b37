"""bgzip VCF -> dosages in HBM: the GPU-inflate route against the host-inflate stream, per staging
size (GPU box): python tools/bgzf_rate.py [MB of text]"""
import os
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from sai_amd.engine import Engine  # noqa: E402
from sai_amd.utils import device_vcf  # noqa: E402


def member(chunk: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = comp.compress(chunk) + comp.flush()
    head = b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1)
    return head + raw + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))


def write_tbi(gz_path: str, chrom: str, pos: np.ndarray, voff: np.ndarray, end_voff: int) -> None:
    """Minimal tabix index (TBI) of one chromosome from the records' positions and virtual offsets:
    one chunk in bin 0 and the linear index (first record per 16 kb window, empty windows filled from
    the next one as htslib does), gzip-compressed."""
    import gzip

    win = (pos.astype(np.int64) - 1) >> 14
    n_win = int(win[-1]) + 1
    ioff = np.zeros(n_win, dtype=np.uint64)
    first = np.flatnonzero(np.diff(win, prepend=-1) > 0)
    ioff[win[first]] = voff[first]
    for w in range(n_win - 2, -1, -1):
        if ioff[w] == 0:
            ioff[w] = ioff[w + 1]
    name = chrom.encode() + b"\0"
    out = b"TBI\1" + struct.pack("<8i", 1, 2, 1, 2, 0, ord("#"), 0, len(name)) + name
    out += struct.pack("<i", 1) + struct.pack("<Ii", 0, 1) + struct.pack("<QQ", int(voff[0]), int(end_voff))
    out += struct.pack("<i", n_win) + ioff.astype("<u8").tobytes()
    with gzip.open(gz_path + ".tbi", "wb") as f:
        f.write(out)


def main() -> None:
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 480
    n_samples = 2002
    rng = np.random.default_rng(1)
    names = [f"i{k}" for k in range(n_samples)]
    calls = np.array([b"0|0", b"0|1", b"1|0", b"1|1", b".|."])
    header = ("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n").encode()
    rows = []
    for _ in range(1000):
        rows.append(b"\tA\tT\t100\tPASS\t.\tGT\t" + b"\t".join(calls[rng.choice(5, size=n_samples, p=[0.7, 0.1, 0.1, 0.095, 0.005])]) + b"\n")
    parts, size, pos = [header], len(header), 0
    while size < mb << 20:
        pos += 7
        line = b"1\t%d\t." % pos + rows[pos % 1000]
        parts.append(line)
        size += len(line)
    text = b"".join(parts)
    n_lines = len(parts) - 1
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(d, "synth.vcf.gz")
    t0 = time.perf_counter()
    with ThreadPoolExecutor(16) as ex:
        blocks = list(ex.map(member, [text[i : i + 65280] for i in range(0, len(text), 65280)]))
    with open(path, "wb") as f:
        f.write(b"".join(blocks) + member(b""))
    print(f"{len(text) / 1e6:.0f} MB of text, {n_lines} lines, {os.path.getsize(path) / 1e6:.1f} MB bgzip ({time.perf_counter() - t0:.1f} s)", flush=True)
    eng = Engine.get(0)
    if "--regions" in sys.argv:
        # region seek through a tabix index: time and bytes read against the size of the region
        line_start = np.cumsum([len(p) for p in parts])[:-1]  # text offset of every record line
        coff = np.concatenate([[0], np.cumsum([len(b) for b in blocks])])
        voff = (coff[line_start // 65280].astype(np.uint64) << np.uint64(16)) | (line_start % 65280).astype(np.uint64)
        all_pos = 7 * np.arange(1, n_lines + 1)
        write_tbi(path, "1", all_pos, voff, int(coff[-1]) << 16)
        size = os.path.getsize(path)
        for label, env in (("host inflate + seek", "0"), ("GPU inflate + seek", "1")):
            os.environ["SAI_AMD_GPU_INFLATE"] = env
            for frac in (0.002, 0.01, 0.05, 0.25, 1.0):
                k = max(int(n_lines * frac), 1)
                a = (n_lines - k) // 2
                start, end = int(all_pos[a]), int(all_pos[a + k - 1])
                best = 1e9
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    pos_d, dos_d, _, _ = device_vcf.load_dosage_device(eng, path, "1", names, [2] * n_samples, start, end)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                assert pos_d.tolist() == all_pos[a : a + k].tolist()
                read = eng._inflate_state["last"]["comp_bytes"] if env == "1" else 0
                print(f"{label}: region of {k:6d} records ({100 * frac:5.1f} %): {1e3 * best:7.2f} ms"
                      + (f", {read / 1e6:6.2f} MB of {size / 1e6:.1f} MB read" if env == "1" else ""), flush=True)
        os.remove(path + ".tbi")
        os.remove(path)
        os.rmdir(d)
        return
    want = None
    for label, env in (("host inflate", "0"), ("GPU inflate", "1")):
        os.environ["SAI_AMD_GPU_INFLATE"] = env
        for cap_mb in ((124, 248) if env == "1" and os.environ.get("SAI_AMD_INGEST_TRACE") else (64, 124, 248, 496)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            device_vcf.load_dosage_device(eng, path, "1", names, [2] * n_samples, buffer_bytes=cap_mb << 20)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pos_d, dos_d, _, _ = device_vcf.load_dosage_device(eng, path, "1", names, [2] * n_samples, buffer_bytes=cap_mb << 20)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            if want is None:
                want = (pos_d.copy(), dos_d.clone())
            assert np.array_equal(pos_d, want[0]) and torch.equal(dos_d, want[1]) and len(pos_d) == n_lines
            print(f"{label}, {cap_mb:3d} MiB staging: {1e3 * best:.1f} ms = {len(text) / best / 1e9:.1f} GB/s of text "
                  f"(first call, buffers allocated and page-locked: {1e3 * first:.0f} ms)", flush=True)
    os.remove(path)
    os.rmdir(d)


if __name__ == "__main__":
    main()
