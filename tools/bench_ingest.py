"""Throughput of the native VCF tokenizer (host), of the Python reader and of the GPU reader on a
synthetic VCF, plain and bgzip."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sai_amd.utils.native_vcf import load_dosage, default_threads
from sai_amd.utils.read_data import _load_python

n_sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 2002
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "synth.vcf")
names = [f"i{k}" for k in range(n_samples)]
calls = np.array(["0|0", "0|1", "1|0", "1|1", ".|."])
with open(path, "w") as f:
    f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
    pos = 0
    for s in range(n_sites):
        pos += int(rng.integers(1, 50))
        row = calls[rng.choice(5, size=n_samples, p=[0.7, 0.1, 0.1, 0.095, 0.005])]
        f.write(f"1\t{pos}\t.\tA\tT\t100\tPASS\t.\tGT\t" + "\t".join(row) + "\n")
size = os.path.getsize(path)
print(f"VCF: {n_sites} sites x {n_samples} samples, {size / 1e6:.1f} MB")
load_dosage(path, "1", names[:1], [2], n_threads=1)  # loads the library (and torch) once
for th in (1, 4, default_threads()):
    t0 = time.perf_counter(); pos_n, dos_n, _, _ = load_dosage(path, "1", names, [2] * n_samples, n_threads=th); dt = time.perf_counter() - t0
    print(f"native, {th:2d} threads: {dt:.3f} s  {size / dt / 1e6:8.1f} MB/s  {n_sites * n_samples / dt / 1e6:8.1f} M genotypes/s")
sub = min(n_sites, 1500)
t0 = time.perf_counter(); pos_p, dos_p, _, _ = _load_python(path, "1", names, 2, None, int(pos_n[sub - 1]), None); dt = time.perf_counter() - t0
frac = sub / n_sites
print(f"python reader on the first {sub} sites: {dt:.3f} s  {size * frac / dt / 1e6:8.1f} MB/s")
assert np.array_equal(dos_p, dos_n[:sub])
# the same file as bgzip (64 KiB members, what real VCFs use): members are inflated in parallel
import struct, zlib
gzpath = path + ".gz"
t0 = time.perf_counter()
with open(path, "rb") as f, open(gzpath, "wb") as g:
    def member(chunk):
        comp = zlib.compressobj(6, zlib.DEFLATED, -15)
        raw = comp.compress(chunk) + comp.flush()
        head = b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1)
        return head + raw + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    while True:
        chunk = f.read(65280)
        if not chunk:
            break
        g.write(member(chunk))
    g.write(member(b""))
gzsize = os.path.getsize(gzpath)
print(f"bgzip copy: {gzsize / 1e6:.1f} MB (written in {time.perf_counter() - t0:.1f} s)")
for th in (1, 4, default_threads()):
    t0 = time.perf_counter(); pos_g, dos_g, _, _ = load_dosage(gzpath, "1", names, [2] * n_samples, n_threads=th); dt = time.perf_counter() - t0
    print(f"native bgzip, {th:2d} threads: {dt:.3f} s  {size / dt / 1e6:8.1f} MB/s of text  {gzsize / dt / 1e6:8.1f} MB/s of file")
assert np.array_equal(dos_g, dos_n) and np.array_equal(pos_g, pos_n)
# the GPU reader (text over PCIe, tokenised on the GPU, the dosages stay in HBM) when a GPU is there
try:
    import torch
    have_gpu = torch.cuda.is_available()
except Exception:
    have_gpu = False
if have_gpu:
    from sai_amd.engine import Engine
    from sai_amd.utils.device_vcf import load_dosage_device
    eng = Engine.get(0)
    for label, pth in (("plain", path), ("bgzip", gzpath)):
        load_dosage_device(eng, pth, "1", names, [2] * n_samples)  # staging buffers are allocated once per engine
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pos_d, dos_d, _, _ = load_dosage_device(eng, pth, "1", names, [2] * n_samples)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert np.array_equal(pos_d, pos_n) and np.array_equal(dos_d.cpu().numpy(), dos_n)
        print(f"GPU reader, {label}: {best:.3f} s  {size / best / 1e6:8.1f} MB/s of text (dosages left in HBM)")
import gzip
t0 = time.perf_counter()
with gzip.open(gzpath, "rb") as f:
    while f.read(1 << 24):
        pass
print(f"python gzip.read of the same file (inflate only, one thread): {size / (time.perf_counter() - t0) / 1e6:.1f} MB/s of text")
os.remove(path); os.remove(gzpath); os.rmdir(d)
