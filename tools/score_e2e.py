"""End-to-end `score` on a synthetic VCF (host parse -> GPU -> TSV): where does the time go?"""
import cProfile, io, os, pstats, sys, tempfile, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))

n_sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50000
n_ref, n_tgt, n_src = 1000, 1000, 2
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
names = [f"i{k}" for k in range(n_ref + n_tgt + n_src)]
calls = np.array(["0|0", "0|1", "1|0", "1|1", ".|."])
vcf = os.path.join(d, "synth.vcf")
t0 = time.perf_counter()
with open(vcf, "w") as f:
    f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
    pos = 0
    for s in range(n_sites):
        pos += int(rng.integers(1, 50))
        intro = rng.random() < 0.01
        p = rng.random() ** 4
        pr = 0.0 if intro else p
        row = np.concatenate([rng.binomial(1, pr, 2 * n_ref), rng.binomial(1, 0.6 if intro else p, 2 * n_tgt),
                              rng.binomial(1, 1.0 if intro else p, 2 * n_src)]).reshape(-1, 2)
        f.write(f"1\t{pos}\t.\tA\tT\t100\tPASS\t.\tGT\t" + "\t".join(f"{a}|{b}" for a, b in row) + "\n")
for grp, sl in (("ref", names[:n_ref]), ("tgt", names[n_ref:n_ref + n_tgt]), ("src", names[n_ref + n_tgt:])):
    with open(os.path.join(d, f"{grp}.list"), "w") as f:
        f.write("".join(f"{grp.upper()}\t{n}\n" for n in sl))
cfg = os.path.join(d, "cfg.yaml")
with open(cfg, "w") as f:
    f.write(f"""statistics:
  U:
    ref: {{REF: 0.01}}
    tgt: {{TGT: 0.5}}
    src: {{SRC: "=1"}}
  Q:
    ref: {{REF: 0.01}}
    tgt: {{TGT: 0.95}}
    src: {{SRC: "=1"}}
ploidies:
  ref: {{REF: 2}}
  tgt: {{TGT: 2}}
  src: {{SRC: 2}}
populations:
  ref: {d}/ref.list
  tgt: {d}/tgt.list
  src: {d}/src.list
""")
print(f"VCF {os.path.getsize(vcf) / 1e6:.0f} MB written in {time.perf_counter() - t0:.1f} s", flush=True)
from sai_amd.sai import score
import torch; torch.cuda.init()
out = os.path.join(d, "out.tsv")
score(vcf_file=vcf, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=None, output_file=out, config=cfg, num_workers=1)  # warm
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
score(vcf_file=vcf, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=None, output_file=out, config=cfg, num_workers=1)
pr.disable()
dt = time.perf_counter() - t0
rows = sum(1 for _ in open(out)) - 1
print(f"score: {dt:.3f} s for {n_sites} sites, {rows} windows -> {os.path.getsize(vcf) / dt / 1e6:.0f} MB/s of VCF")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[-3500:])
print(open(out).read()[:400])
# repeated calls in one process (a per-chromosome loop): any erratic host-side cost shows here
reps = []
for _ in range(12):
    t0 = time.perf_counter()
    score(vcf_file=vcf, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=None, output_file=out, config=cfg, num_workers=1)
    reps.append(time.perf_counter() - t0)
print("12 more calls, ms each:", " ".join(f"{1e3 * t:.1f}" for t in reps))
# every statistic the path offers, polarised: an ancestral-allele BED for every site and an outgroup
# (ten of the reference samples), fd / df / Danc / Dplus / DD next to U and Q
bed = os.path.join(d, "anc.bed")
with open(vcf) as f, open(bed, "w") as g:
    for line in f:
        if line[0] != "#":
            p1 = int(line.split("\t", 2)[1])
            g.write(f"1\t{p1 - 1}\t{p1}\tA\n")
with open(os.path.join(d, "out.list"), "w") as f:
    f.write("".join(f"OUT\t{n}\n" for n in names[:10]))
with open(os.path.join(d, "ref2.list"), "w") as f:
    f.write("".join(f"REF\t{n}\n" for n in names[10:n_ref]))
cfg_all = os.path.join(d, "cfg_all.yaml")
with open(cfg_all, "w") as f:
    f.write(f"""statistics:
  U:
    ref: {{REF: 0.01}}
    tgt: {{TGT: 0.5}}
    src: {{SRC: "=1"}}
  Q:
    ref: {{REF: 0.01}}
    tgt: {{TGT: 0.95}}
    src: {{SRC: "=1"}}
  fd: true
  df: true
  Danc: true
  Dplus: true
  DD: true
ploidies:
  ref: {{REF: 2}}
  tgt: {{TGT: 2}}
  src: {{SRC: 2}}
  outgroup: {{OUT: 2}}
populations:
  ref: {d}/ref2.list
  tgt: {d}/tgt.list
  src: {d}/src.list
  outgroup: {d}/out.list
""")
out_all = os.path.join(d, "out_all.tsv")
score(vcf_file=vcf, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=bed, output_file=out_all, config=cfg_all, num_workers=1)
pr = cProfile.Profile()
reps = []
for k in range(5):
    t0 = time.perf_counter()
    if k == 4:
        pr.enable()
    score(vcf_file=vcf, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=bed, output_file=out_all, config=cfg_all, num_workers=1)
    if k == 4:
        pr.disable()
    reps.append(time.perf_counter() - t0)
print("all seven statistics, polarised, with an outgroup: ms each", " ".join(f"{1e3 * t:.1f}" for t in reps))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[-4200:])
print(open(out_all).read()[:300])
# the same file as bgzip (what real VCFs are): members inflated on the GPU
import struct, zlib
from concurrent.futures import ThreadPoolExecutor
def member(chunk):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = comp.compress(chunk) + comp.flush()
    head = b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(raw) + 8 - 1)
    return head + raw + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
text = open(vcf, "rb").read()
with ThreadPoolExecutor(16) as ex:
    blocks = list(ex.map(member, [text[i : i + 65280] for i in range(0, len(text), 65280)]))
gz = vcf + ".gz"
open(gz, "wb").write(b"".join(blocks) + member(b""))
out_gz = os.path.join(d, "out_gz.tsv")
for label, env in (("GPU inflate", "1"), ("host inflate", "0")):
    os.environ["SAI_AMD_GPU_INFLATE"] = env
    score(vcf_file=gz, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=None, output_file=out_gz, config=cfg, num_workers=1)
    reps = []
    for _ in range(6):
        t0 = time.perf_counter()
        score(vcf_file=gz, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=None, output_file=out_gz, config=cfg, num_workers=1)
        reps.append(time.perf_counter() - t0)
    assert open(out_gz, "rb").read() == open(out, "rb").read()
    print(f"score on the bgzip copy ({os.path.getsize(gz) / 1e6:.1f} MB), {label}: ms each", " ".join(f"{1e3 * t:.1f}" for t in reps),
          f"-> {len(text) / min(reps) / 1e9:.1f} GB/s of text; same bytes as from the plain file")
# ... and with a tabix index next to it: the chromosome's first / last position come from two records
# (host, through the index) instead of a pass over the file, the load seeks
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bgzf_rate import write_tbi
arr = np.frombuffer(text, dtype=np.uint8)
starts = np.concatenate([[0], np.flatnonzero(arr == 10)[:-1] + 1])
starts = starts[arr[starts] != ord("#")]
rec_pos = np.array([int(text[a : a + 24].split(b"\t")[1]) for a in starts.tolist()])
coff = np.concatenate([[0], np.cumsum([len(b) for b in blocks])])
voff = (coff[starts // 65280].astype(np.uint64) << np.uint64(16)) | (starts % 65280).astype(np.uint64)
write_tbi(gz, "1", rec_pos, voff, int(coff[-1]) << 16)
os.environ["SAI_AMD_GPU_INFLATE"] = "1"
score(vcf_file=gz, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=None, output_file=out_gz, config=cfg, num_workers=1)
reps = []
for _ in range(6):
    t0 = time.perf_counter()
    score(vcf_file=gz, chr_name="1", win_len=50000, win_step=25000, anc_allele_file=None, output_file=out_gz, config=cfg, num_workers=1)
    reps.append(time.perf_counter() - t0)
assert open(out_gz, "rb").read() == open(out, "rb").read()
print(f"score on the indexed bgzip copy, GPU inflate: ms each", " ".join(f"{1e3 * t:.1f}" for t in reps),
      f"-> {len(text) / min(reps) / 1e9:.1f} GB/s of text; same bytes as from the plain file")
import shutil; shutil.rmtree(d)
