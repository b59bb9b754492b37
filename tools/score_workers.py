"""`sai score --num-workers N` at size on ONE GPU (every rank on device 0, gloo for the gather): the product's
multi-worker route end to end -- launcher, per-rank region reads of the VCF, sharded scoring, gather, rank 0
writes -- against the one-process run of the same command: identical files, and what the route costs in wall
time when it cannot win anything (one device).  python tools/score_workers.py [n_sites] [workers ...]"""
import hashlib, os, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
workers = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
n_ref, n_tgt, n_src = 1000, 1000, 2
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
names = [f"i{k}" for k in range(n_ref + n_tgt + n_src)]
vcf = os.path.join(d, "synth.vcf")
t0 = time.perf_counter()
pairs = np.array(["0|0", "0|1", "1|0", "1|1"])
with open(vcf, "w") as f:
    f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
    pos = 0
    for s in range(n_sites):
        pos += int(rng.integers(1, 50))
        intro = rng.random() < 0.01
        p = rng.random() ** 4
        hap = np.concatenate([rng.binomial(1, 0.0 if intro else p, 2 * n_ref), rng.binomial(1, 0.6 if intro else p, 2 * n_tgt),
                              rng.binomial(1, 1.0 if intro else p, 2 * n_src)]).reshape(-1, 2)
        f.write(f"1\t{pos}\t.\tA\tT\t100\tPASS\t.\tGT\t" + "\t".join(pairs[hap[:, 0] * 2 + hap[:, 1]]) + "\n")
for grp, sl in (("ref", names[:n_ref]), ("tgt", names[n_ref:n_ref + n_tgt]), ("src", names[n_ref + n_tgt:])):
    with open(os.path.join(d, f"{grp}.list"), "w") as f:
        f.write("".join(f"{grp.upper()}\t{n}\n" for n in sl))
cfg = os.path.join(d, "cfg.yaml")
with open(cfg, "w") as f:
    f.write(f"""statistics:
  U: {{ref: {{REF: 0.01}}, tgt: {{TGT: 0.5}}, src: {{SRC: "=1"}}}}
  Q: {{ref: {{REF: 0.01}}, tgt: {{TGT: 0.95}}, src: {{SRC: "=1"}}}}
ploidies: {{ref: {{REF: 2}}, tgt: {{TGT: 2}}, src: {{SRC: 2}}}}
populations: {{ref: {d}/ref.list, tgt: {d}/tgt.list, src: {d}/src.list}}
""")
print(f"VCF {os.path.getsize(vcf) / 1e6:.0f} MB, {n_sites} sites x {len(names)} samples, written in {time.perf_counter() - t0:.1f} s", flush=True)
env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
env.update(SAI_AMD_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
digests = {}
for n in workers:
    out = os.path.join(d, f"w{n}", "scores.tsv")
    cmd = [sys.executable, "-m", "sai_amd", "score", "--vcf", vcf, "--chr-name", "1", "--win-len", "50000", "--win-step", "25000",
           "--config", cfg, "--output", out, "--num-workers", str(n)]
    t0 = time.perf_counter()
    res = subprocess.run(cmd, env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if res.returncode != 0:
        print(res.stderr[-2000:])
        sys.exit(f"--num-workers {n} failed with {res.returncode}")
    files = sorted(os.listdir(os.path.dirname(out)))
    digests[n] = {f: hashlib.sha256(open(os.path.join(os.path.dirname(out), f), "rb").read()).hexdigest()[:16] for f in files}
    rows = sum(1 for _ in open(out)) - 1
    print(f"--num-workers {n}: {dt:.2f} s wall (interpreter + torch start-up, build check, rendezvous included), {rows} windows, files {digests[n]}", flush=True)
same = all(digests[n] == digests[workers[0]] for n in workers)
print("identical files for every worker count:", same)
sys.exit(0 if same else 1)
