"""`score(..., num_workers=N)` / `sai score --num-workers N` (sai.py:33-42, 86-93): the
caller's process starts the N ranks as one child job and returns its exit code.  No GPU here, so the
launcher's mechanics are checked with a stand-in rank script, the no-GPU failure of the real ranks must
be loud, and the sharded route itself runs over gloo with a stand-in for the chunk processor.  The GPU
counterpart (byte-identical files from two ranks on one GPU) is tests/test_hip_sharded.py."""

import json
import os
import subprocess
import sys
import textwrap

import pytest
from conftest import ROOT

RANK_SCRIPT = textwrap.dedent(
    """
    import json, os, sys
    import torch, torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([dist.get_rank() + 1.0])
    dist.all_reduce(t)
    if os.environ.get("FAIL_RANK") == os.environ["RANK"]:
        sys.exit(7)
    if dist.get_rank() == 0:
        import sai_amd
        print(json.dumps({"sum": t.item(), "world": dist.get_world_size(), "argv": sys.argv[1:], "cwd": os.getcwd(),
                          "master": os.environ["MASTER_ADDR"], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
    dist.destroy_process_group()
    """
)


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SAI_AMD_GPUS")}
    env.update(extra)
    return env


def _launch(tmp_path, n, **extra_env):
    script = tmp_path / "ranks.py"
    script.write_text(RANK_SCRIPT)
    code = (
        "import sys; sys.path.insert(0, %r); from sai_amd.launcher import launch_ranks; "
        "sys.exit(launch_ranks(%d, ['--flag', 'rel/path.tsv'], script=%r, who='t'))" % (str(ROOT), n, str(script))
    )
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=_clean_env(**extra_env),
                          cwd=str(tmp_path))  # fmt: skip


def test_launch_ranks_runs_one_child_job_in_the_callers_directory(tmp_path):
    """Relative paths of the command line must mean in the ranks what they mean in the caller: the job runs in
    the caller's working directory and finds the package through PYTHONPATH."""
    res = _launch(tmp_path, 2)
    assert res.returncode == 0, res.stderr[-2000:]
    (line,) = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    rec = json.loads(line)
    assert rec == {"sum": 3.0, "world": 2, "argv": ["--flag", "rel/path.tsv"], "cwd": str(tmp_path), "master": "127.0.0.1",
                   "ipc": "0"}  # fmt: skip


def test_launch_ranks_returns_the_jobs_exit_code(tmp_path):
    res = _launch(tmp_path, 2, FAIL_RANK="1")
    assert res.returncode != 0 and "t: the 2-rank child job exited with" in res.stderr


def test_launch_ranks_refuses_inside_a_rank_and_bad_counts(monkeypatch):
    from sai_amd import launcher

    with pytest.raises(ValueError):
        launcher.launch_ranks(0, [], module="sai_amd", build=False)
    with pytest.raises(ValueError):
        launcher.rank_command(2, [], module="a", script="b")
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert launcher.in_rank_job()
    with pytest.raises(RuntimeError, match="inside a rank"):
        launcher.launch_ranks(2, [], module="sai_amd", build=False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert not launcher.in_rank_job()
    cmd = launcher.rank_command(3, ["score", "--vcf", "x"], module="sai_amd", port=1234)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                       "--master-port", "1234", "-m", "sai_amd", "score", "--vcf", "x"]  # fmt: skip
    alone = launcher.rank_command(2, ["x"], script="s.py")  # no port: torchrun's own store picks a free one
    assert alone[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--standalone", "--local-addr", "127.0.0.1", "s.py", "x"]
    monkeypatch.setenv("SAI_AMD_GPUS", "4")
    assert launcher.workers_from_env() == 4
    monkeypatch.setenv("SAI_AMD_GPUS", "0")
    with pytest.raises(ValueError):
        launcher.workers_from_env()
    monkeypatch.delenv("SAI_AMD_GPUS")
    assert launcher.workers_from_env() == 1 and launcher.chunks_per_worker() == 8  # sai.py:91


def test_score_starts_the_ranks_with_its_own_arguments(monkeypatch, tmp_path):
    """score(num_workers=3): configuration errors are the caller's own exceptions; otherwise ONE launch of
    `python -m sai_amd score <the same arguments>`; a failed job raises with its exit code."""
    from sai_amd import launcher
    from sai_amd.sai import score

    calls = []

    def fake(n, argv, module=None, script=None, stdout=None, build=True, who=""):
        calls.append((n, list(argv), module))
        return fake.rc

    fake.rc = 0
    monkeypatch.setattr(launcher, "launch_ranks", fake)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    kw = dict(vcf_file="tests/data/test.data.vcf", chr_name=21, win_len=10000, win_step=5000, anc_allele_file=None,
              output_file=str(tmp_path / "o.tsv"), config="tests/data/test.uq.config.yaml")  # fmt: skip
    os.chdir(ROOT)
    with pytest.raises(FileNotFoundError, match="Configuration file 'nope.yaml' not found."):
        score(**{**kw, "config": "nope.yaml"}, num_workers=3)
    with pytest.raises(ValueError, match="num_workers"):
        score(**kw, num_workers=0)
    assert calls == []
    score(**kw, num_workers=3)
    assert calls == [(3, ["score", "--vcf", "tests/data/test.data.vcf", "--chr-name", "21", "--win-len", "10000", "--win-step", "5000",
                          "--output", str(tmp_path / "o.tsv"), "--config", "tests/data/test.uq.config.yaml", "--num-workers", "3"],
                      "sai_amd")]  # fmt: skip
    score(**{**kw, "anc_allele_file": "anc.bed"}, num_workers=2)
    assert calls[-1][1][-2:] == ["--anc-alleles", "anc.bed"]
    fake.rc = 9
    with pytest.raises(launcher.RankJobFailed) as err:
        score(**kw, num_workers=2)
    assert err.value.returncode == 9


def test_cli_has_the_flag_and_the_reference_default(monkeypatch):
    from sai_amd.__main__ import _sai_cli_parser

    base = ["score", "--vcf", str(ROOT / "tests/data/test.data.vcf"), "--chr-name", "21", "--output", "o.tsv", "--config",
            str(ROOT / "tests/data/test.uq.config.yaml")]  # fmt: skip
    from sai_amd.parsers.score_parser import resolve_workers

    monkeypatch.delenv("SAI_AMD_GPUS", raising=False)
    assert resolve_workers(_sai_cli_parser().parse_args(base)) == 1  # score_parser.py:64
    assert resolve_workers(_sai_cli_parser().parse_args([*base, "--num-workers", "8"])) == 8
    monkeypatch.setenv("SAI_AMD_GPUS", "2")
    assert resolve_workers(_sai_cli_parser().parse_args(base)) == 2
    with pytest.raises(SystemExit):
        _sai_cli_parser().parse_args([*base, "--num-workers", "0"])
    # a malformed variable is `score`'s usage error when it runs; the parser, --help and `outlier` never see it
    monkeypatch.setenv("SAI_AMD_GPUS", "many")
    args = _sai_cli_parser().parse_args(base)
    with pytest.raises(SystemExit) as exc:
        resolve_workers(args)
    assert exc.value.code == 2
    assert _sai_cli_parser().parse_args(["outlier", "--score", str(ROOT / "tests/data/test.data.vcf"), "--output-prefix", "x",
                                         "--quantile", "0.9"]).runner is not None  # fmt: skip


def test_plain_cli_with_two_workers_and_no_gpu_fails_loudly(tmp_path):
    """`python -m sai_amd score ... --num-workers 2` in a container without a GPU: the ranks die with the
    library's own message (no CPU fallback), the caller reports the job's failure with a non-zero exit
    code, and no header-only result files are left behind."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: this is then simply a run (tests/test_hip_sharded.py)")
    out = tmp_path / "res" / "scores.tsv"
    res = subprocess.run(
        [sys.executable, "-m", "sai_amd", "score", "--vcf", "tests/data/test.data.vcf", "--chr-name", "21", "--win-len", "10000",
         "--win-step", "5000", "--output", str(out), "--config", "tests/data/test.uq.config.yaml", "--num-workers", "2"],
        cwd=str(ROOT), capture_output=True, text=True, timeout=600, env=_clean_env(SAI_AMD_DIST_TIMEOUT_MIN="2"))  # fmt: skip
    assert res.returncode != 0
    assert "sai score: the 2-rank child job exited with" in res.stderr
    assert "RankJobFailed" in res.stderr or "2-rank job exited" in res.stderr
    assert not out.exists() and not out.with_suffix(".U.log").exists() and not out.with_suffix(".Q.log").exists()


# ---- the sharded route of `score` over gloo, with a stand-in for the GPU chunk processor -----------


def _rank_main(rank, world, port, out_file, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      SAI_AMD_DIST_BACKEND="gloo", SAI_AMD_CHUNKS_PER_WORKER="2")  # fmt: skip
    os.chdir(ROOT)
    import torch.distributed as dist

    import sai_amd.sai as S
    from sai_amd.distributed import ShardFailure

    seen = []

    class Pre:
        def run_compact(self, chr_name, start, end):
            if mode == "raise" and rank == 1:
                raise ValueError("malformed record in this rank's region")
            if mode == "interrupt" and rank == 1:
                raise KeyboardInterrupt()
            seen.append((start, end))
            return f"{chr_name}:{start}-{end}".encode()

        pack_result = staticmethod(lambda r: r)
        unpack_result = staticmethod(lambda raw: bytes(raw))

        def write_results(self, results):
            with open(out_file, "a") as f:
                f.writelines(r.decode() + "\n" for r in results)

    S_chunk = S.chunk_preprocessor_for
    S.chunk_preprocessor_for = lambda *a, **k: Pre()
    try:
        expect = {"ok": None, "raise": ValueError if rank == 1 else ShardFailure,
                  "interrupt": KeyboardInterrupt if rank == 1 else ShardFailure}[mode]  # fmt: skip
        try:
            S.score(vcf_file="tests/data/test.data.vcf", chr_name="21", win_len=10000, win_step=5000, anc_allele_file=None,
                    output_file=out_file, config="tests/data/test.uq.config.yaml", num_workers=world)  # fmt: skip
            raised = None
        except BaseException as exc:  # noqa: BLE001
            raised = type(exc)
        assert raised is expect, (rank, raised, expect)
        assert not dist.is_initialized()  # score leaves the group on every way out
        with open(out_file + f".rank{rank}", "w") as f:
            json.dump(seen, f)
    finally:
        S.chunk_preprocessor_for = S_chunk


def _spawn(world, out_file, mode):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(world, port, out_file, mode), nprocs=world, join=True)


def test_score_inside_a_job_is_a_rank_of_the_sharded_route(tmp_path):
    """WORLD_SIZE=2: `score` cuts the window list into world x SAI_AMD_CHUNKS_PER_WORKER chunks, every rank
    computes its contiguous share, rank 0 writes header + all chunks in order."""
    out = str(tmp_path / "o.tsv")
    _spawn(2, out, "ok")
    from sai_amd.generators import ChunkGenerator

    os.chdir(ROOT)
    chunks = list(ChunkGenerator(vcf_file="tests/data/test.data.vcf", chr_name="21", window_size=10000, step_size=5000, num_chunks=4).get())
    lines = open(out).read().splitlines()
    assert lines[0].startswith("Chrom\tStart\tEnd") and lines[1:] == [f"21:{c['start']}-{c['end']}" for c in chunks]
    seen = [json.load(open(out + f".rank{r}")) for r in range(2)]
    assert [tuple(x) for x in seen[0] + seen[1]] == [(c["start"], c["end"]) for c in chunks] and seen[0] and seen[1]


@pytest.mark.parametrize("mode", ["raise", "interrupt"])
def test_a_failing_rank_ends_the_job_and_removes_the_files(tmp_path, mode):
    """Rank 1 leaves its chunks through an Exception or a BaseException: rank 0 gets ShardFailure
    instead of waiting in the gather, and the header-only TSV / log files are removed."""
    out = tmp_path / "o.tsv"
    _spawn(2, str(out), mode)
    assert not out.exists() and not out.with_suffix(".U.log").exists() and not out.with_suffix(".Q.log").exists()
