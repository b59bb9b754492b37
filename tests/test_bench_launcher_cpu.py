"""`python bench.py --gpus N` without a launcher around it: the parent builds once,
starts the ranks as one child job (torch.distributed.run, rendezvous on 127.0.0.1), hands the child
its own stdout and returns the child's exit code.  Checked here on CPU with a stand-in script for
the ranks (a gloo group that reduces over the ranks and prints one JSON line from rank 0), and with
bench.py itself, whose ranks cannot run without a GPU: the failure must come back as a non-zero
exit code, not as a hang or a zero.  The GPU counterpart is tests/test_hip_sharded.py."""

import json
import os
import subprocess
import sys
import textwrap

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
        print(json.dumps({"sum": t.item(), "world": dist.get_world_size(), "argv": sys.argv[1:],
                          "master": os.environ["MASTER_ADDR"]}), flush=True)
    dist.destroy_process_group()
    """
)


def _launch(tmp_path, n, extra_env=None):
    script = tmp_path / "ranks.py"
    script.write_text(RANK_SCRIPT)
    code = (
        "import os, sys; sys.path.insert(0, %r); import bench; "
        "out = os.fdopen(os.dup(1), 'w'); os.dup2(2, 1); "
        "sys.exit(bench.self_launch(%d, ['--gpus', '%d', '--steps', '2'], out, script=%r))" % (str(ROOT), n, n, str(script))
    )
    env = {**os.environ, **(extra_env or {})}
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_self_launch_starts_the_ranks_and_relays_one_line(tmp_path):
    res = _launch(tmp_path, 3)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec == {"sum": 6.0, "world": 3, "argv": ["--gpus", "3", "--steps", "2"], "master": "127.0.0.1"}


def test_self_launch_returns_the_childs_exit_code(tmp_path):
    res = _launch(tmp_path, 2, {"FAIL_RANK": "1"})
    assert res.returncode != 0  # rank 1 left with 7: torchrun fails the job, the launcher passes that on
    assert "child job exited with" in res.stderr


def test_ranks_of_a_foreign_launcher_get_the_ipc_mode(tmp_path):
    """Ranks started by plain `python -m torch.distributed.run` -- not by sai_amd.launcher, whose children inherit
    HSA_ENABLE_IPC_MODE_LEGACY=0 from it -- in an environment WITHOUT the variable: joining the job through
    sai_amd.distributed.init_process_group puts it in place before the first GPU call (RCCL's IPC needs it on this
    pool)."""
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(
        f"""
        import os, sys
        sys.path.insert(0, {str(ROOT)!r})
        assert "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ
        from sai_amd import distributed as D
        rank, world = D.init_process_group("gloo")
        import torch.distributed as dist
        assert world == 2 and dist.get_world_size() == 2
        open({str(tmp_path)!r} + f"/rank{{rank}}.txt", "w").write(str(os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))  # the ranks share stdout
        D.shutdown_process_group()
        """
    ))
    env = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29631", str(script)], capture_output=True, text=True, timeout=300, env=env)  # fmt: skip
    assert res.returncode == 0, res.stderr[-2000:]
    assert [(tmp_path / f"rank{r}.txt").read_text() for r in (0, 1)] == ["0", "0"]


def test_plain_bench_gpus_2_without_a_gpu_fails_loudly():
    """No GPU here: the ranks die in torch.cuda.set_device / Engine(); `python bench.py --gpus 2` must
    come back non-zero with no JSON line (it used to exit with a usage message before starting anything)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--sites", "5000",
                          "--chroms", "2", "--cpu-sites", "0"], cwd=str(ROOT), capture_output=True, text=True, timeout=600, env=env)  # fmt: skip
    import torch

    if torch.cuda.is_available():  # on a GPU box this is simply a run (RCCL needs one device per rank: may fail too)
        return
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert "launch with torch.distributed.run" not in res.stderr


def test_live_traffic_reads_the_counters_of_two_child_runs(tmp_path, monkeypatch):
    """bench.measure_traffic with a stand-in for rocprofv3 on PATH: one child run per counter (the program
    behind `--`, --kernel-trace as the only other option, the parent's step / baseline flags replaced),
    FETCH_SIZE doubled as the guide prescribes for gfx950, WRITE_SIZE as it is, both KiB; a failing
    profiler is reported, not raised."""
    sys.path.insert(0, str(ROOT))
    import bench

    fake = tmp_path / "bin" / "rocprofv3"
    fake.parent.mkdir()
    log = tmp_path / "calls.txt"
    fake.write_text(textwrap.dedent(
        f"""\
        #!{sys.executable}
        import os, sys
        a = sys.argv[1:]
        open({str(log)!r}, "a").write(" ".join(a) + "\\n")
        if os.environ.get("FAKE_PROF_FAIL"):
            sys.exit(3)
        if os.environ.get("FAKE_PROF_HANG"):  # a profiler that never comes back, with the program it started
            import subprocess, time
            kid = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
            open(os.environ["FAKE_PROF_HANG"], "w").write(f"{{os.getpid()}} {{kid.pid}}")
            time.sleep(600)
        counter, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
        os.makedirs(os.path.join(out, "host"), exist_ok=True)
        rows = ["Kernel_Name,Counter_Name,Counter_Value"]
        for v in ((1000.0, 1004.0) if counter == "FETCH_SIZE" else (10.0, 14.0)):
            rows.append(f'"void (anonymous namespace)::site_counts_kernel<false, true>(A, B)",{{counter}},{{v}}')
        rows.append(f'"(anonymous namespace)::window_lists_kernel(W)",{{counter}},999999')
        open(os.path.join(out, "host", "1_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")
        """
    ))
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", f"{fake.parent}:{os.environ['PATH']}")
    total, how = bench.measure_traffic(["--workload", "c2", "--steps", "50", "--warmup=7", "--cpu-sites", "1e5", "--anc", "false"], "site_counts")
    assert total == int(1002.0 * 1024 * 2 + 12.0 * 1024) and "measured in this run" in how
    calls = log.read_text().splitlines()
    assert len(calls) == 2 and "--pmc FETCH_SIZE --kernel-trace" in calls[0] and "--pmc WRITE_SIZE --kernel-trace" in calls[1]
    child = calls[0].split(" -- ", 1)[1]
    assert child.endswith("--workload c2 --anc false --steps 3 --warmup 1 --cpu-sites 0 --score-path off --traffic off")
    assert "bench.py" in child and "--steps 50" not in child and "--warmup=7" not in child
    monkeypatch.setenv("FAKE_PROF_FAIL", "1")
    total, why = bench.measure_traffic(["--workload", "c2"], "site_counts")
    assert total is None and "failed" in why
    # a run that overstays is ended together with the program it profiles (its own process group)
    monkeypatch.delenv("FAKE_PROF_FAIL")
    pids = tmp_path / "pids.txt"
    monkeypatch.setenv("FAKE_PROF_HANG", str(pids))
    total, why = bench.measure_traffic(["--workload", "c2"], "site_counts", timeout_s=3.0)
    assert total is None and "no result within" in why
    import time

    time.sleep(0.5)
    for pid in map(int, pids.read_text().split()):
        try:
            stat = open(f"/proc/{pid}/stat").read().split()
            assert stat[2] == "Z", (pid, stat[2])  # at most a zombie waiting for init
        except FileNotFoundError:
            pass
