"""Starting the ranks of a multi-GPU job from an ordinary process.

The reference's ``score(..., num_workers)`` owns its workers: the parent holds the task list and a
process pool computes the chunks (sai.py:33-42, 86-93; mp_pool.py:45-73).  The MI355X build keeps that
shape with one process per GPU: the parent builds the library once and starts ONE child job
(``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...``, rendezvous on 127.0.0.1)
whose ranks take the sharded route (``sai_amd.distributed.score_sharded``); it waits for the job and
hands back its exit code.  Nothing in this module touches the GPU -- no ``torch.cuda`` call, no
library context -- and the child is started with ``subprocess``, never exec'd over the caller.
``bench.py --gpus N`` starts its ranks through the same function.
"""

from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path
from typing import Optional, Sequence

PACKAGE_ROOT = Path(__file__).resolve().parent.parent
CHUNKS_PER_WORKER = 8  # the reference's grain: `num_chunks = num_workers * 8` (sai.py:91)


class RankJobFailed(RuntimeError):
    """The child job of ``launch_ranks`` left with a non-zero exit code."""

    def __init__(self, n_ranks: int, returncode: int):
        super().__init__(f"the {n_ranks}-rank job exited with {returncode} (the ranks' own messages say why)")
        self.returncode = returncode


def in_rank_job() -> bool:
    """True in a process that is one of several ranks of a launched job (torchrun's environment)."""
    try:
        return int(os.environ.get("WORLD_SIZE", "1")) > 1
    except ValueError:
        return False


def workers_from_env(default: int = 1) -> int:
    """``SAI_AMD_GPUS``: the number of GPUs (= worker processes) ``sai score`` uses when
    ``--num-workers`` is not given; 1 = the reference's CLI (score_parser.py:64, one process)."""
    raw = os.environ.get("SAI_AMD_GPUS", "")
    if not raw:
        return default
    n = int(raw)
    if n < 1:
        raise ValueError("SAI_AMD_GPUS must be a positive integer")
    return n


def chunks_per_worker() -> int:
    raw = os.environ.get("SAI_AMD_CHUNKS_PER_WORKER", "")
    n = int(raw) if raw else CHUNKS_PER_WORKER
    if n < 1:
        raise ValueError("SAI_AMD_CHUNKS_PER_WORKER must be a positive integer")
    return n


def rank_command(n_ranks: int, argv: Sequence[str], module: Optional[str] = None, script: Optional[str] = None,
                 port: Optional[int] = None) -> list:  # fmt: skip
    """The child job's command line: ``-m module`` or a script path, followed by ``argv``.  Without a ``port``
    the job is a stand-alone one on 127.0.0.1: torchrun's own store binds a free port itself, so two launches at
    the same moment cannot be handed the same port (probing for a free one and closing the probe could)."""
    if (module is None) == (script is None):
        raise ValueError("exactly one of module / script names what the ranks run")
    target = ["-m", module] if module is not None else [str(script)]
    where = ["--master-addr", "127.0.0.1", "--master-port", str(port)] if port else ["--standalone", "--local-addr", "127.0.0.1"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}", *where, *target,
            *map(str, argv)]  # fmt: skip


def launch_ranks(n_ranks: int, argv: Sequence[str], module: Optional[str] = None, script: Optional[str] = None,
                 stdout=None, build: bool = True, who: str = "sai_amd") -> int:  # fmt: skip
    """Build the library once (before N ranks would all find the tree stale), run the N ranks as one
    child process tree and return its exit code.  ``stdout`` = a file object the child's stdout goes
    to (default: this process's)."""
    if n_ranks < 1:
        raise ValueError("the number of ranks must be positive")
    if in_rank_job():
        raise RuntimeError("launch_ranks called from inside a rank of a running job")
    if build:
        from ._build import build as build_library

        build_library()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([str(PACKAGE_ROOT), *filter(None, [env.get("PYTHONPATH")])])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = rank_command(n_ranks, argv, module=module, script=script)
    kw = {}
    if stdout is not None:
        stdout.flush()
        kw["stdout"] = stdout.fileno()
    child = subprocess.run(cmd, env=env, **kw)
    if child.returncode != 0:
        print(f"{who}: the {n_ranks}-rank child job exited with {child.returncode}", file=sys.stderr)
    return child.returncode
