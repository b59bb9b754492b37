"""Multi-GPU execution: one process per GPU (torchrun), windows sharded by contiguous ranges.

The reference's only parallel executor is ``sai.multiprocessing.mp_pool`` (mp_pool.py:45-73): a
process pool over ChunkGenerator chunks whose results return to the parent.  Windows are
independent, so the MI355X build keeps exactly that decomposition: rank r computes the chunks
assigned to it on its own GPU, with no data-path collective, and the per-window results are
gathered once at the end to rank 0 (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU
for the tests).  Results are identical to a single-GPU run by construction: every chunk is
computed by the same code on the same bytes.
"""

from __future__ import annotations

import os
from typing import Any, Optional, Sequence

from .utils.windows import split_index_ranges


def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


_STATUS_GROUP = None  # host-side (gloo) group of the same ranks, for the failure report of run_sharded


def init_process_group(backend: Optional[str] = None, force: bool = False) -> tuple[int, int]:
    """Join the default group when launched under torchrun; returns (rank, world_size).  ``force``
    joins even as the only rank (the collectives then really run: tests of the RCCL calls on one GPU).

    With RCCL as the data backend a second, gloo group of the same ranks is made for the status
    exchange of ``run_sharded``: a rank that failed in a HIP / RCCL call must not report that through
    another collective on the communicator and device that just failed."""
    import datetime

    import torch
    import torch.distributed as dist

    global _STATUS_GROUP
    rank, local_rank, world = env_rank_world()
    # the host driver of the MI355X pool only supports dmabuf IPC (RCCL's set-up fails with `hipIpcGetMemHandle:
    # invalid argument` otherwise): the rank launcher sets this for its children, a rank started by any other
    # launcher gets it here -- read when HIP initialises, i.e. it must be in place before the first GPU call
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:  # SAI_AMD_DIST_BACKEND=gloo: several ranks on one GPU (tests, rehearsals)
            backend = os.environ.get("SAI_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        timeout = datetime.timedelta(minutes=float(os.environ.get("SAI_AMD_DIST_TIMEOUT_MIN", "30")))
        if backend == "nccl":
            n_dev = torch.cuda.device_count()
            if local_rank >= n_dev:
                raise RuntimeError(
                    f"rank {rank} (local rank {local_rank}) has no GPU of its own: {n_dev} device(s) visible; one worker "
                    "process per GPU (SAI_AMD_DIST_BACKEND=gloo lets several ranks share a device for rehearsals)"
                )
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout,
                                    device_id=torch.device("cuda", local_rank))  # fmt: skip
            _STATUS_GROUP = dist.new_group(backend="gloo", timeout=timeout)
        else:
            if torch.cuda.is_available():  # ranks sharing a box's devices round-robin (one GPU: all on it)
                torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world


def shutdown_process_group() -> None:
    import torch.distributed as dist

    global _STATUS_GROUP
    if dist.is_available() and dist.is_initialized():
        _STATUS_GROUP = None
        dist.destroy_process_group()


def my_chunk_indices(n_chunks: int, rank: int, world: int) -> range:
    """Contiguous block of chunk indices for this rank (same balancing rule as
    chunk_generator.py:130-142: the first ``n % world`` ranks get one more)."""
    bounds = split_index_ranges(n_chunks, world)
    if rank >= len(bounds):
        return range(0)
    return range(*bounds[rank])


def _exchange_task_results(data_processor, mine: list, group=None) -> Optional[list]:
    """Bring every rank's per-task results to rank 0, in rank (= task) order.

    A processor that offers ``pack_result`` / ``unpack_result`` (ChunkPreprocessor: the numeric
    ``WindowBatch`` -- fixed window records + CSR candidate lists) travels as ONE byte row per rank:
    sizes are exchanged once, then a single padded ``gather`` moves the rows (RCCL over xGMI when
    the backend is "nccl", the rows then live in HBM; gloo on the host).  Other processors -- user
    plugins with arbitrary Python results -- fall back to ``gather_object`` (pickle), which is what
    the reference's pool does with every result (mp_pool.py:67-73)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if not (hasattr(data_processor, "pack_result") and hasattr(data_processor, "unpack_result")):
        gathered: Optional[list] = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0, group=group)
        return None if rank != 0 else [res for per_rank in gathered for res in per_rank]
    blobs = [data_processor.pack_result(r) for r in mine]
    head = np.array([len(blobs), *map(len, blobs)], dtype=np.int64).tobytes()
    row = np.frombuffer(head + b"".join(blobs), dtype=np.uint8)
    on_gpu = dist.get_backend(group) == "nccl"
    local = torch.from_numpy(row.copy())
    if on_gpu:
        local = local.to(torch.device("cuda", torch.cuda.current_device()))
    per_rank = gather_window_records(local, group)
    if rank != 0:
        return None
    out = []
    for t in per_rank:
        raw = t.cpu().numpy().tobytes()
        n = int(np.frombuffer(raw, dtype=np.int64, count=1)[0])
        sizes = np.frombuffer(raw, dtype=np.int64, count=n, offset=8).tolist()
        o = 8 * (1 + n)
        for sz in sizes:
            out.append(data_processor.unpack_result(raw[o : o + sz]))
            o += sz
    return out


def _finish(data_processor, results: list) -> list:
    """Items of all tasks: a processor with the numeric protocol merges its batches into the
    single-chunk order (combination-major, sai.py:146-151), others are concatenated task by task."""
    if hasattr(data_processor, "items_from_results"):
        return data_processor.items_from_results(results)
    return [it for per_task in results for it in per_task]


class ShardFailure(RuntimeError):
    """Another rank of the sharded job failed; this rank stops instead of waiting for it."""


def _failed_ranks(i_failed: bool, group=None) -> list:
    """Ranks that report a failure (one small all_gather: every rank takes part, failed or not).  Always
    on host tensors: over the gloo side group when the data backend is RCCL (``init_process_group``)."""
    import torch
    import torch.distributed as dist

    status_group, dev = group, torch.device("cpu")
    if dist.get_backend(group) == "nccl":
        if group is None and _STATUS_GROUP is not None:
            status_group = _STATUS_GROUP
        else:  # a caller's own RCCL group without a host-side twin: the report has to travel on it
            dev = torch.device("cuda", torch.cuda.current_device())
    world = dist.get_world_size(status_group)
    mine = torch.tensor([1 if i_failed else 0], dtype=torch.int32, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=status_group)
    return [r for r, t in enumerate(every) if int(t.item())]


def run_sharded(data_processor, data_generator, group=None, as_items: bool = True) -> Optional[list]:
    """mp_pool's decomposition over ranks: every rank runs its share of ``data_generator.get()``
    (``run_compact`` when the processor has the numeric protocol, else ``run``), the per-task results
    come to rank 0 in generator order, and rank 0 calls ``process_items`` on the items -- written in
    the order a one-process run writes them, so the output files are byte-identical for any number
    of ranks.  Returns the items on rank 0.  ``as_items=False`` (``score_sharded``): a processor that
    can write its numeric results directly (``write_results``) does so and the per-task results are
    returned instead of items -- no Python object per window is ever built."""
    import torch.distributed as dist

    tasks = list(data_generator.get())
    compute = getattr(data_processor, "run_compact", None) or data_processor.run
    direct = not as_items and hasattr(data_processor, "write_results") and hasattr(data_processor, "run_compact")
    if not (dist.is_available() and dist.is_initialized()):
        results = [compute(**params) for params in tasks]
        if direct:
            data_processor.write_results(results)
            return results
        items = _finish(data_processor, results)
        data_processor.process_items(items)
        return items
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if hasattr(data_processor, "run_compact") and not hasattr(data_processor, "pack_result"):
        raise TypeError("a processor with run_compact must also offer pack_result / unpack_result")
    # A rank that fails in its own chunks (a malformed VCF line, a CRC, a full candidate buffer) must not
    # leave the others waiting in the gather until the collective times out: every rank reports one
    # status word first, and if any failed nobody enters the data exchange -- the failing rank raises
    # its own error, the others say which ranks failed.
    error, mine = None, []
    try:
        mine = [compute(**tasks[i]) for i in my_chunk_indices(len(tasks), rank, world)]
    except BaseException as exc:  # noqa: BLE001 - any way out (KeyboardInterrupt, SystemExit too): re-raised below, after the status exchange
        error = exc
    failed = _failed_ranks(error is not None, group)
    if failed:
        if error is not None:
            raise error
        raise ShardFailure(f"rank(s) {failed} failed while computing their chunks (their own messages say why)")
    results = _exchange_task_results(data_processor, mine, group)
    if rank != 0:
        return None
    if direct:
        data_processor.write_results(results)
        return results
    items = _finish(data_processor, results)
    data_processor.process_items(items)
    return items


def gather_window_records(local, group=None):
    """Gather fixed-size window records (any 1-D tensor; lengths may differ per rank) to rank 0
    with one size exchange and one padded gather.  Returns the list of per-rank tensors on rank 0
    (in rank order), None elsewhere.  Works on GPU tensors with RCCL and CPU tensors with gloo."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [local]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n_local = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    return gather_padded(local, sizes, group)


def gather_padded(local, sizes: Sequence[int], group=None):
    """The data half of ``gather_window_records`` when every rank already knows all sizes
    (they are fixed for a resident scorer, so the size exchange is paid once, not per step)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()  # gloo gathers host tensors (CPU tests, single-GPU rehearsals)
    cap = max(sizes)
    padded = local
    if local.numel() != cap:
        padded = torch.zeros((cap,), dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local
    out = [torch.empty((cap,), dtype=local.dtype, device=local.device) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, out, dst=0, group=group)
    if rank != 0:
        return None
    return [t[:n] for t, n in zip(out, sizes)]


class RowGather:
    """The per-pass exchange of a sharded resident job: every rank owns a
    ``RowLayout`` (records + CSR candidate lists of its windows; fixed for a resident block), the
    layouts are exchanged ONCE at set-up with one small all_gather, and each pass then costs one
    padded ``gather`` of the byte rows to rank 0 -- RCCL over xGMI for HBM rows (backend "nccl"),
    gloo for host rows.  ``decode`` turns what rank 0 received into per-rank WindowResults."""

    HEADER_LEN = 64  # int64 words: enough for 20 set chunks (320 parameter sets)

    def __init__(self, layout, device, group=None):
        import torch
        import torch.distributed as dist

        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        head = ([1] + layout.header()) if layout is not None else [0]
        if len(head) > self.HEADER_LEN:
            raise ValueError("too many set chunks for the gather header")
        mine = torch.zeros((self.HEADER_LEN,), dtype=torch.int64)
        mine[: len(head)] = torch.tensor(head, dtype=torch.int64)
        if self.on:
            mine = mine.to(device)
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(every, mine, group=group)
            heads = [h.cpu().tolist() for h in every]
        else:
            heads = [mine.tolist()]
        from .resident import RowLayout

        self.layouts = [RowLayout.from_header(h[1:]) if h[0] else None for h in heads]
        self.sizes = [lay.nbytes if lay is not None else 0 for lay in self.layouts]

    def gather(self, row):
        """One pass: ``row`` = this rank's uint8 row (``sizes[rank]`` bytes; any tensor for a rank
        without windows).  Returns the rank-ordered list of rows on rank 0, None elsewhere."""
        if not self.on:
            return [row[: self.sizes[0]]]
        return gather_padded(row[: self.sizes[self.rank]], self.sizes, self.group)

    def decode(self, rows) -> list:
        """Rank 0: WindowResults per rank (None for a rank without windows)."""
        out = []
        for lay, t in zip(self.layouts, rows):
            out.append(None if lay is None else lay.unpack(t.cpu().numpy()))
        return out


def score_sharded(vcf_file: str, chr_name: str, win_len: int, win_step: int, anc_allele_file, output_file: str,
                  config: str, chunks_per_rank: int = 1) -> Optional[list]:  # fmt: skip
    """``score`` over all ranks of the job: the chromosome's window list is cut into
    ``world * chunks_per_rank`` ChunkGenerator chunks, each rank loads and scores only its own
    regions on its GPU, rank 0 writes the reference's TSV / log files (natively, from the gathered
    numeric batches) and gets the batches back; the other ranks get None."""
    import torch.distributed as dist

    from .generators import ChunkGenerator
    from .sai import chunk_preprocessor_for, load_config, require_polarised_input, write_headers

    rank, world = init_process_group()
    cfg = load_config(config)
    require_polarised_input(cfg.statistics, anc_allele_file)
    generator = ChunkGenerator(vcf_file=vcf_file, chr_name=chr_name, window_size=win_len, step_size=win_step,
                               num_chunks=max(world * chunks_per_rank, 1))  # fmt: skip
    preprocessor = chunk_preprocessor_for(cfg, vcf_file, win_len, win_step, output_file, anc_allele_file)
    if rank == 0:
        write_headers(output_file, cfg.statistics, cfg.ploidies)
    if world > 1:
        dist.barrier()
    try:
        return run_sharded(preprocessor, generator, as_items=False)
    except BaseException:
        if rank == 0:  # a header-only TSV and empty logs would look like a finished run without windows
            from pathlib import Path

            out = Path(output_file)
            for f in (out, out.with_suffix(".U.log"), out.with_suffix(".Q.log")):
                f.unlink(missing_ok=True)
        raise
