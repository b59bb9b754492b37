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


def init_process_group(backend: Optional[str] = None) -> tuple[int, int]:
    """Join the default group when launched under torchrun; returns (rank, world_size)."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:  # SAI_AMD_DIST_BACKEND=gloo: several ranks on one GPU (tests, rehearsals)
            backend = os.environ.get("SAI_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def my_chunk_indices(n_chunks: int, rank: int, world: int) -> range:
    """Contiguous block of chunk indices for this rank (same balancing rule as
    chunk_generator.py:130-142: the first ``n % world`` ranks get one more)."""
    bounds = split_index_ranges(n_chunks, world)
    if rank >= len(bounds):
        return range(0)
    return range(*bounds[rank])


def run_sharded(data_processor, data_generator, group=None) -> Optional[list]:
    """mp_pool's contract over ranks: every rank runs ``data_processor.run(**params)`` for its
    share of ``data_generator.get()``, the item lists are gathered to rank 0 in generator order,
    and rank 0 calls ``process_items`` on the concatenation.  Returns the items on rank 0."""
    import torch.distributed as dist

    tasks = list(data_generator.get())
    if not (dist.is_available() and dist.is_initialized()):
        items = [it for params in tasks for it in data_processor.run(**params)]
        data_processor.process_items(items)
        return items
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = [data_processor.run(**tasks[i]) for i in my_chunk_indices(len(tasks), rank, world)]
    gathered: Optional[list] = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0, group=group)
    if rank != 0:
        return None
    items = [it for per_rank in gathered for per_task in per_rank for it in per_task]
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


def score_sharded(vcf_file: str, chr_name: str, win_len: int, win_step: int, anc_allele_file, output_file: str,
                  config: str, chunks_per_rank: int = 1) -> Optional[list[dict[str, Any]]]:  # fmt: skip
    """``score`` over all ranks of the job: the chromosome's window list is cut into
    ``world * chunks_per_rank`` ChunkGenerator chunks, each rank loads and scores only its own
    regions on its GPU, rank 0 writes the reference's TSV / log files."""
    import torch.distributed as dist

    from .generators import ChunkGenerator
    from .preprocessors import ChunkPreprocessor
    from .sai import load_config, write_headers

    rank, world = init_process_group()
    cfg = load_config(config)
    generator = ChunkGenerator(vcf_file=vcf_file, chr_name=chr_name, window_size=win_len, step_size=win_step,
                               num_chunks=max(world * chunks_per_rank, 1))  # fmt: skip
    preprocessor = ChunkPreprocessor(
        vcf_file=vcf_file,
        ref_ind_file=cfg.populations.get_population("ref"),
        tgt_ind_file=cfg.populations.get_population("tgt"),
        src_ind_file=cfg.populations.get_population("src"),
        out_ind_file=cfg.populations.get_population("outgroup"),
        win_len=win_len,
        win_step=win_step,
        output_file=output_file,
        ploidy_config=cfg.ploidies,
        stat_config=cfg.statistics,
        anc_allele_file=anc_allele_file,
    )
    if rank == 0:
        write_headers(output_file, cfg.statistics, cfg.ploidies)
    if world > 1:
        dist.barrier()
    return run_sharded(preprocessor, generator)
