"""``mp_pool`` of the reference (sai/multiprocessing/mp_pool.py:25-73) over GPUs instead of host
processes: one process drives one GPU, so the tasks of ``data_generator.get()`` are not handed to a
``multiprocessing.Pool`` but run in this process, or -- under ``torchrun`` -- in contiguous blocks
on the ranks of the job, whose results come back to rank 0 in task order."""

from __future__ import annotations

from typing import Any

from ..distributed import my_chunk_indices
from ..generators import DataGenerator
from ..preprocessors import DataPreprocessor


def mp_worker(params: tuple[DataPreprocessor, dict]) -> Any:
    """``data_processor.run(**param_dict)`` (mp_pool.py:27-42)."""
    data_processor, param_dict = params
    return data_processor.run(**param_dict)


def mp_pool(data_processor: DataPreprocessor, data_generator: DataGenerator, nprocess: int = 1) -> None:
    """Same contract as the reference: one task per ``data_generator.get()`` entry, results in
    task order, ``data_processor.process_items(results)`` once at the end (on rank 0 of a
    multi-rank job).  ``nprocess`` is accepted for signature compatibility; the degree of
    parallelism is the number of ranks (= GPUs) the job was launched with.  Results travel to rank
    0 as one byte row per rank when the processor has the numeric protocol (ChunkPreprocessor:
    window records + CSR candidate lists through one RCCL gather), pickled otherwise."""
    import torch.distributed as dist

    from ..distributed import _exchange_task_results

    tasks = [(data_processor, params) for params in data_generator.get()]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        data_processor.process_items([mp_worker(t) for t in tasks])
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    compact = hasattr(data_processor, "run_compact") and hasattr(data_processor, "pack_result")
    mine = []
    for i in my_chunk_indices(len(tasks), rank, world):
        mine.append(data_processor.run_compact(**tasks[i][1]) if compact else mp_worker(tasks[i]))
    results = _exchange_task_results(data_processor, mine)
    if rank == 0:
        if compact:  # per-task item lists, as ``run`` returns them
            results = [data_processor.items_from_result(b) for b in results]
        data_processor.process_items(results)
