"""The N > 1 path on CPU: world_size-2 ``gloo`` process groups exercise the same sharding and
gather code the GPUs run over RCCL (sai_amd.distributed).  The compute inside a chunk is a toy
preprocessor here, exactly how the reference tests its executors
(tests/multiprocessing/test_mp_pool.py:25-46)."""

import json
import os
import socket

import numpy as np
import pytest


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))  # fmt: skip
    from sai_amd.distributed import gather_padded, gather_window_records, init_process_group, my_chunk_indices, run_sharded
    from sai_amd.generators import DataGenerator
    from sai_amd.preprocessors import DataPreprocessor

    r, w = init_process_group("gloo")
    assert (r, w) == (rank, world)

    class Gen(DataGenerator):
        def get(self):
            for i in range(7):
                yield {"chr_name": "1", "start": i * 100 + 1, "end": i * 100 + 150}

    class Pre(DataPreprocessor):
        def __init__(self):
            self.written = None

        def run(self, chr_name, start, end):
            return [{"chr": chr_name, "start": start, "end": end, "rank": dist.get_rank(), "k": k} for k in range(2)]

        def process_items(self, items):
            self.written = items

    pre = Pre()
    items = run_sharded(pre, Gen())
    mine = list(my_chunk_indices(7, rank, world))
    if rank == 0:
        assert [it["start"] for it in items] == [i * 100 + 1 for i in range(7) for _ in range(2)]
        assert [it["rank"] for it in items] == [0] * 8 + [1] * 6  # 4 chunks on rank 0, 3 on rank 1
        assert pre.written is items and mine == [0, 1, 2, 3]
    else:
        assert items is None and pre.written is None and mine == [4, 5, 6]

    # the reference's executor name and contract (results per task, unflattened, in task order)
    from sai_amd.multiprocessing import mp_pool

    pre2 = Pre()
    mp_pool(pre2, Gen(), nprocess=4)
    if rank == 0:
        assert [[it["start"] for it in res] for res in pre2.written] == [[i * 100 + 1] * 2 for i in range(7)]
        assert [res[0]["rank"] for res in pre2.written] == [0, 0, 0, 0, 1, 1, 1]
    else:
        assert pre2.written is None

    # a rank that fails inside its chunks: nobody waits in the gather for it -- the failing rank raises its own
    # error, the other says which rank failed (it used to block until the collective timed out)
    class Failing(Pre):
        def run(self, chr_name, start, end):
            if start == 501:  # chunk 5: rank 1's
                raise ValueError("malformed line at 1:501")
            return super().run(chr_name, start, end)

    from sai_amd.distributed import ShardFailure

    bad = Failing()
    with pytest.raises(ValueError if rank == 1 else ShardFailure, match="malformed line" if rank == 1 else r"rank\(s\) \[1\] failed"):
        run_sharded(bad, Gen())
    assert bad.written is None

    # records of different lengths per rank -> rank 0, in rank order
    local = torch.arange(10 + 5 * rank, dtype=torch.uint8) + 100 * rank
    got = gather_window_records(local)
    got2 = gather_padded(local.to(torch.int64), [10, 15])
    if rank == 0:
        assert [g.tolist() for g in got] == [list(range(10)), [100 + i for i in range(15)]]
        assert [g.tolist() for g in got2] == [list(range(10)), [100 + i for i in range(15)]]
    else:
        assert got is None and got2 is None
    dist.barrier()
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_chunk_assignment_covers_everything():
    from sai_amd.distributed import my_chunk_indices

    for n in (0, 1, 5, 8, 17, 110000):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in my_chunk_indices(n, r, world)]
            assert got == list(range(n))
            sizes = [len(my_chunk_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_run_sharded_is_the_serial_loop():
    from sai_amd.distributed import gather_window_records, run_sharded
    from sai_amd.generators import DataGenerator
    from sai_amd.preprocessors import DataPreprocessor

    class Gen(DataGenerator):
        def get(self):
            return iter([{"x": 1}, {"x": 2}])

    class Pre(DataPreprocessor):
        def run(self, x):
            return [x, x * 10]

        def process_items(self, items):
            self.items = items

    p = Pre()
    assert run_sharded(p, Gen()) == [1, 10, 2, 20] and p.items == [1, 10, 2, 20]
    import torch

    t = torch.arange(4)
    assert gather_window_records(t)[0] is t


def test_mp_pool_single_process():
    """Without a process group mp_pool runs the tasks in order in this process
    (reference contract: tests/multiprocessing/test_mp_pool.py:25-46)."""
    from sai_amd.generators import DataGenerator
    from sai_amd.multiprocessing import mp_pool
    from sai_amd.preprocessors import DataPreprocessor

    class Gen(DataGenerator):
        def get(self):
            for i in range(5):
                yield {"a": i, "b": 2 * i}

    class Pre(DataPreprocessor):
        def run(self, a, b):
            return a + b

        def process_items(self, items):
            self.items = items

    pre = Pre()
    mp_pool(pre, Gen(), nprocess=2)
    assert pre.items == [0, 3, 6, 9, 12]


# ---- the numeric transport of the product path (WindowBatch rows) over gloo --------------------


def _batch_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))  # fmt: skip
    from sai_amd.distributed import init_process_group, run_sharded
    from sai_amd.multiprocessing import mp_pool

    init_process_group("gloo")
    pre, gen = _FakeChunkPreprocessor(), _FakeChunks()
    items = run_sharded(pre, gen)
    pre2 = _FakeChunkPreprocessor()
    mp_pool(pre2, _FakeChunks(), nprocess=3)
    if rank == 0:
        with open(os.path.join(out_dir, "items.json"), "w") as f:
            json.dump([items, pre.written, pre2.written], f)
    else:
        assert items is None and pre.written is None and pre2.written is None
    dist.barrier()
    dist.destroy_process_group()


class _FakeChunks:
    def get(self):
        for i in range(7):
            yield {"chr_name": "3", "start": 1 + 100 * i, "end": 100 * (i + 1)}

    def __len__(self):
        return 7


class _FakeChunkPreprocessor:
    """ChunkPreprocessor's numeric protocol (run_compact / pack_result / unpack_result /
    items_from_results) with made-up numbers instead of kernels: two population combinations, three
    windows per chunk."""

    def __init__(self):
        self.written = None

    def run_compact(self, chr_name, start, end):
        from sai_amd.engine import RECORD_DTYPE, WindowResults
        from sai_amd.preprocessors.window_batch import ComboBatch, WindowBatch

        combos = []
        for ci, tgt in enumerate(("t1", "t2")):
            win = np.array([[start + 10 * k, start + 10 * k + 29] for k in range(3)], dtype=np.int64)
            rec = np.zeros((1, 3), dtype=RECORD_DTYPE)
            rec["n_sites"] = 5
            rec["u_count"][0] = [(start + k + ci) % 3 for k in range(3)]
            off = np.zeros((1, 3, 2), dtype=np.int64)
            off[0, :, 0] = np.cumsum(rec["u_count"][0]) - rec["u_count"][0]
            lists = np.arange(int(rec["u_count"].sum()), dtype=np.int32) + start
            combos.append(ComboBatch("r", tgt, ("s",), None, win, np.full(3, 5, np.int32), ["U"],
                                     WindowResults(rec, off, lists, np.zeros(0, np.int32))))  # fmt: skip
        return WindowBatch(chr_name, combos)

    def run(self, **params):
        return self.items_from_result(self.run_compact(**params))

    @staticmethod
    def pack_result(batch):
        return batch.to_bytes()

    @staticmethod
    def unpack_result(raw):
        from sai_amd.preprocessors.window_batch import WindowBatch

        return WindowBatch.from_bytes(raw)

    @staticmethod
    def items_from_result(batch):
        out = []
        for cb in batch.combos:
            for wi in range(len(cb.windows)):
                out.append([cb.tgt_pop, int(cb.windows[wi, 0]), int(cb.uq.records[0, wi]["u_count"]), cb.uq.u_list(0, wi).tolist()])
        return out

    def items_from_results(self, batches):
        out = []
        for k in range(2):  # combination-major, chunks in order: what ONE chunk would emit
            for b in batches:
                out.extend(it for it in self.items_from_result(b) if it[0] == ("t1", "t2")[k])
        return out

    def process_items(self, items):
        self.written = items


@pytest.mark.parametrize("world", [2, 4])
def test_numeric_rows_through_run_sharded_and_mp_pool(tmp_path, world):
    """A processor with the numeric protocol travels as ONE byte row per rank (sizes exchanged,
    one padded gather) and rank 0 emits the items in single-chunk order, for 2 and 4 ranks over 7
    uneven chunks; mp_pool keeps the reference's per-task, task-ordered contract."""
    import torch.multiprocessing as mp

    mp.spawn(_batch_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    items, written, pooled = json.load(open(tmp_path / "items.json"))
    pre, chunks = _FakeChunkPreprocessor(), list(_FakeChunks().get())
    serial = pre.items_from_results([pre.run_compact(**c) for c in chunks])
    assert items == written == serial
    assert [it[0] for it in items] == ["t1"] * 21 + ["t2"] * 21 and [it[1] for it in items[:4]] == [1, 11, 21, 101]
    assert pooled == [pre.run(**c) for c in chunks]  # one list per task, in task order
