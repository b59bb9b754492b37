"""The native output formatter (libsaihip: sai_format_score_rows / sai_format_log_rows, host side):
numbers print exactly as Python's str() prints them, and the rows equal what
FeaturePreprocessor.process_items writes from item dictionaries (whose text is pinned to the
reference's by the golden captures)."""

import ctypes as C
import struct

import numpy as np
import pytest


def native_strs(values: np.ndarray) -> list[str]:
    from sai_amd import _ffi

    lib = _ffi.load_host()
    v = np.ascontiguousarray(values, dtype=np.float64)
    h = C.c_void_p()
    _ffi.check(lib.sai_format_doubles(v.ctypes.data_as(C.c_void_p), v.size, C.byref(h)), lib)
    n = C.c_int64()
    ptr = lib.sai_text_data(h, C.byref(n))
    text = C.string_at(ptr, n.value).decode()
    lib.sai_text_free(h)
    return text.split("\n")[:-1]


def test_doubles_print_like_python_str():
    rng = np.random.default_rng(1)
    special = [0.0, -0.0, 1.0, -1.0, 0.1, 0.9, 0.5, 1e-4, 9.999e-5, 1e-5, 1e15, 9.999999999999998e15, 1e16, 1e17, 1e22, 1e23,
               123456.0, 0.6179, 0.6142000000000001, 0.9666666666666667, 2 / 3, 1 / 3, 5e-324, 2.2250738585072014e-308,
               1.7976931348623157e308, float("nan"), float("inf"), float("-inf"), 1e-4 * (1 - 2**-52), 4.35, 0.1 + 0.2]  # fmt: skip
    bits = rng.integers(0, 2**64, size=200_000, dtype=np.uint64)  # every exponent, every kind of mantissa
    rand = bits.view(np.float64)
    ratios = rng.integers(0, 4001, size=100_000) / rng.integers(1, 4001, size=100_000)  # frequencies and the like
    small = ratios * 10.0 ** rng.integers(-12, 20, size=ratios.size)
    values = np.concatenate([np.array(special), rand, ratios, small, -small])
    got = native_strs(values)
    assert len(got) == values.size
    bad = [(float(v).hex(), g, str(float(v))) for v, g in zip(values.tolist(), got) if g != str(float(v))]
    assert not bad, bad[:5]
    assert str(np.float64(0.6142000000000001)) == native_strs(np.array([0.6142000000000001]))[0]  # numpy scalars print the same


def _random_batch(rng, n_w, n_src, with_extra, pos_dtype):
    from sai_amd.engine import RECORD_DTYPE, WindowResults
    from sai_amd.preprocessors.window_batch import ComboBatch, WindowBatch

    rec = np.zeros((2, n_w), dtype=RECORD_DTYPE)
    nsnps = rng.integers(0, 40, n_w).astype(np.int32)
    nsnps[rng.random(n_w) < 0.15] = 0
    rec["n_sites"] = nsnps
    rec["u_count"][0] = np.where(nsnps > 0, rng.integers(0, 4, n_w), 0)
    rec["n_cond"][1] = np.where(nsnps > 0, rng.integers(0, 3, n_w), 0)
    rec["n_cdd_q"][1] = np.where(rec["n_cond"][1] > 0, rng.integers(1, 3, n_w), 0)
    rec["q"] = rng.integers(0, 2001, (2, n_w)) / 2000.0
    rec["q"][1][rec["n_cond"][1] == 0] = np.nan
    off = np.zeros((2, n_w, 2), dtype=np.int64)
    for k, name in enumerate(("u_count", "n_cdd_q")):
        flat = rec[name].reshape(-1).astype(np.int64)
        off[:, :, k] = (np.cumsum(flat) - flat).reshape(2, n_w)
    uq = WindowResults(rec, off, rng.integers(1, 10**8, int(rec["u_count"].sum())).astype(np.int32),
                       rng.integers(1, 10**8, int(rec["n_cdd_q"].sum())).astype(np.int32))  # fmt: skip
    win = np.stack([np.arange(n_w) * 500 + 1, np.arange(n_w) * 500 + 1000], axis=1).astype(np.int64)
    four = dd = None
    if with_extra:
        four = rng.standard_normal((n_w, n_src, 4)) * 10.0 ** rng.integers(-8, 8, (n_w, n_src, 4))
        four[rng.random(four.shape) < 0.1] = np.nan
        dd = rng.standard_normal((n_w, n_src))
    cb = ComboBatch("refA", "tgtB", tuple(f"s{i}" for i in range(n_src)), "og" if with_extra else None, win, nsnps, ["U", "Q"], uq,
                    four, dd, pos_dtype)  # fmt: skip
    return WindowBatch("chr7", [cb])


@pytest.mark.parametrize("n_src,with_extra,pos_dtype", [(1, False, "int32"), (2, True, "int32"), (1, True, "int64"), (3, False, "int64")])
def test_native_rows_equal_process_items(tmp_path, n_src, with_extra, pos_dtype):
    from sai_amd.configs import StatConfig
    from sai_amd.preprocessors import FeaturePreprocessor

    rng = np.random.default_rng(n_src * 7 + with_extra)
    src = {f"s{i}": "=1" for i in range(n_src)}
    stats = {"U": {"ref": {"refA": 0.1}, "tgt": {"tgtB": 0.5}, "src": dict(src)},
             "Q": {"ref": {"refA": 0.1}, "tgt": {"tgtB": 0.9}, "src": dict(src)}}  # fmt: skip
    if with_extra:
        stats = {"fd": True, "U": stats["U"], "df": True, "Danc": False, "Q": stats["Q"], "Dplus": True, "DD": True}
    # 400 windows: one piece on the calling thread; 9 000: several pieces formatted by the worker pool and joined
    n_w = 9000 if (n_src == 2 or pos_dtype == "int64" and not with_extra) else 400
    batch = _random_batch(rng, n_w, n_src, with_extra, pos_dtype)
    a, b = tmp_path / "items.tsv", tmp_path / "native.tsv"
    fa = FeaturePreprocessor(str(a), StatConfig(dict(stats)))
    fa.process_items(fa.items_from_batch(batch))
    fb = FeaturePreprocessor(str(b), StatConfig(dict(stats)))
    fb.write_batches([batch])
    assert b.read_text() == a.read_text() and len(a.read_text().splitlines()) == n_w
    for k in ("U", "Q"):
        assert b.with_suffix(f".{k}.log").read_text() == a.with_suffix(f".{k}.log").read_text()
    assert "NA\n" in a.with_suffix(".Q.log").read_text() and "\tnan" in a.read_text()


def _text(lib, handle) -> bytes:
    n = C.c_int64()
    ptr = lib.sai_text_data(handle, C.byref(n))
    data = C.string_at(ptr, n.value)
    lib.sai_text_free(handle)
    return data


def _row_args(batch):
    """The C arguments of one combination's rows, as write_batches builds them."""
    from sai_amd import _ffi

    cb = batch.combos[0]
    win = np.ascontiguousarray(cb.windows, dtype=np.int64)
    nsnps = np.ascontiguousarray(cb.nsnps, dtype=np.int32)
    u, q = cb.uq.records[0]["u_count"], cb.uq.records[1]["q"]
    cols = (_ffi.SaiTextColumn * 2)(_ffi.SaiTextColumn(u.ctypes.data, u.strides[0], 0, 0), _ffi.SaiTextColumn(q.ctypes.data, q.strides[0], 1, 0))
    lists = []
    for si, (cnt, k, arr) in enumerate((("u_count", 0, cb.uq.cdd_u), ("n_cdd_q", 1, cb.uq.cdd_q))):
        lists.append((cb.uq.records[si][cnt], np.ascontiguousarray(cb.uq.offsets[si, :, k]), np.ascontiguousarray(arr)))
    return cb, win, nsnps, cols, lists, (u, q)


@pytest.mark.parametrize("n_w", [300, 9000])
def test_written_files_equal_the_in_memory_text(tmp_path, n_w):
    """sai_write_window_rows (what write_batches calls: files written by the library) against
    sai_format_score_rows / sai_format_log_rows (the same rows as text in memory), both sizes of fan-out;
    an fd < 0 leaves that output out; byte counts are reported per output."""
    import os

    from sai_amd import _ffi

    lib = _ffi.load_host()
    batch = _random_batch(np.random.default_rng(n_w), n_w, 1, False, "int32")
    cb, win, nsnps, cols, lists, _keep = _row_args(batch)
    pops = b"refA\ttgtB\ts0\tNA"
    h = C.c_void_p()
    _ffi.check(lib.sai_format_score_rows(b"chr7", pops, n_w, win.ctypes.data_as(C.c_void_p), nsnps.ctypes.data_as(C.c_void_p), 2, cols, C.byref(h)), lib)
    want = [_text(lib, h)]
    for counts, offs, arr in lists:
        h = C.c_void_p()
        _ffi.check(lib.sai_format_log_rows(b"chr7", n_w, win.ctypes.data_as(C.c_void_p), C.c_void_p(counts.ctypes.data), counts.strides[0],
                                           offs.ctypes.data_as(C.c_void_p), 1, C.c_void_p(arr.ctypes.data), 4, C.byref(h)), lib)  # fmt: skip
        want.append(_text(lib, h))
    assert want[0].count(b"\n") == n_w and b"NA\n" in want[2]
    for skip in (None, 0, 2):
        paths = [tmp_path / f"{n_w}_{skip}_{k}" for k in range(3)]
        fds = [-1 if k == skip else os.open(p, os.O_WRONLY | os.O_CREAT | os.O_APPEND) for k, p in enumerate(paths)]
        logs = (_ffi.SaiLogRows * 2)(*[_ffi.SaiLogRows(c.ctypes.data, c.strides[0], o.ctypes.data, 1, a.ctypes.data, 4, fds[1 + k])
                                       for k, (c, o, a) in enumerate(lists)])  # fmt: skip
        nbytes = (C.c_int64 * 3)()
        for _ in range(2):  # appended twice: the pieces' buffers are reused from call to call
            _ffi.check(lib.sai_write_window_rows(b"chr7", pops, n_w, win.ctypes.data_as(C.c_void_p), nsnps.ctypes.data_as(C.c_void_p), 2, cols,
                                                 fds[0], 2, logs, nbytes), lib)  # fmt: skip
        for k, p in enumerate(paths):
            if k == skip:
                assert not p.exists() and nbytes[k] == 0
            else:
                os.close(fds[k])
                assert p.read_bytes() == want[k] * 2 and nbytes[k] == len(want[k])


def test_write_failure_is_reported(tmp_path):
    import os

    from sai_amd import _ffi

    lib = _ffi.load_host()
    batch = _random_batch(np.random.default_rng(3), 50, 1, False, "int32")
    cb, win, nsnps, cols, lists, _keep = _row_args(batch)
    fd = os.open(tmp_path / "ro", os.O_RDONLY | os.O_CREAT)  # not open for writing
    try:
        with pytest.raises(_ffi.SaiHipError, match="write to fd"):
            _ffi.check(lib.sai_write_window_rows(b"c", b"a\tb\tc\tNA", 50, win.ctypes.data_as(C.c_void_p), nsnps.ctypes.data_as(C.c_void_p), 2, cols,
                                                 fd, 0, None, None), lib)  # fmt: skip
    finally:
        os.close(fd)
    with pytest.raises(_ffi.SaiHipError):  # a list that is NULL although a window has entries
        bad = (_ffi.SaiLogRows * 1)(_ffi.SaiLogRows(lists[0][0].ctypes.data, lists[0][0].strides[0], lists[0][1].ctypes.data, 1, None, 4, 1))
        assert lists[0][0].sum() > 0
        _ffi.check(lib.sai_write_window_rows(b"c", b"a\tb\tc\tNA", 50, win.ctypes.data_as(C.c_void_p), nsnps.ctypes.data_as(C.c_void_p), 2, cols,
                                             -1, 1, bad, None), lib)  # fmt: skip
    with pytest.raises(_ffi.SaiHipError):
        _ffi.check(lib.sai_write_window_rows(b"c", b"a", 50, win.ctypes.data_as(C.c_void_p), nsnps.ctypes.data_as(C.c_void_p), 2, cols, -1, 9, None, None), lib)


def test_writer_in_a_forked_child(tmp_path):
    """After a fork the pool's threads are gone: the child formats on its own thread, same bytes."""
    import os
    import time

    from sai_amd.configs import StatConfig
    from sai_amd.preprocessors import FeaturePreprocessor

    stats = {"U": {"ref": {"refA": 0.1}, "tgt": {"tgtB": 0.5}, "src": {"s0": "=1"}}, "Q": {"ref": {"refA": 0.1}, "tgt": {"tgtB": 0.9}, "src": {"s0": "=1"}}}
    batch = _random_batch(np.random.default_rng(11), 9000, 1, False, "int32")
    FeaturePreprocessor(str(tmp_path / "parent.tsv"), StatConfig(dict(stats))).write_batches([batch])  # the pool's threads exist now
    pid = os.fork()
    if pid == 0:
        try:
            FeaturePreprocessor(str(tmp_path / "child.tsv"), StatConfig(dict(stats))).write_batches([batch])
            os._exit(0)
        except BaseException:
            os._exit(1)
    for _ in range(200):
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        time.sleep(0.1)
    else:
        os.kill(pid, 9)
        os.waitpid(pid, 0)
        raise AssertionError("the forked child hung in sai_write_window_rows")
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    for sfx in (".tsv", ".U.log", ".Q.log"):
        assert (tmp_path / f"child{sfx}").read_bytes() == (tmp_path / f"parent{sfx}").read_bytes()


def test_a_failed_output_takes_the_others_back(tmp_path):
    """the TSV and the logs of a run must agree on their last window.  The Q log's descriptor is
    not open for writing: the call fails, and the TSV and the U log -- already written by then -- are cut
    back to where they were before the call (earlier rows stay)."""
    import os

    from sai_amd import _ffi

    lib = _ffi.load_host()
    n_w = 400
    batch = _random_batch(np.random.default_rng(5), n_w, 1, False, "int32")
    cb, win, nsnps, cols, lists, _keep = _row_args(batch)
    paths = [tmp_path / n for n in ("s.tsv", "s.U.log", "s.Q.log")]
    for p in paths:
        p.write_bytes(b"header of an earlier call\n")
    fds = [os.open(paths[0], os.O_WRONLY | os.O_APPEND), os.open(paths[1], os.O_WRONLY | os.O_APPEND), os.open(paths[2], os.O_RDONLY)]
    try:
        logs = (_ffi.SaiLogRows * 2)(*[_ffi.SaiLogRows(c.ctypes.data, c.strides[0], o.ctypes.data, 1, a.ctypes.data, 4, fds[1 + k])
                                       for k, (c, o, a) in enumerate(lists)])  # fmt: skip
        nbytes = (C.c_int64 * 3)()
        with pytest.raises(_ffi.SaiHipError, match="write to fd"):
            _ffi.check(lib.sai_write_window_rows(b"chr7", b"a\tb\tc\tNA", n_w, win.ctypes.data_as(C.c_void_p), nsnps.ctypes.data_as(C.c_void_p), 2,
                                                 cols, fds[0], 2, logs, nbytes), lib)  # fmt: skip
        assert list(nbytes) == [0, 0, 0]
        for p in paths:
            assert p.read_bytes() == b"header of an earlier call\n"
        os.close(fds[2])
        fds[2] = os.open(paths[2], os.O_WRONLY | os.O_APPEND)  # the same call with a usable descriptor: appended behind the headers
        logs[1].fd = fds[2]
        _ffi.check(lib.sai_write_window_rows(b"chr7", b"a\tb\tc\tNA", n_w, win.ctypes.data_as(C.c_void_p), nsnps.ctypes.data_as(C.c_void_p), 2,
                                             cols, fds[0], 2, logs, nbytes), lib)  # fmt: skip
        for k, p in enumerate(paths):
            body = p.read_bytes()
            assert body.startswith(b"header of an earlier call\n") and len(body) == 26 + nbytes[k] and body.count(b"\n") == 1 + n_w
    finally:
        for fd in fds:
            os.close(fd)


def test_fork_while_another_thread_is_writing(tmp_path):
    """a fork() while another thread is inside sai_write_window_rows hands the child a copy of the
    writer's mutex in its locked state.  The child starts with a fresh writer state (pthread_atfork) and
    writes the same bytes instead of waiting for ever."""
    import os
    import threading
    import time

    from sai_amd.configs import StatConfig
    from sai_amd.preprocessors import FeaturePreprocessor

    stats = {"U": {"ref": {"refA": 0.1}, "tgt": {"tgtB": 0.5}, "src": {"s0": "=1"}}, "Q": {"ref": {"refA": 0.1}, "tgt": {"tgtB": 0.9}, "src": {"s0": "=1"}}}
    small = _random_batch(np.random.default_rng(11), 2000, 1, False, "int32")
    big = _random_batch(np.random.default_rng(12), 400000, 1, False, "int32")
    FeaturePreprocessor(str(tmp_path / "parent.tsv"), StatConfig(dict(stats))).write_batches([small])
    stop = threading.Event()

    def keep_writing():  # the library releases the GIL while it formats and writes
        fp = FeaturePreprocessor(str(tmp_path / "busy.tsv"), StatConfig(dict(stats)))
        while not stop.is_set():
            fp.write_batches([big])
            for sfx in (".tsv", ".U.log", ".Q.log"):
                os.truncate(tmp_path / f"busy{sfx}", 0)

    th = threading.Thread(target=keep_writing)
    th.start()
    try:
        hung = 0
        for attempt in range(6):
            time.sleep(0.05 + 0.03 * attempt)  # most forks land while the other thread holds the writer
            pid = os.fork()
            if pid == 0:
                try:
                    FeaturePreprocessor(str(tmp_path / f"child{attempt}.tsv"), StatConfig(dict(stats))).write_batches([small])
                    os._exit(0)
                except BaseException:
                    os._exit(1)
            for _ in range(300):
                done, status = os.waitpid(pid, os.WNOHANG)
                if done:
                    break
                time.sleep(0.1)
            else:
                os.kill(pid, 9)
                os.waitpid(pid, 0)
                hung += 1
                continue
            assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
            for sfx in (".tsv", ".U.log", ".Q.log"):
                assert (tmp_path / f"child{attempt}{sfx}").read_bytes() == (tmp_path / f"parent{sfx}").read_bytes()
        assert hung == 0, f"{hung} forked children hung in sai_write_window_rows"
    finally:
        stop.set()
        th.join()
