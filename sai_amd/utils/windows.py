"""Window grid helpers (mirror of sai/utils/utils.py:558-612 and
sai/generators/chunk_generator.py:111-142)."""

from __future__ import annotations

from typing import Optional, Sequence


def split_genome(pos, window_size: int, step_size: int, start: Optional[int] = None) -> list[tuple]:
    """Inclusive ``(start, end)`` sliding windows covering ``pos[0] .. pos[-1]``.

    Same validation and messages as utils.py:593-598.  The first window is aligned to the step
    grid so that it ends at the first multiple of ``step_size`` above ``pos[0]`` (:601), never
    starts before ``start`` (default 1, :602-604), and windows are emitted while their start is
    <= ``pos[-1]`` (:607-610)."""
    if step_size <= 0 or window_size <= 0:
        raise ValueError("`step_size` and `window_size` must be positive integers.")
    if step_size > window_size:
        raise ValueError("`step_size` cannot be greater than `window_size`.")
    if len(pos) == 0:
        raise ValueError("`pos` array must not be empty.")
    first, last = int(pos[0]), int(pos[-1])
    lower = 1 if start is None else int(start)
    s = max((first + step_size) // step_size * step_size - window_size + 1, lower)
    n = 0 if s > last else (last - s) // step_size + 1
    return [(s + k * step_size, s + k * step_size + window_size - 1) for k in range(n)]


def split_windows_ranges(windows: Sequence[tuple], num_chunks: int) -> list[tuple]:
    """Contiguous window ranges, one per chunk: ``(first.start, last.end)``; the first
    ``len(windows) % num_chunks`` chunks get one extra window and empty chunks are dropped
    (chunk_generator.py:130-142)."""
    base, extra = divmod(len(windows), num_chunks)
    out, i = [], 0
    for c in range(num_chunks):
        j = i + base + (1 if c < extra else 0)
        if j > i:
            out.append((windows[i][0], windows[j - 1][1]))
        i = j
    return out


def split_index_ranges(n_items: int, num_chunks: int) -> list[tuple[int, int]]:
    """The same balancing rule on indices: [(i0, i1), ...] half-open, empty ranges dropped."""
    base, extra = divmod(n_items, num_chunks)
    out, i = [], 0
    for c in range(num_chunks):
        j = i + base + (1 if c < extra else 0)
        if j > i:
            out.append((i, j))
        i = j
    return out
