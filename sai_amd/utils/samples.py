"""Sample-list files (mirror of sai/utils/utils.py:31-75)."""

from __future__ import annotations


import os

_PARSED: dict = {}  # (path, mtime_ns, size) -> parsed lists: one `score` call asks for the same file seven times


def parse_ind_file(filename: str) -> dict[str, list[str]]:
    """``population<ws>sample`` per line -> {population: [samples...]} in file order; lines that
    do not have exactly two fields are skipped; no usable line is a ValueError, a missing file a
    FileNotFoundError (utils.py:53-75)."""
    key = None
    try:
        st = os.stat(filename)
        key = (os.fspath(filename), st.st_mtime_ns, st.st_size)
        if key in _PARSED:
            return {pop: list(names) for pop, names in _PARSED[key].items()}
    except (OSError, TypeError):
        key = None  # not a real path (tests patch open()): parse every time
    samples = _parse_ind_file(filename)
    if key is not None:
        if len(_PARSED) > 64:
            _PARSED.clear()
        _PARSED[key] = {pop: list(names) for pop, names in samples.items()}
    return samples


def _parse_ind_file(filename: str) -> dict[str, list[str]]:
    samples: dict[str, list[str]] = {}
    try:
        with open(filename, "r") as f:
            for line in f:
                parts = line.strip().split()
                if len(parts) != 2:
                    continue
                samples.setdefault(parts[0], []).append(parts[1])
    except FileNotFoundError:
        raise FileNotFoundError(f"File '{filename}' not found. Please check the file path.")
    if not samples:
        raise ValueError(f"No samples found in {filename}. Please check your data.")
    return samples
