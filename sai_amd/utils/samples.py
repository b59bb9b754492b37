"""Sample-list files (mirror of sai/utils/utils.py:31-75)."""

from __future__ import annotations


def parse_ind_file(filename: str) -> dict[str, list[str]]:
    """``population<ws>sample`` per line -> {population: [samples...]} in file order; lines that
    do not have exactly two fields are skipped; no usable line is a ValueError, a missing file a
    FileNotFoundError (utils.py:53-75)."""
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
