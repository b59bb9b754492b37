"""Chunks of the window grid (mirror of sai/generators/chunk_generator.py:26-142)."""

from __future__ import annotations

from typing import Iterator

from ..utils.native_vcf import scan_first_last
from ..utils.windows import split_genome, split_windows_ranges
from .data_generator import DataGenerator


class ChunkGenerator(DataGenerator):
    """Scans the VCF for the first and last position of ``chr_name``, lays the sliding-window
    grid over it and cuts the window list into ``num_chunks`` contiguous ranges; each chunk is
    ``{"chr_name", "start", "end"}`` = first window's start .. last window's end.  The same
    rule shards windows over GPUs (sai_amd.distributed)."""

    def __init__(self, vcf_file: str, chr_name: str, step_size: int, window_size: int, num_chunks: int, span=None):
        """``span`` = (first, last) position of the chromosome when the caller already has it (``score``
        scans a plain-text file while it is being read)."""
        chr_name = str(chr_name)
        first, last = scan_first_last(vcf_file, chr_name) if span is None else span
        if first is None:
            raise ValueError(f"Chromosome {chr_name} not found in VCF.")  # chunk_generator.py:75-76
        self.windows = split_genome([first, last], window_size, step_size)
        self.chunks = self._split_windows_ranges(self.windows, num_chunks)
        self.num_chunks = len(self.chunks)
        self.chr_name = chr_name

    def get(self) -> Iterator[dict]:
        for start, end in self.chunks:
            yield {"chr_name": self.chr_name, "start": start, "end": end}

    def __len__(self) -> int:
        return self.num_chunks

    def _split_windows_ranges(self, windows: list, num_chunks: int) -> list:
        return split_windows_ranges(windows, num_chunks)
