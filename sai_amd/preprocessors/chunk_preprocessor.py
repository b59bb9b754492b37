"""Chunk driver (mirror of sai/preprocessors/chunk_preprocessor.py:28-158): one chromosome
region -> all its windows, computed in batched launches on the GPU."""

from __future__ import annotations

import os
from typing import Any

from ..generators.window_generator import WindowGenerator
from ..utils.samples import parse_ind_file
from .data_preprocessor import DataPreprocessor
from .feature_preprocessor import FeaturePreprocessor


class ChunkPreprocessor(DataPreprocessor):
    def __init__(
        self,
        vcf_file: str,
        ref_ind_file: str,
        tgt_ind_file: str,
        src_ind_file: str,
        out_ind_file: str,
        win_len: int,
        win_step: int,
        output_file: str,
        ploidy_config,
        stat_config,
        anc_allele_file: str = None,
        num_src: int = 1,
    ):
        self.vcf_file = vcf_file
        self.ref_ind_file = ref_ind_file
        self.tgt_ind_file = tgt_ind_file
        self.src_ind_file = src_ind_file
        self.out_ind_file = out_ind_file
        self.win_len = win_len
        self.win_step = win_step
        self.ploidy_config = ploidy_config
        self.anc_allele_file = anc_allele_file
        # chunk_preprocessor.py:94-95: every population of the source file is used together
        self.num_src = len(parse_ind_file(src_ind_file).keys())
        self.feature_preprocessor = FeaturePreprocessor(
            output_file=output_file,
            stat_config=stat_config,
            anc_allele_available=anc_allele_file is not None,
        )

    def run(self, chr_name: str, start: int, end: int) -> list[dict[str, Any]]:
        """Items of all windows of [start, end] (chunk_preprocessor.py:105-147)."""
        return self.feature_preprocessor.items_from_batch(self.run_compact(chr_name, start, end))

    def preload(self, chr_name: str):
        """Every record of the chromosome, read and tokenised into HBM before the chunk bounds are known
        (``read_data_device`` without a region): ``(results, pos_dev, (lowest, highest position))``, the
        span None when nothing was kept.  ``run_compact(..., preloaded=)`` takes it when the chunk turns
        out to contain all of it."""
        from ..engine import Engine
        from ..utils.read_data import read_data_device

        results, pos_dev = read_data_device(
            Engine.get(), vcf_file=self.vcf_file, chr_name=chr_name, start=None, end=None, ref_ind_file=self.ref_ind_file,
            tgt_ind_file=self.tgt_ind_file, src_ind_file=self.src_ind_file, out_ind_file=self.out_ind_file,
            ploidy_config=self.ploidy_config, anc_allele_file=self.anc_allele_file)  # fmt: skip
        span = None
        for data, _ in (results.get(g, (None, None)) for g in ("ref", "tgt", "src", "outgroup")):
            for block in (data or {}).values():
                if block.POS.size:
                    lo, hi = int(block.POS.min()), int(block.POS.max())
                    span = (lo, hi) if span is None else (min(span[0], lo), max(span[1], hi))
        return results, pos_dev, span

    def run_and_write(self, chr_name: str, start: int, end: int, preloaded=None) -> None:
        """``write_results([run_compact(...)])`` for the ONE chunk of a one-process run, the writing overlapped
        with the scoring (``FeaturePreprocessor.score_and_write``): same files."""
        self.feature_preprocessor.score_and_write(self._window_generator(chr_name, start, end, preloaded))

    def run_compact(self, chr_name: str, start: int, end: int, preloaded=None):
        """The same work unit in numeric form (a ``WindowBatch``): what a rank of a sharded run
        computes and sends to rank 0, where ``unpack_results`` turns the batches into items."""
        return self.feature_preprocessor.score_windows(self._window_generator(chr_name, start, end, preloaded))

    def _window_generator(self, chr_name: str, start: int, end: int, preloaded=None):
        return WindowGenerator(
            preloaded=preloaded,
            vcf_file=self.vcf_file,
            chr_name=chr_name,
            start=start,
            end=end,
            ref_ind_file=self.ref_ind_file,
            tgt_ind_file=self.tgt_ind_file,
            src_ind_file=self.src_ind_file,
            out_ind_file=self.out_ind_file,
            win_len=self.win_len,
            win_step=self.win_step,
            ploidy_config=self.ploidy_config,
            anc_allele_file=self.anc_allele_file,
            num_src=self.num_src,
            resident=os.environ.get("SAI_AMD_INGEST", "device") != "host",
        )

    # transport hooks of sai_amd.distributed.run_sharded / sai_amd.multiprocessing.mp_pool
    @staticmethod
    def pack_result(batch) -> bytes:
        return batch.to_bytes()

    @staticmethod
    def unpack_result(raw):
        from .window_batch import WindowBatch

        return WindowBatch.from_bytes(raw)

    def items_from_result(self, batch) -> list[dict[str, Any]]:
        """Items of one task's batch -- what ``run`` returns for that task."""
        return self.feature_preprocessor.items_from_batch(batch)

    def items_from_results(self, batches) -> list[dict[str, Any]]:
        """All items of the tasks' batches in single-chunk order (combination-major)."""
        return self.feature_preprocessor.items_from_batches(batches)

    def process_items(self, items: list[dict[str, Any]]) -> None:
        self.feature_preprocessor.process_items(items)

    def write_results(self, batches) -> None:
        """Write the tasks' batches without going through item dictionaries (same files as
        ``process_items(items_from_results(batches))``)."""
        self.feature_preprocessor.write_batches(batches)
