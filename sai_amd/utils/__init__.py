from .genomic_dataclasses import ChromosomeData
from .natsort_df import natsorted_df
from .read_data import read_data
from .samples import parse_ind_file
from .unique_key_loader import UniqueKeyLoader
from .vcf import read_anc_allele
from .windows import split_genome, split_index_ranges, split_windows_ranges

__all__ = [
    "ChromosomeData",
    "UniqueKeyLoader",
    "natsorted_df",
    "parse_ind_file",
    "read_anc_allele",
    "read_data",
    "split_genome",
    "split_index_ranges",
    "split_windows_ranges",
]
