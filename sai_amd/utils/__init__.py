from .genomic_dataclasses import ChromosomeData
from .natsort_df import natsorted_df
from .geno import (
    check_anc_allele,
    filter_fixed_variants,
    filter_geno_data,
    flip_snps,
    get_ref_alt_allele,
    read_geno_data,
    reshape_genotypes,
)
from .read_data import read_data, read_dosage_data
from .samples import parse_ind_file
from .unique_key_loader import UniqueKeyLoader
from .vcf import read_anc_allele
from .windows import split_genome, split_index_ranges, split_windows_ranges

__all__ = [
    "ChromosomeData",
    "UniqueKeyLoader",
    "check_anc_allele",
    "filter_fixed_variants",
    "filter_geno_data",
    "flip_snps",
    "get_ref_alt_allele",
    "natsorted_df",
    "parse_ind_file",
    "read_anc_allele",
    "read_data",
    "read_dosage_data",
    "read_geno_data",
    "reshape_genotypes",
    "split_genome",
    "split_index_ranges",
    "split_windows_ranges",
]
