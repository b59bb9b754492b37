"""``sai score`` sub-command (mirror of sai/parsers/score_parser.py:27-130): same flags and
defaults."""

from __future__ import annotations

import argparse

from ..launcher import workers_from_env
from ..sai import score
from .argument_validation import existed_file, positive_int


def resolve_workers(args: argparse.Namespace) -> int:
    """``--num-workers``, else $SAI_AMD_GPUS, else 1 (the reference's CLI, score_parser.py:64).  The environment is
    read when `score` runs, not while the parser is built: a malformed value is `sai score`'s usage error only,
    never a traceback from `sai --help` or another sub-command."""
    if args.num_workers is None:
        try:
            args.num_workers = workers_from_env()
        except ValueError:
            args.score_parser.error("SAI_AMD_GPUS must be a positive integer")
    return args.num_workers


def _run_score(args: argparse.Namespace) -> None:
    resolve_workers(args)
    score(
        vcf_file=args.vcf,
        chr_name=args.chr_name,
        win_len=args.win_len,
        win_step=args.win_step,
        anc_allele_file=args.anc_alleles,
        output_file=args.output,
        config=args.config,
        num_workers=args.num_workers,
    )


def add_score_parser(subparsers) -> None:
    parser = subparsers.add_parser("score", help="Run the score command based on specified parameters.")
    parser.add_argument("--vcf", type=existed_file, required=True, help="Path to the VCF file containing variant data.")
    parser.add_argument("--chr-name", dest="chr_name", type=str, required=True,
                        help="Chromosome name to analyze from the VCF file.")  # fmt: skip
    parser.add_argument("--win-len", dest="win_len", type=positive_int, default=50000,
                        help="Length of each genomic window in base pairs. Default: 50,000.")  # fmt: skip
    parser.add_argument("--win-step", dest="win_step", type=positive_int, default=10000,
                        help="Step size in base pairs between consecutive windows. Default: 10,000.")  # fmt: skip
    parser.add_argument("--anc-alleles", dest="anc_alleles", type=existed_file, default=None,
                        help="Path to the BED file with ancestral allele information. Without it, a site is "
                        "tested against both alleles (y and 1 - y) when the statistics are computed. Default: None.")  # fmt: skip
    parser.add_argument("--output", type=str, required=True, help="Output file path for saving results.")
    parser.add_argument("--config", type=existed_file, required=True,
                        help="Path to the YAML configuration file specifying the statistics to compute, ploidy "
                        "settings, and population group file paths.")  # fmt: skip
    # not a flag of the reference, whose CLI passes num_workers=1 (score_parser.py:64): the number of GPUs,
    # one worker process each (sai.py:42); the default, 1, is the reference's behaviour
    parser.add_argument("--num-workers", dest="num_workers", type=positive_int, default=None,
                        help="Number of GPUs to use, one worker process per GPU. Default: $SAI_AMD_GPUS, else 1.")  # fmt: skip
    parser.set_defaults(runner=_run_score, score_parser=parser)
