"""``sai outlier`` sub-command (mirror of sai/parsers/outlier_parser.py:27-77)."""

from __future__ import annotations

import argparse

from ..sai import outlier
from .argument_validation import between_zero_and_one, existed_file


def _run_outlier(args: argparse.Namespace) -> None:
    outlier(score_file=args.score, output_prefix=args.output_prefix, quantile=args.quantile)


def add_outlier_parser(subparsers) -> None:
    parser = subparsers.add_parser("outlier", help="Detect and output outlier rows based on quantile thresholds.")
    parser.add_argument("--score", type=existed_file, required=True, help="Path to the input score file.")
    parser.add_argument("--output-prefix", dest="output_prefix", type=str, required=True, help="Prefix of the output files.")
    parser.add_argument("--quantile", type=between_zero_and_one, default=0.99,
                        help="Quantile threshold for outlier detection, between 0 and 1. Default: 0.99.")  # fmt: skip
    parser.set_defaults(runner=_run_outlier)
