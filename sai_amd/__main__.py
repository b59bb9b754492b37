"""Command line entry: ``python -m sai_amd score ...`` (mirror of sai/__main__.py:27-76)."""

from __future__ import annotations

import argparse

import sai_amd.stats  # noqa: F401  (registers U and Q)
from sai_amd import __version__
from sai_amd.parsers.outlier_parser import add_outlier_parser
from sai_amd.parsers.score_parser import add_score_parser


def _set_sigpipe_handler() -> None:
    import os
    import signal

    if os.name == "posix":
        signal.signal(signal.SIGPIPE, signal.SIG_DFL)


def _sai_cli_parser() -> argparse.ArgumentParser:
    top_parser = argparse.ArgumentParser(description="SAI: Statistics for Adaptive Introgression (MI355X build)")
    top_parser.add_argument("--version", action="version", version=f"{__version__}")
    subparsers = top_parser.add_subparsers(dest="subcommand")
    subparsers.required = True
    add_score_parser(subparsers)
    add_outlier_parser(subparsers)
    return top_parser


def main(arg_list: list = None) -> None:
    _set_sigpipe_handler()
    args = _sai_cli_parser().parse_args(arg_list)
    args.runner(args)


if __name__ == "__main__":
    main()
