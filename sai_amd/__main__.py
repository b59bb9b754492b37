"""``python -m sai_amd <command> ...`` -- the ``sai`` command line (sai/__main__.py:27-76 is the
interface: ``main(arg_list)``, and the two helpers its test patches, ``_sai_cli_parser`` and
``_set_sigpipe_handler``)."""

from __future__ import annotations

import argparse
import signal
from typing import Optional, Sequence

import sai_amd.stats  # noqa: F401  (fills STAT_REGISTRY)
from sai_amd import __version__
from sai_amd.parsers.outlier_parser import add_outlier_parser
from sai_amd.parsers.score_parser import add_score_parser

_COMMANDS = (add_score_parser, add_outlier_parser)


def _set_sigpipe_handler() -> None:
    """`sai ... | head` should end quietly: give SIGPIPE its default action where the platform
    has the signal."""
    sigpipe = getattr(signal, "SIGPIPE", None)
    if sigpipe is not None:
        signal.signal(sigpipe, signal.SIG_DFL)


def _sai_cli_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(prog="sai", description="SAI: Statistics for Adaptive Introgression (MI355X build)")
    parser.add_argument("--version", action="version", version=str(__version__))
    commands = parser.add_subparsers(dest="subcommand", required=True)
    for add_command in _COMMANDS:
        add_command(commands)
    return parser


def main(arg_list: Optional[Sequence[str]] = None) -> None:
    _set_sigpipe_handler()
    parser = _sai_cli_parser()
    args = parser.parse_args(arg_list)
    args.runner(args)


if __name__ == "__main__":
    main()
