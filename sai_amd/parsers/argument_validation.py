"""argparse ``type=`` callables of the command line.  Interface of
sai/parsers/argument_validation.py:26-169 (the names and the complaint texts its tests pin:
"<v> is not a valid integer", "<v> is not a positive number", "<v> is not between 0 and 1",
"<v> is not found"); here every numeric checker is one instance of a single table-driven class."""

from __future__ import annotations

import argparse
import os
from typing import Callable, Optional


class _NumberArg:
    """Convert with ``cast``; complain with "<raw> is not a valid <kind>" when that fails and with
    "<converted> is not <wanted>" when ``accept`` rejects the number.  ``None`` passes through, so
    an option's default is never second-guessed."""

    def __init__(self, cast: Callable, kind: str, accept: Callable[[float], bool], wanted: str):
        self.cast, self.kind, self.accept, self.wanted = cast, kind, accept, wanted
        self.__name__ = kind  # argparse shows it in "invalid <name> value" messages

    def __call__(self, text: Optional[str]):
        if text is None:
            return None
        try:
            number = self.cast(text)
        except ValueError:
            raise argparse.ArgumentTypeError(f"{text} is not a valid {self.kind}") from None
        if not self.accept(number):
            raise argparse.ArgumentTypeError(f"{number} is not {self.wanted}")
        return number


positive_int = _NumberArg(int, "integer", lambda n: n > 0, "a positive integer")
positive_number = _NumberArg(float, "number", lambda n: n > 0, "a positive number")
between_zero_and_one = _NumberArg(float, "number", lambda n: 0 <= n <= 1, "between 0 and 1")


def existed_file(path: Optional[str]) -> Optional[str]:
    """The path itself when it names a regular file (or is None)."""
    if path is None or os.path.isfile(path):
        return path
    raise argparse.ArgumentTypeError(f"{path} is not found")


def validate_stat_type(label: str) -> str:
    """A statistic label of the form letter + two digits: ``U05`` (U with x > 0.05), ``Q95`` (Q at the
    0.95 quantile).  Not used by the current commands; kept because the module's interface has it
    (argument_validation.py:140-169)."""
    letter, digits = label[:1], label[1:]
    if letter in ("U", "Q") and len(digits) == 2 and digits.isascii() and digits.isdigit():
        return label
    raise argparse.ArgumentTypeError(
        f"Invalid --stat-type: {label}. Must be 'UXX' or 'QXX' (e.g., 'U05' for x > 0.05, 'Q95' for quantile = 0.95)."
    )
