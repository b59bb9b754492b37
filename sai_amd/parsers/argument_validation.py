"""argparse type validators (mirror of sai/parsers/argument_validation.py:26-169, the ones the
score command uses)."""

from __future__ import annotations

import argparse
import os


def positive_int(value: str) -> int:
    if value is not None:
        try:
            value = int(value)
        except ValueError:
            raise argparse.ArgumentTypeError(f"{value} is not a valid integer")
        if value <= 0:
            raise argparse.ArgumentTypeError(f"{value} is not a positive integer")
    return value


def positive_number(value: str) -> float:
    if value is not None:
        try:
            value = float(value)
        except ValueError:
            raise argparse.ArgumentTypeError(f"{value} is not a valid number")
        if value <= 0:
            raise argparse.ArgumentTypeError(f"{value} is not a positive number")
    return value


def between_zero_and_one(value: str) -> float:
    if value is not None:
        try:
            value = float(value)
        except ValueError:
            raise argparse.ArgumentTypeError(f"{value} is not a valid number")
        if not (0 <= value <= 1):
            raise argparse.ArgumentTypeError(f"{value} is not between 0 and 1")
    return value


def existed_file(value: str) -> str:
    if value is not None and not os.path.isfile(value):
        raise argparse.ArgumentTypeError(f"{value} is not found")
    return value
