"""Natural ordering of score tables (mirror of sai/utils/utils.py:615-646, which delegates to the
natsort package; the ordering is restated here so the build has no extra dependency)."""

from __future__ import annotations

import re

import pandas as pd

_DIGITS = re.compile(r"(\d+)")


def natural_key(value):
    """Key under which "2" < "10" < "X": numbers compare numerically, text chunk-wise with
    embedded digit runs as integers (natsort's default behaviour for these tables)."""
    if isinstance(value, (int, float)) and not isinstance(value, bool):
        return (0, value)
    parts = _DIGITS.split(str(value))
    return (1, tuple(int(p) if i % 2 else p for i, p in enumerate(parts)))


def natsorted_df(df: pd.DataFrame) -> pd.DataFrame:
    """Rows ordered naturally by (Chrom, Start, End); Start/End are cast to int (utils.py:637-638);
    a missing column is a ValueError naming it (:633-634)."""
    required = {"Chrom", "Start", "End"}
    if missing := required - set(df.columns):
        raise ValueError(f"Missing required columns: {', '.join(missing)}")
    df["Start"] = df["Start"].astype(int)
    df["End"] = df["End"].astype(int)
    order = sorted(
        df.index, key=lambda i: (natural_key(df.at[i, "Chrom"]), natural_key(df.at[i, "Start"]), natural_key(df.at[i, "End"]))
    )
    return df.loc[order].reset_index(drop=True)
