"""Statistics section of the YAML config (mirror of sai/configs/stat_config.py).

Only U and Q are computed by this build; the other names of the reference are recognised so
that a config can say ``fd: False`` etc. without being rejected, but enabling one is reported
as outside the HIP path when ``score`` runs.
"""

from __future__ import annotations

from typing import Dict, Optional, Union

from pydantic import RootModel, field_validator

SUPPORTED_STATISTICS = ["Danc", "DD", "df", "Dplus", "fd", "U", "Q"]  # stat_config.py:20-28
COMPARATORS = ["<=", ">=", "=", "<", ">"]  # search order of stat_config.py:188


class StatConfig(RootModel[Dict[str, Union[bool, Dict[str, Dict[str, Union[float, str]]]]]]):
    """``{"U": {"ref": {pop: w}, "tgt": {pop: x}, "src": {pop: "=1"}}, "Q": {...}, "fd": bool}``.
    Validation rewrites every ``src`` comparator string into an ``(op, float)`` tuple in place,
    as stat_config.py:147-157 does, so ``get_parameters`` hands out tuples."""

    @field_validator("root")
    def check_valid_stat_types(cls, v):
        for name, params in v.items():
            if name not in SUPPORTED_STATISTICS:
                raise ValueError(f"The {name} statistic is not supported.")
            if name in ("U", "Q"):
                cls.check_range_for_u_q(name, params)
        return v

    @staticmethod
    def check_range_for_u_q(stat_name: str, params) -> None:
        if not isinstance(params, dict):
            raise ValueError(f"{stat_name} must map ref/tgt/src to population thresholds.")
        required = {"ref", "tgt", "src"}
        if set(params.keys()) != required:
            raise ValueError(
                f"{stat_name} must have exactly the keys: {required}, but got {set(params.keys())}."
            )
        for group in ("ref", "tgt"):
            for pop, value in params[group].items():
                num = float(value)
                if not (0 <= num <= 1):
                    raise ValueError(
                        f"{group}[{pop}] value must be between 0 and 1 for {stat_name}, got {value}."
                    )
        parsed: Dict[str, tuple[str, float]] = {}
        for pop, expr in params["src"].items():
            if not isinstance(expr, str):
                raise ValueError(f"src[{pop}] value must be a comparator string for {stat_name}.")
            parsed[pop] = StatConfig.check_comparator(expr, stat_name, f"src[{pop}]")
        params["src"] = parsed

    @staticmethod
    def check_comparator(value: str, stat_name: str, param: str) -> tuple[str, float]:
        """``">=0.2"`` -> ``(">=", 0.2)`` (stat_config.py:159-207)."""
        op = next((c for c in COMPARATORS if c in value), None)
        if op is None:
            raise ValueError(
                f"{param} for {stat_name} must contain a valid comparator (e.g., '=0.5', '>=0.2')."
            )
        try:
            num = float(value[len(op):])
        except ValueError:
            raise ValueError(f"{param} value for {stat_name} must be a valid number after the comparator.")
        if not (0 <= num <= 1):
            raise ValueError(f"{param} value must be between 0 and 1 for {stat_name}, but got {num}.")
        return op, num

    def get_parameters(self, stat_name: str) -> Optional[Union[bool, dict]]:
        return self.root.get(stat_name, None)
