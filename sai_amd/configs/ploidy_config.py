"""Ploidy section of the YAML config (mirror of sai/configs/ploidy_config.py:25-96)."""

from __future__ import annotations

from typing import Dict, Union

from pydantic import RootModel, field_validator

_ALLOWED = {"ref", "tgt", "src", "outgroup"}
_REQUIRED = {"ref", "tgt", "src"}


class PloidyConfig(RootModel[Dict[str, Dict[str, int]]]):
    """``{"ref": {pop: ploidy}, "tgt": {...}, "src": {...}[, "outgroup": {...}]}``."""

    @field_validator("root")
    def validate_ploidy_dict(cls, v):
        if extra := set(v.keys()) - _ALLOWED:
            raise ValueError(f"Unsupported ploidy keys: {extra}. Allowed keys are {_ALLOWED}.")
        if missing := _REQUIRED - set(v.keys()):
            raise ValueError(f"Missing required ploidy keys: {missing}.")
        for group, pops in v.items():
            if not isinstance(pops, dict):
                raise ValueError(f"Value for '{group}' must be a dictionary of population -> ploidy.")
            for pop, ploidy in pops.items():
                if not isinstance(ploidy, int) or ploidy <= 0:
                    raise ValueError(f"Ploidy for '{group}:{pop}' must be a positive integer.")
        return v

    def get_ploidy(self, group: str, population: str = None) -> Union[int, list[int], None]:
        """One population's ploidy, or all of a group's in config order (ploidy_config.py:66-96)."""
        if group not in self.root:
            if group == "outgroup":
                return None
            raise KeyError(f"Group '{group}' not found in configuration.")
        if population is None:
            return list(self.root[group].values())
        if population not in self.root[group]:
            raise KeyError(f"Population '{population}' not found under group '{group}'.")
        return self.root[group][population]
