"""Top-level YAML config (mirror of sai/configs/global_config.py:29-99)."""

from __future__ import annotations

from pydantic import BaseModel, model_validator

from .ploidy_config import PloidyConfig
from .pop_config import PopConfig
from .stat_config import StatConfig


class GlobalConfig(BaseModel):
    statistics: StatConfig
    ploidies: PloidyConfig
    populations: PopConfig

    @model_validator(mode="before")
    @classmethod
    def _check_required(cls, data):
        if isinstance(data, dict):
            if missing := sorted({"statistics", "ploidies", "populations"} - set(data.keys())):
                raise ValueError(f"Missing required fields in configuration: {', '.join(missing)}")
        return data

    @model_validator(mode="after")
    def validate_population_in_ploidies(self):
        """Every population named by U/Q must have a ploidy (global_config.py:46-67)."""
        for stat_name, params in self.statistics.root.items():
            if stat_name not in ("U", "Q"):
                continue
            for group in ("ref", "tgt", "src"):
                for pop in params.get(group, {}):
                    if pop not in self.ploidies.root.get(group, {}):
                        raise ValueError(
                            f"Population '{pop}' used in statistics[{stat_name}][{group}] "
                            f"is not defined in ploidies[{group}]"
                        )
        return self

    @model_validator(mode="after")
    def validate_population_in_populations(self):
        """... and must appear in the group's sample file (global_config.py:69-99)."""
        from ..utils import parse_ind_file

        cats = {g: set(parse_ind_file(path).keys()) for g, path in self.populations.root.items()}
        for stat_name, params in self.statistics.root.items():
            if stat_name not in ("U", "Q"):
                continue
            for group in ("ref", "tgt", "src"):
                for pop in params.get(group, {}):
                    if pop not in cats.get(group, set()):
                        raise ValueError(
                            f"Population '{pop}' used in statistics[{stat_name}][{group}] "
                            f"is not found in the population file for group '{group}'."
                        )
        return self
