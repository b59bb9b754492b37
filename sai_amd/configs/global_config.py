"""Top-level YAML config: the three sections and the cross-checks between them
(behaviour of sai/configs/global_config.py:29-99, messages included; own structure)."""

from __future__ import annotations

from typing import Callable, Iterator, NamedTuple

from pydantic import BaseModel, model_validator

from .ploidy_config import PloidyConfig
from .pop_config import PopConfig
from .stat_config import StatConfig

SECTIONS = ("statistics", "ploidies", "populations")
CHECKED_STATS = ("U", "Q")  # the statistics whose populations are named in the YAML
GROUPS = ("ref", "tgt", "src")


class _CrossCheck(NamedTuple):
    """Where the populations a statistic names must also be known, and what to say otherwise."""

    known: Callable[["GlobalConfig"], dict]  # config -> {group: collection of population names}
    complaint: str  # formatted with pop, stat, group


def _ploidy_names(cfg: "GlobalConfig") -> dict:
    return {group: pops.keys() for group, pops in cfg.ploidies.root.items()}


def _sample_file_names(cfg: "GlobalConfig") -> dict:
    from ..utils import parse_ind_file

    return {group: parse_ind_file(path).keys() for group, path in cfg.populations.root.items()}


# in the order the reference reports them: a population without a ploidy first (global_config.py:46-67),
# then one that is missing from its group's sample file (:69-99)
CROSS_CHECKS = (
    _CrossCheck(_ploidy_names, "Population '{pop}' used in statistics[{stat}][{group}] is not defined in ploidies[{group}]"),
    _CrossCheck(_sample_file_names,
                "Population '{pop}' used in statistics[{stat}][{group}] is not found in the population file for group '{group}'."),
)  # fmt: skip


class GlobalConfig(BaseModel):
    statistics: StatConfig
    ploidies: PloidyConfig
    populations: PopConfig

    @model_validator(mode="before")
    @classmethod
    def _sections_present(cls, raw):
        if isinstance(raw, dict):
            absent = [name for name in sorted(SECTIONS) if name not in raw]
            if absent:
                raise ValueError("Missing required fields in configuration: " + ", ".join(absent))
        return raw

    def _named_populations(self) -> Iterator[tuple]:
        """(statistic, group, population) for every population a checked statistic names."""
        for stat, params in self.statistics.root.items():
            if stat in CHECKED_STATS:
                for group in GROUPS:
                    for pop in params.get(group, {}):
                        yield stat, group, pop

    @model_validator(mode="after")
    def _populations_known_everywhere(self):
        named = list(self._named_populations())
        for check in CROSS_CHECKS:
            known = check.known(self)
            for stat, group, pop in named:
                if pop not in known.get(group, ()):
                    raise ValueError(check.complaint.format(pop=pop, stat=stat, group=group))
        return self
