"""``populations`` section of the YAML config: group -> sample-list file.  Interface of
sai/configs/pop_config.py:31-82: groups ref, tgt and src are required, outgroup is optional, any
other key is refused, every file must exist; ``get_population("outgroup")`` is None when no
outgroup is configured."""

from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

from pydantic import RootModel, model_validator

_GROUPS = {"ref": True, "tgt": True, "src": True, "outgroup": False}  # group -> required?


class PopConfig(RootModel[Dict[str, str]]):
    @model_validator(mode="after")
    def _groups_and_files(self) -> "PopConfig":
        given = self.root
        absent = sorted(g for g, required in _GROUPS.items() if required and g not in given)
        if absent:
            raise ValueError(f"Missing required population keys: {absent}")
        unknown = sorted(set(given) - set(_GROUPS))
        if unknown:
            raise ValueError(f"Unsupported population keys: {unknown}")
        for group, file_name in given.items():
            if not Path(file_name).is_file():
                raise ValueError(f"{group} file does not exist: {file_name}")
        return self

    def get_population(self, group: str) -> Optional[str]:
        if group in self.root:
            return self.root[group]
        if _GROUPS.get(group) is False:  # an optional group that was left out
            return None
        raise ValueError(f"Population group '{group}' not found in config.")
