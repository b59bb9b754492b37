"""Population-file section of the YAML config (mirror of sai/configs/pop_config.py:31-82)."""

from __future__ import annotations

import os
from typing import Dict

from pydantic import RootModel, field_validator

REQUIRED_KEYS = {"ref", "tgt", "src"}
OPTIONAL_KEYS = {"outgroup"}
ALLOWED_KEYS = REQUIRED_KEYS | OPTIONAL_KEYS


class PopConfig(RootModel[Dict[str, str]]):
    """``{"ref": path, "tgt": path, "src": path[, "outgroup": path]}``; every file must exist."""

    @field_validator("root")
    def validate_population_keys_and_paths(cls, v):
        keys = set(v.keys())
        if missing := REQUIRED_KEYS - keys:
            raise ValueError(f"Missing required population keys: {missing}")
        if invalid := keys - ALLOWED_KEYS:
            raise ValueError(f"Unsupported population keys: {invalid}")
        for name, path in v.items():
            if not os.path.isfile(path):
                raise ValueError(f"{name} file does not exist: {path}")
        return v

    def get_population(self, group: str) -> str:
        if group not in self.root:
            if group == "outgroup":
                return None
            raise ValueError(f"Population group '{group}' not found in config.")
        return self.root[group]
