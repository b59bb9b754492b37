"""YAML loader that rejects duplicate mapping keys (mirror of
sai/utils/unique_key_loader.py:26-72)."""

from __future__ import annotations

import yaml


class UniqueKeyLoader(yaml.SafeLoader):
    """SafeLoader whose mappings raise ``ValueError("Duplicate key in YAML: ...")``."""


def _mapping_without_duplicates(loader, node, deep=False):
    out = {}
    for key_node, value_node in node.value:
        key = loader.construct_object(key_node, deep=deep)
        if key in out:
            raise ValueError(f"Duplicate key in YAML: {key!r}")
        out[key] = loader.construct_object(value_node, deep=deep)
    return out


UniqueKeyLoader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG, _mapping_without_duplicates)
