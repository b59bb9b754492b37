"""Name -> component registry (mirror of sai/registries/generic_registry.py:25-89)."""

from __future__ import annotations

from abc import ABC
from typing import Any, Callable


class GenericRegistry(ABC):
    """``register(name)`` decorator, ``get(name)``, ``list_registered()``; a duplicate name is a
    ValueError (generic_registry.py:58-59), an unknown one a KeyError (:76-77)."""

    def __init__(self):
        self._registry: dict[str, Any] = {}

    def register(self, name: str) -> Callable:
        def decorator(obj: Any) -> Any:
            self._register(name, obj)
            return obj

        return decorator

    def _register(self, name: str, obj: Any) -> None:
        if name in self._registry:
            raise ValueError(f"{name!r} is already registered.")
        self._registry[name] = obj

    def get(self, name: str) -> Any:
        if name not in self._registry:
            raise KeyError(f"No component registered under name '{name}'")
        return self._registry[name]

    def list_registered(self) -> list[str]:
        return list(self._registry.keys())
