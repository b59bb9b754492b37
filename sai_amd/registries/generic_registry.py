"""Name -> component table behind ``STAT_REGISTRY``.  Interface of
sai/registries/generic_registry.py:25-89: ``register(name)`` as a class decorator, ``get(name)``,
``list_registered()``; registering a name twice is a ValueError (:58-59), asking for an unknown
one a KeyError (:76-77)."""

from __future__ import annotations

from abc import ABC
from typing import Any, Callable, Iterator


class GenericRegistry(ABC):
    def __init__(self):
        self._components: dict[str, Any] = {}

    def register(self, name: str) -> Callable[[Any], Any]:
        """``@registry.register("U")`` files the decorated object under ``name`` and hands it back
        unchanged."""

        def file_under_name(component: Any) -> Any:
            if name in self._components:  # also when it is the very same object again
                raise ValueError(f"{name!r} is already registered.")
            self._components[name] = component
            return component

        return file_under_name

    def get(self, name: str) -> Any:
        try:
            return self._components[name]
        except KeyError:
            raise KeyError(f"No component registered under name '{name}'") from None

    def list_registered(self) -> list[str]:
        return [*self._components]

    def __contains__(self, name: str) -> bool:
        return name in self._components

    def __iter__(self) -> Iterator[str]:
        return iter(self._components)
