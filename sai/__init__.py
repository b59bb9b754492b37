"""``sai`` -- the reference's import name, served by the MI355X build.

``import sai``, ``import sai.stats``, ``from sai.registries.stat_registry import STAT_REGISTRY`` ... resolve to
the SAME module objects as ``sai_amd`` / ``sai_amd.stats`` / ... (one registry, one set of classes), so code
written against xin-huang/sai's plugin surface (sai/__init__.py, sai/stats/__init__.py:20-30,
sai/__main__.py:64-76) runs unchanged on the GPU path.  This package stands in for sai-pg's top-level
package: install one distribution or the other.
"""

from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import sys

import sai_amd
from sai_amd import __version__  # noqa: F401

_PREFIX = __name__ + "."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self.module = module

    def create_module(self, spec):
        return self.module

    def exec_module(self, module):  # already executed under its own name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if not name.startswith(_PREFIX):
            return None
        try:
            module = importlib.import_module(sai_amd.__name__ + name[len(__name__) :])
        except ModuleNotFoundError as exc:
            if exc.name and exc.name.startswith(sai_amd.__name__):
                return None  # no such module in the build: the import fails as a missing `sai.*` module
            raise
        return importlib.machinery.ModuleSpec(name, _AliasLoader(module), is_package=hasattr(module, "__path__"))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())


def __getattr__(attr: str):
    """Sub-packages as attributes (``sai.stats`` after a plain ``import sai``) and anything sai_amd exports."""
    try:
        return importlib.import_module(_PREFIX + attr)
    except ModuleNotFoundError:
        pass
    try:
        return getattr(sai_amd, attr)
    except AttributeError:
        raise AttributeError(f"module {__name__!r} has no attribute {attr!r}") from None
