"""Build hook of the distribution (metadata: pyproject.toml).  Before the package files are collected,
`sai_amd._build.build()` compiles every translation unit for gfx950 and links sai_amd/lib/libsaihip.so in
the source tree (hipcc cross-compiles without a GPU), and the public header is copied next to the sources
so that an installed package can rebuild itself; an editable install (`pip install -e .`) runs the same
build and then imports straight from the tree."""

import shutil
import sys
from pathlib import Path

import setuptools
from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = Path(__file__).resolve().parent


def build_library() -> None:
    sys.path.insert(0, str(ROOT))
    try:
        from sai_amd import _build

        _build.build()
        dst = ROOT / "sai_amd" / "include"
        dst.mkdir(exist_ok=True)
        shutil.copy2(ROOT / "include" / "saihip.h", dst / "saihip.h")  # travels as package data
    finally:
        sys.path.remove(str(ROOT))


class BuildWithLibrary(build_py):
    def run(self):
        build_library()
        super().run()


class DevelopWithLibrary(develop):
    def run(self):
        build_library()
        super().run()


def metadata_for_old_setuptools() -> dict:
    """setuptools < 61 does not read [project] / [tool.setuptools]: hand it the same table."""
    if int(setuptools.__version__.split(".")[0]) >= 61:
        return {}
    try:
        import tomllib as toml
    except ImportError:
        import tomli as toml
    cfg = toml.loads((ROOT / "pyproject.toml").read_text())
    proj, tool = cfg["project"], cfg["tool"]["setuptools"]
    return dict(
        name=proj["name"], version=proj["version"], description=proj["description"], python_requires=proj["requires-python"],
        license=proj["license"]["text"], install_requires=proj["dependencies"], packages=tool["packages"],
        package_data=tool["package-data"], include_package_data=True,
        entry_points={"console_scripts": [f"{k} = {v}" for k, v in proj["scripts"].items()]},
        long_description=(ROOT / "README.md").read_text(), long_description_content_type="text/markdown",
    )  # fmt: skip


setup(cmdclass={"build_py": BuildWithLibrary, "develop": DevelopWithLibrary}, **metadata_for_old_setuptools())
