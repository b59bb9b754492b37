"""The distribution as a drop-in (reference: pyproject.toml:31-39, sai/__main__.py:64-76):
`sai` console script, `sai` import alias, the library built by the install and shipped as package data."""

import os
import subprocess
import sys

import pytest
from conftest import ROOT


def _project():
    try:
        import tomllib as toml
    except ImportError:
        import tomli as toml
    return toml.loads((ROOT / "pyproject.toml").read_text())


def test_entry_point_metadata_names_the_reference_command():
    import importlib

    proj = _project()["project"]
    assert proj["scripts"] == {"sai": "sai_amd.__main__:main"}  # reference: sai = "sai.__main__:main"
    module, func = proj["scripts"]["sai"].split(":")
    assert callable(getattr(importlib.import_module(module), func))
    import sai_amd

    assert proj["version"] == sai_amd.__version__
    pkgs = _project()["tool"]["setuptools"]["packages"]
    on_disk = sorted(str(p.parent.relative_to(ROOT)).replace(os.sep, ".") for p in ROOT.glob("sai_amd/**/__init__.py"))
    assert sorted(p for p in pkgs if p.startswith("sai_amd")) == on_disk and "sai" in pkgs


def test_alias_package_serves_the_same_module_objects():
    code = (
        "import sai, sai.stats, sai_amd.stats, sai.sai\n"
        "from sai.registries.stat_registry import STAT_REGISTRY\n"
        "from sai.stats import UStatistic, QStatistic\n"
        "import sai_amd.registries.stat_registry as R\n"
        "assert sai.stats is sai_amd.stats and STAT_REGISTRY is R.STAT_REGISTRY\n"
        "assert STAT_REGISTRY.get('U') is UStatistic and sai.sai.score is sai_amd.sai.score\n"
        "assert sai.__version__ == sai_amd.__version__ and sai.configs.GlobalConfig is sai_amd.configs.GlobalConfig\n"
        "try:\n    import sai.no_such_module\nexcept ModuleNotFoundError as e:\n    assert 'sai.no_such_module' in str(e)\nelse:\n    raise SystemExit('no error')\n"
        "from sai.__main__ import main\n"
    )
    res = subprocess.run([sys.executable, "-c", code], cwd="/", env={**os.environ, "PYTHONPATH": str(ROOT)}, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]


def test_install_builds_the_library_and_the_sai_command_prints_the_reference_flags(tmp_path):
    """`pip install .` into a scratch prefix (no index, no build isolation: the container has no network): the
    build hook compiles libsaihip.so, the wheel carries it with the sources and the header, and the installed
    `sai` script answers `score --help` with the reference's flags from outside the source tree."""
    target = tmp_path / "site"
    res = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index", "--quiet",
                          "--target", str(target), str(ROOT)], capture_output=True, text=True, timeout=1500)  # fmt: skip
    import shutil

    for left in (ROOT / "build", ROOT / "sai_amd.egg-info"):  # pip builds in the tree: leave it as it was
        shutil.rmtree(left, ignore_errors=True)
    if res.returncode != 0 and "No module named pip" in res.stderr:
        pytest.skip("pip is not available")
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    assert (target / "sai_amd" / "lib" / "libsaihip.so").exists() and (target / "sai_amd" / "include" / "saihip.h").exists()
    assert (target / "sai_amd" / "csrc" / "windows.hip").exists() and (target / "sai" / "__init__.py").exists()
    script = target / "bin" / "sai"
    assert script.exists()
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = str(target)
    res = subprocess.run([sys.executable, str(script), "score", "--help"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    for flag in ("--vcf", "--chr-name", "--win-len", "--win-step", "--anc-alleles", "--output", "--config", "--num-workers"):
        assert flag in res.stdout, flag
    # the installed copy loads ITS library (every symbol of the header) and knows where its sources are
    code = ("import sai_amd, sai_amd._ffi as f, sai_amd._build as b, sai.stats; lib = f.load(); "
            f"assert sai_amd.__file__.startswith({str(target)!r}), sai_amd.__file__; "
            "assert str(f.LIB_PATH).startswith(sai_amd.__path__[0]) and b.INCLUDE.name == 'include' and (b.INCLUDE / 'saihip.h').exists(); "
            "print(lib.sai_abi_version())")  # fmt: skip
    res = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip() == str(__import__("sai_amd._ffi", fromlist=["x"]).SAI_ABI_VERSION), res.stderr[-2000:]
    res = subprocess.run([sys.executable, str(script), "--version"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip() == _project()["project"]["version"]

    # the multi-worker route from the INSTALLED tree, on a machine without hipcc and with a read-only package
    # directory: the launcher's "build once" must take the shipped library as it is, not look for a
    # compiler and recompile every unit into site-packages because the wheel has no object directory
    rank = tmp_path / "rank.py"
    rank.write_text("import os, torch.distributed as dist\ndist.init_process_group('gloo')\n"
                    "print('rank', dist.get_rank(), flush=True)\ndist.destroy_process_group()\n")
    code = ("import sai_amd._build as b, sai_amd.launcher as L, sys; assert b.shipped_library_is_current(); "
            f"sys.exit(L.launch_ranks(2, [], script={str(rank)!r}))")
    no_hipcc = {**env, "PATH": "/usr/bin:/bin", "HIPCC": "/nonexistent/hipcc"}
    lib_dir = target / "sai_amd" / "lib"
    before = sorted(f.name for f in lib_dir.iterdir())
    os.chmod(lib_dir, 0o555)
    try:
        res = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=no_hipcc, capture_output=True, text=True, timeout=300)
    finally:
        os.chmod(lib_dir, 0o755)
    assert res.returncode == 0, res.stderr[-2000:]
    assert sorted(ln for ln in res.stdout.splitlines() if ln.startswith("rank")) == ["rank 0", "rank 1"]
    assert sorted(f.name for f in lib_dir.iterdir()) == before  # nothing was written next to the library
