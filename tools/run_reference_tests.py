"""Run the REFERENCE's own test modules against sai_amd (build container only: the reference checkout
never travels).  `sai` and every `sai.*` import are aliased to `sai_amd` / `sai_amd.*`, then pytest
runs the given test paths from the reference checkout (nothing is written there):

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tools/run_reference_tests.py \
        tests/parsers tests/registries tests/configs tests/test___main__.py \
        tests/multiprocessing/test_mp_pool.py tests/utils/test_unique_key_loader.py

Those modules need no GPU and no scikit-allel; tests/stats, tests/preprocessors, tests/generators
and tests/test_sai.py compute (GPU) or parse VCFs with scikit-allel inside the test itself."""
import importlib
import importlib.machinery
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.dont_write_bytecode = True
import sai_amd  # noqa


class Loader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


class Alias:
    def find_spec(self, name, path=None, target=None):
        if name == "sai" or name.startswith("sai."):
            try:
                mod = importlib.import_module("sai_amd" + name[3:])
            except ImportError:
                return None
            return importlib.machinery.ModuleSpec(name, Loader(mod), is_package=hasattr(mod, "__path__"))
        return None


sys.meta_path.insert(0, Alias())
import pytest  # noqa

import tempfile  # noqa

sys.exit(pytest.main(["-q", "-p", "no:cacheprovider", f"--rootdir={tempfile.mkdtemp()}", *sys.argv[1:]]))
