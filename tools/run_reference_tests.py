"""Run the REFERENCE's own test modules against sai_amd (build container only: the reference checkout
never travels).  `sai` is the distribution's alias package (every `sai.*` import resolves to the `sai_amd.*`
module object), imported here BEFORE the reference checkout's directory can shadow it, then pytest runs
the given test paths from the reference checkout (nothing is written there):

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tools/run_reference_tests.py \
        tests/parsers tests/registries tests/configs tests/test___main__.py \
        tests/multiprocessing/test_mp_pool.py tests/utils/test_unique_key_loader.py

Those modules need no GPU and no scikit-allel; tests/stats, tests/preprocessors, tests/generators
and tests/test_sai.py compute (GPU) or parse VCFs with scikit-allel inside the test itself."""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.dont_write_bytecode = True
import sai  # noqa  -- the distribution's own alias package (sai/__init__.py): sai.* IS sai_amd.*
import sai_amd  # noqa

assert sai.__file__.startswith(__import__("os").path.dirname(sai_amd.__path__[0])), "the reference's sai shadows the alias"
import pytest  # noqa

import tempfile  # noqa

sys.exit(pytest.main(["-q", "-p", "no:cacheprovider", f"--rootdir={tempfile.mkdtemp()}", *sys.argv[1:]]))
