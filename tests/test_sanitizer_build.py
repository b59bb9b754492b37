"""The host-side native code (VCF / BED / tabix reader, int8 narrowing, host generator) under
AddressSanitizer + UndefinedBehaviorSanitizer ("host ASan/UBSan build of the
C-ABI shim for the ctypes tests").  ``build(sanitize=True)`` compiles host_core.cpp +
vcf_ingest.cpp with g++ -fsanitize=address,undefined into libsaihost_san.so; the ingest tests
(awkward, damaged and truncated plain / gzip / bgzip files, synthetic and htslib-written tabix
indexes, the reference's fixtures) and the narrowing test then run against it in a child
interpreter that has the sanitizer runtimes preloaded.  CPU only: GPU sanitizers are not
available on this pool."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT

TARGETS = [
    "tests/test_ingest_native.py",
    "tests/test_ingest_stream.py",
    "tests/test_text_out.py",
    "tests/test_host_logic.py::test_to_int8_dosage",
    "tests/test_sanitizer_build.py::test_host_generator_equals_numpy_statement",
]


def _runtime(name: str) -> str:
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True, check=True).stdout.strip()
    return os.path.realpath(out)


def test_host_generator_equals_numpy_statement():
    """sai_synth_fill_host / sai_synth_gaps_host against the independent numpy definition (also one
    of the targets run under the sanitizers)."""
    import ctypes as C

    import numpy as np

    from oracle import synth_numpy as S
    from sai_amd import _ffi

    lib = _ffi.load_host()
    for seed, chrom, site0, n, stream, n_ind, ploidy, mpm in [(7, 1, 0, 300, 0, 9, 2, 0), (20260633, 3, 1234, 257, 1, 17, 2, 5000),
                                                               (5, 22, 10**6, 129, 2, 3, 4, 100000), (9, 2, 77, 64, 3, 1, 1, 0)]:  # fmt: skip
        out = np.empty((n, n_ind), dtype=np.int8)
        _ffi.check(lib.sai_synth_fill_host(seed, chrom, site0, n, stream, n_ind, ploidy, mpm, out.ctypes.data_as(C.c_void_p)), lib)
        assert np.array_equal(out, S.genotypes(seed, chrom, site0, n, stream, n_ind, ploidy, mpm))
        gaps = np.empty(n, dtype=np.int32)
        _ffi.check(lib.sai_synth_gaps_host(seed, chrom, site0, n, gaps.ctypes.data_as(C.c_void_p)), lib)
        assert np.array_equal(gaps, S.gaps(seed, chrom, site0, n))


@pytest.mark.skipif(os.environ.get("SAI_AMD_HOST_LIB") is not None, reason="already inside the sanitizer run")
def test_ingest_suite_is_clean_under_asan_ubsan(tmp_path):
    import __graft_entry__ as entry

    entry.build(sanitize=True)
    assert entry.SAN_LIB.exists()
    env = dict(os.environ)
    env.update(
        LD_PRELOAD=f"{_runtime('libasan.so')}:{_runtime('libubsan.so')}",
        # CPython keeps arenas alive by design: leak checking would only report the interpreter
        ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97:allocator_may_return_null=1",
        UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98",
        SAI_AMD_HOST_LIB=str(entry.SAN_LIB),
        PYTHONDONTWRITEBYTECODE="1",
    )
    res = subprocess.run(
        [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu", *TARGETS],
        cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900,
    )  # fmt: skip
    (tmp_path / "san.log").write_text(res.stdout + res.stderr)
    text = res.stdout + res.stderr
    assert "AddressSanitizer" not in text and "runtime error:" not in text, text[-4000:]
    assert res.returncode == 0, text[-4000:]
    assert " passed" in text
