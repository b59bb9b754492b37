"""Building libsaihip.so in-tree: every translation unit of sai_amd/csrc for gfx950 (hipcc) or the host
(g++), linked into sai_amd/lib.  Used by ``__graft_entry__.build()``, by the rank launcher (once, before
N ranks would each find the tree stale) and by ``setup.py``'s build step; needs no GPU."""

from __future__ import annotations

import fcntl
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
# the public header: include/ at the repository root, or the copy an installed package carries
INCLUDE = ROOT / "include" if (ROOT / "include" / "saihip.h").exists() else PKG / "include"
HOST_UNITS = ["host_core.cpp", "vcf_ingest.cpp", "vcf_stream.cpp", "bgzf_stream.cpp", "narrow.cpp", "text_out.cpp"]  # plain C++: also built alone under the sanitizers
UNITS = ["core.hip", "site_pass.hip", "site_pass_dd.hip", "packed2.hip", "windows.hip", "single_window.hip", "plan.hip", "fourpop.hip", "dd.hip", "synth.hip", "tokenize.hip", "inflate.hip", "lines.hip", *HOST_UNITS]
LIB = PKG / "lib" / "libsaihip.so"
SAN_LIB = LIB.parent / "libsaihost_san.so"
OBJ = LIB.parent / "obj"
HIPCC_FLAGS = [
    "-O3",
    "--offload-arch=gfx950",
    "-std=c++17",
    "-ffp-contract=off",  # f64 must round like numpy: no fused multiply-add anywhere
    "-fPIC",
]
HOST_FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread"]
SAN_FLAGS = ["-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-omit-frame-pointer",
             "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]  # fmt: skip


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def _run(cmd) -> None:
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{' '.join(map(str, cmd))}\n{res.stdout}{res.stderr}")


def _link(cmd_without_output: list, target: Path) -> None:
    """Link to a scratch name and rename into place: a process that is loading the library never
    sees a half-written file."""
    tmp = target.with_name(f".{target.name}.{os.getpid()}.tmp")
    try:
        _run([*cmd_without_output, "-o", str(tmp)])
        os.replace(tmp, target)
    finally:
        if tmp.exists():
            tmp.unlink()


def shipped_library_is_current() -> bool:
    """An INSTALLED package (pip / a wheel) carries libsaihip.so and the sources but no object directory: every
    unit would look stale, and a rebuild would need hipcc and a writable site-packages -- neither is a given
    where the package was merely installed.  Such a library is taken as it is when it loads with the ABI this
    package expects; a source tree (lib/obj exists) is always checked unit by unit."""
    if OBJ.exists() or not LIB.exists():
        return False
    try:
        from . import _ffi

        _ffi.load()
        return True
    except Exception:  # noqa: BLE001 - anything wrong with it: let the build say what
        return False


def build(force: bool = False, sanitize: bool = False) -> None:
    """Compile every translation unit for gfx950 (in parallel, only the stale ones), link them
    in-tree into libsaihip.so and import the package.  Safe to call from several processes at once
    (the ranks of a torchrun job): they serialise on a file lock and the later ones find the tree
    current.  ``sanitize=True`` additionally builds the host-only part (the VCF / BED reader, the
    int8 narrowing, the host generator) with g++ -fsanitize=address,undefined into
    libsaihost_san.so for the CPU test suite (tests/test_sanitizer_build.py).  The compilers are looked
    for only when something has to be compiled; the library an installed package ships is not rebuilt
    (``shipped_library_is_current``)."""
    from concurrent.futures import ThreadPoolExecutor

    if not force and not sanitize and shipped_library_is_current():
        return
    OBJ.mkdir(parents=True, exist_ok=True)
    with open(LIB.parent / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        headers = [INCLUDE / "saihip.h", *sorted(CSRC.glob("*.hpp"))]
        hipcc = gxx = None
        jobs = []
        for unit in UNITS:
            src, obj = CSRC / unit, OBJ / (Path(unit).stem + ".o")
            if force or _stale(obj, [src, *headers]):
                hipcc = hipcc or _hipcc()  # only now: a current tree builds nothing and needs no compiler
                gxx = gxx or shutil.which("g++") or "g++"
                if unit in HOST_UNITS:  # plain C++, the same sources the sanitizer build compiles
                    jobs.append([gxx, *HOST_FLAGS, f"-I{INCLUDE}", "-c", str(src), "-o", str(obj)])
                else:
                    jobs.append([hipcc, *HIPCC_FLAGS, f"-I{INCLUDE}", "-c", str(src), "-o", str(obj)])
        if jobs:
            with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
                list(pool.map(_run, jobs))
        objs = [OBJ / (Path(u).stem + ".o") for u in UNITS]
        if force or _stale(LIB, objs):
            _link([hipcc or _hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-lz", "-lpthread", "-ldl"], LIB)
        if sanitize:
            srcs = [CSRC / u for u in HOST_UNITS]
            if force or _stale(SAN_LIB, [*srcs, *headers]):
                _link([gxx or shutil.which("g++") or "g++", *SAN_FLAGS, f"-I{INCLUDE}", "-shared", *map(str, srcs), "-lz", "-lpthread", "-ldl"], SAN_LIB)
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import sai_amd  # noqa: F401
    import sai_amd.stats  # noqa: F401
    from sai_amd import _ffi

    _ffi.load()  # every symbol of include/saihip.h resolves
