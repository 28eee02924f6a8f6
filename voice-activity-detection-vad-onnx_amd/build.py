"""Build libvadx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per source, compiled in parallel (objects are cached under csrc/.obj and rebuilt only when the source or a
header is newer), then one link.  `build_test_hooks()` builds tests/hip/*.hip into tests/hip/libvadx_testhooks.so --
test-only entry points that are deliberately NOT part of the product ABI."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIB = os.path.join(HERE, "libvadx.so")
TEST_HOOKS_SRC = os.path.join(ROOT, "tests", "hip")
TEST_HOOKS_LIB = os.path.join(TEST_HOOKS_SRC, "libvadx_testhooks.so")
SOURCES = ["capi.hip", "silero.hip", "silero_split.hip", "silero_h2.hip", "frontend.hip", "fsmn.hip", "firered.hip", "marblenet.hip", "dfsmn.hip", "dfsmn_cfb.hip", "ingest.hip"]
# -fno-slp-vectorize: the SLP vectoriser is what turns pairs of scalar float adds / multiplies into v_pk_*_f32 with swizzled sources; the
# CROSS-swizzled form gives wrong sums inside silero_encode_h2_kernel (profiles/r06_pk_hazard.txt) and is banned from
# the product (tests/test_cabi_cpu.py::test_no_cross_swizzled_packed_f32_in_product_kernels).  A/B over the five BASELINE configs on one box
# (tools/ab_noslp.sh, twice each): 4.41 / 4.43 vs 4.41 / 4.46, 46.0 / 46.2 vs 45.6 / 45.4, 12.3 / 12.5 vs 12.2 / 12.4, 15.1 / 15.1 vs 15.1 / 15.4,
# 2090 / 2106 vs 2098 / 2085 ms -- no difference outside the run-to-run spread.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "vadx.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + _headers())


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _compile(src, obj, verbose):
    cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(lambda j: _compile(j[0], j[1], verbose), jobs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + \
          [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _write_abi()
    return LIB


def _write_abi():
    """The ABI number this library was compiled against, next to it (read by _lib when the header does not travel with the package)."""
    import re
    with open(os.path.join(ROOT, "include", "vadx.h")) as fh:
        n = int(re.search(r"^#define\s+VADX_ABI_VERSION\s+(\d+)\s*$", fh.read(), flags=re.M).group(1))
    with open(os.path.join(HERE, "_abi.py"), "w") as fh:
        fh.write(f'"""written by build.py: the VADX_ABI_VERSION of the include/vadx.h libvadx.so was built from"""\nABI_VERSION = {n}\n')


def build_test_hooks(force=False, verbose=True):
    srcs = sorted(os.path.join(TEST_HOOKS_SRC, f) for f in os.listdir(TEST_HOOKS_SRC) if f.endswith(".hip"))
    if not force and not _stale(TEST_HOOKS_LIB, srcs + _headers()):
        return TEST_HOOKS_LIB
    cmd = [_hipcc()] + FLAGS + ["-shared"] + srcs + ["-o", TEST_HOOKS_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TEST_HOOKS_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_test_hooks(force="--force" in sys.argv)
