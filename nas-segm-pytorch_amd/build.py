"""Build recipe for libnasseg_hip.so (hipcc, gfx950 only, in-tree).

`python nas-segm-pytorch_amd/build.py` or `build_library()` compiles every
csrc/*.hip / *.cpp to an object under csrc/build/ and links them into
nas-segm-pytorch_amd/libnasseg_hip.so.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libnasseg_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]
# experiments only (tools/kbench_*.py variants): extra -D flags; objects are rebuilt by mtime, so touch
# the source that reads the macro
FLAGS += os.environ.get("NASSEG_EXTRA_FLAGS", "").split()


def sources():
    return sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip") or f.endswith(".cpp")
    )


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers_of(path, seen=None):
    """the csrc/*.h files a source includes, directly or through another header"""
    import re

    seen = set() if seen is None else seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        h = os.path.join(CSRC, name)
        if os.path.exists(h) and h not in seen:
            seen.add(h)
            _headers_of(h, seen)
    return seen


# sources whose entry points never touch activations: compiled once (fp32 build only)
FP32_ONLY_SOURCES = ("capi.cpp", "graph.cpp", "miou.hip", "optim.hip")


def _compile(job):
    """job = (source, bf16): every kernel source is compiled twice - activations stored as fp32
    (nasseg_<op>) and as bfloat16 (-DNASSEG_BF16, nasseg_bf16_<op>); see csrc/common.h."""
    src, bf16 = job
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + (".bf16.o" if bf16 else ".o"))
    deps = [src, os.path.abspath(__file__)] + sorted(_headers_of(src))
    if _stale(obj, deps):
        cmd = [HIPCC] + FLAGS + (["-DNASSEG_BF16"] if bf16 else []) + ["-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for {}:\n{}\n{}".format(src, r.stdout, r.stderr))
    return obj


def build_library(force=False, verbose=False):
    """Compile and link; returns the path of the shared library."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    srcs = sources()
    jobs = [(f, False) for f in srcs] + [(f, True) for f in srcs
                                         if os.path.basename(f) not in FP32_ONLY_SOURCES]
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(_compile, jobs))
    if force or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n{}\n{}".format(r.stdout, r.stderr))
    build_ffi(force=force, verbose=verbose)
    if verbose:
        print("built", LIB_PATH)
    return LIB_PATH


def build_ffi(force=False, verbose=False):
    """The native call shim (ffi_gen.py): generated from include/nasseg.h, compiled with the host C compiler.
    Optional - without it the calls go through ctypes (same library, same kernels, ~4 us more host time each)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_nasseg_ffi_gen", os.path.join(HERE, "ffi_gen.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    header = os.path.join(os.path.dirname(HERE), "include", "nasseg.h")
    try:
        path = gen.build(header, HERE, OBJ_DIR, force=force)
    except (RuntimeError, OSError) as e:  # (no compiler / headers: ctypes stays)
        sys.stderr.write("nasseg: call shim not built ({}); using ctypes\n".format(str(e).splitlines()[0]))
        return None
    if verbose and path:
        print("built", path)
    return path


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
