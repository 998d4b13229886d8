"""ctypes binding of libnasseg_hip.so (the C ABI declared in include/nasseg.h).

Prototypes are parsed from the header so the header stays the single source of
truth.  Every call that returns a negative status raises RuntimeError - the
error class the reference's ``try_except`` wrapper (src/helpers/utils.py:172-187)
converts into reward 0.  There is no CPU fallback: if the shared library is
missing, loading fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NASSEG_LIB: another build of the same ABI, for A/B measurements of two libraries on one box)
LIB_PATH = os.environ.get("NASSEG_LIB") or os.path.join(_HERE, "libnasseg_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nasseg.h")

_CTYPE = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}
_PROTO = re.compile(r"^\s*(const char\*|int64_t|int)\s+(nasseg_\w+)\s*\(([^)]*)\)\s*;", re.M | re.S)


def parse_header(path=HEADER_PATH):
    """Return {symbol: (restype, [argtypes])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for ret, name, args in _PROTO.findall(text):
        args = " ".join(args.split())
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const ", "").split()[0]
                    argtypes.append(_CTYPE[base])
        restype = {"const char*": ctypes.c_char_p, "int64_t": ctypes.c_int64, "int": ctypes.c_int}[ret]
        protos[name] = (restype, argtypes)
    return protos


def pointer_access(path=HEADER_PATH):
    """{symbol: [(argument index, "r" | "w" | "t")]} for the pointer arguments of every prototype except `stream`:
    the header's const-ness IS the contract - a `const T*` is only read, a `T*` may be written (engine/graph_dag.py
    derives the dependencies between the launches of a recorded step from it); "t": a table of pointers."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for _, name, args in _PROTO.findall(text):
        args = " ".join(args.split())
        acc = []
        if args and args != "void":
            for i, a in enumerate(args.split(",")):
                a = a.strip()
                if "*" not in a or a[a.rindex("*") + 1:].strip() == "stream":
                    continue
                acc.append((i, "t" if a.count("*") > 1 else ("r" if a.startswith("const ") else "w")))
        out[name] = acc
    return out


def _address_of(obj):
    """Address behind a ctypes array / pointer / c_void_p (the call shim's slow path; ints and None never get
    here)."""
    if isinstance(obj, ctypes.Array):
        return ctypes.addressof(obj)
    if not isinstance(obj, (ctypes._SimpleCData, ctypes._Pointer)):
        raise TypeError("address must be an int, None or a ctypes array / pointer, not {}".format(type(obj).__name__))
    value = getattr(obj, "value", None)  # c_void_p and friends
    if isinstance(value, int):
        return value
    if value is None and isinstance(obj, ctypes.c_void_p):
        return 0
    return ctypes.cast(obj, ctypes.c_void_p).value or 0


class NassegError(RuntimeError):
    """A nasseg C-ABI call failed (bad arguments, HIP launch failure, ...)."""


class LaunchProfiler(object):
    """Optional per-entry-point timing with HIP events on the launch stream.

    While installed (``lib.profiler = LaunchProfiler()``) every status-returning
    call is bracketed by a pair of events recorded on the stream the kernels
    are launched on (torch's current stream, or a registered second one).  ``summary()`` synchronises and
    returns {entry point: (launches, total ms, [per-launch (ms, args)])}.
    Used by bench.py for the roofline figures; off by default (zero overhead).
    """

    streams = {}  # raw handle -> torch stream, for launches that go to another stream than the current one
    #               (functional._wgrad_stream registers its second stream here)

    def __init__(self, names=None):
        self.names = set(names) if names else None
        self.records = []

    def wants(self, name):
        return self.names is None or name in self.names

    def bracket(self, name, args, fn):
        import torch

        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        on = self.streams.get(args[-1]) if args and isinstance(args[-1], int) else None
        if on is None:
            on = torch.cuda.current_stream()
        e0.record(on)
        rc = fn(*args)
        e1.record(on)
        self.records.append((name, args, e0, e1))
        return rc

    def summary(self):
        import torch

        torch.cuda.synchronize()
        out = {}
        for name, args, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            ent = out.setdefault(name, [0, 0.0, []])
            ent[0] += 1
            ent[1] += ms
            ent[2].append((ms, args))
        return out


class _Library(object):
    def __init__(self):
        self._dll = None
        self._fn = {}
        self._ctypes_fn = {}
        self.ffi = None  # "native" | "ctypes" once loaded
        self._memo = {}
        self.profiler = None
        self.recorder = None  # engine/graph_dag.Recorder while a step is being recorded into a hipGraph

    def load(self):
        if self._dll is not None:
            return self
        if not os.path.exists(LIB_PATH):
            raise NassegError(
                "{} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.".format(LIB_PATH)
            )
        dll = ctypes.CDLL(LIB_PATH)
        ffi = self._shim()
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(dll, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
            self._ctypes_fn[name] = fn
            fast = getattr(ffi, name, None) if ffi is not None else None
            if fast is not None and ffi.bind(name, ctypes.cast(fn, ctypes.c_void_p).value):
                fn = fast
            self._fn[name] = fn
        self.ffi = "native" if ffi is not None else "ctypes"
        self._dll = dll
        return self

    @staticmethod
    def _shim():
        """The generated CPython call shim (ffi_gen.py; built by build.py next to the library): same library,
        same symbols - resolved above with ctypes and handed over by address - a fifth of the host time per
        call.  NASSEG_FFI=ctypes keeps ctypes; so does a missing shim, or one generated from other prototypes
        than include/nasseg.h declares now (ffi_gen.abi_hash - not file times)."""
        if os.environ.get("NASSEG_FFI", "native") == "ctypes":
            return None
        try:
            from . import _nasseg_ffi as ffi
        except ImportError:
            return None
        from . import ffi_gen

        if not hasattr(ffi, "abi") or ffi.abi() != ffi_gen.abi_hash(HEADER_PATH):
            return None  # (generated from another header: its argument lists may not match the library's)
        ffi.set_addr_of(_address_of)
        return ffi

    def symbols(self):
        self.load()
        return sorted(self._fn)

    def last_error(self):
        self.load()
        msg = self._fn["nasseg_last_error"]()
        return msg.decode("utf-8", "replace") if msg else ""  # (bytes from either call path)

    def call(self, name, *args):
        """Call a status-returning entry point; raise RuntimeError on failure."""
        fn = self._fn.get(name)
        if fn is None:
            self.load()
            fn = self._fn[name]
        prof = self.profiler
        if self.recorder is not None:
            rc = self.recorder.call(name, args, fn)
        elif prof is not None and prof.wants(name):
            rc = prof.bracket(name, args, fn)
        else:
            rc = fn(*args)
        if rc < 0:
            raise NassegError("{}: {}".format(name, self.last_error()))
        return rc

    def query(self, name, *args):
        """Call a value-returning entry point (workspace sizes, plan queries, versions).  These
        are pure functions of their integer arguments, so results are memoised: a training step
        asks the same few hundred questions every time."""
        key = (name,) + args
        hit = self._memo.get(key)
        if hit is not None:
            return hit
        fn = self._fn.get(name)
        if fn is None:
            self.load()
            fn = self._fn[name]
        value = fn(*args)
        if len(self._memo) < 65536:
            self._memo[key] = value
        return value


lib = _Library()


def ptr(t):
    """Device (or host) address of a tensor, None -> NULL.  (While a step is recorded into a hipGraph the recorder
    also notes the address range of the tensor's storage: what the next call may read or write behind that address.)"""
    if t is None:
        return None
    if lib.recorder is not None:
        return lib.recorder.note(t)
    return t.data_ptr()


def current_stream():
    """Raw handle of torch's current HIP stream on the current device (the C-level accessors:
    torch.cuda.current_stream() builds a Stream object per call, ~9 us of the ~17 us the host
    spends per launch)."""
    import torch

    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:  # private accessors moved: the public, slower way
        return torch.cuda.current_stream().cuda_stream


def require_device(*tensors):
    """The product path never computes on the host: fail loudly instead."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NassegError(
                "nasseg kernels need tensors on a HIP device (got {}); there is no CPU "
                "fallback - use oracle/ for CPU reference results".format(t.device)
            )
