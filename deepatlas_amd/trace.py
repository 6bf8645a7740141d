"""roctx ranges around the phases of a training step ("rocprofv3 counters + roctx ranges"), so that
`rocprofv3 --marker-trace --kernel-trace` attributes kernels to forward / loss / backward / optimiser / all-reduce.

The ranges come from librocprofiler-sdk-roctx (ROCm >= 6.2) or the older libroctx64, loaded with ctypes on first use; they cost two
library calls per range and nothing when neither library is present.  Off unless DA_ROCTX=1 or trace.enable(True): a range pushed on
the host does not synchronise anything, so the timed region of bench.py is unaffected either way."""
import contextlib
import ctypes
import os

_lib = None
_tried = False
enabled = os.environ.get('DA_ROCTX') == '1'


def _load():
    global _lib, _tried
    if not _tried:
        _tried = True
        for name in ('librocprofiler-sdk-roctx.so', 'libroctx64.so'):
            for prefix in ('', '/opt/rocm/lib/'):
                try:
                    lib = ctypes.CDLL(prefix + name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    lib.roctxMarkA.argtypes = [ctypes.c_char_p]
                    _lib = lib
                    return _lib
                except (OSError, AttributeError):
                    continue
    return _lib


def enable(flag=True):
    global enabled
    prev, enabled = enabled, bool(flag)
    return prev


def available():
    return _load() is not None


def push(name):
    if enabled and _load() is not None:
        _lib.roctxRangePushA(name.encode())


def pop():
    if enabled and _load() is not None:
        _lib.roctxRangePop()


def mark(name):
    if enabled and _load() is not None:
        _lib.roctxMarkA(name.encode())


@contextlib.contextmanager
def range(name):
    """with trace.range('seg/forward'): ...  (a no-op context unless enabled and a roctx library is present)"""
    push(name)
    try:
        yield
    finally:
        pop()
