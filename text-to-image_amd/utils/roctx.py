"""roctx ranges for rocprofv3 timelines (`rocprofv3 --marker-trace --kernel-trace -- python bench.py ...`): SURVEY section 5's tracing hook.

The reference has no tracing of its own (TensorFlow's timeline would be the tool); here the training step names its phases — critic
step, generator step, each graph-segment replay, each gradient exchange — so that a trace of a data-parallel run shows which
collectives overlap which segment.  Ranges cost a library call each (~100 ns without a profiler attached); T2I_ROCTX=0 turns them
into no-ops.  The marker library is looked up once: rocprofiler-sdk's (what rocprofv3 listens to), then roctracer's; without either
the ranges are no-ops too — nothing on the compute path depends on them."""
import contextlib
import ctypes
import os

_LIB = [None, False]          # (handle, looked up)


def _lib():
    if not _LIB[1]:
        _LIB[1] = True
        if os.environ.get('T2I_ROCTX', '1') != '0':
            for name in ('librocprofiler-sdk-roctx.so', 'librocprofiler-sdk-roctx.so.1', 'libroctx64.so', 'libroctx64.so.4'):
                try:
                    h = ctypes.CDLL(name)
                    h.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    h.roctxRangePushA.restype = ctypes.c_int
                    h.roctxRangePop.restype = ctypes.c_int
                    _LIB[0] = h
                    break
                except (OSError, AttributeError):
                    continue
    return _LIB[0]


def available():
    return _lib() is not None


def push(name):
    h = _lib()
    if h is not None:
        h.roctxRangePushA(name.encode())


def pop():
    h = _lib()
    if h is not None:
        h.roctxRangePop()


@contextlib.contextmanager
def range(name):
    h = _lib()
    if h is None:
        yield
        return
    h.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        h.roctxRangePop()
