"""Raw HIP events via ctypes on the libamdhip64 that torch already loaded: lets bench.py time a
kernel on the exact stream the C ABI launches on (torch.cuda.Event only sees torch's stream)."""
import ctypes as C

_hip = None


def _rt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        _hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipEventSynchronize.argtypes = [C.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        _hip.hipEventDestroy.argtypes = [C.c_void_p]
    return _hip


HIP_EVENT_DISABLE_TIMING = 0x2


class HipEvent:
    """timing=False: an ordering-only event (hipEventDisableTiming), cheaper to record and to wait for; the search
    loop records one per leaf batch."""

    def __init__(self, timing: bool = True):
        self.h = C.c_void_p()
        rc = (_rt().hipEventCreate(C.byref(self.h)) if timing
              else _rt().hipEventCreateWithFlags(C.byref(self.h), HIP_EVENT_DISABLE_TIMING))
        if rc != 0:
            raise RuntimeError("hipEventCreate failed")

    def record(self, stream_ptr):
        if _rt().hipEventRecord(self.h, C.c_void_p(stream_ptr)) != 0:
            raise RuntimeError("hipEventRecord failed")

    def sync(self):
        _rt().hipEventSynchronize(self.h)

    def elapsed_ms(self, later: "HipEvent") -> float:
        ms = C.c_float()
        if _rt().hipEventElapsedTime(C.byref(ms), self.h, later.h) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        return ms.value

    @property
    def ptr(self):
        return self.h.value
