"""Test plumbing: device memory through the HIP runtime that libmiba.so has already loaded (ctypes), so that a GPU test
needs no second framework in the process.  (PyTorch ships its own copy of the HIP runtime: initialised after libmiba's,
it finds no GPU.)"""
import ctypes as C

import numpy as np

from bundle_adjustment_amd import miba

_hip = None


def hip():
    global _hip
    if _hip is None:
        miba.lib()   # loads libamdhip64 as libmiba's dependency
        path = None
        with open("/proc/self/maps") as fh:
            for line in fh:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        _hip = C.CDLL(path or "libamdhip64.so")
    return _hip


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


def mem_free_bytes():
    free, total = C.c_size_t(), C.c_size_t()
    _chk(hip().hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value


class DeviceArray:
    """a device copy of a numpy array; .ptr is the device address"""

    def __init__(self, host):
        self.host = np.ascontiguousarray(host)
        p = C.c_void_p()
        _chk(hip().hipMalloc(C.byref(p), C.c_size_t(max(1, self.host.nbytes))), "hipMalloc")
        self.ptr = p.value
        _chk(hip().hipMemcpy(C.c_void_p(self.ptr), self.host.ctypes.data_as(C.c_void_p), C.c_size_t(self.host.nbytes), C.c_int(1)),
             "hipMemcpy H2D")

    def get(self):
        out = np.empty_like(self.host)
        _chk(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), C.c_int(2)), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(C.c_void_p(self.ptr))
            self.ptr = None
