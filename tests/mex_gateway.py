"""Build + ctypes access to matlab/SimpleCABACMex.cpp compiled against tests/mexstub/mex.h."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libmex_gateway.so")
SRCS = [os.path.join(ROOT, "matlab", "SimpleCABACMex.cpp"), os.path.join(ROOT, "tests", "mexstub", "mex_driver.cpp")]
DOUBLE, UINT8, UINT32, CHAR, UINT16 = 0, 1, 2, 3, 4
_DT = {DOUBLE: np.float64, UINT8: np.uint8, UINT32: np.uint32, UINT16: np.uint16}


class MexError(RuntimeError):
    pass


class Gateway:
    def __init__(self):
        from isscabac_amd import build
        lib = build.build_lib()
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        deps = SRCS + [os.path.join(ROOT, "tests", "mexstub", "mex.h"), os.path.join(ROOT, "include", "isscabac.h")]
        if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps):
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++11", "-shared", "-fPIC", "-I", os.path.join(ROOT, "tests", "mexstub"),
                                   "-I", os.path.join(ROOT, "include")] + SRCS +
                                  ["-L", os.path.dirname(lib), "-lisscabac_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", OUT])
        L = self.L = C.CDLL(OUT)
        L.drv_str.restype = C.c_void_p
        L.drv_str.argtypes = [C.c_char_p]
        L.drv_numeric.restype = C.c_void_p
        L.drv_numeric.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]
        L.drv_free.argtypes = [C.c_void_p]
        L.drv_class.argtypes = [C.c_void_p]
        L.drv_m.restype = C.c_size_t
        L.drv_m.argtypes = [C.c_void_p]
        L.drv_n.restype = C.c_size_t
        L.drv_n.argtypes = [C.c_void_p]
        L.drv_data.restype = C.c_void_p
        L.drv_data.argtypes = [C.c_void_p]
        L.drv_mex.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.drv_printed.restype = C.c_char_p
        L.drv_locks.restype = C.c_int

    def printed(self, clear=True):
        out = self.L.drv_printed().decode()
        if clear:
            self.L.drv_clear_printed()
        return out

    def locks(self):
        return int(self.L.drv_locks())

    def shutdown(self):
        """what `clear mex` / MATLAB exit does: run the gateway's mexAtExit function"""
        self.L.drv_shutdown()

    def _to_mx(self, a):
        if isinstance(a, str):
            return self.L.drv_str(a.encode())
        a = np.asarray(a)
        if a.dtype == np.uint8:
            cls = UINT8
        elif a.dtype == np.uint32:
            cls = UINT32
        elif a.dtype == np.uint16:
            cls = UINT16
        else:
            cls, a = DOUBLE, a.astype(np.float64)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(1, -1)
        f = np.asfortranarray(a)  # MATLAB arrays are column-major
        return self.L.drv_numeric(cls, f.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1])

    def call(self, *args, nlhs=0):
        """SimpleCABACMex(args...) -> list of nlhs numpy arrays (MATLAB shapes)."""
        prhs = (C.c_void_p * len(args))(*[self._to_mx(a) for a in args])
        plhs = (C.c_void_p * max(1, nlhs))()
        err = C.create_string_buffer(1024)
        rc = self.L.drv_mex(nlhs, plhs, len(args), prhs, err, 1024)
        for p in prhs:
            self.L.drv_free(p)
        if rc:
            raise MexError(err.value.decode())
        outs = []
        for i in range(nlhs):
            p = plhs[i]
            assert p, "gateway did not create output %d" % i
            m, n, cls = self.L.drv_m(p), self.L.drv_n(p), self.L.drv_class(p)
            flat = np.ctypeslib.as_array(C.cast(self.L.drv_data(p), C.POINTER(C.c_uint8)), (m * n * np.dtype(_DT[cls]).itemsize,))
            outs.append(flat.view(_DT[cls]).reshape((m, n), order="F").copy())
            self.L.drv_free(p)
        return outs
