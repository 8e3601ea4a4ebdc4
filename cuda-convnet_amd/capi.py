"""ctypes binding of the C ABI in include/mi_convnet.h (libmiconvnet.so).

The prototypes are parsed from the header itself, so the binding cannot drift
from the declared ABI.  `DeviceBuffer` is a minimal owner of device memory for
tests, the benchmark and tools; the training path itself goes through the
`_ConvNet` extension (C++), not through this module.

Fails loudly when the shared library is missing or cannot be loaded: there is
no CPU fallback anywhere in the product path.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'mi_convnet.h')
LIB_PATH = os.path.join(_HERE, 'libmiconvnet.so')

_CTYPES = {
    'int': C.c_int, 'float': C.c_float, 'size_t': C.c_size_t, 'void*': C.c_void_p, 'void**': C.POINTER(C.c_void_p),
    'const void*': C.c_void_p, 'float*': C.c_void_p, 'const float*': C.c_void_p, 'double*': C.c_void_p,
    'int*': C.POINTER(C.c_int), 'int**': C.POINTER(C.c_void_p), 'const int*': C.c_void_p, 'const float* const*': C.c_void_p, 'float* const*': C.c_void_p, 'const mcn_sgd_item*': C.c_void_p, 'const mcn_row_sum_item*': C.c_void_p, 'const mcn_sgd_partials*': C.c_void_p, 'char*': C.c_char_p, 'const char*': C.c_char_p,
}


class McnError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """Returns {name: (restype_str, [(type_str, arg_name), ...])} for every declared entry point."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'(?:^|\n)\s*(const char\*|int|size_t|uint64_t)\s+(mcn_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), ' '.join(m.group(3).split())
        parsed = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                am = re.match(r'^(.*?)(\w+)(\[\d+\])?$', a)
                typ = am.group(1).strip().replace(' *', '*')
                if am.group(3):
                    typ += '*'
                parsed.append((typ, am.group(2)))
        protos[name] = (ret, parsed)
    return protos


class Lib:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise McnError('%s not found: build it with `make -C cuda-convnet_amd/csrc` '
                           '(or __graft_entry__.build()); there is no CPU fallback' % path)
        self.dll = C.CDLL(path)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.dll, name)  # AttributeError if the symbol is not exported
            fn.restype = {'int': C.c_int, 'size_t': C.c_size_t, 'uint64_t': C.c_uint64, 'const char*': C.c_char_p}[ret]
            fn.argtypes = [_CTYPES[t] for t, _ in args]

    def raw(self, name):
        return getattr(self.dll, name)

    def call(self, name, *args):
        """Call an int-returning entry point; raise McnError on a non-zero status."""
        conv = [a.ptr if isinstance(a, DeviceBuffer) else a for a in args]
        rc = getattr(self.dll, name)(*conv)
        if self.protos[name][0] == 'int' and rc != 0:
            raise McnError('%s failed (%d): %s' % (name, rc, self.dll.mcn_last_error().decode()))
        return rc


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = Lib()
    return _LIB


class DeviceBuffer:
    """Owns `nbytes` of device memory obtained through mcn_alloc."""

    def __init__(self, nbytes=None, array=None, shape=None):
        L = lib()
        if array is not None:
            array = np.ascontiguousarray(array)
            nbytes = array.nbytes
            shape = array.shape
            self.dtype = array.dtype
        else:
            self.dtype = np.dtype(np.float32)
            if shape is not None and nbytes is None:
                nbytes = int(np.prod(shape)) * 4
        self.nbytes, self.shape = int(nbytes), shape
        p = C.c_void_p()
        L.call('mcn_alloc', C.byref(p), max(self.nbytes, 4))
        self.ptr = p.value
        if array is not None:
            self.upload(array)

    def upload(self, array):
        array = np.ascontiguousarray(array)
        assert array.nbytes <= self.nbytes
        lib().call('mcn_h2d', self.ptr, array.ctypes.data, array.nbytes, None)
        lib().call('mcn_device_sync')

    def numpy(self, shape=None, dtype=None):
        shape = shape if shape is not None else self.shape
        out = np.empty(shape, dtype or self.dtype)
        lib().call('mcn_device_sync')
        lib().call('mcn_d2h', out.ctypes.data, self.ptr, out.nbytes, None)
        lib().call('mcn_device_sync')
        return out

    def zero(self):
        lib().call('mcn_memset', self.ptr, 0, self.nbytes, None)

    def offset(self, nbytes):
        return self.ptr + nbytes

    def free(self):
        if self.ptr:
            lib().dll.mcn_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def zeros(shape):
    b = DeviceBuffer(shape=shape)
    b.zero()
    return b


def workspace(nbytes):
    return DeviceBuffer(nbytes=max(int(nbytes), 256))
