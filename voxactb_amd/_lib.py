"""ctypes binding of libvoxactb_hip.so (the C ABI declared in include/voxactb_hip.h).

There is deliberately NO fallback: if the library is missing or a kernel returns an error the
caller gets an exception.  torch is imported first so that the HIP runtime the library resolves
(`libamdhip64.so.7`) is the one torch already loaded -- device pointers and streams are then shared.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libvoxactb_hip.so')

_ERR = {-1: 'bad argument', -2: 'unsupported size', -3: 'workspace too small', -4: 'HIP launch error'}


class VoxactbHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoxactbHipError(
                'HIP extension %s not found: run `python -m voxactb_amd.csrc.build` '
                '(there is no CPU fallback for the product path)' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(rc, what):
    if rc != 0:
        raise VoxactbHipError('%s failed: %s (code %d)' % (what, _ERR.get(rc, 'unknown'), rc))


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VoxactbHipError('voxactb_amd kernels need tensors on a HIP device (got %s); '
                                  'there is no CPU fallback' % t.device)


c_int, c_i64, c_sz, c_f, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p


def _declare(L):
    L.vxb_abi_version.restype = c_int
    L.vxb_voxelize_workspace_bytes.restype = c_sz
    L.vxb_voxelize_workspace_bytes.argtypes = [c_int, c_int, c_int]
    L.vxb_voxelize_f32.restype = c_int
    L.vxb_voxelize_f32.argtypes = [c_p, c_p, c_int, c_int, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                   c_p, c_int, c_int, c_p, c_p, c_sz, c_p]
    for name, (res, args) in _EXTRA_DECLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


# filled by the op modules (voxactb_amd/ops_*.py) before first use of lib()
_EXTRA_DECLS = {}


def declare(name, restype, argtypes):
    _EXTRA_DECLS[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
