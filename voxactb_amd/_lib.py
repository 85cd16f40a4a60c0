"""ctypes binding of libvoxactb_hip.so (the C ABI declared in include/voxactb_hip.h).

There is deliberately NO fallback: if the library is missing or a kernel returns an error the
caller gets an exception.  torch is imported first so that the HIP runtime the library resolves
(`libamdhip64.so.7`) is the one torch already loaded -- device pointers and streams are then shared.
Prototypes are parsed from the header itself, so the binding cannot drift from the ABI.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VOXACTB_HIP_LIB: A/B runs of two builds inside one process launch on the same GPU box -- boxes differ by several per cent)
LIB_PATH = os.environ.get('VOXACTB_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libvoxactb_hip.so')
HEADER = os.path.join(_HERE, '..', 'include', 'voxactb_hip.h')

_ERR = {-1: 'bad argument', -2: 'unsupported size', -3: 'workspace too small', -4: 'HIP launch error'}


class VoxactbHipError(RuntimeError):
    pass


_CT = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'size_t': ctypes.c_size_t, 'float': ctypes.c_float,
       'double': ctypes.c_double, 'uint32_t': ctypes.c_uint32, 'int32_t': ctypes.c_int32, 'vxb_stream_t': ctypes.c_void_p, 'void': None}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every `vxb_*` prototype in the header."""
    txt = open(path).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int|size_t)\s+(vxb_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        for a in [x.strip() for x in args.split(',') if x.strip()]:
            if a == 'void':
                continue
            if '*' in a:
                argtypes.append(ctypes.c_void_p)
            else:
                t = a.replace('const', '').split()[0]
                argtypes.append(_CT[t])
        protos[name] = (_CT[ret], argtypes)
    return protos


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoxactbHipError(
                'HIP extension %s not found: run `python -m voxactb_amd.csrc.build` '
                '(there is no CPU fallback for the product path)' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in parse_header().items():
            fn = getattr(L, name)       # AttributeError here == header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        if os.environ.get('VOXACTB_WGRAD_CHUNKS'):   # experiment switch: chunks per workgroup of the LDS-halo weight gradient
            L.vxb_debug_set_wgrad_halo_chunks(int(os.environ['VOXACTB_WGRAD_CHUNKS']))
        if os.environ.get('VOXACTB_HALO_WN'):      # experiment switch: wave layout of the LDS-halo conv (conv_halo_bf16.hip)
            L.vxb_debug_set_halo_wn(int(os.environ['VOXACTB_HALO_WN']))
        if os.environ.get('VOXACTB_HALO_DBG'):     # experiment bits of conv_halo_bf16.hip (4 = no skipping of depth-edge waves; 1, 2: timing only, WRONG results)
            L.vxb_debug_set_halo_experiment(int(os.environ['VOXACTB_HALO_DBG']))
        if os.environ.get('VOXACTB_WGRAD_LIN'):    # A/B switch of the linear layers' fp16 weight-gradient kernels (wgrad_bf16.hip): 0 generic, 1 pipelined, 2 wide
            L.vxb_debug_set_wgrad_lin(int(os.environ['VOXACTB_WGRAD_LIN']))
        if os.environ.get('VOXACTB_WIDE_WAVES'):   # A/B switch of the wide linear-layer GEMMs (gemm_wide.hip): 8 or 4 waves per workgroup
            L.vxb_debug_set_gemm_wide_waves(int(os.environ['VOXACTB_WIDE_WAVES']))
        if os.environ.get('VOXACTB_WIDE_MIN_M'):   # rows from which the wide linear-layer kernels are dispatched: ops.WIDE_MIN_M reads the same variable
            L.vxb_debug_set_wide_min_rows(int(os.environ['VOXACTB_WIDE_MIN_M']))
        if os.environ.get('VOXACTB_WIDE_DBG'):     # gemm_wide.hip experiment bits (32: row blocks fastest in the grid, round 3's order)
            L.vxb_debug_set_gemm_wide_experiment(int(os.environ['VOXACTB_WIDE_DBG']))
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise VoxactbHipError('%s failed: %s (code %d)' % (what, _ERR.get(rc, 'unknown'), rc))


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the launching stream (bench.py uses it to price
    the dominant kernel against its roofline).  `meta` = (label, flops, bytes) supplied by the ops wrappers."""

    def __init__(self, only=None):
        self.records = []          # (label, name, start_event, stop_event, flops, bytes)
        self.only = only           # None = every launch; a set of labels = only those (two events per launch are not free)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for label, name, e0, e1, fl, by in self.records:
            d = agg.setdefault(label, dict(entry=name, calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d['calls'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['flops'] += fl
            d['bytes'] += by
        return agg


TIMER = None          # set to a KernelTimer to enable
_META = None


def set_meta(label, flops=0.0, nbytes=0.0):
    global _META
    _META = (label, float(flops), float(nbytes))


def call(name, *args):
    """Call a C-ABI entry point; tensors are passed as device pointers, None as NULL; the current stream is appended."""
    global _META
    meta, _META = _META, None
    if TIMER is not None and (TIMER.only is None or (meta is not None and meta[0] in TIMER.only)):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        _call(name, *args)
        e1.record()
        label, fl, by = meta if meta is not None else (name, 0.0, 0.0)
        TIMER.records.append((label, name, e0, e1, fl, by))
        return
    _call(name, *args)


_FNS = {}
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _call(name, *args):
    # (hot: ~550 calls per training step -- plain ints for pointers, the raw current-stream handle of the first tensor's
    # device; ctypes converts them through the argtypes parsed from the header)
    fn = _FNS.get(name)
    if fn is None:
        fn = _FNS[name] = getattr(lib(), name)
    conv = []
    dev = -1
    for a in args:
        if isinstance(a, torch.Tensor):
            if not a.is_cuda:
                raise VoxactbHipError('%s: tensor on %s -- the kernels need a HIP device (no CPU fallback)' % (name, a.device))
            conv.append(a.data_ptr())
            if dev < 0:
                dev = a.get_device()
        elif a is None:
            conv.append(0)
        else:
            conv.append(a)
    if dev >= 0 and dev != _current_device():
        # a kernel launch goes to the CURRENT device (the null stream is per device, and a stream handle of another device
        # is an invalid resource): make the tensors' device current for the launch.  Agents call torch.cuda.set_device in
        # build(), so this branch is the safety net for callers that hold tensors on several devices in one process.
        with torch.cuda.device(dev):
            check(fn(*conv, _stream_of(dev)), name)
        return
    check(fn(*conv, _stream_of(dev)), name)


_current_device = torch.cuda.current_device


def _stream_of(dev):
    if _raw_stream is not None and dev >= 0:
        return _raw_stream(dev)
    return torch.cuda.current_stream(dev if dev >= 0 else None).cuda_stream


class on_device:
    """`with on_device(tensor_or_device):` -- make that HIP device current for raw C-ABI launches (no-op when it already is)."""

    def __init__(self, where):
        dev = where.device if isinstance(where, torch.Tensor) else torch.device(where)
        self._idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self._ctx = None

    def __enter__(self):
        if self._idx != torch.cuda.current_device():
            self._ctx = torch.cuda.device(self._idx)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VoxactbHipError('voxactb_amd kernels need tensors on a HIP device (got %s); '
                                  'there is no CPU fallback' % t.device)
