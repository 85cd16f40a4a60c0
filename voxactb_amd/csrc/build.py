"""Build libvoxactb_hip.so (all gfx950 kernels + the C ABI) in-tree with hipcc.

    python -m voxactb_amd.csrc.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU
box with the gpurun snapshot.  Objects are rebuilt only when their source (or a header) is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libvoxactb_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wno-unused-result']


# scheduler choice per source, measured on the kernels' own micro-benchmarks (tools/bench_*.py)
_MINREG = ['-mllvm', '-amdgpu-sched-strategy=iterative-minreg']
PER_FILE_FLAGS = {
    # weight gradient, 2x8x8 tiles: the same register count with either scheduler since round 2 (no spills); minreg kept.
    # The 4x4x8 unit (wgrad_halo_t44.hip) spills 2 VGPRs with minreg and none with the default scheduler -- and a scratch
    # reload in the prefetch section is an s_waitcnt vmcnt(0), see wgrad_halo.hip -- so it takes the default.
    # conv_halo_bf16.hip: + 1-2 % on the dense kernels but - 5 % on the tap-list variant -> default.  max-ilp: c1_conv.hip - 9 %.
    'wgrad_halo.hip': _MINREG,
}


INCLUDES_SOURCE = {'wgrad_halo_t44.hip': ['wgrad_halo.hip']}      # translation units that #include another source


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))


def _headers_mtime():
    hs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith('.h')]
    hs.append(os.path.join(HERE, '..', '..', 'include', 'voxactb_hip.h'))
    hs.append(os.path.join(HERE, 'exports.map'))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(HERE, src[:-4] + '.o')
    s = os.path.join(HERE, src)
    src_mtime = max([os.path.getmtime(s)] + [os.path.getmtime(os.path.join(HERE, d)) for d in INCLUDES_SOURCE.get(src, [])])
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > src_mtime
            and os.path.getmtime(obj) > _headers_mtime()):
        return obj, False
    cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(src, []) + os.environ.get('VXB_EXTRA_FLAGS', '').split() + ['-c', s, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--version-script=' + os.path.join(HERE, 'exports.map'), '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
        if verbose:
            print('built %s from %d sources' % (LIB, len(srcs)))
    elif verbose:
        print('%s up to date' % LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
