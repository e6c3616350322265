"""Build libdcunet.so (gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m deep_calcium_amd._build
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libdcunet.so')
SOURCES = ['common.cpp', 'comm.cpp', 'nf_score.cpp', 'tape.cpp', 'igemm_conv.hip', 'igemm_f16x3.hip', 'igemm_pp.hip', 'wgrad.hip', 'wgrad_f16x3.hip', 'bwd_joint.hip', 'conv_c1.hip', 'elementwise.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
HOST_ONLY = {'nf_score.cpp': ['-ffp-contract=off']}      # host arithmetic that must round like numpy's: no fused multiply-add


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _newest_src():
    t = 0.0
    for root, _, files in os.walk(CSRC):
        for f in files:
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    t = max(t, os.path.getmtime(os.path.join(HERE, '..', 'include', 'dcunet.h')))
    return t


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_src():
        return LIB
    hipcc = _hipcc()
    from . import _gen_tape
    _gen_tape.generate()              # csrc/tape_tramp.inc: one trampoline per launch entry point of include/dcunet.h
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        cmd = [hipcc] + FLAGS + HOST_ONLY.get(src, []) + ['-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s' % r.stderr[-6000:])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
