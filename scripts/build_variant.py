"""Build an A/B copy of the library: deep_calcium_amd/lib/libdcunet_<tag>.so = the shipped objects with the named sources
re-compiled under extra -D flags (kernel experiments and ablation builds; run HERE so the .so travels to the GPU box).
    python scripts/build_variant.py <tag> <source.hip>[,<source.hip>...] [-DNAME=VALUE ...]
Select it with DC_LIB_PATH=<path> (deep_calcium_amd/_lib.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deep_calcium_amd import _build            # noqa: E402


def build_variant(tag, sources, defines, verbose=False):
    _build.build()                              # the shipped objects (lib/obj/*.o) are the base
    objdir = os.path.join(_build.LIBDIR, 'obj')
    vdir = os.path.join(_build.LIBDIR, 'obj_' + tag)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for src in _build.SOURCES:
        base = os.path.splitext(src)[0] + '.o'
        if src in sources:
            obj = os.path.join(vdir, base)
            cmd = [_build._hipcc()] + _build.FLAGS + _build.HOST_ONLY.get(src, []) + list(defines) + \
                  ['-x', 'hip', '-c', os.path.join(_build.CSRC, src), '-o', obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('hipcc failed for %s %s:\n%s' % (src, defines, r.stderr[-6000:]))
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
            objs.append(obj)
        else:
            objs.append(os.path.join(objdir, base))
    lib = os.path.join(_build.LIBDIR, 'libdcunet_%s.so' % tag)
    r = subprocess.run([_build._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s' % r.stderr[-6000:])
    return lib


if __name__ == '__main__':
    tag, srcs = sys.argv[1], sys.argv[2].split(',')
    print(build_variant(tag, srcs, sys.argv[3:], verbose=True))
