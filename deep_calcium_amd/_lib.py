"""ctypes binding of libdcunet.so (the C ABI declared in include/dcunet.h).

The prototypes are read from the header itself, so the Python side cannot drift
from the ABI.  There is NO fallback: if the library is missing or a call fails
the product raises (`DcunetError`); nothing here ever routes to a CPU path.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, '..', 'include', 'dcunet.h')
LIB_PATH = os.environ.get('DC_LIB_PATH') or os.path.join(HERE, 'lib', 'libdcunet.so')   # DC_LIB_PATH: A/B builds


class DcunetError(RuntimeError):
    pass


_CTYPES = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'uint64_t': ctypes.c_uint64, 'dc_stream_t': ctypes.c_void_p,
}


def _ctype_of(decl):
    decl = decl.strip()
    if '*' in decl:
        if decl.startswith('void**') or decl.replace(' ', '').startswith('void**'):
            return ctypes.POINTER(ctypes.c_void_p)
        return ctypes.c_void_p                      # device pointers travel as integers
    base = decl.replace('const', '').split()[0]
    return _CTYPES[base]


def parse_header(path=HEADER):
    """Returns {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'^\s*(const char\*|int|long)\s+(dc_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = ' '.join(args.split())
        argtypes, argnames = [], []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                nm = re.search(r'(\w+)$', a).group(1)
                argtypes.append(_ctype_of(a[:len(a) - len(nm)]))
                argnames.append(nm)
        restype = ctypes.c_char_p if ret.startswith('const char') else (ctypes.c_long if ret == 'long' else ctypes.c_int)
        protos[name] = (restype, argtypes, argnames)
    return protos


def header_abi_version(path=HEADER):
    m = re.search(r'^#define\s+DC_ABI_VERSION\s+(\d+)', open(path).read(), flags=re.M)
    if not m:
        raise DcunetError('include/dcunet.h carries no DC_ABI_VERSION')
    return int(m.group(1))


class _Lib(object):
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise DcunetError('libdcunet.so not found at %s -- run `python -m deep_calcium_amd._build` '
                              '(there is no CPU fallback)' % path)
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.protos = parse_header()
        # the header's argument lists are only valid for a library of the same ABI revision (DC_LIB_PATH A/B builds!)
        want = header_abi_version()
        self.cdll.dc_version.restype = ctypes.c_int
        got = self.cdll.dc_version()
        if got != want:
            raise DcunetError('%s is ABI revision %d, include/dcunet.h declares DC_ABI_VERSION %d -- rebuild the library '
                              '(`python -m deep_calcium_amd._build --force`)' % (path, got, want))
        # names whose int return is a count/size, not a status code
        self._plain = set(n for n, (rt, _, _) in self.protos.items()
                          if rt is not ctypes.c_int or n.endswith('_tiles') or n.endswith('_blocks') or n.endswith('_floats') or n == 'dc_version')
        for name, (restype, argtypes, _) in self.protos.items():
            fn = getattr(self.cdll, name)             # AttributeError here = ABI/header mismatch
            fn.restype = restype
            fn.argtypes = argtypes

    def __getattr__(self, name):
        fn = getattr(self.cdll, name)
        if name in self._plain:
            return fn

        def checked(*args):
            rc = fn(*args)
            if rc != 0:
                raise DcunetError('%s failed (%d): %s' % (name, rc, self.cdll.dc_last_error().decode()))
            return rc
        checked.__name__ = name
        setattr(self, name, checked)
        return checked


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
