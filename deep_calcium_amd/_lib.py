"""ctypes binding of libdcunet.so (the C ABI declared in include/dcunet.h).

The prototypes are read from the header itself, so the Python side cannot drift
from the ABI.  There is NO fallback: if the library is missing or a call fails
the product raises (`DcunetError`); nothing here ever routes to a CPU path.
"""
import ctypes
import os
import re
import struct
import threading

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


class Var(object):
    """A launch argument that changes from step to step (a dropout seed, Adam's lr_t, the batch pointers): recorded on a tape
    by KEY, supplied by value at every replay."""
    __slots__ = ('key', 'value')

    def __init__(self, key, value):
        self.key, self.value = key, value


MARK = '__mark__'        # a recorded position where the caller runs Python between two replayed segments (a collective)


def _bits_of_double(x):
    return struct.unpack('<q', struct.pack('<d', float(x)))[0]


def _pack(value, ctype):
    """One argument as the signed 8-byte slot dc_tape_append / dc_tape_replay take."""
    if ctype in (ctypes.c_float, ctypes.c_double):
        return _bits_of_double(value)
    if value is None:
        return 0
    v = int(value)
    return v - (1 << 64) if v >= (1 << 63) else v


def ops_equal(a, b):
    """Two recordings of the same phase of a step: the same entry points with the same arguments (Vars: the same keys)."""
    if len(a) != len(b):
        return False
    for (na, xa), (nb, xb) in zip(a, b):
        if na != nb:
            return False
        if na == MARK:
            if xa != xb:
                return False
            continue
        if len(xa) != len(xb):
            return False
        for u, v in zip(xa, xb):
            if type(u) is Var or type(v) is Var:
                if type(u) is not type(v) or u.key != v.key:
                    return False
            elif u != v:
                return False
    return True


class Tape(object):
    """A recorded enqueue sequence [(entry point name, args) | (MARK, tag)] as a dc_tape_* object (include/dcunet.h):
    replay(values) re-issues it from C -- values: {Var key: current value} --, calling on_mark(tag) between the segments."""

    def __init__(self, lib, ops):
        self.lib = lib
        self.handle = ctypes.c_void_p()
        lib.dc_tape_create(ctypes.byref(self.handle))
        self.keys, self.ctypes_of, self.segments = [], [], []
        slot_of = {}
        n = first = 0
        for name, args in ops:
            if name == MARK:
                self.segments.append((first, n - first, args))
                first = n
                continue
            argtypes = lib.protos[name][1]
            packed, patches = [], []
            for i, (a, t) in enumerate(zip(args, argtypes)):
                if type(a) is Var:
                    if a.key not in slot_of:
                        slot_of[a.key] = len(self.keys)
                        self.keys.append(a.key)
                        self.ctypes_of.append(t)
                    patches.append((i, slot_of[a.key]))
                    a = a.value
                packed.append(_pack(a, t))
            lib.dc_tape_append(self.handle, name.encode(), (ctypes.c_long * len(packed))(*packed), len(packed))
            for i, slot in patches:
                lib.dc_tape_patch(self.handle, n, i, slot)
            n += 1
        self.segments.append((first, n - first, None))
        self.n = n
        self._vals = (ctypes.c_long * max(len(self.keys), 1))()
        self._replay = lib.dc_tape_replay

    def replay(self, values, on_mark=None):
        vals = self._vals
        for i, (k, t) in enumerate(zip(self.keys, self.ctypes_of)):
            vals[i] = _pack(values[k], t)
        nv = len(self.keys)
        for first, count, tag in self.segments:
            if count:
                self._replay(self.handle, first, count, vals, nv)
            if tag is not None:
                on_mark(tag)

    def __del__(self):
        try:
            self.lib.cdll.dc_tape_destroy(self.handle)
        except Exception:
            pass


class _Lib(object):
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise DcunetError('libdcunet.so not found at %s -- run `python -m deep_calcium_amd._build` '
                              '(there is no CPU fallback)' % path)
        self.path = path
        # torch FIRST: the PyTorch-ROCm wheel carries its own libamdhip64; once it is loaded, libdcunet.so's dependency on that
        # soname resolves to the same runtime.  Loaded the other way round (this library before torch: /opt/rocm's copy, then
        # torch's) the process holds two HIP runtimes and every launch from here fails with "no ROCm-capable device is detected".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
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
                          if rt is not ctypes.c_int or n.endswith('_tiles') or n.endswith('_blocks') or n.endswith('_floats') or n.endswith('_rows') or n == 'dc_version')
        for name, (restype, argtypes, _) in self.protos.items():
            fn = getattr(self.cdll, name)             # AttributeError here = ABI/header mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        self._tl = threading.local()                  # per-thread recording list (Tape): None = launches are not recorded

    # ---- recording (deep_calcium_amd/net.py _taped): every checked launch of THIS thread is also appended to a list ----
    def record_begin(self):
        self._tl.rec = []

    def record_end(self):
        rec, self._tl.rec = self._tl.rec, None
        return rec

    def recording(self):
        return getattr(self._tl, 'rec', None) is not None

    def mark(self, tag):
        rec = getattr(self._tl, 'rec', None)
        if rec is not None:
            rec.append((MARK, tag))

    def __getattr__(self, name):
        fn = getattr(self.cdll, name)
        if name in self._plain:
            return fn

        tl = self._tl

        def checked(*args):
            rec = getattr(tl, 'rec', None)
            if rec is not None:
                rec.append((name, args))
                args = tuple(a.value if type(a) is Var else a for a in args)
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
