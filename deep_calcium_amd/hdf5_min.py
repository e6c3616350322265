"""A minimal HDF5 reader / writer in pure numpy: exactly the subset Keras-2.0.x model files use.

The reference stores its models as Keras HDF5 files (`ModelCheckpoint`, `load_model`:
/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:28,:423-424 and
/root/reference/deepcalcium/utils/keras_helpers.py:24-68), written by h5py with libhdf5's default ("earliest")
file-format settings.  h5py is not installable next to this build's interpreter, so the container format is
restated here from the HDF5 File Format Specification (version 1/2 structures):

  read : superblock v0/v1; old-style groups (symbol table + v1 B-tree + local heap) and compact new-style groups
         (link messages); version-1 object headers with continuation blocks; attributes (message versions 1-3) holding
         fixed-length strings, variable-length strings (global heap), integers and floats, scalar or n-d;
         datasets with contiguous or compact layout of little-endian integers / IEEE floats.
  write: superblock v0, old-style groups, version-1 object headers, version-1 attributes (fixed-length strings, numeric
         scalars / arrays), contiguous datasets -- the structures h5py itself emits for such a file, so h5py / Keras read it.

Not supported (raises Hdf5Error, never guesses): chunked / compressed datasets, dense (fractal-heap) link or attribute
storage, version-2 object headers (libver='latest' files), big-endian data, compound types.
"""
import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b'\x89HDF\r\n\x1a\n'


class Hdf5Error(ValueError):
    pass


def is_hdf5(path):
    try:
        with open(path, 'rb') as fp:
            return fp.read(8) == SIGNATURE
    except OSError:
        return False


# =====================================================================================================================
# reading
# =====================================================================================================================
class _Datatype(object):
    """Decoded datatype message: kind in {'int', 'float', 'str', 'vstr'}; numpy dtype for the fixed-size kinds."""

    def __init__(self, buf, off):
        b0, = struct.unpack_from('<B', buf, off)
        self.version, self.cls = b0 >> 4, b0 & 0x0F
        bits = buf[off + 1:off + 4]
        self.size, = struct.unpack_from('<I', buf, off + 4)
        self.base = None
        if self.cls == 0:                                   # fixed point
            if bits[0] & 1:
                raise Hdf5Error('big-endian integers are not supported')
            self.kind, self.dtype = 'int', np.dtype('<%s%d' % ('i' if bits[0] & 8 else 'u', self.size))
            self.msg_size = 8 + 4
        elif self.cls == 1:                                 # IEEE float
            if bits[0] & 1:
                raise Hdf5Error('big-endian floats are not supported')
            if self.size not in (2, 4, 8):
                raise Hdf5Error('unsupported float size %d' % self.size)
            self.kind, self.dtype = 'float', np.dtype('<f%d' % self.size)
            self.msg_size = 8 + 12
        elif self.cls == 3:                                 # fixed-length string
            self.kind, self.dtype = 'str', np.dtype('S%d' % self.size)
            self.pad = bits[0] & 0x0F                       # 0 null-terminated, 1 null-padded, 2 space-padded
            self.msg_size = 8
        elif self.cls == 9:                                 # variable length (sequence or string)
            self.is_vstr = (bits[0] & 0x0F) == 1
            self.base = _Datatype(buf, off + 8)
            if not self.is_vstr:
                raise Hdf5Error('variable-length sequences are not supported')
            self.kind, self.dtype = 'vstr', None
            self.msg_size = 8 + self.base.msg_size
        else:
            raise Hdf5Error('unsupported datatype class %d' % self.cls)


def _dataspace(buf, off):
    """-> shape tuple (() for scalar, None for a null dataspace)."""
    version, rank, flags = struct.unpack_from('<BBB', buf, off)
    if version == 1:
        o = off + 8
    elif version == 2:
        stype = buf[off + 3]
        if stype == 2:
            return None
        o = off + 4
    else:
        raise Hdf5Error('unsupported dataspace version %d' % version)
    return tuple(struct.unpack_from('<%dQ' % rank, buf, o)) if rank else ()


class _Object(object):
    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.msgs = f._read_object_header(addr)

    # -- attributes ------------------------------------------------------------------------------------------------
    @property
    def attrs(self):
        out = {}
        for mtype, flags, body in self.msgs:
            if mtype == 0x000C:
                name, value = self.f._decode_attribute(body)
                out[name] = value
            elif mtype == 0x0015:                           # attribute info: dense storage?
                o = 2 + (2 if body[1] & 1 else 0)
                if struct.unpack_from('<Q', body, o)[0] != UNDEF:
                    raise Hdf5Error('dense attribute storage (fractal heap) is not supported')
        return out


class Group(_Object):
    def _links(self):
        """name -> object header address.  Walked once per group address and kept on the (read-only) File: a Keras model
        file is ~140 lookups below `model_weights`, each of which used to re-walk the B-trees of every level."""
        cache = self.f._link_cache
        links = cache.get(self.addr)
        if links is None:
            links = cache[self.addr] = self._read_links()
        return links

    def _read_links(self):
        links = {}
        for mtype, flags, body in self.msgs:
            if mtype == 0x0011:                             # symbol table: old-style group
                btree, heap = struct.unpack_from('<QQ', body, 0)
                links.update(self.f._walk_group_btree(btree, heap))
            elif mtype == 0x0006:                           # link message: compact new-style group
                name, addr = self.f._decode_link(body)
                if addr is not None:
                    links[name] = addr
            elif mtype == 0x0002:                           # link info: dense storage?
                lflags = body[1]
                o = 2 + (8 if lflags & 1 else 0)
                fheap, = struct.unpack_from('<Q', body, o)
                if fheap != UNDEF:
                    raise Hdf5Error('dense link storage (fractal heap) is not supported')
        return links

    def keys(self):
        return sorted(self._links())

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self.f.root if path.startswith('/') else self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._links()
            if part not in links:
                raise KeyError(path)
            node = self.f._open(links[part])
        return node


class Dataset(_Object):
    def _meta(self):
        dt = shape = layout = None
        for mtype, flags, body in self.msgs:
            if mtype == 0x0003:
                dt = _Datatype(body, 0)
            elif mtype == 0x0001:
                shape = _dataspace(body, 0)
            elif mtype == 0x0008:
                layout = body
            elif mtype == 0x000B:
                raise Hdf5Error('filtered (compressed) datasets are not supported')
        if dt is None or layout is None:
            raise Hdf5Error('object at %#x is not a dataset' % self.addr)
        return dt, shape, layout

    @property
    def shape(self):
        return self._meta()[1]

    @property
    def dtype(self):
        return self._meta()[0].dtype

    def read(self):
        dt, shape, layout = self._meta()
        if dt.kind not in ('int', 'float', 'str'):
            raise Hdf5Error('unsupported dataset element type')
        if shape is None:
            return np.zeros((0,), dt.dtype)
        n = int(np.prod(shape)) if shape else 1
        version = layout[0]
        if version != 3:
            raise Hdf5Error('unsupported data layout message version %d' % version)
        cls = layout[1]
        if cls == 1:                                        # contiguous
            addr, size = struct.unpack_from('<QQ', layout, 2)
            raw = b'\0' * (n * dt.dtype.itemsize) if addr == UNDEF else self.f._bytes(addr, n * dt.dtype.itemsize)
        elif cls == 0:                                      # compact
            size, = struct.unpack_from('<H', layout, 2)
            raw = bytes(layout[4:4 + size])
        else:
            raise Hdf5Error('chunked datasets are not supported (Keras weight files are contiguous)')
        return np.frombuffer(raw, dtype=dt.dtype, count=n).reshape(shape).copy()

    def __array__(self, dtype=None, copy=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)


class File(Group):
    """Read-only view of an HDF5 file: `f['a/b']` -> Group / Dataset, `.attrs`, `Dataset.read()`."""

    def __init__(self, path):
        import mmap
        with open(path, 'rb') as fp:
            # memory-mapped: the reference's dataset files carry a multi-GB series/raw next to the small arrays read here
            self.buf = mmap.mmap(fp.fileno(), 0, access=mmap.ACCESS_READ) if os.fstat(fp.fileno()).st_size else b''
        if self.buf[:8] != SIGNATURE:
            raise Hdf5Error('%s is not an HDF5 file' % path)
        version = self.buf[8]
        if version not in (0, 1):
            raise Hdf5Error('superblock version %d (libver="latest" file) is not supported' % version)
        if self.buf[13] != 8 or self.buf[14] != 8:
            raise Hdf5Error('only 8-byte offsets / lengths are supported')
        o = 24 + (4 if version == 1 else 0)
        self.base, _, self.eof, _ = struct.unpack_from('<QQQQ', self.buf, o)
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        _, root_addr = struct.unpack_from('<QQ', self.buf, o + 32)
        self.f = self
        self.root = self
        self._gcol = {}
        self._link_cache = {}
        self._obj_cache = {}
        _Object.__init__(self, self, root_addr)

    def close(self):
        if hasattr(self.buf, 'close'):
            self._gcol.clear()
            self._link_cache.clear()
            self._obj_cache.clear()
            self.buf.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # -- low level ---------------------------------------------------------------------------------------------------
    def _bytes(self, addr, size):
        a = self.base + addr
        if a + size > len(self.buf):
            raise Hdf5Error('truncated file: %d bytes at %#x' % (size, a))
        return self.buf[a:a + size]

    def _open(self, addr):
        o = self._obj_cache.get(addr)
        if o is None:
            obj = _Object(self, addr)
            kind = Dataset if any(m[0] == 0x0008 for m in obj.msgs) else Group
            o = kind.__new__(kind)
            o.f, o.addr, o.msgs = self, addr, obj.msgs
            self._obj_cache[addr] = o
        return o

    def _read_object_header(self, addr):
        head = self._bytes(addr, 16)
        if head[:4] == b'OHDR':
            raise Hdf5Error('version-2 object headers (libver="latest" file) are not supported')
        version, _, nmsgs, _, hsize = struct.unpack_from('<BBHII', head, 0)
        if version != 1:
            raise Hdf5Error('unsupported object header version %d at %#x' % (version, addr))
        msgs = []
        blocks = [(addr + 16, hsize)]
        while blocks and len(msgs) < nmsgs:
            baddr, bsize = blocks.pop(0)
            block = self._bytes(baddr, bsize)
            o = 0
            while o + 8 <= bsize and len(msgs) < nmsgs:
                mtype, msize, mflags = struct.unpack_from('<HHB', block, o)
                body = block[o + 8:o + 8 + msize]
                o += 8 + msize
                if mtype == 0x0010:                         # continuation
                    caddr, csize = struct.unpack_from('<QQ', body, 0)
                    blocks.append((caddr, csize))
                if mflags & 2:
                    raise Hdf5Error('shared object header messages are not supported')
                msgs.append((mtype, mflags, body))
        return msgs

    def _local_heap_data(self, addr):
        head = self._bytes(addr, 32)
        if head[:4] != b'HEAP':
            raise Hdf5Error('bad local heap at %#x' % addr)
        size, _, data_addr = struct.unpack_from('<QQQ', head, 8)
        return self._bytes(data_addr, size)

    def _walk_group_btree(self, btree_addr, heap_addr):
        heap = self._local_heap_data(heap_addr)
        links = {}

        def name_at(off):
            end = heap.index(b'\0', off)
            return heap[off:end].decode('utf8')

        def walk(addr):
            head = self._bytes(addr, 24)
            if head[:4] == b'SNOD':
                nsym, = struct.unpack_from('<H', head, 6)
                node = self._bytes(addr, 8 + nsym * 40)
                for i in range(nsym):
                    name_off, ohdr = struct.unpack_from('<QQ', node, 8 + i * 40)
                    links[name_at(name_off)] = ohdr
                return
            if head[:4] != b'TREE':
                raise Hdf5Error('bad group B-tree node at %#x' % addr)
            ntype, level, used = struct.unpack_from('<BBH', head, 4)
            if ntype != 0:
                raise Hdf5Error('unexpected B-tree node type %d in a group' % ntype)
            node = self._bytes(addr, 24 + (2 * used + 1) * 8)
            for i in range(used):
                child, = struct.unpack_from('<Q', node, 24 + 8 + i * 16)
                walk(child)

        if btree_addr != UNDEF:
            walk(btree_addr)
        return links

    def _decode_link(self, body):
        version, flags = body[0], body[1]
        o = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[o]; o += 1
        if flags & 0x04:
            o += 8
        if flags & 0x10:
            o += 1
        lsize = 1 << (flags & 3)
        nlen = int.from_bytes(body[o:o + lsize], 'little'); o += lsize
        name = body[o:o + nlen].decode('utf8'); o += nlen
        if ltype != 0:
            return name, None                               # soft / external links are ignored
        addr, = struct.unpack_from('<Q', body, o)
        return name, addr

    def _global_heap_object(self, coll_addr, index):
        if coll_addr not in self._gcol:
            head = self._bytes(coll_addr, 16)
            if head[:4] != b'GCOL':
                raise Hdf5Error('bad global heap collection at %#x' % coll_addr)
            csize, = struct.unpack_from('<Q', head, 8)
            data = self._bytes(coll_addr, csize)
            objs, o = {}, 16
            while o + 16 <= csize:
                idx, _, _, osize = struct.unpack_from('<HHIQ', data, o)
                if idx == 0:
                    break
                objs[idx] = data[o + 16:o + 16 + osize]
                o += 16 + ((osize + 7) // 8) * 8
            self._gcol[coll_addr] = objs
        return self._gcol[coll_addr][index]

    def _decode_attribute(self, body):
        version = body[0]
        if version == 1:
            nsize, dsize, ssize = struct.unpack_from('<HHH', body, 2)
            o = 8
            pad = lambda n: (n + 7) // 8 * 8
            enc = 'ascii'
        elif version in (2, 3):
            nsize, dsize, ssize = struct.unpack_from('<HHH', body, 2)
            o = 8 + (1 if version == 3 else 0)
            pad = lambda n: n
            if body[1] & 3:
                raise Hdf5Error('shared attribute datatypes are not supported')
        else:
            raise Hdf5Error('unsupported attribute message version %d' % version)
        name = body[o:o + nsize].split(b'\0')[0].decode('utf8'); o += pad(nsize)
        dt = _Datatype(body, o); o += pad(dsize)
        shape = _dataspace(body, o); o += pad(ssize)
        if shape is None:
            return name, None
        n = int(np.prod(shape)) if shape else 1
        if dt.kind == 'vstr':
            vals = []
            for i in range(n):
                length, caddr, idx = struct.unpack_from('<IQI', body, o + 16 * i)
                vals.append(self._global_heap_object(caddr, idx)[:length].decode('utf8') if length else '')
            arr = np.array(vals, dtype=object).reshape(shape)
            return name, (arr.item() if shape == () else arr)
        raw = body[o:o + n * dt.dtype.itemsize]
        arr = np.frombuffer(raw, dtype=dt.dtype, count=n).reshape(shape).copy()
        if dt.kind == 'str':
            arr = np.char.rstrip(arr, b'\0') if shape else arr
            return name, (bytes(arr.item()).rstrip(b'\0') if shape == () else arr)
        return name, (arr.item() if shape == () else arr)


# =====================================================================================================================
# writing
# =====================================================================================================================
def _dt_msg(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == 'S':
        return struct.pack('<BBBBI', 0x13, 0x00, 0, 0, dtype.itemsize)             # string, null-terminated, ASCII
    if dtype.kind == 'f':
        size = dtype.itemsize
        exp_loc, exp_size, man_size, bias = {2: (10, 5, 10, 15), 4: (23, 8, 23, 127), 8: (52, 11, 52, 1023)}[size]
        return struct.pack('<BBBBIHHBBBBI', 0x11, 0x20, size * 8 - 1, 0, size, 0, size * 8, exp_loc, exp_size, 0, man_size,
                           bias)
    if dtype.kind in 'iu':
        return struct.pack('<BBBBIHH', 0x10, 0x08 if dtype.kind == 'i' else 0x00, 0, 0, dtype.itemsize, 0,
                           dtype.itemsize * 8)
    raise Hdf5Error('cannot write dtype %r' % dtype)


def _space_msg(shape):
    if shape == ():
        return struct.pack('<BBBB4x', 1, 0, 0, 0)
    return struct.pack('<BBBB4x', 1, len(shape), 1, 0) + b''.join(struct.pack('<Q', s) for s in shape) * 2   # dims + max dims


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _attr_msg(name, value):
    if isinstance(value, str):
        value = value.encode('utf8')
    if isinstance(value, bytes):
        arr = np.array(value + b'\0', dtype='S%d' % (len(value) + 1))
    else:
        arr = np.asarray(value)
        if arr.dtype.kind == 'U':
            arr = np.char.encode(arr, 'utf8')
        if arr.dtype.kind == 'S':
            arr = arr.astype('S%d' % (arr.dtype.itemsize + 1))      # room for the terminator of the longest entry
        elif arr.dtype.kind == 'f':
            arr = arr.astype('<f%d' % arr.dtype.itemsize)
        elif arr.dtype.kind in 'iu':
            arr = arr.astype('<%s%d' % (arr.dtype.kind, arr.dtype.itemsize))
        elif arr.dtype.kind == 'b':
            arr = arr.astype('<i1')
    nm = name.encode('utf8') + b'\0'
    dt, sp = _dt_msg(arr.dtype), _space_msg(arr.shape)
    body = struct.pack('<BBHHH', 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + arr.tobytes()
    return body


class _WNode(object):
    def __init__(self):
        self.attrs = {}


class _WGroup(_WNode):
    def __init__(self):
        _WNode.__init__(self)
        self.children = {}

    def create_group(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            nxt = node.children.get(part)
            if nxt is None:
                nxt = node.children[part] = _WGroup()
            if not isinstance(nxt, _WGroup):
                raise Hdf5Error('%r is a dataset' % part)
            node = nxt
        return node

    def create_dataset(self, path, data=None, shape=None, dtype=None):
        """data given: written with the file.  shape + dtype instead: a DEFERRED dataset -- save() only reserves its
        (zero-filled) extent behind the metadata and Writer.open_deferred(d) maps it for writing afterwards, so a series
        of thousands of frames never has to sit in memory."""
        parts = [p for p in path.split('/') if p]
        g = self.create_group('/'.join(parts[:-1])) if len(parts) > 1 else self
        if data is None:
            if shape is None or dtype is None:
                raise Hdf5Error('create_dataset needs data, or shape and dtype')
            d = _WDataset(None, tuple(int(x) for x in shape), np.dtype(dtype))
        else:
            d = _WDataset(np.asarray(data).copy(order='C'))    # (ascontiguousarray would turn a scalar into shape (1,))
        g.children[parts[-1]] = d
        return d


class _WDataset(_WNode):
    def __init__(self, data, shape=None, dtype=None):
        _WNode.__init__(self)
        dt = data.dtype if data is not None else dtype
        if dt.kind == 'f':
            dt = np.dtype('<f%d' % dt.itemsize)
        elif dt.kind in 'iu':
            dt = np.dtype('<%s%d' % (dt.kind, dt.itemsize))
        else:
            raise Hdf5Error('cannot write a dataset of dtype %r' % dt)
        self.data = data.astype(dt) if data is not None else None
        self.shape = self.data.shape if data is not None else shape
        self.dtype = dt
        self.file_offset = None         # deferred datasets: byte offset of the extent, known after save()

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize


class Writer(_WGroup):
    """Build a tree (`create_group`, `create_dataset`, `.attrs[...] = ...`) and `save(path)` it as an HDF5 file."""

    LEAF_K = 4          # symbol-table node holds up to 2K entries; group B-tree rank (internal K) 16
    INTERNAL_K = 16

    def save(self, path):
        self._buf = bytearray(b'\0' * 96)                  # superblock (v0) is patched in at the end
        self._deferred = []
        root_ohdr, root_btree, root_heap = self._write_group(self)
        # deferred extents follow the metadata; their layout messages carried a sentinel address until now
        self._buf += b'\0' * (-len(self._buf) % 8)
        eof = len(self._buf)
        for i, d in enumerate(self._deferred):
            sentinel = struct.pack('<Q', self._SENTINEL + i)
            at = bytes(self._buf).find(sentinel)
            if at < 0 or bytes(self._buf).find(sentinel, at + 1) >= 0:
                raise Hdf5Error('internal: deferred dataset address not found exactly once')
            d.file_offset = eof
            self._buf[at:at + 8] = struct.pack('<Q', eof)
            eof += d.nbytes + (-d.nbytes % 8)
        self._path = path
        sb = bytearray()
        sb += SIGNATURE
        sb += struct.pack('<BBBBBBBB', 0, 0, 0, 0, 0, 8, 8, 0)      # versions, sizes of offsets / lengths
        sb += struct.pack('<HHI', self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack('<QQQQ', 0, UNDEF, eof, UNDEF)  # base, free-space, end of file, driver info
        sb += struct.pack('<QQII', 0, root_ohdr, 1, 0) + struct.pack('<QQ', root_btree, root_heap)
        assert len(sb) == 96
        self._buf[:96] = sb
        with open(path, 'wb') as fp:
            fp.write(bytes(self._buf))
            if eof > len(self._buf):
                fp.truncate(eof)                            # zero-filled (sparse where the filesystem allows)

    _SENTINEL = 0x5EED5EED00000000

    def open_deferred(self, d):
        """Writable memory map of a deferred dataset of the file save() just wrote (flush / delete it when done)."""
        if d.file_offset is None:
            raise Hdf5Error('open_deferred: save() first')
        return np.memmap(self._path, dtype=d.dtype, mode='r+', offset=d.file_offset, shape=d.shape)

    def _alloc(self, data):
        self._buf += b'\0' * (-len(self._buf) % 8)
        addr = len(self._buf)
        self._buf += data
        return addr

    def _object_header(self, msgs):
        """msgs: [(type, body bytes)] -> address.  One block (h5py readers accept any size); every message must stay
        below the 64 KiB limit of a version-1 header message."""
        blob = bytearray()
        for mtype, body in msgs:
            body = _pad8(bytes(body))
            if len(body) >= 65536:
                raise Hdf5Error('object header message of %d bytes exceeds the 64 KiB limit (attribute too large)' % len(body))
            blob += struct.pack('<HHB3x', mtype, len(body), 0) + body
        head = struct.pack('<BBHII4x', 1, 0, len(msgs), 1, len(blob))
        return self._alloc(head + bytes(blob))

    def _write_group(self, g):
        entries = []
        for name in sorted(g.children):                    # B-tree keys are ordered by name
            child = g.children[name]
            if isinstance(child, _WGroup):
                ohdr, bt, hp = self._write_group(child)
                entries.append((name, ohdr, 1, bt, hp))
            else:
                entries.append((name, self._write_dataset(child), 0, 0, 0))
        # local heap: offset 0 holds the empty string (the B-tree's left-most key)
        heap = bytearray(b'\0' * 8)
        offs = {}
        for name, *_ in entries:
            offs[name] = len(heap)
            heap += name.encode('utf8') + b'\0'
            heap += b'\0' * (-len(heap) % 8)
        free_off = len(heap)
        heap += struct.pack('<QQ', 1, 16)                   # one free block: next = 1 (none), size 16
        data_addr = self._alloc(bytes(heap))
        heap_addr = self._alloc(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap), free_off, data_addr))
        # symbol-table nodes of <= 2K entries under one B-tree node of level 0
        per = 2 * self.LEAF_K
        chunks = [entries[i:i + per] for i in range(0, len(entries), per)] or [[]]
        if len(chunks) > 2 * self.INTERNAL_K:
            raise Hdf5Error('group with %d links needs a multi-level B-tree (not written by this subset)' % len(entries))
        snods, keys = [], [0]
        for ch in chunks:
            node = bytearray(b'SNOD' + struct.pack('<BBH', 1, 0, len(ch)))
            for name, ohdr, cache, bt, hp in ch:
                node += struct.pack('<QQII', offs[name], ohdr, cache, 0)
                node += struct.pack('<QQ', bt, hp) if cache == 1 else b'\0' * 16
            node += b'\0' * (40 * (per - len(ch)))
            snods.append(self._alloc(bytes(node)))
            keys.append(offs[ch[-1][0]] if ch else 0)
        used = len(chunks) if entries else 0
        tree = bytearray(b'TREE' + struct.pack('<BBHQQ', 0, 0, used, UNDEF, UNDEF))
        tree += struct.pack('<Q', keys[0])
        for i in range(used):
            tree += struct.pack('<QQ', snods[i], keys[i + 1])
        tree += b'\0' * (16 * (2 * self.INTERNAL_K - used))
        btree_addr = self._alloc(bytes(tree))
        msgs = [(0x0011, struct.pack('<QQ', btree_addr, heap_addr))]
        msgs += [(0x000C, _attr_msg(k, v)) for k, v in g.attrs.items()]
        return self._object_header(msgs), btree_addr, heap_addr

    def _write_dataset(self, d):
        if d.data is None:
            nbytes = d.nbytes
            addr = self._SENTINEL + len(self._deferred) if nbytes else UNDEF
            if nbytes:
                self._deferred.append(d)
        else:
            raw = d.data.tobytes()
            nbytes = len(raw)
            addr = self._alloc(raw) if raw else UNDEF
        msgs = [(0x0001, _space_msg(d.shape)),
                (0x0003, _dt_msg(d.dtype)),
                (0x0005, struct.pack('<BBBB', 2, 2, 0, 0)),                       # fill value v2: late alloc, never write, undefined
                (0x0008, struct.pack('<BBQQ', 3, 1, addr, nbytes))]
        msgs += [(0x000C, _attr_msg(k, v)) for k, v in d.attrs.items()]
        return self._object_header(msgs)
