"""A reader for the HDF5 files the reference's preprocessing writes - own code, because h5py is not in this image.

The reference stores its labels with `h5py.File(path, 'w')` + `hf.create_dataset(name=f'{fn}/adpit/se', data=..., dtype=np.bool_)`
(preproc/preprocess.py:88-129, 197-209, 449-459: bool / int16 / int8 / float32 arrays under `<recording>/<method>/<se|azi|ele>` or
`<recording>/<sed_label|doa_label>`; :559-560 a float32 `feature` array) and reads slices of them back in its datasets
(data/data.py:82-96, 150-161, 210-224). With h5py's defaults that is the ORIGINAL HDF5 layout: superblock version 0, version-1 object
headers, groups as symbol tables (B-tree v1 + local heap + symbol nodes), contiguous little-endian datasets, np.bool_ as an ENUM over int8
{FALSE = 0, TRUE = 1}. This module reads exactly that subset (plus the version-2/3 superblock, version-2 object headers and compact
link-message groups that `libver='latest'` files of this size use) and says so when a file uses anything else (chunked / filtered data,
dense groups, big-endian or compound types): NotImplementedError, never a wrong array.

Pinned by `tests/golden/hdf5/*.h5`: written by the HDF5 C library itself (libhdf5 1.10 of this image through ctypes,
`tests/golden/make_hdf5_golden.py`, with the calls h5py makes for `create_dataset`) from arrays the reference's own label functions
produced (`tests/test_hdf5_lite.py`).

    with File(path) as hf:
        se = hf[f'{fn}/adpit/se'][b:e]          # numpy array (bool for h5py's boolean enum); slicing reads only the rows asked for
        list(hf.keys()); 'name' in hf; hf['grp'].keys(); hf['x'].shape / .dtype
"""
import struct

import numpy as np

_SIG = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(ValueError):
    pass


_MAX_HEADER_BLOCKS, _MAX_HEADER_MESSAGES, _MAX_BTREE_DEPTH = 4096, 1 << 16, 64


class _Reader:
    def __init__(self, f):
        self.f = f
        self.O = self.L = 8
        self.base = 0
        f.seek(0, 2)
        self.size = f.tell()

    def at(self, addr, n):
        # label files are external input to a training run: every block is checked against the file before it is read (a negative or
        # oversized length from a damaged header must raise, not allocate or hang - ADVICE r5)
        if n < 0 or addr < 0 or self.base + addr + n > self.size:
            raise Hdf5FormatError(f'block at {addr} (+{n}) lies outside the file ({self.size} bytes)')
        self.f.seek(self.base + addr)
        b = self.f.read(n)
        if len(b) != n:
            raise Hdf5FormatError(f'short read at {addr} (+{n})')
        return b

    def uint(self, b, off, size):
        return int.from_bytes(b[off:off + size], 'little')


def _numpy_dtype(msg):
    """Datatype message -> (numpy dtype, is_bool_enum)."""
    cls, ver = msg[0] & 0x0F, msg[0] >> 4
    bits0, bits1, _bits2 = msg[1], msg[2], msg[3]
    size = struct.unpack_from('<I', msg, 4)[0]
    if cls == 0:                                     # fixed point
        if bits0 & 1:
            raise NotImplementedError('big-endian integers')
        signed = bool(bits0 & 8)
        if size not in (1, 2, 4, 8):
            raise NotImplementedError(f'{size}-byte integers')
        return np.dtype(('<i' if signed else '<u') + str(size)), False
    if cls == 1:                                     # floating point
        if bits0 & 1:
            raise NotImplementedError('big-endian floats')
        if size not in (2, 4, 8):
            raise NotImplementedError(f'{size}-byte floats')
        return np.dtype('<f' + str(size)), False
    if cls == 8:                                     # enumeration: base type, names, values
        nmemb = bits0 | (bits1 << 8)
        base, _ = _numpy_dtype(msg[8:])
        off = 8 + _datatype_len(msg[8:])
        names = []
        for _ in range(nmemb):
            end = msg.index(b'\x00', off)
            names.append(msg[off:end].decode())
            n = end - off + 1
            off += n if ver >= 3 else (n + 7) // 8 * 8
        vals = np.frombuffer(msg, base, nmemb, off)
        is_bool = base.itemsize == 1 and sorted(zip(vals.tolist(), names)) == [(0, 'FALSE'), (1, 'TRUE')]
        return base, is_bool
    raise NotImplementedError(f'HDF5 datatype class {cls} (only integers, floats and the boolean enum are read)')


def _datatype_len(msg):
    """Bytes a datatype message of class 0 / 1 occupies (header + properties): needed to step over an enum's base type."""
    cls = msg[0] & 0x0F
    if cls == 0:
        return 8 + 4
    if cls == 1:
        return 8 + 12
    raise NotImplementedError('enum over a non-numeric base type')


class Dataset:
    def __init__(self, rd, name, shape, dtype, is_bool, layout):
        self._rd, self.name, self.shape, self._dt, self._bool, self._layout = rd, name, tuple(shape), dtype, is_bool, layout
        self.dtype = np.dtype(bool) if is_bool else dtype

    def __len__(self):
        return self.shape[0]

    def _rows(self, lo, hi):
        """Rows [lo, hi) of the first axis as an array (all of it for a scalar / 1-row request)."""
        kind, a, b = self._layout
        inner = int(np.prod(self.shape[1:], dtype=np.int64)) if len(self.shape) > 1 else 1
        n = (hi - lo) * inner
        if kind == 'compact':
            raw = a[lo * inner * self._dt.itemsize:(lo * inner + n) * self._dt.itemsize]
        else:
            if a == _UNDEF or n == 0:                # never written (no storage allocated): the fill value, zero
                raw = bytes(n * self._dt.itemsize)
            else:
                raw = self._rd.at(a + lo * inner * self._dt.itemsize, n * self._dt.itemsize)
        arr = np.frombuffer(raw, self._dt, n).reshape((hi - lo,) + self.shape[1:])
        return arr.astype(bool) if self._bool else arr.copy()

    def __getitem__(self, key):
        if not self.shape:                           # scalar dataset
            v = self.__class__(self._rd, self.name, (1,), self._dt, self._bool, self._layout)._rows(0, 1)[0]
            return v if key == () else v[key]
        if not isinstance(key, tuple):
            key = (key,)
        first = key[0] if key else slice(None)
        if first is Ellipsis:
            first, key = slice(None), (slice(None),) + tuple(key)
        if isinstance(first, slice):
            lo, hi, step = first.indices(self.shape[0])
            if step == 1:
                out = self._rows(lo, max(lo, hi))
                return out[(slice(None),) + tuple(key[1:])] if len(key) > 1 else out
        elif isinstance(first, (int, np.integer)):
            i = int(first) + (self.shape[0] if first < 0 else 0)
            if not 0 <= i < self.shape[0]:
                raise IndexError(first)
            out = self._rows(i, i + 1)[0]
            return out[tuple(key[1:])] if len(key) > 1 else out
        return self._rows(0, self.shape[0])[key]     # anything else: read it all, let numpy index

    def __array__(self, dtype=None, copy=None):
        a = self[...] if self.shape else np.asarray(self[()])
        return a.astype(dtype) if dtype is not None else a


class Group:
    def __init__(self, file, name, links):
        self._file, self.name, self._links = file, name, links

    def keys(self):
        return list(self._links)

    def __iter__(self):
        return iter(self._links)

    def __len__(self):
        return len(self._links)

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(f'{path!r}: no {part!r} in {node.name!r}')
            node = self._file._object(node._links[part], (node.name.rstrip('/') + '/' + part))
        return node


class File(Group):
    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise NotImplementedError('hdf5_lite reads; the reference writes its label files once, offline')
        self._fh = open(path, 'rb')
        self._rd = _Reader(self._fh)
        self._cache = {}
        try:
            root = self._superblock()
        except Exception:
            self._fh.close()
            raise
        Group.__init__(self, self, '/', root)

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- superblock ---------------------------------------------------------------------------------------------------------------
    def _superblock(self):
        rd, off = self._rd, 0
        while True:                                   # the signature sits at 0, 512, 1024, 2048, ...
            rd.f.seek(off)
            head = rd.f.read(9)
            if head[:8] == _SIG:
                break
            if len(head) < 9 or off > (1 << 24):
                raise Hdf5FormatError('not an HDF5 file (no signature)')
            off = 512 if off == 0 else off * 2
        ver = head[8]
        if ver in (0, 1):
            b = rd.at(off, 24 + (4 if ver == 1 else 0))
            rd.O, rd.L = b[13], b[14]
            p = off + 24 + (4 if ver == 1 else 0)
            addrs = rd.at(p, 4 * rd.O)
            rd.base = rd.uint(addrs, 0, rd.O)
            entry = rd.at(p + 4 * rd.O - rd.base, 2 * rd.O + 8 + 16)
            ohdr = rd.uint(entry, rd.O, rd.O)
            cache_type = struct.unpack_from('<I', entry, 2 * rd.O)[0]
            if cache_type == 1:                       # B-tree and heap addresses cached in the entry's scratch pad
                return self._symbol_table(rd.uint(entry, 2 * rd.O + 8, rd.O), rd.uint(entry, 3 * rd.O + 8, rd.O))
            return self._group_links(ohdr)
        if ver in (2, 3):
            b = rd.at(off, 12)
            rd.O, rd.L = b[9], b[10]
            addrs = rd.at(off + 12, 4 * rd.O)
            rd.base = rd.uint(addrs, 0, rd.O)
            return self._group_links(rd.uint(addrs, 3 * rd.O, rd.O))
        raise NotImplementedError(f'HDF5 superblock version {ver}')

    # ---- object headers -----------------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, bytes)] of the object header at addr, continuation blocks followed."""
        rd = self._rd
        head = rd.at(addr, 16)
        out = []
        if head[:4] == b'OHDR':                       # version 2
            flags = head[5]
            p = 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
            szlen = 1 << (flags & 3)
            head = rd.at(addr, p + szlen)
            size0 = rd.uint(head, p, szlen)
            blocks = [(addr + p + szlen, size0)]
            track = bool(flags & 4)
            seen = 0
            while blocks:
                seen += 1
                if seen > _MAX_HEADER_BLOCKS or len(out) > _MAX_HEADER_MESSAGES:          # (a continuation chain that loops or never ends)
                    raise Hdf5FormatError(f'object header at {addr}: more than {_MAX_HEADER_BLOCKS} continuation blocks / {_MAX_HEADER_MESSAGES} messages')
                a, n = blocks.pop(0)
                b = rd.at(a, n)
                q = 0
                while q + 4 <= n:
                    t, sz, _fl = b[q], struct.unpack_from('<H', b, q + 1)[0], b[q + 3]
                    q += 4 + (2 if track else 0)
                    body = b[q:q + sz]
                    q += sz
                    if t == 0x10:
                        ca, cl = rd.uint(body, 0, rd.O), rd.uint(body, rd.O, rd.L)
                        blocks.append((ca + 4, cl - 8))           # "OCHK" ... checksum
                    elif t != 0:
                        out.append((t, body))
            return out
        if head[0] != 1:
            raise Hdf5FormatError(f'object header at {addr}: version {head[0]}')
        nmsg = struct.unpack_from('<H', head, 2)[0]
        size = struct.unpack_from('<I', head, 8)[0]
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and len(out) < nmsg + 64:
            seen += 1
            if seen > _MAX_HEADER_BLOCKS:
                raise Hdf5FormatError(f'object header at {addr}: more than {_MAX_HEADER_BLOCKS} continuation blocks')
            a, n = blocks.pop(0)
            b = rd.at(a, n)
            q = 0
            while q + 8 <= n:
                t, sz = struct.unpack_from('<HH', b, q)
                body = b[q + 8:q + 8 + sz]
                q += 8 + sz
                if t == 0x10:
                    blocks.append((rd.uint(body, 0, rd.O), rd.uint(body, rd.O, rd.L)))
                elif t != 0:
                    out.append((t, body))
        return out

    def _object(self, addr, name):
        if addr in self._cache:
            return self._cache[addr]
        msgs = self._messages(addr)
        types = {t for t, _ in msgs}
        if 0x11 in types or 0x02 in types or (0x06 in types and 0x08 not in types) or not (types & {0x01, 0x03, 0x08}):
            obj = Group(self, name, self._group_links(addr, msgs))
        else:
            obj = self._dataset(msgs, name)
        self._cache[addr] = obj
        return obj

    # ---- groups -------------------------------------------------------------------------------------------------------------------
    def _group_links(self, addr, msgs=None):
        msgs = self._messages(addr) if msgs is None else msgs
        links = {}
        for t, body in msgs:
            if t == 0x11:                              # symbol table message: the original group layout
                links.update(self._symbol_table(self._rd.uint(body, 0, self._rd.O), self._rd.uint(body, self._rd.O, self._rd.O)))
            elif t == 0x06:                            # link message (compact new-style group)
                name, target = self._link(body)
                links[name] = target
            elif t == 0x02:                            # link info: a fractal-heap ("dense") group when the heap address is set
                rd = self._rd
                p = 2 + (8 if body[1] & 1 else 0)
                if rd.uint(body, p, rd.O) != _UNDEF:
                    raise NotImplementedError('dense (fractal heap) groups; the reference writes original-format files')
        return links

    def _link(self, body):
        rd = self._rd
        flags = body[1]
        p = 2
        ltype = 0
        if flags & 8:
            ltype = body[p]; p += 1
        if flags & 4:
            p += 8
        if flags & 16:
            p += 1
        n = 1 << (flags & 3)
        ln = rd.uint(body, p, n); p += n
        name = body[p:p + ln].decode(); p += ln
        if ltype != 0:
            raise NotImplementedError('soft / external links')
        return name, rd.uint(body, p, rd.O)

    def _symbol_table(self, btree, heap):
        rd = self._rd
        h = rd.at(heap, 8 + 2 * rd.L + rd.O)
        if h[:4] != b'HEAP':
            raise Hdf5FormatError('local heap signature')
        seg_size, seg = rd.uint(h, 8, rd.L), rd.uint(h, 8 + 2 * rd.L, rd.O)
        names = rd.at(seg, seg_size)
        links = {}
        visited = set()

        def walk(addr, depth=0):
            if addr in visited or depth > _MAX_BTREE_DEPTH:               # (a child pointer that leads back into the tree)
                raise Hdf5FormatError(f'group B-tree: node {addr} revisited or deeper than {_MAX_BTREE_DEPTH} levels')
            visited.add(addr)
            b = rd.at(addr, 8 + 2 * rd.O)
            if b[:4] == b'TREE':
                if b[4] != 0:
                    raise Hdf5FormatError('group B-tree node type')
                used = struct.unpack_from('<H', b, 6)[0]
                body = rd.at(addr + 8 + 2 * rd.O, (used + 1) * rd.L + used * rd.O)         # key 0, child 0, key 1, ..., key `used`
                q = rd.L
                for _ in range(used):
                    child = rd.uint(body, q, rd.O)
                    q += rd.O + rd.L
                    walk(child, depth + 1)                  # (level > 0: further TREE nodes; level 0: symbol nodes)
            elif b[:4] == b'SNOD':
                n = struct.unpack_from('<H', b, 6)[0]
                ent = rd.at(addr + 8, n * (2 * rd.O + 24))
                for i in range(n):
                    e = i * (2 * rd.O + 24)
                    off, ohdr = rd.uint(ent, e, rd.O), rd.uint(ent, e + rd.O, rd.O)
                    links[names[off:names.index(b'\x00', off)].decode()] = ohdr
            else:
                raise Hdf5FormatError(f'group node signature {b[:4]!r}')
        walk(btree)
        return links

    # ---- datasets -----------------------------------------------------------------------------------------------------------------
    def _dataset(self, msgs, name):
        rd = self._rd
        shape = dtype = layout = None
        is_bool = False
        for t, body in msgs:
            if t == 0x01:
                ver, rank, flags = body[0], body[1], body[2]
                p = 8 if ver == 1 else 4
                shape = tuple(rd.uint(body, p + i * rd.L, rd.L) for i in range(rank))
            elif t == 0x03:
                dtype, is_bool = _numpy_dtype(bytes(body))
            elif t == 0x08:
                ver = body[0]
                if ver in (3, 4):                      # (4 = the 1.10 message: same fields for compact and contiguous storage)
                    cls = body[1]
                    if cls == 0:
                        n = struct.unpack_from('<H', body, 2)[0]
                        layout = ('compact', bytes(body[4:4 + n]), n)
                    elif cls == 1:
                        layout = ('contiguous', rd.uint(body, 2, rd.O), rd.uint(body, 2 + rd.O, rd.L))
                    else:
                        raise NotImplementedError(f'{name}: chunked storage (the reference writes contiguous datasets: no chunks / compression given)')
                elif ver in (1, 2):
                    rank, cls = body[1], body[2]
                    if cls == 1:
                        layout = ('contiguous', rd.uint(body, 8, rd.O), 0)
                    elif cls == 0:
                        p = 8 + 4 * rank
                        n = struct.unpack_from('<I', body, p)[0]
                        layout = ('compact', bytes(body[p + 4:p + 4 + n]), n)
                    else:
                        raise NotImplementedError(f'{name}: chunked storage')
                else:
                    raise NotImplementedError(f'{name}: data layout message version {ver}')
            elif t == 0x0B:
                raise NotImplementedError(f'{name}: filtered (compressed) data')
        if shape is None or dtype is None or layout is None:
            raise Hdf5FormatError(f'{name}: dataset without dataspace / datatype / layout')
        return Dataset(rd, name, shape, dtype, is_bool, layout)
