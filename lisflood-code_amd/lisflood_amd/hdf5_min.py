"""A minimal HDF5 writer / reader in pure Python (struct + zlib): just enough of the format for the netCDF-4 files the
reference writes with `writenet` (global_modules/netcdf.py:432-583): float / int datasets, contiguous or chunked with
shuffle + deflate, fill values, string / numeric attributes, and the dimension-scale attributes (CLASS, NAME,
_Netcdf4Dimid, REFERENCE_LIST, DIMENSION_LIST) that make a netCDF-4 reader see named dimensions.

File layout (HDF5 File Format Specification 2.0, the "earliest" structures every library version reads): superblock
version 1, one old-style root group (version-1 object header with a symbol-table message, one B-tree node, one symbol
node, a local heap), version-1 object headers for the datasets, version-3 layout messages, version-1 B-tree for the
chunks (one leaf: the superblock's indexed-storage K is raised to fit), version-1 filter pipeline, one global heap
collection for the variable-length object references of DIMENSION_LIST.  Nothing is ever modified in place: the whole
file is laid out in memory and written once.

The reader handles what this writer and libhdf5's `libver='earliest'` produce (continuation blocks, multi-level chunk
B-trees, contiguous data) -- it is there so that the files can be checked without an HDF5 library in the image."""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


# ---- datatype messages ---------------------------------------------------------------------------------------------
def _dt_float(size):
    if size == 8:
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)


def _dt_int32():
    return struct.pack("<BBBBIHH", 0x10, 0x08, 0, 0, 4, 0, 32)


def _dt_string(n):        # fixed length, null terminated, ASCII
    return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, n)


def _dt_objref():
    return struct.pack("<BBBBI", 0x17, 0, 0, 0, 8)


def _dt_vlen_objref():
    return struct.pack("<BBBBI", 0x19, 0, 0, 0, 16) + _dt_objref()


def _dt_reference_list():  # compound {dataset: object reference @0, dimension: int32 @8}, 16 bytes
    def member(name, offset, dt):
        return _pad8(name + b"\0") + struct.pack("<IB3xI4x16x", offset, 0, 0) + dt
    return (struct.pack("<BBBBI", 0x16, 2, 0, 0, 16) + member(b"dataset", 0, _dt_objref()) +
            member(b"dimension", 8, _dt_int32()))


def _dt_of(arr):
    if arr.dtype == np.float64:
        return _dt_float(8)
    if arr.dtype == np.float32:
        return _dt_float(4)
    if arr.dtype == np.int32:
        return _dt_int32()
    raise TypeError("unsupported dtype %s" % arr.dtype)


def _dataspace(shape):
    if len(shape) == 0:
        return struct.pack("<BBB5x", 1, 0, 0)
    return struct.pack("<BBB5x", 1, len(shape), 1) + b"".join(struct.pack("<Q", n) for n in shape) * 2


def _attr_message(name, dt, ds, data):
    nm = name.encode() + b"\0"
    return struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + data


def _attr_value(name, value):
    """-> attribute message body for a python / numpy value"""
    if isinstance(value, (bytes, str)):
        b = value.encode() if isinstance(value, str) else value
        b = b + b"\0"
        return _attr_message(name, _dt_string(len(b)), _dataspace(()), b)
    a = np.asarray(value)
    if a.dtype.kind in "iub":
        a = a.astype(np.int32)
    elif a.dtype.kind == "f" and a.dtype != np.float32:
        a = a.astype(np.float64)
    return _attr_message(name, _dt_of(a), _dataspace(a.shape if a.ndim else ()), a.astype(a.dtype.newbyteorder("<")).tobytes())


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages):
    data = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(data)) + data


class Dataset:
    """One variable of the file.  `dims`: names of its dimensions (each must be a 1-D dataset of the file = its dimension
    scale), `chunks` + `deflate` (+ `shuffle`) for compressed chunked storage, `fill` for the HDF5 fill value.
    `data` None with `shape` and `dtype`: a streamed dataset -- chunked, its chunks handed to Writer.write_chunk one at
    a time after the file's metadata is on disk (chunks never written read back as the fill value)."""

    def __init__(self, name, data, dims=(), attrs=None, chunks=None, deflate=None, shuffle=False, fill=None,
                 shape=None, dtype=None):
        if data is None:
            if shape is None or dtype is None or chunks is None:
                raise ValueError("dataset %s: a streamed dataset needs shape, dtype and chunks" % name)
            self.data, self.shape, self.dtype = None, tuple(int(n) for n in shape), np.dtype(dtype)
        else:
            data = np.asarray(data)
            self.data = data if data.flags.c_contiguous else data.copy(order="C")
            self.shape, self.dtype = self.data.shape, self.data.dtype
        self.name, self.dims = name, tuple(dims)
        self.attrs = dict(attrs or {})
        self.chunks, self.deflate, self.shuffle, self.fill = chunks, deflate, shuffle, fill
        if self.dtype not in (np.float64, np.float32, np.int32):
            raise TypeError("dataset %s: dtype %s not supported" % (name, self.dtype))
        if chunks is not None and len(chunks) != len(self.shape):
            raise ValueError("dataset %s: chunk rank" % name)
        if (deflate is not None or shuffle) and chunks is None:
            raise ValueError("dataset %s: filters need chunked storage" % name)

    @property
    def streamed(self):
        return self.data is None

    @property
    def chunk_counts(self):
        return tuple(-(-n // c) for n, c in zip(self.shape, self.chunks))


def _encode_block(ds, block):
    """one chunk (an array of the chunk's shape) -> its bytes in the file: little endian, shuffle, deflate"""
    raw = np.ascontiguousarray(block, dtype=ds.dtype.newbyteorder("<")).tobytes()
    if ds.shuffle:
        raw = np.frombuffer(raw, np.uint8).reshape(-1, ds.dtype.itemsize).T.tobytes()
    if ds.deflate is not None:
        raw = zlib.compress(raw, ds.deflate)
    return raw


def _encode_chunks(ds):
    """-> list of (element offsets, filtered bytes) in B-tree key order (row-major chunk index)"""
    a, ch = ds.data, ds.chunks
    out = []
    for idx in np.ndindex(*ds.chunk_counts):
        off = tuple(i * c for i, c in zip(idx, ch))
        block = np.full(ch, ds.fill if ds.fill is not None else 0, dtype=a.dtype)
        sl = tuple(slice(o, min(o + c, n)) for o, c, n in zip(off, ch, a.shape))
        block[tuple(slice(0, s.stop - s.start) for s in sl)] = a[sl]
        out.append((off, _encode_block(ds, block)))
    return out


def _chunk_node(ds, entries):
    """the (single) chunk B-tree node of dataset ds; entries: [(element offsets, byte size, address)] in key order"""
    node = b"TREE" + struct.pack("<BBHQQ", 1, 0, len(entries), UNDEF, UNDEF)
    for off, size, at in entries:
        node += struct.pack("<II", size, 0) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", 0)
        node += struct.pack("<Q", at)
    # the key behind the last chunk: its offsets plus one chunk in every dimension, as libhdf5 writes it
    last = entries[-1][0] if entries else (0,) * len(ds.chunks)
    end_key = tuple(o + c for o, c in zip(last, ds.chunks)) if entries else last
    return node + struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", o) for o in end_key) + \
        struct.pack("<Q", ds.dtype.itemsize)


def write(path, datasets, root_attrs=None):
    """Write `datasets` (list of Dataset; dimension scales before their users is the netCDF habit, not a requirement) and
    the root group's attributes to `path`."""
    if any(d.streamed for d in datasets):
        raise ValueError("streamed datasets are written through Writer")
    Writer(path, datasets, root_attrs).close()


class Writer:
    """The file with every in-memory dataset written at construction; the chunks of streamed datasets (Dataset(data=None,
    shape=, dtype=, chunks=)) follow one by one through write_chunk and land behind everything else; close() (or flush())
    rewrites the chunk B-tree nodes of the streamed datasets and the end-of-file address.  Memory: one chunk."""

    def __init__(self, path, datasets, root_attrs=None):
        self._f = None
        self._layout(path, datasets, root_attrs)

    def write_chunk(self, name, index, block):
        """chunk `index` (chunk coordinates, e.g. (t, 0, 0)) of the streamed dataset `name`; `block`: the chunk's shape
        (smaller at the upper edges of the dataset: the rest is the fill value)"""
        if self._f is None:
            raise ValueError("the file is closed")
        d = self._streamed[name]
        index = tuple(int(i) for i in index)
        if len(index) != len(d.chunks) or any(i < 0 or i >= n for i, n in zip(index, d.chunk_counts)):
            raise IndexError("dataset %s: chunk %s outside %s" % (name, index, d.chunk_counts))
        if index in self._entries[name]:
            raise ValueError("dataset %s: chunk %s written twice" % (name, index))
        block = np.asarray(block)
        if block.shape != tuple(d.chunks):
            full = np.full(d.chunks, d.fill if d.fill is not None else 0, dtype=d.dtype)
            if block.ndim != full.ndim or any(b > c for b, c in zip(block.shape, d.chunks)):
                raise ValueError("dataset %s: block shape %s, chunk %s" % (name, block.shape, tuple(d.chunks)))
            full[tuple(slice(0, n) for n in block.shape)] = block
            block = full
        raw = _encode_block(d, block)
        self._f.seek(self._eof)
        self._f.write(raw)
        self._entries[name][index] = (len(raw), self._eof)
        self._eof += len(raw)

    def flush(self):
        """B-tree nodes and end-of-file address as of now: the file on disk is complete up to the chunks written so far"""
        if self._f is None:
            return
        for name, d in self._streamed.items():
            ent = [(tuple(i * c for i, c in zip(idx, d.chunks)), size, at)
                   for idx, (size, at) in sorted(self._entries[name].items())]
            self._f.seek(self._where[("btree", name)])
            self._f.write(_chunk_node(d, ent))
        eof = (self._eof + 7) // 8 * 8
        if eof > self._eof:
            self._f.seek(self._eof)
            self._f.write(b"\0" * (eof - self._eof))
        self._f.seek(44)                                             # superblock version 1: end-of-file address
        self._f.write(struct.pack("<Q", eof))
        self._f.flush()

    def close(self):
        if self._f is not None:
            self.flush()
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):   # a writer dropped without close() (an exception in the caller's step loop): the chunk index and
        try:             # the end-of-file address still reach the file, the steps written so far stay readable
            self.close()
        except Exception:
            pass

    def _layout(self, path, datasets, root_attrs):
        names = [d.name for d in datasets]
        if len(set(names)) != len(names):
            raise ValueError("duplicate dataset names")
        by_name = {d.name: d for d in datasets}
        dimid = {}
        for d in datasets:                       # dimension scales: every name used as a dimension
            for n in d.dims:
                if n not in by_name or len(by_name[n].shape) != 1:
                    raise ValueError("dimension %s of %s is not a 1-D dataset of the file" % (n, d.name))
                dimid.setdefault(n, None)
        for i, n in enumerate(n for n in names if n in dimid):
            dimid[n] = i                        # netCDF numbers dimensions in creation order
        leaf_k = 16
        if len(datasets) > 2 * leaf_k:
            raise ValueError("too many datasets for one symbol node")
        chunks = {d.name: _encode_chunks(d) for d in datasets if d.chunks is not None and not d.streamed}
        nchunks = [int(np.prod(d.chunk_counts)) for d in datasets if d.chunks is not None]
        istore_k = max(32, max([(n + 1) // 2 + 1 for n in nchunks] or [0]))
        if istore_k > 65535:
            raise ValueError("too many chunks for a single B-tree node")
        # users of every dimension scale: (dataset name, axis) -> REFERENCE_LIST
        users = {n: [] for n in dimid}
        for d in datasets:
            for ax, n in enumerate(d.dims):
                if n != d.name:
                    users[n].append((d.name, ax))

        def header_of(d, addr, where):
            """object header bytes of dataset d; addr: name -> object header address, where: data / btree / heap addresses"""
            a = d
            msgs = [_message(0x01, _dataspace(a.shape)), _message(0x03, _dt_of(a), 1)]
            if d.fill is not None:
                fv = np.array(d.fill, dtype=a.dtype.newbyteorder("<")).tobytes()
                msgs.append(_message(0x05, struct.pack("<BBBBI", 2, 3 if d.chunks else 2, 0, 1, len(fv)) + fv, 1))
                msgs.append(_message(0x04, struct.pack("<I", len(fv)) + fv, 1))
            else:
                msgs.append(_message(0x05, struct.pack("<BBBBI", 2, 3 if d.chunks else 2, 2, 1, 0)[:8], 1))
            if d.chunks is not None:
                flt = b""
                nf = 0
                if d.shuffle:
                    flt += struct.pack("<HHHH", 2, 8, 1, 1) + b"shuffle\0" + struct.pack("<I4x", a.dtype.itemsize)
                    nf += 1
                if d.deflate is not None:
                    flt += struct.pack("<HHHH", 1, 8, 1, 1) + b"deflate\0" + struct.pack("<I4x", d.deflate)
                    nf += 1
                if nf:
                    msgs.append(_message(0x0B, struct.pack("<BB6x", 1, nf) + flt, 1))
                msgs.append(_message(0x08, struct.pack("<BBBQ", 3, 2, len(a.shape) + 1, where.get(("btree", d.name), 0)) +
                                     b"".join(struct.pack("<I", c) for c in d.chunks) + struct.pack("<I", a.dtype.itemsize)))
            else:
                msgs.append(_message(0x08, struct.pack("<BBQQ", 3, 1, where.get(("data", d.name), 0), d.data.nbytes)))
            if d.name in dimid:
                msgs.append(_message(0x0C, _attr_value("CLASS", "DIMENSION_SCALE")))
                msgs.append(_message(0x0C, _attr_value("NAME", d.name)))
                msgs.append(_message(0x0C, _attr_value("_Netcdf4Dimid", np.int32(dimid[d.name]))))
                if users[d.name]:
                    body = b"".join(struct.pack("<Qi4x", addr.get(u, 0), ax) for u, ax in users[d.name])
                    msgs.append(_message(0x0C, _attr_message("REFERENCE_LIST", _dt_reference_list(),
                                                             _dataspace((len(users[d.name]),)), body)))
            real_dims = [n for n in d.dims]
            if real_dims and not (len(real_dims) == 1 and real_dims[0] == d.name):
                body = b"".join(struct.pack("<IQI", 1, where.get("gcol", 0), where.get(("gidx", d.name, ax), 0))
                                for ax in range(len(real_dims)))
                msgs.append(_message(0x0C, _attr_message("DIMENSION_LIST", _dt_vlen_objref(), _dataspace((len(real_dims),)), body)))
                if len(real_dims) > 1:
                    msgs.append(_message(0x0C, _attr_value("_Netcdf4Coordinates", np.array([dimid[n] for n in real_dims], np.int32))))
            for k, v in d.attrs.items():
                msgs.append(_message(0x0C, _attr_value(k, v)))
            return _object_header(msgs)

        # ---- pass 1: sizes with placeholder addresses, then the layout ----
        pos = 100                                                    # superblock version 1
        addr, where = {}, {}

        def take(n, align=8):
            nonlocal pos
            pos = (pos + align - 1) // align * align
            at = pos
            pos += n
            return at

        root_msgs_len = None
        heap_names = [b""] + [n.encode() for n in names]
        heap_off, blob = {}, b""
        for n in heap_names:
            heap_off[n] = len(blob)
            blob += _pad8(n + b"\0")
        heap_data = blob + struct.pack("<QQ", 1, 32) + b"\0" * 16     # one free block of 32 bytes behind the names
        snod_size = 8 + 40 * 2 * leaf_k
        btree_size = 24 + 2 * 16 * 16 + 8                             # group node: 2 * internal K (16) entries
        root_attr_msgs = [_message(0x0C, _attr_value(k, v)) for k, v in (root_attrs or {}).items()]

        def root_header(bt, hp):
            return _object_header([_message(0x11, struct.pack("<QQ", bt, hp))] + root_attr_msgs)

        root_at = take(len(root_header(0, 0)))
        bt_at = take(btree_size)
        hp_at = take(32)
        hd_at = take(len(heap_data))
        sn_at = take(snod_size)
        # global heap: one object (8-byte reference) per (dataset, axis) with dimensions
        gobjs = [(d.name, ax, n) for d in datasets for ax, n in enumerate(d.dims) if not (len(d.dims) == 1 and d.dims[0] == d.name)]
        gsize = max(4096, 16 + 24 * len(gobjs) + 16)
        if gobjs:
            where["gcol"] = take(gsize)
            for i, (dn, ax, _n) in enumerate(gobjs):
                where[("gidx", dn, ax)] = i + 1
        for d in datasets:
            addr[d.name] = take(len(header_of(d, {}, where)))
            if d.chunks is None:
                where[("data", d.name)] = take(d.data.nbytes)
            else:
                rank1 = len(d.shape) + 1
                # a node is always read at its full size: 2K (key, child) pairs and the closing key
                where[("btree", d.name)] = take(24 + 2 * istore_k * (8 + 8 * rank1 + 8) + (8 + 8 * rank1))
                for i, (_off, raw) in enumerate(chunks.get(d.name, ())):
                    where[("chunk", d.name, i)] = take(len(raw), 1)
        eof = (pos + 7) // 8 * 8

        # ---- pass 2: serialise ----
        out = bytearray(eof)

        def put(at, b):
            out[at:at + len(b)] = b

        sb = (SIGNATURE + struct.pack("<BBBBBBBB", 1, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", leaf_k, 16, 0) +
              struct.pack("<HH", istore_k, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF) +
              struct.pack("<QQII", 0, root_at, 1, 0) + struct.pack("<QQ", bt_at, hp_at))
        assert len(sb) == 100
        put(0, sb)
        put(root_at, root_header(bt_at, hp_at))
        ordered = sorted(names, key=lambda n: n.encode())
        last_key = heap_off[ordered[-1].encode()] if ordered else 0
        put(bt_at, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, sn_at, last_key))
        put(hp_at, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), len(blob), hd_at))
        put(hd_at, heap_data)
        sn = b"SNOD" + struct.pack("<BxH", 1, len(ordered))
        for n in ordered:
            sn += struct.pack("<QQII16x", heap_off[n.encode()], addr[n], 0, 0)
        put(sn_at, sn)
        if gobjs:
            g = b"GCOL" + struct.pack("<B3xQ", 1, gsize)
            for i, (_dn, _ax, scale) in enumerate(gobjs):
                g += struct.pack("<HH4xQ", i + 1, 0, 8) + struct.pack("<Q", addr[scale])
            g += struct.pack("<HH4xQ", 0, 0, gsize - len(g))
            put(where["gcol"], g)
        for d in datasets:
            put(addr[d.name], header_of(d, addr, where))
            if d.chunks is None:
                put(where[("data", d.name)], d.data.astype(d.data.dtype.newbyteorder("<")).tobytes())
            elif d.streamed:
                put(where[("btree", d.name)], _chunk_node(d, []))
            else:
                ent = []
                for i, (off, raw) in enumerate(chunks[d.name]):
                    put(where[("chunk", d.name, i)], raw)
                    ent.append((off, len(raw), where[("chunk", d.name, i)]))
                put(where[("btree", d.name)], _chunk_node(d, ent))
        self._streamed = {d.name: d for d in datasets if d.streamed}
        self._entries = {n: {} for n in self._streamed}
        self._where, self._eof = where, eof
        self._f = open(path, "wb")
        self._f.write(bytes(out))


# =====================================================================================================================
# reader
# =====================================================================================================================
class _Reader:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.d = f.read()
        d = self.d
        if d[:8] != SIGNATURE or d[8] > 1:
            raise ValueError("not an HDF5 file with a version 0/1 superblock")
        o = 24 if d[8] == 0 else 28
        ste = o + 32
        self.root = self.u(ste + 8, 8)
        self.objects = {}
        for _t, body, _s in self.messages(self.root):
            if _t == 0x11:
                self._walk_group(self.u(body, 8), self.u(body + 8, 8))

    def u(self, off, n):
        return int.from_bytes(self.d[off:off + n], "little")

    def messages(self, a):
        """-> [(type, body offset, size)] of the version-1 object header at a, continuation blocks included"""
        d = self.d
        if d[a] != 1:
            raise ValueError("object header version %d not supported" % d[a])
        nmsg, size = self.u(a + 2, 2), self.u(a + 8, 4)
        blocks, out = [(a + 16, a + 16 + size)], []
        while blocks and len(out) < nmsg:
            p, end = blocks.pop(0)
            while p + 8 <= end and len(out) < nmsg:
                t, s = self.u(p, 2), self.u(p + 2, 2)
                if t == 0x10:
                    ca, cl = self.u(p + 8, 8), self.u(p + 16, 8)
                    blocks.append((ca, ca + cl))
                out.append((t, p + 8, s))
                p += 8 + s
        return out

    def _walk_group(self, bt, hp):
        d = self.d
        hd = self.u(hp + 24, 8)
        if d[bt:bt + 4] != b"TREE":
            raise ValueError("bad group B-tree")
        level, used = d[bt + 5], self.u(bt + 6, 2)
        for i in range(used):
            child = self.u(bt + 24 + 16 * i + 8, 8)
            if level > 0:
                self._walk_group(child, hp)
                continue
            for j in range(self.u(child + 6, 2)):
                e = child + 8 + 40 * j
                name = d[hd + self.u(e, 8):].split(b"\0", 1)[0].decode()
                self.objects[name] = self.u(e + 8, 8)

    @staticmethod
    def _dtype(body, d):
        cls, size = d[body] & 0x0F, int.from_bytes(d[body + 4:body + 8], "little")
        if cls == 1:
            return np.dtype("<f%d" % size)
        if cls == 0:
            return np.dtype("<i%d" % size if d[body + 1] & 0x08 else "<u%d" % size)
        if cls == 3:
            return np.dtype("S%d" % size)
        return None

    def _shape(self, body):
        rank = self.d[body + 1]
        return tuple(self.u(body + 8 + 8 * i, 8) for i in range(rank))

    def attrs(self, name=None):
        a = self.root if name is None else self.objects[name]
        out = {}
        for t, body, s in self.messages(a):
            if t != 0x0C:
                continue
            ns, ts, ss = self.u(body + 2, 2), self.u(body + 4, 2), self.u(body + 6, 2)
            p = body + 8
            nm = self.d[p:p + ns].split(b"\0", 1)[0].decode()
            p += (ns + 7) // 8 * 8
            dt = self._dtype(p, self.d)
            dtp = p
            p += (ts + 7) // 8 * 8
            shape = self._shape(p)
            p += (ss + 7) // 8 * 8
            if dt is None:
                out[nm] = ("opaque", self.d[dtp] & 0x0F, shape, self.d[p:body + s])
                continue
            n = int(np.prod(shape)) if shape else 1
            v = np.frombuffer(self.d, dt, n, p).reshape(shape)
            if dt.kind == "S":
                v = v.reshape(-1)[0].split(b"\0", 1)[0].decode() if not shape else [x.decode() for x in v]
            elif not shape:
                v = v.reshape(-1)[0]
            out[nm] = v
        return out

    def info(self, name):
        """-> dict(shape, dtype, chunks, filters [(id, values)], fill) of a dataset: the storage side of its header"""
        d = self.d
        out = dict(shape=None, dtype=None, chunks=None, filters=[], fill=None)
        for t, body, _s in self.messages(self.objects[name]):
            if t == 0x01:
                out["shape"] = self._shape(body)
            elif t == 0x03:
                out["dtype"] = self._dtype(body, d)
            elif t == 0x05 and d[body] == 2 and d[body + 3] == 1 and self.u(body + 4, 4) > 0 and out["dtype"] is not None:
                out["fill"] = np.frombuffer(d, out["dtype"], 1, body + 8)[0]
            elif t == 0x08 and d[body + 1] == 2:
                rank1 = d[body + 2]
                out["chunks"] = tuple(self.u(body + 11 + 4 * i, 4) for i in range(rank1 - 1))
            elif t == 0x0B:
                p = body + 8
                for _ in range(d[body + 1]):
                    fid, nlen, _fl, nv = (self.u(p + 2 * k, 2) for k in range(4))
                    p += 8 + (nlen + 7) // 8 * 8
                    out["filters"].append((fid, [self.u(p + 4 * k, 4) for k in range(nv)]))
                    p += 4 * nv + (4 if nv % 2 else 0)
        return out

    def dataset(self, name):
        d = self.d
        dt = shape = layout = None
        filters = []
        for t, body, _s in self.messages(self.objects[name]):
            if t == 0x01:
                shape = self._shape(body)
            elif t == 0x03:
                dt = self._dtype(body, d)
            elif t == 0x08:
                layout = body
            elif t == 0x0B:
                p = body + 8
                for _ in range(d[body + 1]):
                    fid, nlen, _fl, nv = (self.u(p + 2 * k, 2) for k in range(4))
                    p += 8 + (nlen + 7) // 8 * 8
                    vals = [self.u(p + 4 * k, 4) for k in range(nv)]
                    p += 4 * nv + (4 if nv % 2 else 0)
                    filters.append((fid, vals))
        if d[layout] != 3:
            raise ValueError("layout version %d not supported" % d[layout])
        if d[layout + 1] == 1:
            at = self.u(layout + 2, 8)
            return np.frombuffer(d, dt, int(np.prod(shape)), at).reshape(shape).copy()
        rank1 = d[layout + 2]
        bt = self.u(layout + 3, 8)
        cdims = tuple(self.u(layout + 11 + 4 * i, 4) for i in range(rank1 - 1))
        # chunks that were never written (libhdf5's incremental allocation) have no entry in the B-tree: they read as the
        # dataset's fill value -- -9999 in LISFLOOD's maps, the cold-start marker of read_state_maps -- not as zeros
        fill = self.info(name)["fill"]
        out = np.zeros(shape, dt) if fill is None else np.full(shape, fill, dt)
        if bt == 0xFFFFFFFFFFFFFFFF:      # undefined address: no chunk allocated at all
            return out

        def walk(node):
            level, used = d[node + 5], self.u(node + 6, 2)
            p = node + 24
            ksz = 8 + 8 * rank1
            for i in range(used):
                size = self.u(p, 4)
                off = tuple(self.u(p + 8 + 8 * k, 8) for k in range(rank1 - 1))
                child = self.u(p + ksz, 8)
                p += ksz + 8
                if level > 0:
                    walk(child)
                    continue
                raw = d[child:child + size]
                for fid, vals in reversed(filters):
                    if fid == 1:
                        raw = zlib.decompress(raw)
                    elif fid == 2:
                        raw = np.frombuffer(raw, np.uint8).reshape(vals[0], -1).T.tobytes()
                    else:
                        raise ValueError("filter %d not supported" % fid)
                block = np.frombuffer(raw, dt).reshape(cdims)
                sl = tuple(slice(o, min(o + c, n)) for o, c, n in zip(off, cdims, shape))
                out[sl] = block[tuple(slice(0, s.stop - s.start) for s in sl)]
        walk(bt)
        return out


def read(path):
    """-> _Reader: .objects (names), .dataset(name) -> ndarray, .attrs(name or None) -> dict"""
    return _Reader(path)
