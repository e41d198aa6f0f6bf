"""libhdf5-free reader/writer for the DOLFIN ``HDF5File`` mesh layout.

The simulation stage of the reference reads its input mesh with
``HDF5File.read(mesh, "/mesh")``, ``read(boundaries, "/boundaries")``,
``read(domains, "/domains")`` [REF src/vasp/simulations/offset_stenosis.py:88-94] and every
post-processing tool reads the same names back with h5py
[REF src/vasp/postprocessing/postprocessing_common.py:38-40].  Neither libhdf5 nor h5py exist in
the build image or on the GPU box, so this module parses (and emits) the subset of the HDF5 file
format those files use: superblock v0, v1 object headers, old-style (symbol-table) groups,
contiguous datasets, fixed-point / IEEE-float / fixed-length-string datatypes and v1 attribute
messages.

Only the structures are handled that DOLFIN's serial writer produces; anything else raises
``H5Error`` instead of guessing.
"""
from __future__ import annotations

import struct
import time
from typing import Dict, Optional, Tuple, Union

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(RuntimeError):
    pass


class Dataset:
    """A contiguous dataset: ``data`` is a numpy array, ``attrs`` a dict of numpy scalars/arrays/bytes."""

    def __init__(self, data: np.ndarray, attrs: Optional[Dict[str, object]] = None):
        self.data = data
        self.attrs = dict(attrs or {})

    @property
    def shape(self):
        return self.data.shape

    def __getitem__(self, idx):
        return self.data[idx]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.data, dtype=dtype)


class Group(dict):
    """An old-style group: mapping name -> Group | Dataset (sorted by name on write)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.attrs: Dict[str, object] = {}

    def visit(self, prefix=""):
        for name in sorted(self):
            node = self[name]
            path = f"{prefix}/{name}"
            yield path, node
            if isinstance(node, Group):
                yield from node.visit(path)


# --------------------------------------------------------------------------------------------
# reader
# --------------------------------------------------------------------------------------------

class _Reader:
    def __init__(self, buf: bytes):
        self.b = buf
        if buf[:8] != SIGNATURE:
            raise H5Error("not an HDF5 file (bad signature)")
        sb_version = buf[8]
        if sb_version != 0:
            raise H5Error(f"superblock version {sb_version} not supported (DOLFIN writes v0)")
        self.so, self.sl = buf[13], buf[14]
        if (self.so, self.sl) != (8, 8):
            raise H5Error("only 8-byte offsets/lengths supported")
        (self.base,) = struct.unpack_from("<Q", buf, 24)
        # root symbol table entry at 56
        self.root_header = struct.unpack_from("<Q", buf, 56 + 8)[0]

    # -- low level ------------------------------------------------------------------------
    def _messages(self, addr: int):
        b = self.b
        version = b[addr]
        if version != 1:
            raise H5Error(f"object header version {version} at {addr:#x} not supported")
        nmsg, = struct.unpack_from("<H", b, addr + 2)
        hsize, = struct.unpack_from("<I", b, addr + 8)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, pos)
                body = pos + 8
                if mtype == 0x0010:  # continuation
                    coff, clen = struct.unpack_from("<QQ", b, body)
                    blocks.append((coff + self.base, clen))
                out.append((mtype, b[body:body + msize]))
                pos = body + msize
        return out

    @staticmethod
    def _parse_datatype(m: bytes) -> Tuple[Union[np.dtype, str], int]:
        cls = m[0] & 0x0F
        bits0 = m[1]
        size, = struct.unpack_from("<I", m, 4)
        if cls == 0:
            if bits0 & 1:
                raise H5Error("big-endian integers not supported")
            signed = bool(bits0 & 0x08)
            return np.dtype(f"<{'i' if signed else 'u'}{size}"), size
        if cls == 1:
            if bits0 & 1:
                raise H5Error("big-endian floats not supported")
            return np.dtype(f"<f{size}"), size
        if cls == 3:
            return "string", size
        raise H5Error(f"datatype class {cls} not supported")

    @staticmethod
    def _parse_dataspace(m: bytes) -> Tuple[int, ...]:
        version = m[0]
        rank = m[1]
        if version == 1:
            off = 8
        elif version == 2:
            off = 4
        else:
            raise H5Error(f"dataspace version {version}")
        return tuple(struct.unpack_from(f"<{rank}Q", m, off)) if rank else ()

    def _parse_attribute(self, m: bytes):
        version = m[0]
        if version != 1:
            raise H5Error(f"attribute message version {version} not supported")
        nsize, tsize, ssize = struct.unpack_from("<HHH", m, 2)
        pad = lambda n: (n + 7) & ~7
        pos = 8
        name = m[pos:pos + nsize].split(b"\0")[0].decode()
        pos += pad(nsize)
        dt, esize = self._parse_datatype(m[pos:pos + tsize])
        pos += pad(tsize)
        shape = self._parse_dataspace(m[pos:pos + ssize])
        pos += pad(ssize)
        n = int(np.prod(shape)) if shape else 1
        raw = m[pos:pos + n * esize]
        if dt == "string":
            val = raw.split(b"\0")[0].decode()
        else:
            val = np.frombuffer(raw, dtype=dt, count=n).reshape(shape).copy()
        return name, val

    def _read_object(self, addr: int):
        msgs = self._messages(addr)
        types = [t for t, _ in msgs]
        attrs = {}
        for t, m in msgs:
            if t == 0x000C:
                k, v = self._parse_attribute(m)
                attrs[k] = v
        if 0x0011 in types:  # group
            m = dict(msgs)[0x0011]
            btree, heap = struct.unpack_from("<QQ", m, 0)
            g = Group()
            g.attrs = attrs
            for name, child in self._group_entries(btree + self.base, heap + self.base):
                g[name] = self._read_object(child + self.base)
            return g
        md = {}
        for t, m in msgs:
            md.setdefault(t, m)
        if 0x0001 not in md or 0x0003 not in md or 0x0008 not in md:
            raise H5Error(f"object at {addr:#x} is neither an old-style group nor a simple dataset")
        shape = self._parse_dataspace(md[0x0001])
        dt, esize = self._parse_datatype(md[0x0003])
        lay = md[0x0008]
        if lay[0] != 3:
            raise H5Error(f"layout message version {lay[0]} not supported")
        n = int(np.prod(shape)) if shape else 1
        if lay[1] == 1:
            daddr, dsize = struct.unpack_from("<QQ", lay, 2)
            if daddr == UNDEF:
                raw = b"\0" * (n * esize)
            else:
                raw = self.b[daddr + self.base: daddr + self.base + n * esize]
        elif lay[1] == 0:
            csize, = struct.unpack_from("<H", lay, 2)
            raw = lay[4:4 + csize]
        else:
            raise H5Error("chunked datasets not supported (DOLFIN serial output is contiguous)")
        if dt == "string":
            data = np.array([raw[i * esize:(i + 1) * esize].split(b"\0")[0].decode() for i in range(n)]).reshape(shape)
        else:
            data = np.frombuffer(raw, dtype=dt, count=n).reshape(shape).copy()
        return Dataset(data, attrs)

    def _heap_string(self, heap_addr: int, off: int) -> str:
        b = self.b
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise H5Error("bad local heap signature")
        data_addr, = struct.unpack_from("<Q", b, heap_addr + 24)
        start = data_addr + self.base + off
        end = b.find(b"\0", start)
        return bytes(b[start:end]).decode()

    def _group_entries(self, btree: int, heap: int):
        b = self.b
        if b[btree:btree + 4] != b"TREE":
            raise H5Error("bad B-tree signature")
        ntype, level, used = struct.unpack_from("<BBH", b, btree + 4)
        if ntype != 0:
            raise H5Error("not a group B-tree")
        pos = btree + 24
        children = []
        for i in range(used):
            pos += 8  # key
            child, = struct.unpack_from("<Q", b, pos)
            pos += 8
            children.append(child + self.base)
        for c in children:
            if level > 0:
                yield from self._group_entries(c, heap)
            else:
                if b[c:c + 4] != b"SNOD":
                    raise H5Error("bad symbol node signature")
                nsym, = struct.unpack_from("<H", b, c + 6)
                for k in range(nsym):
                    e = c + 8 + 40 * k
                    name_off, ohdr = struct.unpack_from("<QQ", b, e)
                    yield self._heap_string(heap, name_off), ohdr

    def read(self) -> Group:
        root = self._read_object(self.root_header + self.base)
        if not isinstance(root, Group):
            raise H5Error("root object is not a group")
        return root


def read_h5(path) -> Group:
    """Read a whole (small, contiguous) HDF5 file into a tree of ``Group``/``Dataset``."""
    with open(path, "rb") as f:
        buf = f.read()
    return _Reader(buf).read()


# --------------------------------------------------------------------------------------------
# writer (same subset): used for Mesh/mesh.h5, Visualization/*.h5, Checkpoint/*.h5
# --------------------------------------------------------------------------------------------

def _pad8(n: int) -> int:
    return (n + 7) & ~7


def _dtype_message(dt) -> bytes:
    if isinstance(dt, tuple) and dt[0] == "string":
        size = dt[1]
        # class 3 (string), version 1; null-terminated, ASCII
        return struct.pack("<BBBBI", 0x13, 0x00, 0x00, 0x00, size)
    dt = np.dtype(dt)
    if dt.kind == "f":
        if dt.itemsize == 8:
            return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        if dt.itemsize == 4:
            return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if dt.kind in "iu":
        bits0 = 0x08 if dt.kind == "i" else 0x00
        return struct.pack("<BBBBI", 0x10, bits0, 0x00, 0x00, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    raise H5Error(f"cannot encode dtype {dt}")


def _dataspace_message(shape) -> bytes:
    rank = len(shape)
    if rank == 0:
        return struct.pack("<BBB5x", 1, 0, 0)
    return struct.pack("<BBB5x", 1, rank, 1) + struct.pack(f"<{rank}Q", *shape) + struct.pack(f"<{rank}Q", *shape)


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = body + b"\0" * (_pad8(len(body)) - len(body))
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attribute_message(name: str, value) -> bytes:
    nm = name.encode() + b"\0"
    if isinstance(value, (str, bytes)):
        raw = value.encode() if isinstance(value, str) else value
        dtm = _dtype_message(("string", len(raw)))
        dsm = _dataspace_message(())
    else:
        arr = np.ascontiguousarray(value)
        dtm = _dtype_message(arr.dtype)
        dsm = _dataspace_message(arr.shape)
        raw = arr.tobytes()
    body = struct.pack("<BxHHH", 1, len(nm), len(dtm), len(dsm))
    for part in (nm, dtm, dsm):
        body += part + b"\0" * (_pad8(len(part)) - len(part))
    body += raw
    return _message(0x000C, body)


class _Writer:
    """Lays out: superblock | root group | (recursively) groups and datasets; everything 8-aligned."""

    LEAF_K = 4
    INTERNAL_K = 16

    def __init__(self, origin: int = 0):
        self.buf = bytearray()
        self.origin = origin              # file offset of buf[0] (8-aligned): addresses handed out are absolute

    def _alloc(self, n: int) -> int:
        pos = _pad8(len(self.buf))
        self.buf.extend(b"\0" * (pos - len(self.buf) + n))
        return pos + self.origin

    def _put(self, pos: int, data: bytes):
        pos -= self.origin
        self.buf[pos:pos + len(data)] = data

    def _object_header(self, messages) -> int:
        body = b"".join(messages)
        hdr = struct.pack("<BxHII4x", 1, len(messages), 1, len(body))
        pos = self._alloc(len(hdr) + len(body))
        self._put(pos, hdr + body)
        return pos

    def _dataset_header(self, shape, dtype, daddr: int, nbytes: int, attrs) -> int:
        """Object header of a contiguous dataset whose raw data sits at ``daddr`` (already in the file or in this buffer)."""
        msgs = [
            _message(0x0001, _dataspace_message(shape)),
            _message(0x0003, _dtype_message(dtype), flags=1),
            _message(0x0005, bytes([2, 2, 2, 0])),  # fill value v2: late alloc, write-time ifset, undefined
            _message(0x0008, struct.pack("<BBQQ", 3, 1, daddr, nbytes)),
            _message(0x0012, struct.pack("<B3xI", 1, int(time.time()) & 0xFFFFFFFF)),
        ]
        for k, v in (attrs or {}).items():
            msgs.append(_attribute_message(k, v))
        return self._object_header(msgs)

    def _write_dataset(self, ds: Dataset) -> int:
        arr = np.ascontiguousarray(ds.data)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        raw = arr.tobytes()
        daddr = self._alloc(len(raw)) if raw else UNDEF
        if raw:
            self._put(daddr, raw)
        return self._dataset_header(arr.shape, arr.dtype, daddr, len(raw), ds.attrs)

    def _write_group(self, g: Group) -> int:
        names = sorted(g.keys())
        child_addr, child_bt_heap = {}, {}
        for n in names:
            node = g[n]
            child_addr[n] = self._write_group(node) if isinstance(node, Group) else self._write_dataset(
                node if isinstance(node, Dataset) else Dataset(np.asarray(node)))
            if isinstance(node, Group):
                child_bt_heap[n] = node._bt_heap
        g._bt_heap = self._group_tables(names, child_addr, child_bt_heap)
        g._snod_entry = dict(self._last_entry_pos)      # name -> file offset of its 40-byte symbol-table entry
        msgs = [_message(0x0011, struct.pack("<QQ", *g._bt_heap))]
        for k, v in g.attrs.items():
            msgs.append(_attribute_message(k, v))
        g._header = self._object_header(msgs)
        return g._header

    def _group_tables(self, names, child_addr, child_bt_heap):
        """Local heap, symbol nodes and B-tree of an old-style group whose children (``names``, sorted) already have
        object headers at ``child_addr``; returns (B-tree address, heap address)."""
        self._last_entry_pos = {}
        # local heap: first 8 bytes reserved (empty string at offset 0)
        heap_data = bytearray(b"\0" * 8)
        name_off = {}
        for n in names:
            name_off[n] = len(heap_data)
            enc = n.encode() + b"\0"
            heap_data.extend(enc + b"\0" * (_pad8(len(enc)) - len(enc)))
        free_off = len(heap_data)
        heap_data.extend(struct.pack("<QQ", 1, 16))  # one free block: next=1 (end), size=16
        heap_data_addr = self._alloc(len(heap_data))
        self._put(heap_data_addr, bytes(heap_data))
        heap_addr = self._alloc(32)
        self._put(heap_addr, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, heap_data_addr))
        # symbol nodes (<= 2*LEAF_K entries each)
        cap = 2 * self.LEAF_K
        chunks = [names[i:i + cap] for i in range(0, len(names), cap)] or [[]]
        entries = []                      # (address, heap offset of the largest name below it)
        for ch in chunks:
            pos = self._alloc(8 + 40 * cap)
            body = b"SNOD" + struct.pack("<BxH", 1, len(ch))
            for n in ch:
                self._last_entry_pos[n] = pos + len(body)
                if n in child_bt_heap:
                    body += struct.pack("<QQII", name_off[n], child_addr[n], 1, 0) + struct.pack("<QQ", *child_bt_heap[n])
                else:
                    body += struct.pack("<QQII16x", name_off[n], child_addr[n], 0, 0)
            self._put(pos, body)
            if ch:
                entries.append((pos, name_off[ch[-1]]))
        # B-tree over the symbol nodes: as many levels as needed (<= 2*INTERNAL_K children per node)
        fan = 2 * self.INTERNAL_K
        node_size = 24 + (fan + 1) * 8 + fan * 8
        level = 0
        while True:
            groups = [entries[i:i + fan] for i in range(0, len(entries), fan)] or [[]]
            addrs = [self._alloc(node_size) for _ in groups]
            nxt, left_key = [], 0
            for j, grp in enumerate(groups):
                left = addrs[j - 1] if j > 0 else UNDEF
                right = addrs[j + 1] if j + 1 < len(groups) else UNDEF
                body = b"TREE" + struct.pack("<BBHQQ", 0, level, len(grp), left, right) + struct.pack("<Q", left_key)
                for addr, key in grp:
                    body += struct.pack("<QQ", addr, key)
                self._put(addrs[j], body)
                if grp:
                    left_key = grp[-1][1]
                    nxt.append((addrs[j], left_key))
            if len(groups) == 1:
                bt = addrs[0]
                break
            entries, level = nxt, level + 1
        return bt, heap_addr

    def write(self, root: Group) -> bytes:
        self._alloc(96)  # superblock (56) + root symbol-table entry (40)
        root_hdr = self._write_group(root)
        eof = _pad8(len(self.buf))
        self.buf.extend(b"\0" * (eof - len(self.buf)))
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0) + struct.pack("<QQ", *root._bt_heap)
        self._put(0, sb)
        return bytes(self.buf)


def write_h5(path, root: Group) -> None:
    data = _Writer().write(root)
    with open(path, "wb") as f:
        f.write(data)


class H5Series:
    """One HDF5 file with a group of datasets that GROWS: ``append`` adds ``/<series>/<name>`` without rewriting what is
    already there.

    DOLFIN's ``XDMFFile.write(u, t)`` appends every frame of a run to ``<name>.h5:/VisualisationVector/<k>``, and the
    reference's consumers rely on it: ``vasp-predeform-mesh`` takes "the last dataset of displacement.h5"
    [REF src/vasp/postprocessing/predeform_mesh.py:55-57], ``vasp-create-hdf5`` opens the file a frame's XDMF entry names
    [REF src/vasp/postprocessing/postprocessing_fenics/create_hdf5.py:89-98,139-160].  So a series has to live in ONE file
    however long the run is, at O(1 frame) of memory.

    On disk (same subset as ``write_h5``: superblock v0, old-style groups, contiguous datasets): the static part (e.g.
    ``/Mesh/0/mesh``) and an empty series group are written once.  A frame then costs
      1. its raw data and its dataset object header at the end of the file (never touched again);
      2. the series group's tables - local heap, symbol nodes, B-tree; ~60 bytes per frame, entries sorted by name as the
         format requires - written afresh into the inactive one of two metadata regions (doubling in size when full);
      3. three in-place patches: the group's symbol-table message, the cached copy of it in the parent's symbol-table entry,
         and the end-of-file address in the superblock.
    A reader that opens the file between two appends sees a complete file with k frames.
    """

    def __init__(self, path, static_root: Group, series: str = "VisualisationVector", reserve: int = 1 << 14):
        if series in static_root:
            raise H5Error(f"/{series} is managed by H5Series")
        root = Group(static_root)
        root.attrs = dict(getattr(static_root, "attrs", {}))
        sg = Group()
        root[series] = sg
        w = _Writer()
        data = w.write(root)
        self.path, self.series = path, series
        self._group_header = sg._header                      # object header of the series group: message body at + 16 + 8
        self._parent_entry = root._snod_entry[series]        # its symbol-table entry in the root group (cached bt / heap at + 24)
        self.names, self.addr = [], {}
        self._f = open(path, "w+b")
        self._f.write(data)
        self._eof = len(data)
        self._regions, self._cap, self._active = None, 0, 0
        self._reserve = reserve
        self._f.flush()

    def __len__(self):
        return len(self.names)

    def _write_at(self, pos: int, data: bytes):
        self._f.seek(pos)
        self._f.write(data)

    def append(self, name: str, arr: np.ndarray, attrs: Optional[Dict[str, object]] = None) -> None:
        if name in self.addr:
            raise H5Error(f"/{self.series}/{name} exists")
        arr = np.ascontiguousarray(arr)
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        # 1. raw data + dataset header at the end of the file
        daddr = _pad8(self._eof)
        self._write_at(daddr, arr.tobytes())
        w = _Writer(origin=_pad8(daddr + arr.nbytes))
        hdr = w._dataset_header(arr.shape, arr.dtype, daddr if arr.nbytes else UNDEF, arr.nbytes, attrs)
        self._write_at(w.origin, bytes(w.buf))
        self._eof = w.origin + len(w.buf)
        self.names.append(name)
        self.addr[name] = hdr
        # 2. the group's tables into the inactive metadata region
        names = sorted(self.names)
        probe = _Writer(origin=0)
        probe._group_tables(names, self.addr, {})
        need = _pad8(len(probe.buf))
        if self._regions is None or need > self._cap:
            self._cap = max(self._reserve, 2 * need)
            a = _pad8(self._eof)
            self._regions = (a, a + self._cap)
            self._eof = a + 2 * self._cap
            self._write_at(self._eof - 1, b"\0")              # the file really is that long
            self._active = 1
        region = self._regions[1 - self._active]
        t = _Writer(origin=region)
        bt, heap = t._group_tables(names, self.addr, {})
        self._write_at(region, bytes(t.buf))
        self._active = 1 - self._active
        # 3. switch over: group header, parent's cached copy, end-of-file address
        both = struct.pack("<QQ", bt, heap)
        self._write_at(self._group_header + 16 + 8, both)
        self._write_at(self._parent_entry + 24, both)
        self._write_at(40, struct.pack("<Q", self._eof))     # superblock: base 24, free-space 32, end-of-file 40
        self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
