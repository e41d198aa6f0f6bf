"""Independent check of the libhdf5-free writer: the reference's own mesh fixtures are read with
``h5lite``, written back with ``h5lite``'s WRITER, and the two files are compared with a structural walker that shares no
code with either (it parses superblock, object headers, B-tree / SNOD / local-heap groups and messages itself, from the
HDF5 file-format specification): same tree, same dataset names, the same datatype / dataspace / layout messages byte for
byte, the same attributes, the same raw data - i.e. what h5py / DOLFIN's HDF5File see is identical
[layout of REF src/vasp/preprocessing/preprocessing_common.py:198-247 files; SURVEY.md Appendix B]."""
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from vasp_amd.h5lite import read_h5, write_h5

UNDEF = 0xFFFFFFFFFFFFFFFF


class Walker:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        b = self.b
        assert b[:8] == b"\x89HDF\r\n\x1a\n"
        self.super = dict(version=b[8], free_v=b[9], root_v=b[10], shm_v=b[12], off=b[13], len=b[14],
                          leaf_k=struct.unpack_from("<H", b, 16)[0], int_k=struct.unpack_from("<H", b, 18)[0])
        assert self.super["version"] == 0 and self.super["off"] == 8 and self.super["len"] == 8
        base, free, eof, drv = struct.unpack_from("<QQQQ", b, 24)
        self.super.update(base=base, eof_is_len=eof == len(b), free_undef=free == UNDEF, drv_undef=drv == UNDEF)
        # root symbol-table entry at 56: link name offset, object header address, cache type, reserved, scratch
        self.root = struct.unpack_from("<QQ", b, 56)[1]

    def messages(self, addr):
        b = self.b
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHIi", b, addr)
        assert ver == 1
        out, blocks = [], [(addr + 16, hsize)]
        while blocks:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                                   # continuation
                    blocks.append(struct.unpack("<QQ", body[:16]))
                out.append((mtype, flags, body))
        return out

    def heap_str(self, heap, off):
        assert self.b[heap:heap + 4] == b"HEAP"
        data = struct.unpack_from("<Q", self.b, heap + 24)[0]
        s = self.b[data + off:]
        return s[:s.index(b"\0")].decode()

    def entries(self, btree, heap):
        b = self.b
        assert b[btree:btree + 4] == b"TREE"
        ntype, level, nent = struct.unpack_from("<BBH", b, btree + 4)
        assert ntype == 0
        out, pos = [], btree + 24
        for i in range(nent):
            child = struct.unpack_from("<Q", b, pos + 8)[0]             # key, child, key, child ...
            pos += 16
            if level > 0:
                out += self.entries(child, heap)
                continue
            assert b[child:child + 4] == b"SNOD"
            nsym = struct.unpack_from("<H", b, child + 6)[0]
            for k in range(nsym):
                name_off, ohdr = struct.unpack_from("<QQ", b, child + 8 + 40 * k)
                out.append((self.heap_str(heap, name_off), ohdr))
        return out

    def tree(self, addr=None, prefix=""):
        """{path: description}; datasets carry their messages and raw bytes."""
        addr = self.root if addr is None else addr
        msgs = self.messages(addr)
        types = [m[0] for m in msgs]
        out = {}
        if 0x0011 in types:
            bt, heap = struct.unpack("<QQ", [m for m in msgs if m[0] == 0x0011][0][2][:16])
            out[prefix or "/"] = dict(kind="group", header_version=1, message_types=sorted(t for t in types if t))
            for name, ohdr in self.entries(bt, heap):
                out.update(self.tree(ohdr, f"{prefix}/{name}"))
            return out
        md = {m[0]: m for m in msgs}
        lay = md[0x0008][2]
        lver, lclass = lay[0], lay[1]
        daddr, dsize = struct.unpack_from("<QQ", lay, 2)
        attrs = sorted(bytes(m[2]) for m in msgs if m[0] == 0x000C)
        out[prefix] = dict(kind="dataset", header_version=1,
                           message_types=sorted(t for t in types if t not in (0x0000, 0x0010)),
                           dataspace=bytes(md[0x0001][2]), datatype=bytes(md[0x0003][2]), datatype_flags=md[0x0003][1],
                           fill=bytes(md[0x0005][2]), layout=(lver, lclass, dsize), attributes=attrs,
                           raw=self.b[daddr:daddr + dsize])
        return out


FIXTURES = [GOLDEN / "cylinder" / "cylinder.h5", GOLDEN / "offset_stenosis" / "offset_stenosis.h5",
            GOLDEN / "aneurysm" / "small_aneurysm.h5"]


@pytest.mark.parametrize("fixture", FIXTURES, ids=lambda p: p.stem)
def test_rewritten_fixture_has_the_reference_files_structure(fixture, tmp_path):
    out = tmp_path / "rewritten.h5"
    write_h5(out, read_h5(fixture))
    ref, new = Walker(fixture), Walker(out)
    for key in ("version", "free_v", "root_v", "shm_v", "off", "len", "leaf_k", "int_k", "base", "eof_is_len", "free_undef",
                "drv_undef"):
        assert ref.super[key] == new.super[key], key
    tr, tn = ref.tree(), new.tree()
    assert list(tr) == list(tn)                                          # same objects, same (name-sorted) order
    assert set(tr) >= {"/mesh/coordinates", "/mesh/topology", "/boundaries/values", "/domains/values"}
    for path in tr:
        a, b = tr[path], tn[path]
        assert a["kind"] == b["kind"] and a["header_version"] == b["header_version"], path
        if a["kind"] == "group":
            assert a["message_types"] == b["message_types"] == [0x0011], path      # old-style group: one symbol-table message
            continue
        assert a["message_types"] == b["message_types"], (path, a["message_types"], b["message_types"])
        for key in ("dataspace", "datatype", "datatype_flags", "layout", "attributes"):
            assert a[key] == b[key], (path, key)
        assert a["fill"][:3] == b["fill"][:3], path                       # version 2, late allocation, write-time ifset
        assert a["raw"] == b["raw"], path                                  # the numbers themselves, byte for byte


def test_mesh_h5_of_a_run_has_the_structure_of_a_dolfin_written_mesh(tmp_path):
    """``Mesh/mesh.h5`` as the driver writes it vs the reference's own DOLFIN-written ``Mesh/mesh.h5`` (a 4-rank run:
    REF tests/test_data/hemodynamics_data/Mesh/mesh.h5 is not in the golden folder, so the comparison is against the
    input fixture, which DOLFIN's HDF5File wrote through the same code path, convert_xml_mesh_to_hdf5)."""
    from vasp_amd.mesh import FsiMesh
    m = FsiMesh.read(FIXTURES[0])
    m.write(tmp_path / "mesh.h5")
    tr, tn = Walker(FIXTURES[0]).tree(), Walker(tmp_path / "mesh.h5").tree()
    assert list(tr) == list(tn)
    for path in tr:
        if tr[path]["kind"] == "dataset":
            for key in ("dataspace", "datatype", "layout", "attributes", "raw"):
                assert tr[path][key] == tn[path][key], (path, key)
