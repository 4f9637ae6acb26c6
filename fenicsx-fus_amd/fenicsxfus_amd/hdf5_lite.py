"""Minimal read-only HDF5 parser for the mesh files the reference ships.

The reference reads its Gmsh-generated meshes through DOLFINx's XDMF/HDF5 reader
(``io::XDMFFile::read_mesh``, cpp/fenicsx-sf/benchmarks/PH1/BM7-SC1/main.cpp:55-64); neither h5py
nor libhdf5 headers exist where this repository is built, so this module implements just the part
of the HDF5 file format those files use: version-0 superblock, version-1 object headers, old-style
groups (symbol-table B-tree + local heap), fixed-point / IEEE-float datatypes and contiguous or
unfiltered chunked dataset layouts.  Anything else raises ``NotImplementedError``.

    f = H5File("mesh.h5");  x = f["/Mesh/hex/geometry"];  f.keys("/Mesh")
"""
from __future__ import annotations

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if b[:8] != _SIG:
            raise ValueError("not an HDF5 file")
        if b[8] != 0:
            raise NotImplementedError(f"superblock version {b[8]} (only 0 is supported)")
        if b[13] != 8 or b[14] != 8:
            raise NotImplementedError("only 8-byte offsets/lengths are supported")
        self.base = struct.unpack_from("<Q", b, 24)[0]
        root_ohdr = struct.unpack_from("<Q", b, 64)[0]
        cache_type = struct.unpack_from("<I", b, 72)[0]
        if cache_type == 1:
            self.root = struct.unpack_from("<QQ", b, 80)       # (btree, heap)
        else:
            self.root = self._group_of(root_ohdr)

    # ---- object headers ---------------------------------------------------------------------
    def _messages(self, addr: int):
        """Yield (type, data bytes) of a version-1 object header, following continuations."""
        b = self.buf
        addr += self.base
        if b[addr] != 1:
            raise NotImplementedError(f"object header version {b[addr]} (only 1 is supported)")
        nmsg = struct.unpack_from("<H", b, addr + 2)[0]
        hsize = struct.unpack_from("<I", b, addr + 8)[0]
        blocks = [(addr + 16, hsize)]
        seen = 0
        while blocks and seen < nmsg:
            pos, length = blocks.pop(0)
            end = pos + length
            while pos + 8 <= end and seen < nmsg:
                mtype, msize = struct.unpack_from("<HH", b, pos)
                data = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                seen += 1
                if mtype == 0x0010:                              # continuation
                    off, ln = struct.unpack_from("<QQ", data, 0)
                    blocks.append((off + self.base, ln))
                else:
                    yield mtype, data

    def _group_of(self, ohdr_addr: int):
        for mtype, data in self._messages(ohdr_addr):
            if mtype == 0x0011:                                  # symbol table message
                return struct.unpack_from("<QQ", data, 0)
        raise KeyError("object is not an old-style group")

    # ---- groups -----------------------------------------------------------------------------
    def _heap_name(self, heap_addr: int, off: int) -> str:
        b = self.buf
        h = heap_addr + self.base
        assert b[h:h + 4] == b"HEAP"
        data_addr = struct.unpack_from("<Q", b, h + 24)[0] + self.base
        end = b.index(b"\0", data_addr + off)
        return b[data_addr + off:end].decode()

    def _entries(self, btree_addr: int, heap_addr: int):
        """Name -> object header address for every link of a group."""
        b = self.buf
        out = {}

        def walk(node):
            n = node + self.base
            assert b[n:n + 4] == b"TREE" and b[n + 4] == 0
            level = b[n + 5]
            used = struct.unpack_from("<H", b, n + 6)[0]
            pos = n + 24                                         # after the two sibling addresses
            for k in range(used):
                child = struct.unpack_from("<Q", b, pos + 8)[0]  # key_k (8) then child_k (8)
                pos += 16
                if level > 0:
                    walk(child)
                else:
                    s = child + self.base
                    assert b[s:s + 4] == b"SNOD"
                    nsym = struct.unpack_from("<H", b, s + 6)[0]
                    for e in range(nsym):
                        name_off, ohdr = struct.unpack_from("<QQ", b, s + 8 + 40 * e)
                        out[self._heap_name(heap_addr, name_off)] = ohdr

        walk(btree_addr)
        return out

    def _resolve(self, path: str) -> int:
        group = self.root
        ohdr = None
        parts = [p for p in path.split("/") if p]
        for i, name in enumerate(parts):
            ents = self._entries(*group)
            if name not in ents:
                raise KeyError(path)
            ohdr = ents[name]
            if i < len(parts) - 1:
                group = self._group_of(ohdr)
        if ohdr is None:
            raise KeyError(path)
        return ohdr

    def keys(self, path: str = "/"):
        parts = [p for p in path.split("/") if p]
        group = self.root if not parts else self._group_of(self._resolve(path))
        return sorted(self._entries(*group))

    # ---- datasets ---------------------------------------------------------------------------
    def __getitem__(self, path: str) -> np.ndarray:
        b = self.buf
        shape = dtype = layout = None
        for mtype, data in self._messages(self._resolve(path)):
            if mtype == 0x0001:                                  # dataspace
                ver, rank = data[0], data[1]
                off = 8 if ver == 1 else 4
                shape = struct.unpack_from(f"<{rank}Q", data, off)
            elif mtype == 0x0003:                                # datatype
                cls = data[0] & 0x0F
                size = struct.unpack_from("<I", data, 4)[0]
                if data[1] & 1:
                    raise NotImplementedError("big-endian data")
                if cls == 0:
                    dtype = np.dtype(f"<{'i' if data[1] & 8 else 'u'}{size}")
                elif cls == 1:
                    dtype = np.dtype(f"<f{size}")
                else:
                    raise NotImplementedError(f"datatype class {cls}")
            elif mtype == 0x0008:                                # data layout
                layout = data
            elif mtype == 0x000B:
                raise NotImplementedError("filtered (compressed) datasets")
        if shape is None or dtype is None or layout is None:
            raise KeyError(f"{path} is not a simple dataset")
        if layout[0] != 3:
            raise NotImplementedError(f"data layout message version {layout[0]}")
        cls = layout[1]
        count = int(np.prod(shape))
        if cls == 1:                                             # contiguous
            addr, _size = struct.unpack_from("<QQ", layout, 2)
            if addr == _UNDEF:
                return np.zeros(shape, dtype=dtype)
            return np.frombuffer(b, dtype=dtype, count=count, offset=addr + self.base).reshape(shape).copy()
        if cls == 2:                                             # chunked, no filters
            rank1 = layout[2]
            btree = struct.unpack_from("<Q", layout, 3)[0]
            cdims = struct.unpack_from(f"<{rank1}I", layout, 11)[:-1]
            out = np.zeros(shape, dtype=dtype)
            self._read_chunks(btree, rank1, cdims, out)
            return out
        raise NotImplementedError("compact layout")

    def _read_chunks(self, node, rank1, cdims, out):
        b = self.buf
        n = node + self.base
        assert b[n:n + 4] == b"TREE" and b[n + 4] == 1
        level = b[n + 5]
        used = struct.unpack_from("<H", b, n + 6)[0]
        keysize = 8 + 8 * rank1
        pos = n + 24
        for _ in range(used):
            csize, _mask = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from(f"<{rank1}Q", b, pos + 8)[:-1]
            child = struct.unpack_from("<Q", b, pos + keysize)[0]
            pos += keysize + 8
            if level > 0:
                self._read_chunks(child, rank1, cdims, out)
            else:
                chunk = np.frombuffer(b, dtype=out.dtype, count=int(np.prod(cdims)),
                                      offset=child + self.base).reshape(cdims)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
                out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
