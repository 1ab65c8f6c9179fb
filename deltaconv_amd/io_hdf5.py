"""Minimal read-only HDF5 reader (numpy only) -- enough for the ScanObjectNN files the reference opens with h5py
(experiments/datasets/scanobjectnn.py:90-95: ``h5py.File(path)['data']`` / ``['label']``), because h5py is not part
of this image.

Supported (the "earliest" file format every h5py / libhdf5 writes by default): superblock version 0 / 1, groups stored
as symbol tables (v1 B-tree + local heap), version-1 object headers with continuation blocks, simple dataspaces,
fixed-point and floating-point datatypes of either byte order, data layout version 3 -- contiguous, compact and
chunked (v1 chunk B-tree) -- with the deflate and shuffle filters.  Anything else (new-style groups, fractal heaps,
compound / variable-length types, external storage, other filters) raises ``NotImplementedError`` naming the
feature, so a file this reader cannot handle is never misread silently.

    f = File(path); f.keys(); arr = f['data'][...]          # datasets come back as numpy arrays

Format reference: "HDF5 File Format Specification Version 2.0" (sections III.A-III.D, IV.A.2).  Pinned against files
written by libhdf5 1.10.6 / h5py 3.3 (tests/golden/h5/, generator tests/golden/make_golden_h5.py).
"""
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Dataset:
    def __init__(self, f, name, shape, dtype, layout, filters):
        self._f, self.name, self.shape, self.dtype, self._layout, self._filters = f, name, tuple(shape), dtype, layout, filters

    def __len__(self):
        return self.shape[0] if self.shape else 0

    def __repr__(self):
        return f'<HDF5 dataset "{self.name}": shape {self.shape}, type "{self.dtype.str}">'

    def read(self):
        kind = self._layout[0]
        n = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        if kind == "compact":
            raw = self._layout[1]
            return np.frombuffer(raw, self.dtype, n).reshape(self.shape).copy()
        if kind == "contiguous":
            addr, size = self._layout[1], self._layout[2]
            if addr == _UNDEF:                       # never written: fill value (zeros)
                return np.zeros(self.shape, self.dtype)
            raw = self._f._read(addr, n * self.dtype.itemsize)
            return np.frombuffer(raw, self.dtype, n).reshape(self.shape).copy()
        return self._read_chunked()

    def __getitem__(self, key):
        return self.read()[key]

    def __array__(self, dtype=None):
        a = self.read()
        return a if dtype is None else a.astype(dtype)

    def _read_chunked(self):
        _, btree, chunk = self._layout
        out = np.zeros(self.shape, self.dtype)
        if btree == _UNDEF:
            return out
        rank = len(self.shape)
        for offs, size, mask, addr in self._f._chunks(btree, rank):
            raw = self._f._read(addr, size)
            for i, (fid, cvals) in reversed(list(enumerate(self._filters))):
                if mask & (1 << i):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:                      # shuffle: byte planes of the elements
                    es = cvals[0] if cvals else self.dtype.itemsize
                    a = np.frombuffer(raw, np.uint8)
                    ne = a.size // es
                    raw = a[:ne * es].reshape(es, ne).T.tobytes() + a[ne * es:].tobytes()
                elif fid == 3:                      # fletcher32 checksum appended
                    raw = raw[:-4]
                else:
                    raise NotImplementedError(f"HDF5 filter id {fid}")
            blk = np.frombuffer(raw, self.dtype, int(np.prod(chunk))).reshape(chunk)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, self.shape))
            sel_in = tuple(slice(0, so.stop - so.start) for so in sel_out)
            out[sel_out] = blk[sel_in]
        return out


class Group:
    def __init__(self, f, name, links):
        self._f, self.name, self._links = f, name, links

    def keys(self):
        return list(self._links)

    def __contains__(self, k):
        return k in self._links

    def __iter__(self):
        return iter(self._links)

    def __getitem__(self, key):
        node = self
        for part in [p for p in key.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links:
                raise KeyError(key)
            node = node._f._object(node._links[part], (node.name.rstrip("/") + "/" + part))
        return node


class File(Group):
    def __init__(self, path, mode="r"):
        assert mode == "r", "read-only"
        with open(path, "rb") as fh:
            self._buf = fh.read()
        b = self._buf
        base = 0
        while b[base:base + 8] != _SIG:            # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base >= len(b):
                raise ValueError("not an HDF5 file")
        ver = b[base + 8]
        if ver not in (0, 1):
            raise NotImplementedError(f"HDF5 superblock version {ver} (only the classic versions 0 / 1 are supported)")
        self._O, self._L = b[base + 13], b[base + 14]
        if self._O != 8 or self._L != 8:
            raise NotImplementedError("HDF5 offsets / lengths other than 8 bytes")
        p = base + 24 + (4 if ver == 1 else 0)
        self._base = self._u(p, 8)
        p += 32                                      # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        root_hdr = self._u(p + 8, 8)
        Group.__init__(self, self, "/", {})
        root = self._object(root_hdr, "/")
        self._links = root._links

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    # ---- low level
    def _read(self, addr, n):
        a = self._base + addr
        if a + n > len(self._buf):
            raise ValueError("HDF5 file truncated")
        return self._buf[a:a + n]

    def _u(self, pos, n):
        return int.from_bytes(self._buf[pos:pos + n], "little")

    def _messages(self, addr):
        """(type, flags, bytes) of every message of a version-1 object header (continuations followed)."""
        b, p = self._buf, self._base + addr
        if b[p:p + 4] == b"OHDR":
            raise NotImplementedError("HDF5 version-2 object headers (file written with libver='latest')")
        if b[p] != 1:
            raise NotImplementedError(f"HDF5 object header version {b[p]}")
        nmsg = self._u(p + 2, 2)
        size = self._u(p + 8, 4)
        blocks = [(p + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self._u(q, 2), self._u(q + 2, 2), b[q + 4]
                body = b[q + 8:q + 8 + msize]
                if mtype == 0x10:
                    blocks.append((self._base + int.from_bytes(body[:8], "little"), int.from_bytes(body[8:16], "little")))
                out.append((mtype, flags, body))
                q += 8 + msize
        return out

    def _heap_name(self, heap_addr, off):
        p = self._base + heap_addr
        assert self._buf[p:p + 4] == b"HEAP"
        data = self._base + self._u(p + 24, 8)
        end = self._buf.index(b"\0", data + off)
        return self._buf[data + off:end].decode()

    def _symbols(self, btree, heap):
        """{name: object header address} of a symbol-table group (v1 B-tree of SNOD leaves)."""
        b, p = self._buf, self._base + btree
        assert b[p:p + 4] == b"TREE" and b[p + 4] == 0
        level, used = b[p + 5], self._u(p + 6, 2)
        q = p + 24
        out = {}
        for i in range(used):
            child = self._u(q + 8, 8)                # key_i (8), child_i (8)
            q += 16
            if level > 0:
                out.update(self._symbols(child, heap))
                continue
            s = self._base + child
            assert b[s:s + 4] == b"SNOD"
            for j in range(self._u(s + 6, 2)):
                e = s + 8 + 40 * j
                out[self._heap_name(heap, self._u(e, 8))] = self._u(e + 8, 8)
        return out

    def _chunks(self, btree, rank):
        b, p = self._buf, self._base + btree
        assert b[p:p + 4] == b"TREE" and b[p + 4] == 1
        level, used = b[p + 5], self._u(p + 6, 2)
        ksize = 8 + 8 * (rank + 1)
        q = p + 24
        for i in range(used):
            size, mask = self._u(q, 4), self._u(q + 4, 4)
            offs = [self._u(q + 8 + 8 * d, 8) for d in range(rank)]
            child = self._u(q + ksize, 8)
            q += ksize + 8
            if level > 0:
                yield from self._chunks(child, rank)
            else:
                yield offs, size, mask, child

    def _object(self, addr, name):
        msgs = self._messages(addr)
        kinds = {t for t, _, _ in msgs}
        if 0x11 in kinds:                            # symbol table message -> group
            body = next(m for t, _, m in msgs if t == 0x11)
            links = self._symbols(int.from_bytes(body[:8], "little"), int.from_bytes(body[8:16], "little"))
            return Group(self, name, links)
        if 0x02 in kinds or 0x06 in kinds:
            raise NotImplementedError("HDF5 new-style (link message / fractal heap) groups")
        shape = dtype = layout = None
        filters = []
        for t, _, m in msgs:
            if t == 0x01:                            # dataspace
                ver, rank, flags = m[0], m[1], m[2]
                q = 8 if ver == 1 else 4
                shape = [int.from_bytes(m[q + 8 * d:q + 8 * d + 8], "little") for d in range(rank)]
            elif t == 0x03:                          # datatype
                cls, ver = m[0] & 15, m[0] >> 4
                bits0, size = m[1], int.from_bytes(m[4:8], "little")
                order = ">" if (bits0 & 1) else "<"
                if cls == 0:
                    dtype = np.dtype(f"{order}{'i' if bits0 & 8 else 'u'}{size}")
                elif cls == 1:
                    dtype = np.dtype(f"{order}f{size}")
                else:
                    raise NotImplementedError(f"HDF5 datatype class {cls} (only integers and floats)")
            elif t == 0x08:                          # data layout
                if m[0] != 3:
                    raise NotImplementedError(f"HDF5 data layout message version {m[0]}")
                if m[1] == 0:
                    n = int.from_bytes(m[2:4], "little")
                    layout = ("compact", bytes(m[4:4 + n]))
                elif m[1] == 1:
                    layout = ("contiguous", int.from_bytes(m[2:10], "little"), int.from_bytes(m[10:18], "little"))
                elif m[1] == 2:
                    nd = m[2]
                    dims = [int.from_bytes(m[11 + 4 * d:15 + 4 * d], "little") for d in range(nd)]
                    layout = ("chunked", int.from_bytes(m[3:11], "little"), tuple(dims[:-1]))
                else:
                    raise NotImplementedError(f"HDF5 layout class {m[1]}")
            elif t == 0x0B:                          # filter pipeline
                ver, nf = m[0], m[1]
                q = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = int.from_bytes(m[q:q + 2], "little")
                    if ver == 1 or fid >= 256:
                        nlen = int.from_bytes(m[q + 2:q + 4], "little")
                        q += 4
                    else:
                        nlen = 0
                        q += 2
                    nvals = int.from_bytes(m[q + 2:q + 4], "little")
                    q += 4
                    q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    cvals = [int.from_bytes(m[q + 4 * i:q + 4 * i + 4], "little") for i in range(nvals)]
                    q += 4 * nvals
                    if ver == 1 and nvals % 2:
                        q += 4
                    filters.append((fid, cvals))
        if shape is None or dtype is None or layout is None:
            raise NotImplementedError(f"HDF5 object '{name}' is neither a classic group nor a simple dataset")
        return Dataset(self, name, shape, dtype, layout, filters)
