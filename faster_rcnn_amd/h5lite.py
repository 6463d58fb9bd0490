"""Minimal HDF5 reader and writer for Keras 2.0.x weight / model files.

The reference loads its checkpoints with ``model.load_weights(path, by_name)`` and
``keras.models.load_model`` (resnet.py:32-61, 481-485; train_det_step2.py:110), i.e. through h5py,
which this image's interpreter does not have.  Keras 2.0.8 + h5py 2.7 (requirements.txt:9,22) write
the oldest on-disk format: superblock version 0, version-1 object headers, symbol-table groups
(v1 B-tree + local heap), contiguous little-endian float datasets and fixed-length string attributes.
That subset is what this module parses, straight from the published HDF5 File Format Specification;
anything outside it (chunked or compressed datasets, new-style groups, dense attribute storage,
variable-length strings) raises ``H5Error`` naming the feature instead of guessing.

    weights = read_keras_weights("model_frcnn_step2.h5")    # {layer_name: [arrays in get_weights() order]}
    write_keras_weights("step2.h5", weights)                 # the same subset; h5py / Keras read it (tests/test_h5lite.py)
"""
import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
VLEN_STR = "vlen-str"       # marker returned by _datatype for variable-length strings (global-heap references)


class H5Error(ValueError):
    pass


def is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(8) == SIGNATURE


class _Reader:
    def __init__(self, buf):
        self.b = buf
        if buf[:8] != SIGNATURE:
            raise H5Error("not an HDF5 file (no signature at offset 0)")
        ver = buf[8]
        if ver not in (0, 1):
            raise H5Error("superblock version %d is not supported (Keras 2.0.x / h5py 2.7 write version 0)" % ver)
        self.O, self.L = buf[13], buf[14]                   # size of offsets / lengths
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5Error("unsupported offset/length size %d/%d" % (self.O, self.L))
        p = 24 + (4 if ver == 1 else 0)
        self.base = self.uint(p, self.O)
        p += 4 * self.O                                      # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self.root = self.uint(p + self.O, self.O)

    def uint(self, off, size):
        return int.from_bytes(self.b[off:off + size], "little")

    # ------------------------------------------------------------------ object headers
    def messages(self, addr):
        """Yield (type, flags, payload memoryview) of a version-1 object header, following continuations."""
        b = self.b
        a = addr + self.base
        if b[a] != 1:
            if b[a:a + 4] == b"OHDR":
                raise H5Error("version-2 object headers (libver='latest') are not supported")
            raise H5Error("unknown object header version %d at %#x" % (b[a], addr))
        nmsg = self.uint(a + 2, 2)
        size = self.uint(a + 8, 4)
        blocks = [(a + 16, size)]
        seen = 0
        while blocks and seen < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and seen < nmsg:
                mtype, msize, flags = self.uint(p, 2), self.uint(p + 2, 2), b[p + 4]
                body = p + 8
                seen += 1
                if mtype == 0x10:                            # continuation
                    blocks.append((self.uint(body, self.O) + self.base, self.uint(body + self.O, self.L)))
                elif mtype != 0:
                    if flags & 0x02:
                        raise H5Error("shared object-header messages are not supported")
                    yield mtype, flags, memoryview(b)[body:body + msize]
                p = body + msize

    # ------------------------------------------------------------------ groups (symbol tables)
    def _heap_name(self, heap_data, off):
        end = self.b.index(b"\x00", heap_data + off)
        return self.b[heap_data + off:end].decode("utf-8")

    def _btree_entries(self, addr, heap_data, out):
        a = addr + self.base
        if self.b[a:a + 4] == b"SNOD":
            n = self.uint(a + 6, 2)
            p = a + 8
            for _ in range(n):
                name_off = self.uint(p, self.O)
                obj = self.uint(p + self.O, self.O)
                out[self._heap_name(heap_data, name_off)] = obj
                p += 2 * self.O + 24
            return
        if self.b[a:a + 4] != b"TREE":
            raise H5Error("expected a v1 B-tree or symbol node at %#x" % addr)
        if self.b[a + 4] != 0:
            raise H5Error("B-tree node type %d inside a group" % self.b[a + 4])
        used = self.uint(a + 6, 2)
        p = a + 8 + 2 * self.O                               # past the sibling addresses
        for i in range(used):
            child = self.uint(p + self.L, self.O)            # key_i (L bytes) then child_i
            self._btree_entries(child, heap_data, out)
            p += self.L + self.O

    def links(self, addr):
        """{name: object header address} of the group whose object header is at addr."""
        for mtype, _, body in self.messages(addr):
            if mtype == 0x11:                                # symbol table message
                btree = int.from_bytes(body[:self.O], "little")
                heap = int.from_bytes(body[self.O:2 * self.O], "little") + self.base
                if self.b[heap:heap + 4] != b"HEAP":
                    raise H5Error("bad local heap signature")
                heap_data = self.uint(heap + 8 + 2 * self.L, self.O) + self.base
                out = {}
                self._btree_entries(btree, heap_data, out)
                return out
            if mtype in (0x02, 0x06):
                raise H5Error("new-style groups (link messages) are not supported")
        raise H5Error("object at %#x is not a group" % addr)

    def resolve(self, addr, path):
        for part in path.split("/"):
            if part:
                table = self.links(addr)
                if part not in table:
                    raise KeyError(path)
                addr = table[part]
        return addr

    # ------------------------------------------------------------------ dataspace / datatype / data
    def _dataspace(self, body):
        ver, rank, flags = body[0], body[1], body[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
            if body[3] == 2:                                 # null dataspace
                return None
        else:
            raise H5Error("dataspace message version %d" % ver)
        return tuple(int.from_bytes(body[p + i * self.L:p + (i + 1) * self.L], "little") for i in range(rank))

    @staticmethod
    def _datatype(body):
        cls, ver = body[0] & 0x0F, body[0] >> 4
        bits0 = body[1]
        size = int.from_bytes(body[4:8], "little")
        if ver not in (1, 2, 3):
            raise H5Error("datatype message version %d" % ver)
        order = ">" if bits0 & 1 else "<"
        if cls == 1:
            if size not in (2, 4, 8):
                raise H5Error("float of %d bytes" % size)
            return np.dtype(order + "f%d" % size)
        if cls == 0:
            return np.dtype(order + ("i" if bits0 & 0x08 else "u") + "%d" % size)
        if cls == 3:
            return np.dtype("S%d" % size)
        if cls == 9:
            if (bits0 & 0x0F) == 1:
                return VLEN_STR                              # newer h5py stores a list of bytes this way
            raise H5Error("variable-length sequences are not supported")
        raise H5Error("datatype class %d is not supported" % cls)

    def _raw(self, off, nbytes):
        if off + nbytes > len(self.b):
            raise H5Error("dataset extends past the end of the file")
        return self.b[off:off + nbytes]

    def dataset(self, addr):
        shape = dtype = None
        layout = None
        for mtype, _, body in self.messages(addr):
            if mtype == 0x01:
                shape = self._dataspace(body)
            elif mtype == 0x03:
                dtype = self._datatype(body)
            elif mtype == 0x08:
                layout = bytes(body)
            elif mtype == 0x0B:
                raise H5Error("filtered (compressed) datasets are not supported")
        if shape is None or dtype is None or layout is None:
            raise H5Error("object at %#x is not a dataset" % addr)
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        nbytes = n * dtype.itemsize
        ver = layout[0]
        if ver == 3:
            cls = layout[1]
            if cls == 1:
                a = int.from_bytes(layout[2:2 + self.O], "little")
                if a == UNDEF >> (64 - 8 * self.O):
                    data = bytes(nbytes)                     # never written: fill value 0
                else:
                    data = self._raw(a + self.base, nbytes)
            elif cls == 0:
                size = int.from_bytes(layout[2:4], "little")
                data = layout[4:4 + size][:nbytes]
            else:
                raise H5Error("chunked datasets are not supported (Keras 2.0.x writes contiguous ones)")
        elif ver in (1, 2):
            rank, cls = layout[1], layout[2]
            if cls != 1:
                raise H5Error("layout class %d in a version-%d layout message" % (cls, ver))
            a = int.from_bytes(layout[8:8 + self.O], "little")
            data = self._raw(a + self.base, nbytes)
        else:
            raise H5Error("layout message version %d" % ver)
        return np.frombuffer(data, dtype=dtype, count=n).reshape(shape).copy()

    def attrs(self, addr):
        out = {}
        for mtype, _, body in self.messages(addr):
            if mtype == 0x15:
                raise H5Error("dense attribute storage is not supported")
            if mtype != 0x0C:
                continue
            ver = body[0]
            name_sz = int.from_bytes(body[2:4], "little")
            dt_sz = int.from_bytes(body[4:6], "little")
            ds_sz = int.from_bytes(body[6:8], "little")
            if ver == 1:
                pad = lambda v: (v + 7) & ~7
                p = 8
            elif ver in (2, 3):
                if body[1] & 0x03:
                    raise H5Error("shared attribute datatype/dataspace")
                pad = lambda v: v
                p = 8 + (1 if ver == 3 else 0)
            else:
                raise H5Error("attribute message version %d" % ver)
            name = bytes(body[p:p + name_sz]).split(b"\x00")[0].decode("utf-8")
            p += pad(name_sz)
            try:
                dtype = self._datatype(body[p:p + dt_sz])
            except H5Error:
                continue                                     # e.g. a variable-length model_config string: not needed
            p += pad(dt_sz)
            shape = self._dataspace(body[p:p + ds_sz])
            p += pad(ds_sz)
            n = int(np.prod(shape, dtype=np.int64)) if shape else 1
            if dtype is VLEN_STR:
                esz = 8 + self.O                             # length(4), global heap collection address, object index(4)
                items = [self._global_heap_object(int.from_bytes(body[q + 4:q + 4 + self.O], "little"),
                                                  int.from_bytes(body[q + 4 + self.O:q + esz], "little"),
                                                  int.from_bytes(body[q:q + 4], "little"))
                         for q in range(p, p + n * esz, esz)]
                val = np.array(items, dtype=object)
            else:
                val = np.frombuffer(bytes(body[p:p + n * dtype.itemsize]), dtype=dtype, count=n)
            out[name] = val.reshape(shape) if shape else val[0]
        return out

    def _global_heap_object(self, addr, index, length):
        a = addr + self.base
        if self.b[a:a + 4] != b"GCOL":
            raise H5Error("bad global heap collection signature at %#x" % addr)
        end = a + self.uint(a + 8, self.L)
        p = a + 8 + self.L
        while p + 8 + self.L <= end:
            idx, size = self.uint(p, 2), self.uint(p + 8, self.L)
            if idx == index:
                return self.b[p + 8 + self.L:p + 8 + self.L + length]
            if idx == 0:
                break
            p += 8 + self.L + ((size + 7) & ~7)
        raise H5Error("global heap object %d not found" % index)


def _names(attr):
    arr = np.atleast_1d(attr)
    return [(v.decode("utf-8") if isinstance(v, bytes) else str(v)) for v in arr.tolist()]


def read_keras_weights(path):
    """Keras ``save_weights`` file or full ``model.save`` file -> {layer_name: [arrays]} in ``get_weights()`` order
    (layers without weights are dropped).  Layout: keras/engine/topology.py save_weights_to_hdf5_group -- root (or
    group ``model_weights``) attribute ``layer_names``; per layer a group with attribute ``weight_names`` whose
    entries are dataset paths relative to that group (``conv1/kernel:0``)."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    root = r.root
    if "model_weights" in r.links(root):
        root = r.resolve(root, "model_weights")
    attrs = r.attrs(root)
    if "layer_names" not in attrs:
        raise H5Error("%s has no layer_names attribute: not a Keras weight file" % path)
    out = {}
    for lname in _names(attrs["layer_names"]):
        lg = r.resolve(root, lname)
        wn = r.attrs(lg).get("weight_names")
        if wn is None:
            continue
        arrays = [r.dataset(r.resolve(lg, w)) for w in _names(wn)]
        if arrays:
            out[lname] = arrays
    return out


# ----------------------------------------------------------------------------- writer (same on-disk subset)
def _weight_names(layer, arrays):
    """Keras 2.0.x variable names for the layer kinds this package holds (Conv2D / Dense, BatchNormalization, the
    reference's Scale layer custom_layers.py:93-101)."""
    nd = [a.ndim for a in arrays]
    if len(arrays) == 4 and all(d == 1 for d in nd):
        return ["%s/%s:0" % (layer, k) for k in ("gamma", "beta", "moving_mean", "moving_variance")]
    if len(arrays) == 2 and all(d == 1 for d in nd):
        return ["%s/%s_%s:0" % (layer, layer, k) for k in ("gamma", "beta")]
    if nd[0] >= 2:
        return ["%s/%s:0" % (layer, k) for k in ("kernel", "bias")[:len(arrays)]]
    return ["%s/param_%d:0" % (layer, i) for i in range(len(arrays))]


class _Writer:
    """Lays an HDF5 file out in memory: superblock 0, version-1 object headers, symbol-table groups (one B-tree node,
    symbol nodes of up to 64 entries), contiguous datasets, version-1 attribute messages with fixed-length strings."""
    O = L = 8
    LEAF_K, INTERNAL_K = 32, 16
    UNDEF8 = b"\xff" * 8

    def __init__(self):
        self.buf = bytearray(96)                             # superblock placeholder

    def _alloc(self, data):
        while len(self.buf) % 8:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    @staticmethod
    def _pad8(b):
        return b + b"\x00" * (-len(b) % 8)

    def _msg(self, mtype, body):
        body = self._pad8(body)
        return mtype.to_bytes(2, "little") + len(body).to_bytes(2, "little") + b"\x00\x00\x00\x00" + body

    def _object_header(self, msgs):
        body = b"".join(msgs)
        return b"\x01\x00" + len(msgs).to_bytes(2, "little") + (1).to_bytes(4, "little") + len(body).to_bytes(4, "little") + b"\x00" * 4 + body

    @staticmethod
    def _dataspace(shape):
        return b"\x01" + bytes([len(shape)]) + b"\x00" * 6 + b"".join(int(d).to_bytes(8, "little") for d in shape)

    @staticmethod
    def _dtype_f32():
        return (bytes([0x11, 0x20, 0x1F, 0x00]) + (4).to_bytes(4, "little") + (0).to_bytes(2, "little") + (32).to_bytes(2, "little")
                + bytes([23, 8, 0, 23]) + (127).to_bytes(4, "little"))

    @staticmethod
    def _dtype_str(n):
        return bytes([0x13, 0x01, 0x00, 0x00]) + int(n).to_bytes(4, "little")

    def _attr_strings(self, name, values):
        """1-D array of fixed-length (null-padded) byte strings; a zero-length array keeps itemsize 1."""
        vals = [v.encode("utf8") if isinstance(v, str) else bytes(v) for v in values]
        width = max([len(v) for v in vals] + [1])
        data = b"".join(v.ljust(width, b"\x00") for v in vals)
        return self._attr(name, self._dtype_str(width), self._dataspace((len(vals),)), data)

    def _attr_scalar_string(self, name, value):
        v = value.encode("utf8")
        return self._attr(name, self._dtype_str(len(v)), self._dataspace(()), v)

    def _attr(self, name, dtype, dspace, data):
        nm = name.encode("utf8") + b"\x00"
        body = (b"\x01\x00" + len(nm).to_bytes(2, "little") + len(dtype).to_bytes(2, "little") + len(dspace).to_bytes(2, "little")
                + self._pad8(nm) + self._pad8(dtype) + self._pad8(dspace) + data)
        if len(body) > 0xFFF0:
            raise H5Error("attribute %s is too large for a compact object-header message (%d bytes)" % (name, len(body)))
        return self._msg(0x0C, body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, dtype="<f4")
        data_addr = self._alloc(arr.tobytes())
        layout = b"\x03\x01" + data_addr.to_bytes(8, "little") + int(arr.nbytes).to_bytes(8, "little")
        hdr = self._object_header([self._msg(0x01, self._dataspace(arr.shape)), self._msg(0x03, self._dtype_f32()), self._msg(0x08, layout)])
        return self._alloc(hdr)

    def group(self, links, attr_msgs=()):
        """links: {name: (object header address, (btree, heap) or None)} -> (header address, (btree, heap))."""
        names = sorted(links, key=lambda s: s.encode("utf8"))
        heap = bytearray(8)                                   # offset 0: the empty name every B-tree's first key points at
        offs = {}
        for n in names:
            offs[n] = len(heap)
            heap += self._pad8(n.encode("utf8") + b"\x00")
        heap_data = self._alloc(bytes(heap))
        # free-list head = 1 is the library's "no free block" value (H5HL_FREE_NULL), not the undefined address
        heap_hdr = self._alloc(b"HEAP" + b"\x00" * 4 + len(heap).to_bytes(8, "little") + (1).to_bytes(8, "little") + heap_data.to_bytes(8, "little"))
        per = 2 * self.LEAF_K
        n_nodes = max(1, -(-len(names) // per))
        if n_nodes > 2 * self.INTERNAL_K:
            raise H5Error("group with %d links needs a two-level B-tree" % len(names))
        chunks = [names[i * len(names) // n_nodes:(i + 1) * len(names) // n_nodes] for i in range(n_nodes)]
        keys, children = [0], []
        for ch in chunks:
            node = bytearray(b"SNOD\x01\x00" + len(ch).to_bytes(2, "little"))
            for n in ch:
                addr, sub = links[n]
                node += offs[n].to_bytes(8, "little") + addr.to_bytes(8, "little")
                if sub is not None:
                    node += (1).to_bytes(4, "little") + b"\x00" * 4 + sub[0].to_bytes(8, "little") + sub[1].to_bytes(8, "little")
                else:
                    node += b"\x00" * 24
            node += b"\x00" * (8 + per * 40 - len(node))
            children.append(self._alloc(bytes(node)))
            keys.append(offs[ch[-1]] if ch else 0)
        tree = bytearray(b"TREE\x00\x00" + len(children).to_bytes(2, "little") + self.UNDEF8 + self.UNDEF8)
        for i, c in enumerate(children):
            tree += keys[i].to_bytes(8, "little") + c.to_bytes(8, "little")
        tree += keys[len(children)].to_bytes(8, "little")
        tree += b"\x00" * (24 + (4 * self.INTERNAL_K + 1) * 8 - len(tree))
        btree = self._alloc(bytes(tree))
        stab = self._msg(0x11, btree.to_bytes(8, "little") + heap_hdr.to_bytes(8, "little"))
        return self._alloc(self._object_header([stab] + list(attr_msgs))), (btree, heap_hdr)

    def finish(self, root_addr, root_sub):
        eof = len(self.buf) + (-len(self.buf) % 8)
        self.buf += b"\x00" * (eof - len(self.buf))
        sb = bytearray(SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]))
        sb += self.LEAF_K.to_bytes(2, "little") + self.INTERNAL_K.to_bytes(2, "little") + (0).to_bytes(4, "little")
        sb += (0).to_bytes(8, "little") + self.UNDEF8 + eof.to_bytes(8, "little") + self.UNDEF8
        sb += (0).to_bytes(8, "little") + root_addr.to_bytes(8, "little") + (1).to_bytes(4, "little") + b"\x00" * 4
        sb += root_sub[0].to_bytes(8, "little") + root_sub[1].to_bytes(8, "little")
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def write_keras_weights(path, weights, layer_order=None, full_model=False):
    """{layer_name: [arrays]} -> a Keras 2.0.x ``save_weights`` file (or, with ``full_model``, the ``model_weights`` group
    of a ``model.save`` file) that Keras / h5py and ``read_keras_weights`` read back."""
    order = list(layer_order) if layer_order is not None else list(weights)
    w = _Writer()
    links = {}
    for lname in order:
        arrays = weights[lname]
        names = _weight_names(lname, arrays)
        inner = {}
        for wn, arr in zip(names, arrays):
            inner[wn.split("/", 1)[1]] = (w.dataset(arr), None)
        sub_addr, sub = w.group(inner)                          # the "<layer>/" part of the variable names is a nested group
        g_addr, g_sub = w.group({lname: (sub_addr, sub)} if arrays else {}, [w._attr_strings("weight_names", names)])
        links[lname] = (g_addr, g_sub)
    attrs = [w._attr_strings("layer_names", order), w._attr_scalar_string("backend", "tensorflow"), w._attr_scalar_string("keras_version", "2.0.8")]
    top, top_sub = w.group(links, attrs)
    if full_model:
        top, top_sub = w.group({"model_weights": (top, top_sub)},
                               [w._attr_scalar_string("keras_version", "2.0.8"), w._attr_scalar_string("backend", "tensorflow")])
    with open(path, "wb") as f:
        f.write(w.finish(top, top_sub))
