"""Independent (pure-Python struct) reader/writer of the reference's ``archer_index`` file.
TEST INFRASTRUCTURE ONLY.  Follows core/aio/archer_tensor_index.cpp:
  Serialize/Deserialize :101-132   u32 count, then (u32 key, TensorStorageMeta) pairs
  operator<< / operator>> :51-86   u32 file_id, i64 offset, u64 size, i64 ndim, i64 dims[ndim], options
  write_options/read_options :11-49  bool pinned, bool requires_grad, i8 dtype, i8 device_index,
                                      i8 device_type, i8 layout
(native little-endian; the reference writes raw in-memory representations)."""
import struct


def write_index(path, entries):
    """entries: {tensor_id: dict(file_id, offset, size, shape, pinned, requires_grad, dtype, device_index, device_type, layout)}"""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(entries)))
        for key, m in entries.items():
            f.write(struct.pack("<I", key))
            f.write(struct.pack("<IqQq", m["file_id"], m["offset"], m["size"], len(m["shape"])))
            for d in m["shape"]:
                f.write(struct.pack("<q", d))
            f.write(struct.pack("<??bbbb", bool(m.get("pinned", False)), bool(m.get("requires_grad", False)), m["dtype"],
                                m.get("device_index", -1), m.get("device_type", 0), m.get("layout", 0)))


def read_index(path):
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<I", f.read(4))
        for _ in range(n):
            (key,) = struct.unpack("<I", f.read(4))
            file_id, offset, size, nd = struct.unpack("<IqQq", f.read(28))
            shape = [struct.unpack("<q", f.read(8))[0] for _ in range(nd)]
            pinned, rg, dtype, di, dty, layout = struct.unpack("<??bbbb", f.read(6))
            out[key] = dict(file_id=file_id, offset=offset, size=size, shape=shape, pinned=pinned, requires_grad=rg,
                            dtype=dtype, device_index=di, device_type=dty, layout=layout)
    return out
