"""Zero-copy view of a client update as it arrives on the wire.

APPFL clients send `torch.save(state_dict)` bytes (comm/mpi/serializer.py:46-50,
comm/grpc/utils.py:20-24) and the server turns them back into CPU tensors with `torch.load`
(agent/server.py:505-513, comm/mpi/serializer.py:70-82) -- one full copy of every tensor before
the aggregator sees it.  `bytes_to_model` parses the zip container and its pickle itself and
returns tensors that are VIEWS of the received buffer, so the only host copy left is the
aggregator's pack into its pinned staging row (`ClientSlab.upload`).

`model_to_bytes` is the other direction (comm/mpi/serializer.py:46-50 `model_to_byte`,
comm/grpc/utils.py:20-24 `serialize_model`): it writes the same archive layout `torch.save`
does -- so any client's `torch.load` reads it -- but copies each storage once, straight into the
output buffer, with the copy and the CRC-32 of the big records spread over threads.

The pickle is read by a restricted unpickler that knows exactly the constructs `torch.save`
emits for a dict of plain tensors (OrderedDict / dict, `_rebuild_tensor_v2`, `_rebuild_parameter`,
typed storages) and refuses everything else, so unlike a plain `torch.load(weights_only=False)`
it cannot be made to run code from the file.
"""

from __future__ import annotations

import collections
import functools
import io
import pickle
import struct
import warnings
import zlib
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Dict, Tuple, Union

import torch

_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
    "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
    "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8,
    "BoolStorage": torch.bool,
}


class WireFormatError(ValueError):
    pass


class _StorageType:
    def __init__(self, dtype):
        self.dtype = dtype


class _StorageRef:
    def __init__(self, key: str, dtype: torch.dtype, numel: int):
        self.key, self.dtype, self.numel = key, dtype, numel


class _TensorRef:
    def __init__(self, storage: _StorageRef, offset: int, size, stride):
        self.storage, self.offset = storage, int(offset)
        self.size, self.stride = tuple(int(s) for s in size), tuple(int(s) for s in stride)


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False,
                       backward_hooks=None, metadata=None):
    if not isinstance(storage, _StorageRef):
        raise WireFormatError("tensor without a typed storage")
    return _TensorRef(storage, storage_offset, size, stride)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


class _Unpickler(pickle.Unpickler):
    _ALLOWED = {
        ("collections", "OrderedDict"): collections.OrderedDict,
        ("torch._utils", "_rebuild_tensor_v2"): _rebuild_tensor_v2,
        ("torch._utils", "_rebuild_parameter"): _rebuild_parameter,
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return self._ALLOWED[(module, name)]
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(_STORAGE_DTYPES[name])
        raise WireFormatError(f"refusing to unpickle {module}.{name}: not part of a plain state dict")

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage"
                and isinstance(pid[1], _StorageType)):
            raise WireFormatError("unexpected persistent id in the state-dict pickle")
        return _StorageRef(str(pid[2]), pid[1].dtype, int(pid[4]))


_EOCD = struct.Struct("<IHHHHIIH")          # end of central directory record
_EOCD64_LOC = struct.Struct("<IIQI")        # zip64 locator
_EOCD64 = struct.Struct("<IQHHIIQQQQ")      # zip64 end of central directory
_CDIR = struct.Struct("<IHHHHHHIIIHHHHHII")  # central directory file header


def _members(buf: memoryview) -> Dict[str, Tuple[int, int]]:
    """{member name: (start, end)} of the STORED members of a zip archive held in `buf`: the
    central directory is walked directly (no per-member objects, no copy of the archive);
    `torch.save` writes every record uncompressed.  Raises WireFormatError for anything else."""
    n = len(buf)
    if n < _EOCD.size:
        raise WireFormatError("not a torch.save zip archive (legacy formats are not supported)")
    # the end record sits at the very end unless the archive carries a comment (torch.save writes none)
    tail_from = max(0, n - (_EOCD.size + 65535))
    tail = bytes(buf[tail_from:])
    at = tail.rfind(b"PK\x05\x06")
    if at < 0:
        raise WireFormatError("not a torch.save zip archive (legacy formats are not supported)")
    eocd = tail_from + at
    (_, _, _, _, count, cd_size, cd_off, _) = _EOCD.unpack_from(buf, eocd)
    if count == 0xFFFF or cd_size == 0xFFFFFFFF or cd_off == 0xFFFFFFFF:
        loc = eocd - _EOCD64_LOC.size
        if loc < 0:
            raise WireFormatError("truncated zip64 archive")
        sig, _, eocd64, _ = _EOCD64_LOC.unpack_from(buf, loc)
        if sig != 0x07064B50 or eocd64 + _EOCD64.size > n:
            raise WireFormatError("bad zip64 locator")
        (sig, _, _, _, _, _, _, count, cd_size, cd_off) = _EOCD64.unpack_from(buf, eocd64)
        if sig != 0x06064B50:
            raise WireFormatError("bad zip64 end record")
    if cd_off + cd_size > n:
        raise WireFormatError("central directory outside the buffer")
    out: Dict[str, Tuple[int, int]] = {}
    pos = cd_off
    unpack_cd, unpack_lh = _CDIR.unpack_from, struct.Struct("<IHHHHHIIIHH").unpack_from
    for _ in range(count):
        (sig, _, _, _, method, _, _, _, csize, usize, name_len, extra_len, comment_len, _, _, _,
         header) = unpack_cd(buf, pos)
        if sig != 0x02014B50:
            raise WireFormatError("bad central directory entry")
        name = bytes(buf[pos + 46:pos + 46 + name_len]).decode("utf-8")
        if usize == 0xFFFFFFFF or csize == 0xFFFFFFFF or header == 0xFFFFFFFF:
            # zip64 extra field: the 64-bit values, in this order, for the fields that overflowed
            ex, ex_end = pos + 46 + name_len, pos + 46 + name_len + extra_len
            while ex + 4 <= ex_end:
                tag, size = struct.unpack_from("<HH", buf, ex)
                if tag == 1:
                    vals = struct.unpack_from("<%dQ" % (size // 8), buf, ex + 4)
                    it = iter(vals)
                    if usize == 0xFFFFFFFF:
                        usize = next(it)
                    if csize == 0xFFFFFFFF:
                        csize = next(it)
                    if header == 0xFFFFFFFF:
                        header = next(it)
                    break
                ex += 4 + size
        pos += 46 + name_len + extra_len + comment_len
        if method != 0 or csize != usize:
            raise WireFormatError("compressed archive members are not supported")
        if header + 30 > n:
            raise WireFormatError("archive member outside the buffer")
        lh = unpack_lh(buf, header)
        if lh[0] != 0x04034B50:
            raise WireFormatError("bad local file header")
        start = header + 30 + lh[9] + lh[10]
        if start + usize > n:
            raise WireFormatError("archive member reaches past the buffer")
        out[name] = (start, start + usize)
    return out


def _contiguous(size, stride) -> bool:
    expect = 1
    for s, st in zip(reversed(size), reversed(stride)):
        if s != 1 and st != expect:
            return False
        expect *= s
    return True


def bytes_to_model(model_bytes: Union[bytes, bytearray, memoryview]) -> Dict[str, torch.Tensor]:
    """`torch.save`d state dict -> {name: CPU tensor}.  Contiguous tensors are read-only views
    of `model_bytes` (which must stay alive as long as they are used); anything else (strided
    tensors, unaligned storages) is copied.  Drop-in for `ServerAgent._bytes_to_model` when
    compression is off."""
    try:
        return _bytes_to_model(memoryview(model_bytes).cast("B"))
    except WireFormatError:
        raise
    except (struct.error, pickle.UnpicklingError, UnicodeDecodeError, StopIteration, EOFError,
            AttributeError, IndexError, KeyError, TypeError, ValueError, RuntimeError, OverflowError,
            NotImplementedError) as e:
        # malformed or unusual archives must take the caller's safe fallback, not escape
        raise WireFormatError(f"cannot read the archive in place: {type(e).__name__}: {e}") from e


def _bytes_to_model(buf: memoryview) -> Dict[str, torch.Tensor]:
    members = _members(buf)
    pkl = [n for n in members if n.endswith("/data.pkl") or n == "data.pkl"]
    if len(pkl) != 1:
        raise WireFormatError("archive has no data.pkl")
    prefix = pkl[0][: -len("data.pkl")]
    a, b = members[pkl[0]]
    obj = _Unpickler(io.BytesIO(buf[a:b])).load()
    if not isinstance(obj, dict):
        raise WireFormatError(f"expected a state dict, got {type(obj).__name__}")
    out: Dict[str, Any] = type(obj)() if isinstance(obj, collections.OrderedDict) else {}
    with warnings.catch_warnings():
        # a bytes object is read-only; the views are only ever read
        warnings.filterwarnings("ignore", message="The given buffer is not writable")
        _fill_views(out, obj, buf, members, prefix)
    return out


_ITEMSIZE = {dt: torch.empty((), dtype=dt).element_size() for dt in _STORAGE_DTYPES.values()}


def _fill_views(out, obj, buf, members, prefix) -> None:
    data_prefix = prefix + "data/"
    frombuffer = torch.frombuffer
    for name, ref in obj.items():
        if not isinstance(ref, _TensorRef):
            raise WireFormatError(f"entry {name!r} is not a tensor")
        storage = ref.storage
        span = members.get(data_prefix + storage.key)
        if span is None:
            raise WireFormatError(f"storage {storage.key} missing from the archive")
        s0, s1 = span
        size, stride = ref.size, ref.stride
        if storage.numel < 0 or ref.offset < 0 or len(size) != len(stride):
            raise WireFormatError(f"entry {name!r} has a negative or inconsistent size, stride or offset")
        numel = 1
        for d in size:
            if d < 0:
                raise WireFormatError(f"entry {name!r} has a negative or inconsistent size, stride or offset")
            numel *= d
        for d in stride:
            if d < 0:
                raise WireFormatError(f"entry {name!r} has a negative or inconsistent size, stride or offset")
        dtype = storage.dtype
        item = _ITEMSIZE[dtype]
        if s1 - s0 < storage.numel * item:
            raise WireFormatError(f"storage {storage.key} is truncated")
        start = s0 + ref.offset * item
        if numel == 0:
            out[name] = torch.empty(size, dtype=dtype)
        elif _contiguous(size, stride):
            if start + numel * item > s1:
                raise WireFormatError(f"entry {name!r} reaches past its storage")
            t = frombuffer(buf, dtype=dtype, count=numel, offset=start)
            out[name] = t.view(size) if len(size) != 1 else t
        else:
            last = ref.offset + sum((d - 1) * st for d, st in zip(size, stride))
            if last >= storage.numel:
                raise WireFormatError(f"entry {name!r} reaches past its storage")
            whole = frombuffer(buf, dtype=dtype, count=storage.numel, offset=s0)
            out[name] = torch.as_strided(whole, size, stride, ref.offset).clone()


# ---------------------------------------------------------------------------------------------
# state dict -> torch.save-compatible bytes

_DTYPE_STORAGES = {v: k for k, v in _STORAGE_DTYPES.items()}
_ALIGN = 64                      # torch.save aligns record payloads to 64 bytes
_CHUNK = 8 << 20                 # bytes per copy / CRC task
WRITE_THREADS = 8


_NATIVE_CHUNK = 1 << 20          # bytes per CRC piece of the native fused copy


def _native_copy_crc32():
    """`fa_host_copy_crc32` of the device library when it is built (host code only: no GPU is
    touched), else None -- serialising a state dict must also work where only this module is used."""
    try:
        from . import _native
        return _native.load().fa_host_copy_crc32
    except Exception:
        return None


class _StorageMarker:
    def __init__(self, key: str, dtype: torch.dtype, numel: int):
        self.key, self.dtype, self.numel = key, dtype, numel


def _gf2_times(mat, vec):
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _gf2_square(mat):
    return [_gf2_times(mat, mat[i]) for i in range(32)]


@functools.lru_cache(maxsize=64)
def _append_zeros_operator(nbytes: int):
    """GF(2) matrix that turns crc32(A) into crc32(A + nbytes zero bytes) (before the final
    xor with crc32(B)); cached because a model's records have the same lengths every round."""
    op = [0xEDB88320] + [1 << i for i in range(31)]    # one zero bit
    for _ in range(3):
        op = _gf2_square(op)                           # one zero byte
    result = None
    n = nbytes
    while n:
        if n & 1:
            result = op if result is None else [_gf2_times(op, col) for col in result]
        n >>= 1
        if n:
            op = _gf2_square(op)
    return tuple(result)


def crc32_combine(crc1: int, crc2: int, len2: int) -> int:
    """CRC-32 of A+B from crc32(A), crc32(B) and len(B) (what zlib's crc32_combine computes)."""
    if len2 <= 0:
        return crc1
    return _gf2_times(_append_zeros_operator(len2), crc1) ^ crc2


class _Pickler(pickle.Pickler):
    """Emits what torch.save emits for a dict of CPU tensors: `_rebuild_tensor_v2` calls over
    persistent storage ids; tensors sharing a storage share its record."""

    def __init__(self, file):
        super().__init__(file, protocol=2)
        self.records = []            # (key, uint8 tensor over the storage bytes, keep-alive)
        self._by_storage = {}

    def persistent_id(self, obj):
        if isinstance(obj, _StorageMarker):
            return ("storage", getattr(torch, _DTYPE_STORAGES[obj.dtype]), obj.key, "cpu", obj.numel)
        return None

    def reducer_override(self, obj):
        if not isinstance(obj, torch.Tensor):
            return NotImplemented
        t = obj.detach()
        if t.device.type != "cpu" or t.layout != torch.strided or t.dtype not in _DTYPE_STORAGES:
            raise WireFormatError(f"cannot serialise a {t.dtype} tensor on {t.device}")
        st = t.untyped_storage()
        item = t.element_size()
        ident = (st.data_ptr(), st.nbytes(), t.dtype) if st.nbytes() else (id(t), 0, t.dtype)
        marker = self._by_storage.get(ident)
        if marker is None:
            marker = _StorageMarker(str(len(self.records)), t.dtype, st.nbytes() // item)
            self._by_storage[ident] = marker
            raw = torch.empty(0, dtype=torch.uint8).set_(st) if st.nbytes() else torch.empty(0, dtype=torch.uint8)
            self.records.append((marker.key, raw, t))
        return (torch._utils._rebuild_tensor_v2,
                (marker, t.storage_offset(), tuple(t.shape), tuple(t.stride()), False,
                 collections.OrderedDict()))


def _local_header(name: bytes, size: int, crc: int, pad: int) -> bytes:
    extra = b"" if pad == 0 else b"FB" + struct.pack("<H", pad - 4) + b"Z" * (pad - 4)
    return struct.pack("<IHHHHHIIIHH", 0x04034B50, 0, 0x0800, 0, 0, 0x21, crc, size, size,
                       len(name), len(extra)) + name + extra


def _central_header(name: bytes, size: int, crc: int, offset: int) -> bytes:
    return struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 0, 0, 0x0800, 0, 0, 0x21, crc, size, size,
                       len(name), 0, 0, 0, 0, 0, offset) + name


def model_to_bytes(state_dict: Dict[str, torch.Tensor], prefix: str = "archive") -> memoryview:
    """State dict of CPU tensors -> the bytes `torch.save(state_dict, f)` would put in `f`, as far
    as `torch.load` can tell (same records, pickle and alignment; views of one storage share
    one record).  Returns a writable byte memoryview (buffer protocol; wrap in `bytes()` where an
    immutable object is required, e.g. a protobuf field)."""
    import numpy as np

    if not isinstance(state_dict, dict):
        raise WireFormatError("expected a state dict")
    pkl = io.BytesIO()
    pickler = _Pickler(pkl)
    pickler.dump(state_dict)
    small = [("data.pkl", pkl.getvalue()), (".format_version", b"1"), (".storage_alignment", b"64"),
             ("byteorder", b"little")]
    tail = [("version", b"3\n")]
    # layout pass
    entries = []                     # (name bytes, size, header offset, pad, payload or record index)
    pos = 0

    def place(name: str, size: int, payload):
        nonlocal pos
        nm = f"{prefix}/{name}".encode()
        data_at = pos + 30 + len(nm)
        pad = (-(data_at + 4)) % _ALIGN + 4          # extra field is at least its 4-byte header
        entries.append([nm, size, pos, pad, payload])
        pos = data_at + pad + size

    for name, data in small:
        place(name, len(data), data)
    for key, raw, _ in pickler.records:
        place(f"data/{key}", raw.numel(), raw)
    for name, data in tail:
        place(name, len(data), data)
    cd_at = pos
    cd_size = sum(46 + len(e[0]) for e in entries)
    total = cd_at + cd_size + 22
    if total >= 0xFFFFFFFF or len(entries) >= 0xFFFF:
        raise WireFormatError("archives of 4 GiB or more need zip64; use torch.save")
    # uninitialised, so the pages are first touched by the copy threads rather than zero-filled
    view = np.empty(total, dtype=np.uint8)
    out = memoryview(view)
    # payload pass: big records are copied and checksummed in chunks -- by the native library's
    # host threads in one fused pass where it is built, else on a Python thread pool
    tasks = []                       # (entry index, chunk start, chunk end)
    crcs = [0] * len(entries)
    native = _native_copy_crc32()
    for i, (nm, size, at, pad, payload) in enumerate(entries):
        start = at + 30 + len(nm) + pad
        if isinstance(payload, bytes):
            out[start:start + size] = payload
            crcs[i] = zlib.crc32(payload)
        elif native is not None and size >= (1 << 20):
            import ctypes as C
            n_chunks = (size + _NATIVE_CHUNK - 1) // _NATIVE_CHUNK
            parts = (C.c_uint32 * n_chunks)()
            rc = native(C.c_void_p(view.ctypes.data + start), C.c_void_p(payload.data_ptr()), size,
                        _NATIVE_CHUNK, parts, WRITE_THREADS)
            if rc != 0:
                raise WireFormatError(f"fa_host_copy_crc32 failed ({rc})")
            c = parts[0]
            for k in range(1, n_chunks):
                c = crc32_combine(c, parts[k], min(_NATIVE_CHUNK, size - k * _NATIVE_CHUNK))
            crcs[i] = c
        else:
            for a in range(0, size, _CHUNK):
                tasks.append((i, start, a, min(a + _CHUNK, size)))

    def work(task):
        i, start, a, b = task
        src = entries[i][4].numpy()[a:b]
        view[start + a:start + b] = src
        return i, b - a, zlib.crc32(view[start + a:start + b])

    if tasks:
        if len(tasks) == 1:
            done = [work(tasks[0])]
        else:
            with ThreadPoolExecutor(max_workers=WRITE_THREADS) as pool:
                done = list(pool.map(work, tasks))
        started = set()
        for i, n, c in done:
            crcs[i] = c if i not in started else crc32_combine(crcs[i], c, n)
            started.add(i)
    cd = bytearray()
    for i, (nm, size, at, pad, _) in enumerate(entries):
        hdr = _local_header(nm, size, crcs[i], pad)
        out[at:at + len(hdr)] = hdr
        cd += _central_header(nm, size, crcs[i], at)
    out[cd_at:cd_at + cd_size] = bytes(cd)
    out[cd_at + cd_size:] = struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, len(entries), len(entries),
                                        cd_size, cd_at, 0)
    return out
