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
import zipfile
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


class _BufferFile:
    """Read-only, seekable file over a memoryview -- just enough for zipfile to find the central
    directory without copying the archive (io.BytesIO would copy all of it)."""

    def __init__(self, buf: memoryview):
        self._buf, self._pos = buf, 0

    def seekable(self):
        return True

    def readable(self):
        return True

    def tell(self):
        return self._pos

    def seek(self, off, whence=0):
        base = 0 if whence == 0 else (self._pos if whence == 1 else len(self._buf))
        self._pos = max(0, base + off)
        return self._pos

    def read(self, n=-1):
        end = len(self._buf) if n is None or n < 0 else min(len(self._buf), self._pos + n)
        out = bytes(self._buf[self._pos:end])
        self._pos = max(self._pos, end)
        return out

    def close(self):
        pass


def _stored_range(buf: memoryview, info: zipfile.ZipInfo) -> Tuple[int, int]:
    """Byte range of an uncompressed member inside the archive buffer."""
    if info.compress_type != zipfile.ZIP_STORED:
        raise WireFormatError("compressed archive members are not supported")
    h = info.header_offset
    if h < 0 or info.file_size < 0 or h + 30 > len(buf):
        raise WireFormatError("archive member outside the buffer")
    sig, = struct.unpack_from("<I", buf, h)
    if sig != 0x04034B50:
        raise WireFormatError("bad local file header")
    name_len, extra_len = struct.unpack_from("<HH", buf, h + 26)
    start = h + 30 + name_len + extra_len
    if start + info.file_size > len(buf):
        raise WireFormatError("archive member reaches past the buffer")
    return start, start + info.file_size


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
    except (struct.error, pickle.UnpicklingError, zipfile.BadZipFile, zipfile.LargeZipFile, EOFError,
            AttributeError, IndexError, KeyError, TypeError, ValueError, RuntimeError, OverflowError,
            NotImplementedError) as e:
        # malformed or unusual archives must take the caller's safe fallback, not escape
        raise WireFormatError(f"cannot read the archive in place: {type(e).__name__}: {e}") from e


def _bytes_to_model(buf: memoryview) -> Dict[str, torch.Tensor]:
    try:
        zf = zipfile.ZipFile(_BufferFile(buf))
    except zipfile.BadZipFile as e:
        raise WireFormatError("not a torch.save zip archive (legacy formats are not supported)") from e
    members = {i.filename: i for i in zf.infolist()}
    pkl = [n for n in members if n.endswith("/data.pkl") or n == "data.pkl"]
    if len(pkl) != 1:
        raise WireFormatError("archive has no data.pkl")
    prefix = pkl[0][: -len("data.pkl")]
    a, b = _stored_range(buf, members[pkl[0]])
    obj = _Unpickler(io.BytesIO(buf[a:b])).load()
    if not isinstance(obj, dict):
        raise WireFormatError(f"expected a state dict, got {type(obj).__name__}")
    out: Dict[str, Any] = type(obj)() if isinstance(obj, collections.OrderedDict) else {}
    with warnings.catch_warnings():
        # a bytes object is read-only; the views are only ever read
        warnings.filterwarnings("ignore", message="The given buffer is not writable")
        _fill_views(out, obj, buf, members, prefix)
    return out


def _fill_views(out, obj, buf, members, prefix) -> None:
    for name, ref in obj.items():
        if not isinstance(ref, _TensorRef):
            raise WireFormatError(f"entry {name!r} is not a tensor")
        member = members.get(f"{prefix}data/{ref.storage.key}")
        if member is None:
            raise WireFormatError(f"storage {ref.storage.key} missing from the archive")
        s0, s1 = _stored_range(buf, member)
        if (ref.storage.numel < 0 or ref.offset < 0 or any(d < 0 for d in ref.size)
                or any(d < 0 for d in ref.stride) or len(ref.size) != len(ref.stride)):
            raise WireFormatError(f"entry {name!r} has a negative or inconsistent size, stride or offset")
        item = torch.empty((), dtype=ref.storage.dtype).element_size()
        if s1 - s0 < ref.storage.numel * item:
            raise WireFormatError(f"storage {ref.storage.key} is truncated")
        numel = 1
        for s in ref.size:
            numel *= s
        start = s0 + ref.offset * item
        if numel == 0:
            out[name] = torch.empty(ref.size, dtype=ref.storage.dtype)
        elif _contiguous(ref.size, ref.stride):
            if start + numel * item > s1:
                raise WireFormatError(f"entry {name!r} reaches past its storage")
            t = torch.frombuffer(buf, dtype=ref.storage.dtype, count=numel, offset=start)
            out[name] = t.view(ref.size)
        else:
            last = ref.offset + sum((d - 1) * st for d, st in zip(ref.size, ref.stride))
            if last >= ref.storage.numel:
                raise WireFormatError(f"entry {name!r} reaches past its storage")
            whole = torch.frombuffer(buf, dtype=ref.storage.dtype, count=ref.storage.numel, offset=s0)
            out[name] = torch.as_strided(whole, ref.size, ref.stride, ref.offset).clone()


# ---------------------------------------------------------------------------------------------
# state dict -> torch.save-compatible bytes

_DTYPE_STORAGES = {v: k for k, v in _STORAGE_DTYPES.items()}
_ALIGN = 64                      # torch.save aligns record payloads to 64 bytes
_CHUNK = 8 << 20                 # bytes per copy / CRC task
WRITE_THREADS = 8


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
    # payload pass: big records are copied and checksummed in chunks on a thread pool
    tasks = []                       # (entry index, chunk start, chunk end)
    crcs = [0] * len(entries)
    for i, (nm, size, at, pad, payload) in enumerate(entries):
        start = at + 30 + len(nm) + pad
        if isinstance(payload, bytes):
            out[start:start + size] = payload
            crcs[i] = zlib.crc32(payload)
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
