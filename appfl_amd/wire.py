"""Zero-copy view of a client update as it arrives on the wire.

APPFL clients send `torch.save(state_dict)` bytes (comm/mpi/serializer.py:46-50,
comm/grpc/utils.py:20-24) and the server turns them back into CPU tensors with `torch.load`
(agent/server.py:505-513, comm/mpi/serializer.py:70-82) -- one full copy of every tensor before
the aggregator sees it.  `bytes_to_model` parses the zip container and its pickle itself and
returns tensors that are VIEWS of the received buffer, so the only host copy left is the
aggregator's pack into its pinned staging row (`ClientSlab.upload`).

The pickle is read by a restricted unpickler that knows exactly the constructs `torch.save`
emits for a dict of plain tensors (OrderedDict / dict, `_rebuild_tensor_v2`, `_rebuild_parameter`,
typed storages) and refuses everything else, so unlike a plain `torch.load(weights_only=False)`
it cannot be made to run code from the file.
"""

from __future__ import annotations

import collections
import io
import pickle
import struct
import warnings
import zipfile
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


def _stored_range(buf: memoryview, info: zipfile.ZipInfo) -> Tuple[int, int]:
    """Byte range of an uncompressed member inside the archive buffer."""
    if info.compress_type != zipfile.ZIP_STORED:
        raise WireFormatError("compressed archive members are not supported")
    h = info.header_offset
    sig, = struct.unpack_from("<I", buf, h)
    if sig != 0x04034B50:
        raise WireFormatError("bad local file header")
    name_len, extra_len = struct.unpack_from("<HH", buf, h + 26)
    start = h + 30 + name_len + extra_len
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
    buf = memoryview(model_bytes).cast("B")
    try:
        zf = zipfile.ZipFile(io.BytesIO(buf))     # BytesIO shares nothing, but only reads headers
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
            whole = torch.frombuffer(buf, dtype=ref.storage.dtype, count=ref.storage.numel, offset=s0)
            out[name] = torch.as_strided(whole, ref.size, ref.stride, ref.offset).clone()
