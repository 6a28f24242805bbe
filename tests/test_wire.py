"""Zero-copy parsing of torch.save'd state dicts (no GPU needed)."""

import collections
import io
import pickle
import warnings

import pytest
import torch

import scenarios as S
from appfl_amd import wire

warnings.filterwarnings("ignore", message="The given buffer is not writable")


def _save(obj) -> bytes:
    b = io.BytesIO()
    torch.save(obj, b)
    return b.getvalue()


def test_views_equal_torch_load_and_share_the_buffer():
    model = S.ToyModel()
    sd = model.state_dict()
    sd["strided"] = torch.arange(24.0).reshape(4, 6).t()          # non-contiguous
    sd["sliced"] = torch.arange(100, dtype=torch.int32)[10:20]       # storage offset
    sd["empty"] = torch.empty(0, 3)
    raw = _save(sd)
    got = wire.bytes_to_model(raw)
    want = torch.load(io.BytesIO(raw), map_location="cpu", weights_only=True)
    assert isinstance(got, collections.OrderedDict) and list(got) == list(want)
    base = memoryview(raw)
    lo = torch.frombuffer(base, dtype=torch.uint8, count=1, offset=0).data_ptr()
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert torch.equal(got[k], want[k]), k
        if k not in ("strided", "empty"):
            assert lo <= got[k].data_ptr() < lo + len(raw), f"{k} was copied"


def test_plain_dict_and_parameters():
    lin = torch.nn.Linear(3, 2)
    raw = _save({"w": lin.weight, "b": lin.bias.detach(), "n": torch.tensor(7)})
    got = wire.bytes_to_model(raw)
    assert type(got) is dict and torch.equal(got["w"], lin.weight.detach())
    assert got["n"].dtype == torch.int64 and int(got["n"]) == 7


class _Evil:
    def __reduce__(self):
        import os
        return (os.system, ("echo pwned",))


def test_refuses_anything_but_a_plain_state_dict():
    with pytest.raises(wire.WireFormatError):
        wire.bytes_to_model(_save({"x": _Evil()}))
    with pytest.raises(wire.WireFormatError):
        wire.bytes_to_model(_save([torch.ones(3)]))
    with pytest.raises(wire.WireFormatError):
        wire.bytes_to_model(pickle.dumps({"a": 1}))
    with pytest.raises(wire.WireFormatError):
        wire.bytes_to_model(_save({"x": "not a tensor"}))


def _state_for_writing():
    flat = torch.arange(40, dtype=torch.float32) * 0.25
    sd = collections.OrderedDict()
    sd["view_a"] = flat[:24].view(4, 6)                 # two views of one storage
    sd["view_b"] = flat[24:]
    sd["strided"] = torch.arange(12.0).reshape(3, 4).t()
    sd["scalar"] = torch.tensor(5, dtype=torch.int64)
    sd["i32"] = torch.arange(7, dtype=torch.int32)
    sd["empty"] = torch.empty(0, 3)
    sd["param"] = torch.nn.Parameter(torch.ones(2, 2))
    return sd


def test_written_archive_loads_with_torch_load_and_has_valid_checksums():
    import zipfile
    sd = _state_for_writing()
    raw = wire.model_to_bytes(sd)
    assert zipfile.ZipFile(io.BytesIO(raw)).testzip() is None        # every CRC-32 checks out
    for weights_only in (True, False):
        got = torch.load(io.BytesIO(raw), map_location="cpu", weights_only=weights_only)
        assert isinstance(got, collections.OrderedDict) and list(got) == list(sd)
        for k in sd:
            assert got[k].dtype == sd[k].dtype and got[k].shape == sd[k].shape, k
            assert torch.equal(got[k], sd[k].detach()), k
        assert got["view_a"].untyped_storage().data_ptr() == got["view_b"].untyped_storage().data_ptr()
    back = wire.bytes_to_model(raw)
    assert all(torch.equal(back[k], sd[k].detach()) for k in sd)
    # same records as torch.save writes, each payload 64-byte aligned
    ref = io.BytesIO()
    torch.save(sd, ref)
    names = lambda b: sorted(i.filename for i in zipfile.ZipFile(io.BytesIO(b)).infolist()
                             if "serialization_id" not in i.filename)
    assert names(raw) == names(ref.getvalue())
    spans = wire._members(memoryview(raw))
    assert sorted(spans) == sorted(i.filename for i in zipfile.ZipFile(io.BytesIO(raw)).infolist())
    for name, (a, b) in spans.items():
        assert a % 64 == 0, name
        assert bytes(raw[a:b]) == zipfile.ZipFile(io.BytesIO(raw)).read(name), name   # the walk agrees with zipfile


def test_large_record_is_checksummed_in_chunks():
    import zipfile
    import zlib
    big = torch.randn((wire._CHUNK // 4) * 3 + 12345)
    raw = wire.model_to_bytes({"w": big, "n": torch.tensor(3)})
    zf = zipfile.ZipFile(io.BytesIO(raw))
    assert zf.testzip() is None
    assert zf.getinfo("archive/data/0").CRC == zlib.crc32(big.numpy().tobytes())
    assert torch.equal(torch.load(io.BytesIO(raw), weights_only=True)["w"], big)


def test_crc32_combine_matches_zlib():
    import zlib
    a, b = bytes(range(256)) * 37, b"federated" * 1001
    assert wire.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert wire.crc32_combine(zlib.crc32(a), zlib.crc32(b""), 0) == zlib.crc32(a)


def test_corrupted_archives_raise_wire_format_error_only():
    """Whatever is wrong with a payload, bytes_to_model raises WireFormatError (so callers take
    their torch.load fallback) and nothing else."""
    import io
    import struct
    state = {"w": torch.arange(12, dtype=torch.float32).reshape(3, 4), "n": torch.tensor(3)}
    buf = io.BytesIO()
    torch.save(state, buf)
    good = buf.getvalue()
    assert torch.equal(wire.bytes_to_model(good)["w"], state["w"])
    bad_cases = [good[: len(good) // 2], good[:-30] + b"\x00" * 30, b"", b"PK\x03\x04" + b"\x00" * 40]
    # a central-directory size that does not match the archive
    cd = good.rfind(b"PK\x01\x02")
    bad_cases.append(good[:cd + 20] + struct.pack("<II", 2 ** 31, 2 ** 31) + good[cd + 28:])
    # garbage in the pickle stream
    p = good.find(b"data.pkl") + len("data.pkl")
    bad_cases.append(good[:p + 40] + bytes(reversed(good[p + 40:p + 90])) + good[p + 90:])
    for blob in bad_cases:
        try:
            out = wire.bytes_to_model(blob)
        except wire.WireFormatError:
            continue
        # a corruption that happens to leave a readable archive must still give tensors
        assert all(isinstance(t, torch.Tensor) for t in out.values())


def test_archive_walk_reads_zip64_records():
    """Archives beyond 4 GiB (a 1.3 B-parameter update is 5.2 GB) carry their sizes and offsets in
    zip64 records; a hand-built miniature with every field overflowed must walk the same way."""
    import struct
    payload = b"0123456789abcdef" * 3
    name = b"archive/data/0"
    local = struct.pack("<IHHHHHIIIHH", 0x04034B50, 45, 0, 0, 0, 0, 0, 0xFFFFFFFF, 0xFFFFFFFF, len(name), 20)
    local += name + struct.pack("<HHQQ", 1, 16, len(payload), len(payload))
    body = b"\0" * 7 + local + payload                 # the member does not start at offset 0
    header_at = 7
    extra = struct.pack("<HHQQQ", 1, 24, len(payload), len(payload), header_at)
    cdir = struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 45, 45, 0, 0, 0, 0, 0, 0xFFFFFFFF, 0xFFFFFFFF,
                       len(name), len(extra), 0, 0, 0, 0, 0xFFFFFFFF) + name + extra
    cd_off = len(body)
    eocd64 = struct.pack("<IQHHIIQQQQ", 0x06064B50, 44, 45, 45, 0, 0, 1, 1, len(cdir), cd_off)
    locator = struct.pack("<IIQI", 0x07064B50, 0, cd_off + len(cdir), 1)
    eocd = struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, 0xFFFF, 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0)
    raw = body + cdir + eocd64 + locator + eocd
    spans = wire._members(memoryview(raw))
    assert list(spans) == ["archive/data/0"]
    a, b = spans["archive/data/0"]
    assert raw[a:b] == payload
    # a directory entry pointing outside the buffer is refused, not followed
    bad = raw.replace(struct.pack("<Q", header_at), struct.pack("<Q", 10 ** 9), 1)
    with pytest.raises(wire.WireFormatError):
        wire._members(memoryview(bad))
