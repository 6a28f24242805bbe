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
