"""Registration of the device aggregators into an APPFL-like aggregator module (no GPU needed:
only class wiring is checked)."""

import types

from appfl_amd import plugin


def _fake_appfl_module():
    m = types.ModuleType("appfl.algorithm.aggregator")

    class BaseAggregator:
        pass

    class FedAvgAggregator(BaseAggregator):
        def set_model_chunk_size(self, n):
            raise AssertionError("the stock method must not be reached")

        def aggregate(self, *a, **k):
            raise AssertionError("the stock method must not be reached")

    class FedAdamAggregator(FedAvgAggregator):
        pass

    m.BaseAggregator, m.FedAvgAggregator, m.FedAdamAggregator = BaseAggregator, FedAvgAggregator, FedAdamAggregator
    return m


def test_register_adds_prefixed_classes_that_pass_isinstance_checks():
    import appfl_amd.aggregator as A
    m = _fake_appfl_module()
    stock_fedavg = m.FedAvgAggregator
    reg = plugin.register(module=m)
    assert set(reg) == {"MI355X" + n for n in plugin.NAMES}
    cls = m.MI355XFedAdamAggregator
    assert issubclass(cls, A.FedAdamAggregator) and issubclass(cls, stock_fedavg)
    # every public method resolves to the device implementation, not the stock one
    for meth in ("aggregate", "get_parameters", "set_model_chunk_size", "compute_steps"):
        assert getattr(cls, meth) is getattr(A.FedAdamAggregator, meth)
    assert m.FedAvgAggregator is stock_fedavg            # untouched without replace=True


def test_register_replace_swaps_stock_names():
    m = _fake_appfl_module()
    stock = m.FedAvgAggregator
    plugin.register(module=m, replace=True)
    assert m.FedAvgAggregator is not stock and m.StockFedAvgAggregator is stock
    assert issubclass(m.FedAvgAggregator, stock)
    assert m.FedAsyncAggregator.__name__ == "MI355XFedAsyncAggregator"


_REAL_MODULE_CHECK = r"""
import inspect, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests/golden")
import gen_golden
stock_mod = gen_golden.import_reference()            # the reference's appfl.algorithm.aggregator
from appfl.misc.utils import get_appfl_aggregator
import appfl_amd.aggregator as A
from appfl_amd import plugin
stock = {{n: getattr(stock_mod, n) for n in plugin.NAMES}}
reg = plugin.register(replace=True)
for name in plugin.NAMES:
    cls = getattr(stock_mod, name)
    ours = getattr(A, name)
    assert cls is reg[name] and getattr(stock_mod, "Stock" + name) is stock[name], name
    mro = cls.__mro__
    assert mro.index(ours) < mro.index(stock[name]), name            # device methods win
    assert issubclass(cls, stock[name]) and issubclass(cls, stock_mod.BaseAggregator), name
    for meth in ("aggregate", "get_parameters", "set_client_sample_size"):
        impl = next(k for k in mro if meth in vars(k))
        assert impl.__module__.startswith("appfl_amd"), (name, meth, impl)
# the gRPC communicator forwards the chunk size only to isinstance(agg, FedAvgAggregator)
# (comm/grpc/grpc_server_communicator.py:72-88) -- checked against the STOCK class object
for name in ("FedAvgAggregator", "FedAvgMAggregator", "FedAdamAggregator", "FedYogiAggregator", "FedAdagradAggregator"):
    assert issubclass(getattr(stock_mod, name), stock["FedAvgAggregator"]), name
    assert "set_model_chunk_size" in vars(A.FedAvgAggregator)
# call shapes of the reference's schedulers bind to the device signatures
def binds(cls, meth, *a, **k):
    inspect.signature(getattr(cls, meth)).bind(object(), *a, **k)
m, cid = {{"w": 0}}, "c0"
for name in ("FedAvgAggregator", "FedAdamAggregator"):          # sync_scheduler.py:74-76
    binds(getattr(stock_mod, name), "aggregate", {{cid: m}}, val_accuracy={{cid: 0.5}}, _chunk_idx={{cid: 0}})
for name in ("FedAsyncAggregator", "FedBuffAggregator"):        # async_scheduler.py:35
    binds(getattr(stock_mod, name), "aggregate", cid, m, local_steps=5)
    binds(getattr(stock_mod, name), "get_parameters", client_id=cid, init_model=False)
binds(stock_mod.FedCompassAggregator, "aggregate", cid, m, staleness=1, local_steps=7)       # compass_scheduler.py:171-178
binds(stock_mod.FedCompassAggregator, "aggregate", local_models={{cid: m}}, staleness={{cid: 1}}, local_steps={{cid: 7}})
binds(stock_mod.FedQueueAggregator, "aggregate", local_models={{cid: m}}, staleness={{cid: 1}}, local_steps={{cid: 3}})  # queue_scheduler.py:174-178
for name in plugin.NAMES:
    binds(getattr(stock_mod, name), "set_client_sample_size", cid, 10)
# the loader the server agent uses resolves the stock NAME to the device class (misc/utils.py:132-144);
# constructing it needs a GPU, so only the lookup is exercised here
import importlib
assert getattr(importlib.import_module("appfl.algorithm.aggregator"), "FedAdamAggregator") is reg["FedAdamAggregator"]
assert callable(get_appfl_aggregator)
print("ok")
"""


def test_register_against_the_reference_aggregator_module():
    """Container-only: the reference's real `appfl.algorithm.aggregator` (imported from the
    read-only checkout with the two placeholder modules the golden generator uses), in a child
    process so that its modules do not leak into this one."""
    import os
    import subprocess
    import sys

    import pytest
    if not os.path.isdir("/root/reference/src/appfl"):
        pytest.skip("the reference checkout is not present on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _REAL_MODULE_CHECK.format(root=root)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
