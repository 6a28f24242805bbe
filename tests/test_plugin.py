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
