"""Staleness weights s(u) of the asynchronous aggregators: host scalars in double precision that
scale a client's mixing weight by how many global updates (u) it is behind.

Formulas of the reference (their double results must match, since they become fp32 kernel
scalars): algorithm/aggregator/fedasync_aggregator.py:266-277 (constant, polynomial, hinge) and
algorithm/aggregator/fedqueue_aggregator.py:177-191 (constant, harmonic, exponential).
"""

from __future__ import annotations

from typing import Callable, Dict, Tuple

# name -> (required keyword names, builder of s(u) from those keywords)
_FEDASYNC: Dict[str, Tuple[Tuple[str, ...], Callable[..., Callable]]] = {
    "constant": ((), lambda: (lambda u: 1)),
    "polynomial": (("a",), lambda a: (lambda u: (u + 1) ** (-a))),
    "hinge": (("a", "b"), lambda a, b: (lambda u: 1 if u <= b else 1.0 / (a * (u - b) + 1.0))),
}


def _exponential_unrunnable(beta):
    def s(u):
        # The reference evaluates torch.exp(-beta * u) on a Python number, which raises TypeError
        # on every call (fedqueue_aggregator.py:189-191); whether exp(-beta*u) in double or in
        # fp32 was meant cannot be told from code that never ran, so the same error is raised.
        raise TypeError("FedQueue staleness_fn 'exponential' calls torch.exp on a Python number in "
                        "the reference and cannot run there either; use 'constant' or 'harmonic'")
    return s


_FEDQUEUE: Dict[str, Tuple[Tuple[str, ...], Callable[..., Callable]]] = {
    "constant": ((), lambda: (lambda u: 1.0)),
    "harmonic": (("beta",), lambda beta: (lambda u: 1.0 / (1.0 + beta * u))),
    "exponential": (("beta",), _exponential_unrunnable),
}


def _build(table, name, kwargs) -> Callable:
    if name not in table:
        raise NotImplementedError
    needed, builder = table[name]
    return builder(*[kwargs[k] for k in needed])      # a missing keyword is a KeyError, as in the reference


def fedasync_staleness(name: str, **kwargs) -> Callable:
    """s(u) for FedAsync / FedBuff / FedCompass; unknown names raise NotImplementedError."""
    return _build(_FEDASYNC, name, kwargs)


def fedqueue_staleness(name: str, **kwargs) -> Callable:
    """s(u) for FedQueue; unknown names raise NotImplementedError."""
    return _build(_FEDQUEUE, name, kwargs)
