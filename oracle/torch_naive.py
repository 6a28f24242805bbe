"""Per-tensor PyTorch-CPU restatement of the reference's aggregation loops, kept as close to the
reference's op sequence as a restatement can be (K x T Python loop of `sub`, `mul`, `add_`,
fresh temporaries, the reference's `gc.collect()` placement).  TEST INFRASTRUCTURE ONLY: it is
what the benchmark times as the CPU baseline on the GPU box's host cores, and it is checked
against the golden vectors like the other oracles.  Never imported by the product package.

Follows (relative to src/appfl/): algorithm/aggregator/fedavg_aggregator.py:276-299, 329-368;
fedadam_aggregator.py:41-113; fedyogi_aggregator.py:60-112; fedavgm_aggregator.py:37-66;
fedadagrad_aggregator.py:59-103; fedasync_aggregator.py:243-258, 156-162;
misc/memory_utils.py:15-36, 75-131.
"""

from __future__ import annotations

import gc
from typing import Dict, Optional

import torch


def _is_int(t: torch.Tensor) -> bool:
    return t.dtype in (torch.int32, torch.int64)


def clone_state(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    with torch.no_grad():
        for k, v in state.items():
            out[k] = v.clone().detach()
    gc.collect()
    return out


class NaiveFedAvg:
    """FedAvg / FedAvgM / FedAdam / FedYogi / FedAdagrad on CPU state dicts (`opt` selects)."""

    def __init__(self, opt: str = "none", cfg: Optional[dict] = None, named_parameters=None):
        self.opt, self.cfg = opt, cfg or {}
        self.named_parameters = named_parameters
        self.global_state: Optional[Dict[str, torch.Tensor]] = None
        self.step: Dict[str, torch.Tensor] = {}
        self.m_vector: Dict[str, torch.Tensor] = {}
        self.v_vector: Dict[str, torch.Tensor] = {}

    def compute_steps(self, local_models, weights):
        with torch.no_grad():
            for name in self.global_state:
                if (self.named_parameters is not None and name not in self.named_parameters) \
                        or _is_int(self.global_state[name]):
                    continue
                self.step[name] = torch.zeros_like(self.global_state[name])
            for cid, model in local_models.items():
                weight = weights[cid]
                for name in model:
                    if name in self.step:
                        diff = model[name] - self.global_state[name]
                        weighted_diff = diff * weight
                        self.step[name].add_(weighted_diff)
                        del diff, weighted_diff
        if self.opt != "none":
            self._server_opt()

    def _server_opt(self):
        c = self.cfg
        with torch.no_grad():
            if self.opt == "avgm":
                if not self.v_vector:
                    for name in self.step:
                        self.v_vector[name] = torch.zeros_like(self.step[name])
                    gc.collect()
                for name in self.step:
                    momentum_term = self.v_vector[name] * c["server_momentum_param_1"]
                    self.v_vector[name] = momentum_term.add_(self.step[name])
                    self.step[name] = self.v_vector[name].clone()
                gc.collect()
                return
            if not self.m_vector:
                for name in self.step:
                    self.m_vector[name] = torch.zeros_like(self.step[name])
                    self.v_vector[name] = torch.zeros_like(self.step[name]) + c["server_adapt_param"] ** 2
                gc.collect()
            for name in self.step:
                momentum_term = self.m_vector[name] * c["server_momentum_param_1"]
                step_term = self.step[name] * (1 - c["server_momentum_param_1"])
                self.m_vector[name] = momentum_term.add_(step_term)
                step_squared = torch.square(self.step[name])
                if self.opt == "adam":
                    v_term = self.v_vector[name] * c["server_momentum_param_2"]
                    squared_term = step_squared * (1 - c["server_momentum_param_2"])
                    self.v_vector[name] = v_term.add_(squared_term)
                elif self.opt == "yogi":
                    sign_term = torch.sign(self.v_vector[name] - step_squared)
                    yogi_scaled = torch.mul(step_squared, sign_term) * (1 - c["server_momentum_param_2"])
                    self.v_vector[name].sub_(yogi_scaled)
                else:
                    self.v_vector[name].add_(step_squared)
                numerator = c["server_learning_rate"] * self.m_vector[name]
                denominator = torch.sqrt(self.v_vector[name]) + c["server_adapt_param"]
                self.step[name] = numerator.div_(denominator)
            gc.collect()

    def aggregate(self, local_models, weights):
        if self.global_state is None:
            first = next(iter(local_models.values()))
            self.global_state = {}
            with torch.no_grad():
                for name, tensor in first.items():
                    self.global_state[name] = tensor.detach().clone()
            gc.collect()
        self.compute_steps(local_models, weights)
        with torch.no_grad():
            for name in self.global_state:
                if name in self.step:
                    self.global_state[name].add_(self.step[name])
                else:
                    param_sum = torch.zeros_like(self.global_state[name])
                    for _, model in local_models.items():
                        if _is_int(param_sum):
                            param_sum = param_sum + model[name]
                        else:
                            param_sum.add_(model[name])
                    if _is_int(param_sum):
                        self.global_state[name] = torch.div(param_sum, len(local_models)).type(param_sum.dtype)
                    else:
                        self.global_state[name] = param_sum.div_(len(local_models))
        gc.collect()
        self.step.clear()
        return clone_state(self.global_state)


def naive_async_mix(global_state: Dict[str, torch.Tensor], local_model: Dict[str, torch.Tensor],
                    alpha_t: float) -> Dict[str, torch.Tensor]:
    """One FedAsync arrival (model-based, optimize_memory): mutates `local_model` like the
    reference and returns the cloned new global state."""
    with torch.no_grad():
        for name in global_state:
            if _is_int(global_state[name]):
                global_state[name] = local_model[name]
                continue
            diff = local_model[name].sub_(global_state[name], alpha=1)
            step = torch.zeros_like(diff).add_(diff, alpha=alpha_t)
            global_state[name].add_(step, alpha=1)
    gc.collect()
    return clone_state(global_state)
