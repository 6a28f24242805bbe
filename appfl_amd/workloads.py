"""State-dict shape tables of the benchmark / test workloads, generated programmatically
(torchvision is not a dependency): torchvision-style ResNet-18 / ResNet-50 / ViT-B/16, the MNIST
CNN of the reference's tests (tests/resources/model/cnn.py:6-37) and a flat vector.

Each table is an ordered list of (name, shape, dtype, is_parameter) in ``state_dict()`` order.
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import torch

Row = Tuple[str, Tuple[int, ...], torch.dtype, bool]


def _bn(prefix: str, c: int) -> List[Row]:
    return [
        (f"{prefix}.weight", (c,), torch.float32, True),
        (f"{prefix}.bias", (c,), torch.float32, True),
        (f"{prefix}.running_mean", (c,), torch.float32, False),
        (f"{prefix}.running_var", (c,), torch.float32, False),
        (f"{prefix}.num_batches_tracked", (), torch.int64, False),
    ]


def _conv(name: str, cout: int, cin: int, k: int) -> List[Row]:
    return [(f"{name}.weight", (cout, cin, k, k), torch.float32, True)]


def resnet(depth: int, num_classes: int = 1000) -> List[Row]:
    """torchvision.models.resnet18 / resnet50 state-dict layout."""
    bottleneck = depth >= 50
    blocks = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3)}[depth]
    exp = 4 if bottleneck else 1
    rows = _conv("conv1", 64, 3, 7) + _bn("bn1", 64)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), blocks), start=1):
        for b in range(nblk):
            stride = 2 if (li > 1 and b == 0) else 1
            p = f"layer{li}.{b}"
            if bottleneck:
                rows += _conv(f"{p}.conv1", planes, inplanes, 1) + _bn(f"{p}.bn1", planes)
                rows += _conv(f"{p}.conv2", planes, planes, 3) + _bn(f"{p}.bn2", planes)
                rows += _conv(f"{p}.conv3", planes * 4, planes, 1) + _bn(f"{p}.bn3", planes * 4)
            else:
                rows += _conv(f"{p}.conv1", planes, inplanes, 3) + _bn(f"{p}.bn1", planes)
                rows += _conv(f"{p}.conv2", planes, planes, 3) + _bn(f"{p}.bn2", planes)
            if b == 0 and (stride != 1 or inplanes != planes * exp):
                rows += _conv(f"{p}.downsample.0", planes * exp, inplanes, 1)
                rows += _bn(f"{p}.downsample.1", planes * exp)
            inplanes = planes * exp
    rows += [("fc.weight", (num_classes, 512 * exp), torch.float32, True),
             ("fc.bias", (num_classes,), torch.float32, True)]
    return rows


def vit_b_16(num_classes: int = 1000) -> List[Row]:
    """torchvision.models.vit_b_16 state-dict layout (152 tensors, no buffers)."""
    d, mlp, layers, seq = 768, 3072, 12, 197
    f = torch.float32
    rows: List[Row] = [
        ("class_token", (1, 1, d), f, True),
        ("conv_proj.weight", (d, 3, 16, 16), f, True),
        ("conv_proj.bias", (d,), f, True),
        ("encoder.pos_embedding", (1, seq, d), f, True),
    ]
    for i in range(layers):
        p = f"encoder.layers.encoder_layer_{i}"
        rows += [
            (f"{p}.ln_1.weight", (d,), f, True), (f"{p}.ln_1.bias", (d,), f, True),
            (f"{p}.self_attention.in_proj_weight", (3 * d, d), f, True),
            (f"{p}.self_attention.in_proj_bias", (3 * d,), f, True),
            (f"{p}.self_attention.out_proj.weight", (d, d), f, True),
            (f"{p}.self_attention.out_proj.bias", (d,), f, True),
            (f"{p}.ln_2.weight", (d,), f, True), (f"{p}.ln_2.bias", (d,), f, True),
            (f"{p}.mlp.0.weight", (mlp, d), f, True), (f"{p}.mlp.0.bias", (mlp,), f, True),
            (f"{p}.mlp.3.weight", (d, mlp), f, True), (f"{p}.mlp.3.bias", (d,), f, True),
        ]
    rows += [("encoder.ln.weight", (d,), f, True), ("encoder.ln.bias", (d,), f, True),
             ("heads.head.weight", (num_classes, d), f, True),
             ("heads.head.bias", (num_classes,), f, True)]
    return rows


def mnist_cnn(num_channel: int = 1, num_classes: int = 10, num_pixel: int = 28) -> List[Row]:
    x = num_pixel
    x = (x - 4) // 2
    x = (x - 4) // 2
    f = torch.float32
    return [
        ("conv1.weight", (32, num_channel, 5, 5), f, True), ("conv1.bias", (32,), f, True),
        ("conv2.weight", (64, 32, 5, 5), f, True), ("conv2.bias", (64,), f, True),
        ("fc1.weight", (512, 64 * x * x), f, True), ("fc1.bias", (512,), f, True),
        ("fc2.weight", (num_classes, 512), f, True), ("fc2.bias", (num_classes,), f, True),
    ]


def flat(n: int) -> List[Row]:
    return [("flat", (n,), torch.float32, True)]


def as_dtype(rows: List[Row], dtype: torch.dtype) -> List[Row]:
    """The same table with every fp32 tensor held in `dtype` (a model after `.to(dtype)`)."""
    return [(n, shp, dtype if dt == torch.float32 else dt, p) for n, shp, dt, p in rows]


def numel(rows: List[Row], params_only: bool = False) -> int:
    total = 0
    for _, shape, _, is_param in rows:
        if params_only and not is_param:
            continue
        n = 1
        for s in shape:
            n *= s
        total += n
    return total


# name -> (table builder, aggregator class name, clients per round / buffer, description)
WORKLOADS: Dict[str, Dict] = {
    "mnist_cnn_fedavg_k2": {"table": mnist_cnn, "cls": "FedAvgAggregator", "K": 2},
    "resnet18_fedavg_k100": {"table": lambda: resnet(18), "cls": "FedAvgAggregator", "K": 100},
    "resnet50_fedadam_k64": {"table": lambda: resnet(50), "cls": "FedAdamAggregator", "K": 64},
    "resnet50_fedyogi_k64": {"table": lambda: resnet(50), "cls": "FedYogiAggregator", "K": 64},
    # mid-size client lists of a two-moment optimizer (the persistent kernel with the late epilogue)
    "resnet50_fedadam_k16": {"table": lambda: resnet(50), "cls": "FedAdamAggregator", "K": 16},
    "resnet50_fedadam_k24": {"table": lambda: resnet(50), "cls": "FedAdamAggregator", "K": 24},
    "resnet50_fedadam_k32": {"table": lambda: resnet(50), "cls": "FedAdamAggregator", "K": 32},
    # a bfloat16 model: rows and state stay 2 bytes per element, every operation is rounded to bfloat16
    "resnet18_fedavg_k100_bf16": {"table": lambda: as_dtype(resnet(18), torch.bfloat16),
                                  "cls": "FedAvgAggregator", "K": 100},
    "resnet50_fedadam_k64_bf16": {"table": lambda: as_dtype(resnet(50), torch.bfloat16),
                                  "cls": "FedAdamAggregator", "K": 64},
    "resnet50_fedadam_k64_f16": {"table": lambda: as_dtype(resnet(50), torch.float16),
                                 "cls": "FedAdamAggregator", "K": 64},
    "vitb16_fedasync_bf16": {"table": lambda: as_dtype(vit_b_16(), torch.bfloat16), "cls": "FedAsyncAggregator", "K": 1},
    "vitb16_fedbuff_flush_bf16": {"table": lambda: as_dtype(vit_b_16(), torch.bfloat16), "cls": "FedBuffAggregator",
                                  "K": 16, "kind": "flush"},
    "vitb16_fedcompass_group_k8_bf16": {"table": lambda: as_dtype(vit_b_16(), torch.bfloat16),
                                        "cls": "FedCompassAggregator", "K": 8, "mode": "convex"},
    "resnet50_fedavg_k64": {"table": lambda: resnet(50), "cls": "FedAvgAggregator", "K": 64},
    "resnet50_fedavgm_k64": {"table": lambda: resnet(50), "cls": "FedAvgMAggregator", "K": 64},
    "vitb16_fedasync": {"table": vit_b_16, "cls": "FedAsyncAggregator", "K": 1},
    "vitb16_fedbuff_k16": {"table": vit_b_16, "cls": "FedBuffAggregator", "K": 16},
    # ADMM server updates: K clients' primal, dual and previous-primal rows resident in HBM
    "resnet50_iiadmm_k16": {"table": lambda: resnet(50), "cls": "IIADMMAggregator", "K": 16},
    "resnet50_iceadmm_k16": {"table": lambda: resnet(50), "cls": "ICEADMMAggregator", "K": 16},
    # one arrival of a FedBuff buffer of each kind (bytes per parameter: the first arrival of a
    # buffer reads x, g and writes the buffer = 12; a middle one also reads the buffer = 16; the
    # flushing one also writes g = 20)
    "vitb16_fedbuff_first": {"table": vit_b_16, "cls": "FedBuffAggregator", "K": 16, "kind": "first"},
    "vitb16_fedbuff_middle": {"table": vit_b_16, "cls": "FedBuffAggregator", "K": 16, "kind": "middle"},
    "vitb16_fedbuff_flush": {"table": vit_b_16, "cls": "FedBuffAggregator", "K": 16, "kind": "flush"},
    # FedCompass group mix g = (1 - sum w) g + sum w_k x_k and FedQueue replace g = sum w_k x_k
    # (fa_weighted_mix) over K rows of the model's parameters
    "vitb16_fedcompass_group_k8": {"table": vit_b_16, "cls": "FedCompassAggregator", "K": 8, "mode": "convex"},
    "vitb16_fedqueue_replace_k16": {"table": vit_b_16, "cls": "FedQueueAggregator", "K": 16, "mode": "replace"},
    "flat162m_fedavg_k32": {"table": lambda: flat(162_500_000), "cls": "FedAvgAggregator", "K": 32},
    # the whole 1.3 B-parameter vector on ONE GPU: 166 GB of client rows in HBM
    "flat1300m_fedavg_k32": {"table": lambda: flat(1_300_000_000), "cls": "FedAvgAggregator", "K": 32},
}

# server-optimizer settings of the reference's example configurations
# (examples/resources/configs/mnist/server_fedadam.yaml:53-59, server_fedyogi.yaml:53-59,
#  server_fedavgm.yaml:53-56, server_fedasync.yaml:55-61, server_fedbuff.yaml:55-62)
EXAMPLE_CONFIGS = {
    "FedAvgAggregator": {"client_weights_mode": "equal"},
    "FedAvgMAggregator": {"client_weights_mode": "equal", "server_momentum_param_1": 0.9},
    "FedAdamAggregator": {"client_weights_mode": "equal", "server_learning_rate": 0.1,
                          "server_adapt_param": 0.001, "server_momentum_param_1": 0.9,
                          "server_momentum_param_2": 0.99},
    "FedYogiAggregator": {"client_weights_mode": "equal", "server_learning_rate": 0.01,
                          "server_adapt_param": 0.001, "server_momentum_param_1": 0.9,
                          "server_momentum_param_2": 0.99},
    "FedAdagradAggregator": {"client_weights_mode": "equal", "server_learning_rate": 0.1,
                             "server_adapt_param": 0.001, "server_momentum_param_1": 0.9},
    "FedAsyncAggregator": {"client_weights_mode": "equal", "staleness_fn": "polynomial",
                           "staleness_fn_kwargs": {"a": 0.5}, "alpha": 0.9, "num_clients": 16},
    "FedBuffAggregator": {"client_weights_mode": "equal", "staleness_fn": "polynomial",
                          "staleness_fn_kwargs": {"a": 0.5}, "alpha": 0.9, "num_clients": 16,
                          "K": 16},
}
