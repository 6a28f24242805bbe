"""bench.py end to end on the GPU: the one-GPU line and, through torch.distributed.run with ONE
nccl rank, the block-dealt multi-GPU form with its three timed variants."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_one_rank_nccl_run_reports_the_three_sharded_forms():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded", "--workload",
           "resnet50_fedadam_k64", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["scaling"] == "strong" and out["n_gpus"] == 1
    # the process group's own size and every rank's device identity (host | uuid | PCI address)
    assert out["ranks"] == 1 and len(out["devices"]) == 1 and out["devices"][0].count("|") == 2
    for form in ("no_collective", "allgather", "allgather_pipelined"):
        assert out[form]["value"] > 0 and out[form]["ms_per_step"] > 0
    assert out["value"] == out["allgather_pipelined"]["value"]
    assert out["reduce_scatter_whole_updates"]["value"] > 0           # the whole-update mode, same clients
    assert out["config"]["params_total"] == 25557032                # parameters, without BN statistics
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1
    assert rf["algorithmic_bytes_per_launch"] == 4 * (64 + 6) * out["config"]["params_per_gpu"]
    # the other ResNet-50 K = 64 optimizer rides along, dealt to the ranks the same way
    (yogi,) = out["workloads"]
    assert yogi["workload"] == "resnet50_fedyogi_k64" and yogi["params_total"] == 25557032
    for form in ("no_collective", "allgather", "allgather_pipelined"):
        assert yogi[form]["value"] > 0
    assert yogi["value"] == yogi["allgather_pipelined"]["value"] and 0 < yogi["frac"] < 1


def test_one_gpu_line_lists_every_benchmark_configuration():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-big", "--no-end-to-end"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_line(r.stdout)
    assert out["config"]["workload"] == "resnet18_fedavg_k100" and out["scaling"] == "weak"
    names = [w["workload"] for w in out["workloads"]]
    assert names == ["resnet50_fedadam_k64", "resnet50_fedyogi_k64", "resnet50_fedavgm_k64",
                     "resnet18_fedavg_k100_bf16", "resnet50_fedadam_k64_bf16", "resnet50_fedadam_k64_f16",
                     "vitb16_fedasync_bf16",
                     "vitb16_fedcompass_group_k8_bf16", "flat162m_fedavg_k32", "vitb16_fedasync", "vitb16_fedbuff_k16",
                     "vitb16_fedbuff_first", "vitb16_fedbuff_middle", "vitb16_fedbuff_flush",
                     "vitb16_fedcompass_group_k8", "vitb16_fedqueue_replace_k16"]
    by_name = {w["workload"]: w for w in out["workloads"]}
    for w in out["workloads"]:
        assert 0 < w["frac"] < 1
        assert abs(w["frac"] - w["algorithmic_bytes_per_launch"] / (w["kernel_avg_ms"] * 1e-3) / 8e12) < 2e-3
    n_vit = by_name["vitb16_fedbuff_k16"]["algorithmic_bytes_per_launch"] // 16
    assert [by_name["vitb16_fedbuff_" + k]["algorithmic_bytes_per_launch"] // n_vit
            for k in ("first", "middle", "flush")] == [12, 16, 20]
    assert by_name["vitb16_fedcompass_group_k8"]["algorithmic_bytes_per_launch"] == 4 * (8 + 2) * n_vit
    assert by_name["vitb16_fedqueue_replace_k16"]["algorithmic_bytes_per_launch"] == 4 * (16 + 1) * n_vit
    # the bfloat16 model moves 2-byte elements: half the algorithmic bytes of the fp32 one
    assert by_name["resnet50_fedadam_k64_bf16"]["algorithmic_bytes_per_launch"] * 2 == \
        by_name["resnet50_fedadam_k64"]["algorithmic_bytes_per_launch"]
    assert by_name["resnet50_fedadam_k64_bf16"]["kernel"] == "reduce_late_half_kernel<adam,bfloat16>"
    assert by_name["resnet18_fedavg_k100_bf16"]["kernel"] == "reduce_late_half_kernel<none,bfloat16>"
    assert by_name["resnet50_fedadam_k64_f16"]["kernel"] == "reduce_late_half_kernel<adam,float16>"
    assert out["north_star_target"]["workload"] == "resnet50_fedadam_k64"
    assert out["north_star_target"]["frac"] == out["workloads"][0]["frac"]
    # cold-state figures (state vectors rotated over independent sets) beside the warm ones
    assert out["ranks"] == 1 and len(out["devices"]) == 1
    for rec in (out["roofline"], by_name["resnet50_fedadam_k64"], by_name["flat162m_fedavg_k32"]):
        assert 0 < rec["frac_cold"] < 1 and rec["state_sets"] >= 3
    assert out["north_star_target"]["frac_cold"] == by_name["resnet50_fedadam_k64"]["frac_cold"]
    assert out["roofline"]["traffic"] is None or out["roofline"]["traffic_source"]
