import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for th in ("4", "8", "12", "16", "8", "16"):
    env = dict(os.environ, APPFL_AMD_PACK_THREADS=th)
    code = ("import sys; sys.path.insert(0, %r); import bench, json, torch; from appfl_amd import workloads as W; torch.cuda.set_device(0);"
            "print(json.dumps({k: bench.end_to_end(W.WORKLOADS[k], rounds=3, arrivals=5) for k in ('resnet18_fedavg_k100','resnet50_fedadam_k64')}))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    if r.returncode:
        print(th, "FAILED", r.stderr[-400:]); continue
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(th, {k: (v["ms_per_round"], v["after_last_arrival_ms"], v["stage_GBps"]) for k, v in d.items()}, flush=True)
