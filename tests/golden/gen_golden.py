"""Generates tests/golden/golden.npz by running the REFERENCE aggregator classes (imported from
/root/reference, never copied) over tests/scenarios.py.  Run in the build container only:

    python tests/golden/gen_golden.py            # golden.npz: every scenario but the 16-bit ones
    python tests/golden/gen_golden.py --half     # golden_half.npz: the bfloat16 / float16 scenarios

The reference needs `omegaconf` and `opacus` at import time, which are not installed here; two
placeholder modules (a dict-with-attributes DictConfig, an empty PrivacyEngine) satisfy the
imports.  Neither takes part in the arithmetic.

Per scenario the file holds, for every record, a SHA-256 of the concatenated raw bytes of its
tensors (sorted by name) and -- except for `hash_only` scenarios and tensors above MAX_STORED
elements, of which every 97th element is kept -- the tensors themselves.
"""

import hashlib
import json
import os
import sys
import types

import numpy as np

MAX_STORED = 4096
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class DictConfig(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = DictConfig(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def import_reference():
    om = types.ModuleType("omegaconf")
    om.DictConfig = DictConfig
    om.OmegaConf = object
    sys.modules["omegaconf"] = om
    op = types.ModuleType("opacus")
    op.PrivacyEngine = object
    sys.modules["opacus"] = op
    sys.path.insert(0, "/root/reference/src")
    import appfl.algorithm.aggregator as A
    return A


def record_hash(d) -> str:
    h = hashlib.sha256()
    for k in sorted(d):
        a = np.ascontiguousarray(d[k])
        h.update(k.encode())
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def main():
    import scenarios as S
    A = import_reference()

    from appfl.algorithm.scheduler import AsyncScheduler, SyncScheduler
    from appfl.misc.utils import get_appfl_aggregator

    def factory(cls, model, cfg):
        return getattr(A, cls)(model, DictConfig(cfg), None)

    def loader_factory(cls, model, cfg):
        # the route the server agent takes (agent/server.py:416-428 -> misc/utils.py:132-144)
        return get_appfl_aggregator(cls, model, DictConfig(cfg), None)

    def schedulers(kind, agg, num_clients=0):
        if kind == "sync":
            return SyncScheduler(DictConfig({"num_clients": num_clients, "same_init_model": True}), agg, None)
        return AsyncScheduler(DictConfig({"num_clients": num_clients}), agg, None)

    half = "--half" in sys.argv[1:]
    arrays, index = {}, {}
    for sc in (S.half_scenarios() if half else S.base_scenarios()):
        if sc["kind"].startswith("sched"):
            rec = S.run_scenario(loader_factory, sc, schedulers)
        else:
            rec = S.run_scenario(factory, sc)
        entries = []
        for i, (label, d) in enumerate(rec):
            entries.append({"label": label, "sha256": record_hash(d), "names": sorted(d)})
            if not sc.get("hash_only"):
                for k, a in d.items():
                    if a.size <= MAX_STORED:
                        arrays[f"{sc['id']}|{i}|{k}"] = a
                    else:
                        arrays[f"{sc['id']}|{i}|{k}|s97"] = a.ravel()[::97].copy()
        index[sc["id"]] = entries
        print(f"{sc['id']}: {len(rec)} records")
    arrays["__index__"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    out = os.path.join(HERE, "golden_half.npz" if half else "golden.npz")
    np.savez_compressed(out, **arrays)
    print("wrote", out, os.path.getsize(out), "bytes,", len(index), "scenarios")


if __name__ == "__main__":
    main()
