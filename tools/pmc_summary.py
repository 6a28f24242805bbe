"""Development tool: fold rocprofv3 outputs of tools/profile.sh into profiles/ (tracked).

usage: python tools/pmc_summary.py <round-tag> <workload>=<profile dir> ...
Writes profiles/<tag>_<workload>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary),
profiles/<tag>_<workload>_pmc.json and updates profiles/pmc_traffic.json (read by bench.py).

HBM bytes per launch of the dominant kernel = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB;
on gfx950 FETCH_SIZE counts a 16-byte-per-lane streaming read at half its bytes, WRITE_SIZE is
exact for 16-byte-per-lane stores).
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avg(path, counter, kernel_substr):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name = max(vals, key=lambda k: sum(vals[k]))
    v = vals[name]
    return name, sum(v) / len(v), len(v)


def main():
    tag = sys.argv[1]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    tpath = os.path.join(prof, "pmc_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for spec in sys.argv[2:]:
        wl, d = spec.split("=")
        shutil.copy(os.path.join(d, "trace", "t_kernel_stats.csv"),
                    os.path.join(prof, f"{tag}_{wl}_kernel_stats.csv"))
        stats = list(csv.DictReader(open(os.path.join(d, "trace", "t_kernel_stats.csv"))))
        top = stats[0]
        m = re.search(r"::(\w+)\s*[<(]", top["Name"])
        sub = m.group(1) if m else top["Name"]
        kname, fetch_kib, nf = counter_avg(os.path.join(d, "fetch", "f_counter_collection.csv"), "FETCH_SIZE", sub)
        _, write_kib, nw = counter_avg(os.path.join(d, "write", "w_counter_collection.csv"), "WRITE_SIZE", sub)
        extra = {}
        for sub_dir, pre, names in (("stall", "s", ("TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
                                                    "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_CYCLE_sum",
                                                    "GRBM_GUI_ACTIVE")),
                                    ("l2", "h", ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_LEVEL_sum",
                                                 "TCC_EA0_RDREQ_sum"))):
            path = os.path.join(d, sub_dir, f"{pre}_counter_collection.csv")
            if os.path.exists(path):
                for cname in names:
                    try:
                        extra[cname] = counter_avg(path, cname, sub)[1]
                    except ValueError:
                        pass
        if "TCC_HIT_sum" in extra and "TCC_MISS_sum" in extra:
            extra["l2_hit_rate"] = extra["TCC_HIT_sum"] / max(1.0, extra["TCC_HIT_sum"] + extra["TCC_MISS_sum"])
        if "TCC_CYCLE_sum" in extra:
            for k in ("TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"):
                if k in extra:
                    extra[k.replace("_sum", "_per_cycle")] = extra[k] / extra["TCC_CYCLE_sum"]
        if "TCC_EA0_RDREQ_LEVEL_sum" in extra and extra.get("TCC_EA0_RDREQ_sum"):
            extra["read_latency_l2_cycles"] = extra["TCC_EA0_RDREQ_LEVEL_sum"] / extra["TCC_EA0_RDREQ_sum"]
        rec = {
            "workload": wl, "kernel": kname,
            "kernel_avg_ns": float(top["AverageNs"]), "kernel_calls": int(top["Calls"]),
            "FETCH_SIZE_KiB_per_launch_raw": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
            "launches_sampled": [nf, nw],
            "hbm_read_bytes_per_launch": 2 * fetch_kib * 1024,
            "hbm_write_bytes_per_launch": write_kib * 1024,
            "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
            "correction": "FETCH_SIZE doubled (gfx950 counts 128-B streaming read requests at 64 B)",
            "l2_fabric_counters_per_launch": extra,
        }
        json.dump(rec, open(os.path.join(prof, f"{tag}_{wl}_pmc.json"), "w"), indent=1)
        traffic[wl] = {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "source": f"{tag}_{wl}_pmc.json"}
        print(wl, rec["kernel_avg_ns"], rec["hbm_bytes_per_launch"])
    json.dump(traffic, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
