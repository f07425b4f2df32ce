#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/run_profiles.sh) into profiles/<tag>_pmc_summary.json (+ profiles/traffic.json):
per kernel, per launch: duration from the kernel trace and the mean of every collected counter.
HBM traffic follows MI355X_MICROARCH.md "HBM": bytes = FETCH_SIZE*1024 (x2 on gfx950: the counter tallies 128-B requests as 64 B
for wide reads) + WRITE_SIZE*1024; both raw and corrected values are kept."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    out = defaultdict(lambda: {"launches": 0})
    for f in glob.glob(os.path.join(base, "trace", "*", "*_kernel_trace.csv")):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in dur.items():
            out[k]["launches"] = len(v)
            out[k]["avg_duration_us"] = sum(v) / len(v) / 1e3
            out[k]["min_duration_us"] = min(v) / 1e3
    for f in glob.glob(os.path.join(base, "*", "*", "*_counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[short(r["Kernel_Name"])].setdefault("vgpr", int(r.get("VGPR_Count", 0) or 0))
            out[short(r["Kernel_Name"])].setdefault("lds_bytes", int(r.get("LDS_Block_Size", 0) or 0))
        for k, cs in acc.items():
            for c, v in cs.items():
                out[k][c] = sum(v) / len(v)
    for k, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_raw"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
            d["hbm_bytes_corrected"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        if "SQ_ACTIVE_INST_VALU" in d and "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
            d["valu_active_frac_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]
        if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d and d["SQ_ACTIVE_INST_VALU"] > 0:
            d["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_ACTIVE_INST_VALU"] * 64.0)
        # vector memory path: busy cycles of the texture-address / texture-data units summed over 256 CUs against the kernel's cycles
        # (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        if "TD_TD_BUSY_sum" in d and d.get("GRBM_GUI_ACTIVE", 0) > 0:
            d["td_busy_frac"] = (d["TD_TD_BUSY_sum"] / 256.0) / (d["GRBM_GUI_ACTIVE"] / 8.0)
            d["ta_busy_frac"] = (d["TA_TA_BUSY_sum"] / 256.0) / (d["GRBM_GUI_ACTIVE"] / 8.0)
            if d.get("TA_FLAT_READ_WAVEFRONTS_sum", 0) > 0:
                d["td_busy_cycles_per_vmem_wave_instruction"] = d["TD_TD_BUSY_sum"] / d["TA_FLAT_READ_WAVEFRONTS_sum"]
    keep = {k: v for k, v in out.items() if k.startswith("k_")}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(keep, open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 3 and sys.argv[3].startswith("big="):
        # a large-scene pass (tools/run_profiles_big.sh): its kernels go to profiles/traffic_big.json under the scene's name
        scene_key = sys.argv[3][4:]
        path = os.path.join(ROOT, "profiles", "traffic_big.json")
        big = json.load(open(path)) if os.path.exists(path) else {}
        fields = ("hbm_bytes_corrected", "hbm_bytes_raw", "l2_hit_rate", "valu_active_frac_of_wave_cycles", "valu_lane_utilisation", "td_busy_frac",
                  "ta_busy_frac", "avg_duration_us", "launches", "vgpr", "lds_bytes", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum",
                  "TCP_UTCL1_REQUEST_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TA_FLAT_READ_WAVEFRONTS_sum")
        entry = {"source": "profiles/%s_pmc_summary.json: rocprofv3 --pmc passes over tools/bench_bigscenes.py %s (separate runs per counter group; "
                           "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB per MI355X_MICROARCH.md's gfx950 correction)" % (tag, scene_key)}
        for k, v in keep.items():
            if k.startswith("k_raycast_step<false") or k.startswith("k_raycast_step<true") or k.startswith("k_integrate") or k.startswith("k_raycast_persist"):
                e = {f: v[f] for f in fields if f in v}
                if "hbm_bytes_corrected" in e:
                    e["hbm_bytes_per_launch"] = e["hbm_bytes_corrected"]
                    e["hbm_GBps"] = e["hbm_bytes_corrected"] / (e["avg_duration_us"] * 1e-6) / 1e9 if e.get("avg_duration_us") else None
                entry[k] = e
        big[scene_key] = entry
        json.dump(big, open(path, "w"), indent=1, sort_keys=True)
        for k, v in sorted(keep.items()):
            print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
        return
    # what bench.py reads: one entry per kernel of the bench line's rooflines
    traffic = {}
    want = {"k_raycast_closest": "k_raycast_step<false", "k_raycast_anyhit": "k_raycast_step<true", "k_integrate": "k_integrate",
            "k_trace_frame": "k_trace_frame"}
    fields = ("hbm_bytes_corrected", "hbm_bytes_raw", "l2_hit_rate", "valu_active_frac_of_wave_cycles", "valu_lane_utilisation", "td_busy_frac", "ta_busy_frac",
              "avg_duration_us", "launches", "vgpr", "lds_bytes")
    for key, prefix in want.items():
        for k, v in keep.items():
            if k.startswith(prefix):
                e = {f: v[f] for f in fields if f in v}
                if "hbm_bytes_corrected" in e:
                    e["hbm_bytes_per_launch"] = e["hbm_bytes_corrected"]
                e["kernel"] = k
                traffic[key] = e
    if "k_integrate" in traffic:
        traffic["k_integrate"]["spp"] = int(sys.argv[2]) if len(sys.argv) > 2 else 16  # spp of the profiled pass (tools/profile_kernels.py)
    traffic["source"] = ("profiles/%s_pmc_summary.json: rocprofv3 --pmc passes over tools/profile_kernels.py (FETCH_SIZE / WRITE_SIZE / TCC / SQ in "
                         "separate runs; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB per MI355X_MICROARCH.md's gfx950 correction)" % tag)
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(keep.items()):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
