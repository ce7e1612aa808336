#!/usr/bin/env python3
"""Per-launch means of the counters tools/pmc_kernel.sh collected, one column per tag, for the trace_* kernel with
the most time in each tag:   python tools/pmc_table.py gpurun_out/<tag> [...]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(tag_dir):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(tag_dir, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "trace_" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if not acc:
        return None, {}, 0.0
    k = max(acc, key=lambda k: sum(dur[k]))
    return k, {c: sum(v) / len(v) for c, v in acc[k].items()}, sum(dur[k]) / len(dur[k])


def main():
    cols = []
    for d in sys.argv[1:]:
        k, m, ms = load(d)
        cols.append((os.path.basename(d.rstrip("/")), k, m, ms))
    names = sorted({c for _, _, m, _ in cols for c in m})
    print("%-28s" % "counter" + "".join("%22s" % t for t, _, _, _ in cols))
    for t, k, _, ms in cols:
        print("#", t, k, "profiled ms %.3f" % ms)
    for c in names:
        print("%-28s" % c + "".join("%22.4g" % m.get(c, float("nan")) for _, _, m, _ in cols))
    # derived, per wave
    print("--- per wave (SQ_* cycle counters are quad-cycles: x4) ---")
    for t, k, m, ms in cols:
        w = m.get("SQ_WAVES", 0)
        if not w:
            continue
        d = {}
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_SALU",
                  "SQ_INST_CYCLES_SMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
            if c in m:
                d[c + "_cyc"] = round(4 * m[c] / w)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_IFETCH"):
            if c in m:
                d[c] = round(m[c] / w, 1)
        if "SQ_INST_LEVEL_SMEM" in m and m.get("SQ_INSTS_SMEM"):
            d["smem_latency_cyc(level/insts, x4?)"] = round(m["SQ_INST_LEVEL_SMEM"] / m["SQ_INSTS_SMEM"], 1)
        if "SQ_IFETCH_LEVEL" in m and m.get("SQ_IFETCH"):
            d["ifetch_latency(level/fetches)"] = round(m["SQ_IFETCH_LEVEL"] / m["SQ_IFETCH"], 1)
        print(t, json.dumps(d))


if __name__ == "__main__":
    main()
