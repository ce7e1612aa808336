#!/usr/bin/env python3
"""Turns the rocprofv3 CSV output of one round (gpurun_out/<dir>) into the committed
summaries under profiles/: per-kernel time (kernel-trace --stats) and per-launch PMC
counters of the trace kernel, plus profiles/pmc_traffic.json that bench.py reports as
roofline.traffic.

    python tools/summarize_prof.py gpurun_out/r1 r01 [--config C3] [--kernel trace_packet]

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads, so the read side is given both raw and doubled (the doubled figure is the
upper bound used for `traffic`); WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, tag = sys.argv[1], sys.argv[2]
    config = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "C3"
    kernel = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "trace_packet"
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    summary = {"round": tag, "config": config, "kernel_filter": kernel}
    for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
        rows = list(csv.DictReader(open(f)))
        summary["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows]
    bench = os.path.join(src, "kt_bench.json")
    if os.path.exists(bench):
        summary["bench_line_under_rocprof"] = json.load(open(bench))
    counters = defaultdict(list)
    durations = defaultdict(list)  # per counter: dispatch durations (ns) in the pass that collected it
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                durations[r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                summary["kernel_name"] = r["Kernel_Name"]
                summary.setdefault("launch", {"grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"], "lds": r["LDS_Block_Size"],
                                              "vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "scratch": r["Scratch_Size"]})
    pmc = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
    summary["pmc_per_launch_mean"] = pmc
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        rd_raw, wr = pmc["FETCH_SIZE"] * 1024.0, pmc["WRITE_SIZE"] * 1024.0
        summary["hbm_bytes_per_launch"] = {"read_raw": rd_raw, "read_x2_gfx950": 2 * rd_raw, "write": wr,
                                           "total_upper": 2 * rd_raw + wr}
    # effective shader clock of the profiled dispatches (guide, "DVFS give-back"): GRBM_GUI_ACTIVE sums the 8 XCDs
    if "GRBM_GUI_ACTIVE" in pmc:
        clk = [c / 8.0 / d for c, d in zip(counters["GRBM_GUI_ACTIVE"], durations["GRBM_GUI_ACTIVE"]) if d > 0]  # GHz
        summary["effective_clock_ghz"] = sum(clk) / len(clk)
        summary["clock_pass_kernel_ms"] = sum(durations["GRBM_GUI_ACTIVE"]) / len(clk) / 1e6
    # VALU issue roof: a wave64 vector instruction occupies its SIMD-32 for 2 cycles (guide, cycle constants);
    # 4 SIMDs per CU.  floor_ms = the time the launch's vector instructions need at full issue rate.
    if "SQ_INSTS_VALU" in pmc:
        n_simd = 256 * 4
        clk_ghz = summary.get("effective_clock_ghz", 2.4)
        floor_ms = pmc["SQ_INSTS_VALU"] * 2.0 / n_simd / (clk_ghz * 1e9) * 1e3
        kt = [r for r in summary.get("kernel_stats", []) if kernel in r["Name"]]
        kernel_ms = float(kt[0]["AverageNs"]) / 1e6 if kt else None
        summary["valu_issue"] = {"insts_per_launch": pmc["SQ_INSTS_VALU"], "cycles_per_inst": 2, "simds": n_simd,
                                 "clock_ghz": clk_ghz, "clock_source": "GRBM_GUI_ACTIVE pass" if "effective_clock_ghz" in summary else "max clock (no clock pass)",
                                 "floor_ms": floor_ms, "kernel_ms": kernel_ms,
                                 "frac": floor_ms / kernel_ms if kernel_ms else None}
        if "SQ_INSTS_SALU" in pmc:  # one scalar unit per CU, ~1 instruction per cycle (tools/ubench/salu_rate.hip)
            summary["salu_issue_floor_ms"] = (pmc["SQ_INSTS_SALU"] + pmc.get("SQ_INSTS_SMEM", 0.0)) / 256 / (clk_ghz * 1e9) * 1e3
    if "TCC_HIT_sum" in pmc:
        summary["l2_hit_rate"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in pmc:
        wc = pmc["SQ_WAVE_CYCLES"]
        summary["wave_cycle_shares"] = {k: pmc[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in pmc}
        if "SQ_WAVES" in pmc:
            summary["per_wave"] = {k: pmc[k] / pmc["SQ_WAVES"] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS",
                                                                           "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if k in pmc}
    if "hbm_bytes_per_launch" in summary:
        tp = os.path.join(out_dir, "pmc_traffic.json")
        traffic = json.load(open(tp)) if os.path.exists(tp) else {}
        hb = summary["hbm_bytes_per_launch"]
        rec = {"hbm_bytes_per_launch": hb["total_upper"], "read_raw": hb["read_raw"], "write": hb["write"], "round": tag,
               "kernel": summary.get("kernel_name"),
               "note": "FETCH_SIZE*1024*2 (gfx950 half-count correction) + WRITE_SIZE*1024, mean over profiled launches"}
        b = summary.get("bench_line_under_rocprof", {})
        if b.get("roofline", {}).get("rays_per_step_rank0"):
            rec["rays_per_launch"] = b["roofline"]["rays_per_step_rank0"]
        # what bench.py checks before it reports these figures: the tree the passes ran on and the commit they belong to
        rec["source_sha16"] = b.get("source_sha16")
        try:
            import subprocess
            head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
            dirty = subprocess.run(["git", "status", "--porcelain", "--", "messyerraytracer_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
            rec["commit"] = head + ("+uncommitted kernel sources" if dirty else "")
        except OSError:
            rec["commit"] = None
        if "SQ_INSTS_VALU" in pmc:
            rec["valu_insts_per_launch"] = pmc["SQ_INSTS_VALU"]
        if "effective_clock_ghz" in summary:
            rec["clock_ghz"] = summary["effective_clock_ghz"]
        traffic[config] = rec
        json.dump(traffic, open(tp, "w"), indent=1, sort_keys=True)
    with open(os.path.join(out_dir, f"{tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
