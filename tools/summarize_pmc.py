#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (tools/profile_gpu.sh) into profiles/<name>.md + pmc json.

Per sdfr kernel: average duration (kernel trace) and every PMC counter averaged per launch.
FETCH_SIZE is doubled on gfx950 as MI355X_MICROARCH.md (HBM section) prescribes; both
FETCH_SIZE and WRITE_SIZE are in KiB.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("render_forward_kernel", "render_backward_kernel", "forward_prologue_kernel",
              "backward_prologue_kernel", "pack_cells_kernel", "view_setup_kernel",
              "pose_reduce_kernel", "pc_loss", "sampler", "decoder", "fillBuffer"):
        if k in name:
            return k + ("<64>" if "<64>" in name else "")
    return None


def main():
    src, name = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    lines = [f"# {name}: rocprofv3 summary (per-launch averages)", ""]
    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        d = dur[k]
        # the timed steps are the last 20 launches (tools/profile_gpu.sh pre-warms the trace run); without a
        # pre-warm drop the warm-up launch
        dd = d[-20:] if len(d) > 40 else (d[1:] if len(d) > 1 else d)
        avg = sum(dd) / len(dd)
        out[k] = {"calls": len(d), "avg_ns": avg}
        lines.append(f"## {k}: {len(d)} launches, avg of the last {len(dd)} {avg/1e3:.1f} us (min {min(dd)/1e3:.1f}, max {max(dd)/1e3:.1f})")
        for c in sorted(pmc.get(k, {})):
            v = pmc[k][c]
            a = sum(v) / len(v)
            out[k][c] = a
            lines.append(f"- {c}: {a:,.0f}")
        if "FETCH_SIZE" in out[k] or "WRITE_SIZE" in out[k]:
            fetch = out[k].get("FETCH_SIZE", 0.0) * 1024 * 2   # gfx950: FETCH_SIZE reads 1/2
            write = out[k].get("WRITE_SIZE", 0.0) * 1024
            out[k]["hbm_bytes"] = fetch + write
            lines.append(f"- **HBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE)**: {fetch+write:,.0f} B "
                         f"(read {fetch:,.0f}, write {write:,.0f})")
        lines.append("")
    if stats:
        lines += ["## kernel_stats.csv (rocprofv3 --kernel-trace --stats)", "", "```"]
        for r in csv.DictReader(open(stats[0])):
            lines.append(f"{r['Name'][:90]:90s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):>12.0f} pct={r['Percentage']}")
        lines.append("```")
    # What BINDS the image kernels: not HBM but the CU's vector-memory pipeline
    # (TA address -> TCP tags/cache -> TD data return).  From the counters of the same passes, per launch:
    #   cycles               = GRBM_GUI_ACTIVE / 8           (the counter is summed over the 8 XCDs)
    #   unit busy fraction   = {TA_TA_BUSY, TCP_GATE_EN1, TD_TD_BUSY}_sum / (cycles * 256 CUs)
    #   valu_busy_frac       = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)   (rocprof's VALUBusy)
    #   forward only (every vector read is one of the march's two 16-byte gathers per step; TCP_TOTAL_READ counts
    #   64 lane slots per wave instruction, active or not: TCP_TOTAL_READ == 64 * SQ_INSTS_VMEM_RD):
    #   bytes through the return path = 16 B * TCP_TOTAL_READ;  achieved = that / (cycles * 256)  B/clk/CU
    #   against the 64 B/clk/CU the path is wide (a 64-lane dwordx4 instruction = 1024 B = 16 clocks);
    #   active-lane bytes = the same * SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)  (the VALU's lane
    #   utilisation as the estimate of the gathers': no counter gives the active lanes of a VMEM instruction)
    N_XCD, N_CU, N_SIMD, PATH_B_PER_CLK = 8, 256, 1024, 64.0
    for k, v in out.items():
        if "GRBM_GUI_ACTIVE" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / N_XCD
        b = {"cycles_per_launch": round(cyc), "clock_mhz_from_avg_duration": round(cyc / v["avg_ns"] * 1e3, 1)}
        for unit, c in (("TA", "TA_TA_BUSY_sum"), ("TCP", "TCP_GATE_EN1_sum"), ("TD", "TD_TD_BUSY_sum")):
            if c in v:
                b.setdefault("unit_busy_frac", {})[unit] = round(v[c] / (cyc * N_CU), 4)
        if "SQ_ACTIVE_INST_VALU" in v:
            b["valu_busy_frac"] = round(4.0 * v["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc), 4)
        if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
            b["valu_lane_utilisation"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4)
        if k.startswith("render_forward_kernel") and "TCP_TOTAL_READ_sum" in v:
            by = 16.0 * v["TCP_TOTAL_READ_sum"]
            ach = by / (cyc * N_CU)
            b.update({"resource": "CU vector-memory return path (TA -> TCP -> TD), 64 B/clk/CU",
                      "bytes_per_launch_lane_slots": int(by), "achieved_B_per_clk_per_CU": round(ach, 2),
                      "peak": PATH_B_PER_CLK, "frac": round(ach / PATH_B_PER_CLK, 4)})
            if "valu_lane_utilisation" in b:
                b["achieved_active_lanes_B_per_clk_per_CU"] = round(ach * b["valu_lane_utilisation"], 2)
            if v.get("SQ_INSTS_VMEM_RD"):
                b["tcp_busy_cycles_per_read_instruction"] = round(v.get("TCP_GATE_EN1_sum", 0.0) / v["SQ_INSTS_VMEM_RD"], 1)
                b["cache_line_accesses_per_read_instruction"] = round(v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)
                                                                      / v["SQ_INSTS_VMEM_RD"], 1)
        v["binding"] = b
        lines.insert(lines.index(next(l for l in lines if l.startswith(f"## {k}:"))) + 1,
                     "- **binding** " + json.dumps(b))
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    open(os.path.join(root, "profiles", name + ".md"), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(root, "profiles", name + ".json"), "w"), indent=1)
    # what bench.py reports as roofline.traffic: HBM bytes per launch of the two image kernels
    traffic = {"source": f"profiles/{name}.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; "
                         "2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)"}
    sys.path.insert(0, root)
    from tools.bench_extra import kernel_sources_sha
    traffic["kernel_sources_sha16"] = kernel_sources_sha()   # bench.py reports the traffic only for this build
    for k, v in out.items():
        if "hbm_bytes" in v:
            traffic[k.split("<")[0]] = int(v["hbm_bytes"])
        if k.startswith(("render_forward_kernel", "render_backward_kernel", "forward_prologue_kernel", "pose_reduce_kernel")):
            traffic.setdefault("kernel_avg_us_from_trace", {})[k.split("<")[0]] = round(v["avg_ns"] / 1e3, 2)
        if "binding" in v and k.startswith(("render_forward_kernel", "render_backward_kernel")):
            traffic.setdefault("binding", {})[k.split("<")[0]] = v["binding"]
    if len(traffic) > 2:
        json.dump(traffic, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
