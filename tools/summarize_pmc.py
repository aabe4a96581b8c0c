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
    if len(traffic) > 2:
        json.dump(traffic, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
