#!/bin/bash
# Run on the GPU box: kernel trace of tools/profile_loop.py, then the launches of the LAST captured iteration in
# order, with durations and the idle gap before each one.  tools/loop_sequence.sh <tag>   (PROFILE_SCRIPT=tools/profile_multi.py K=8: the K-object loop)
TAG=${1:-loopseq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 "$ROOT/${PROFILE_SCRIPT:-tools/profile_loop.py}" > $OUT/cmd.log 2>&1 || echo "trace failed"
python3 - <<PY > $OUT/sequence.md
import csv, glob, re
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam_step_kernel" in r[2] or "loop_tail_kernel" in r[2]]
a, b = ends[-2] + 1, ends[-1] + 1
print("| # | kernel | threads | duration us | gap before us |\n|---|---|---|---|---|")
tot = gap = 0.0
for i in range(a, b):
    s, e, n, g = rows[i]
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    k = m.group(1) if m else n[:70]
    gp = (s - rows[i - 1][1]) / 1e3
    tot += (e - s) / 1e3; gap += gp
    print(f"| {i - a + 1} | \`{k}\` | {g} | {(e - s) / 1e3:.2f} | {gp:.2f} |")
print(f"\n{b - a} launches, kernels {tot:.1f} us + gaps {gap:.1f} us = {(rows[b - 1][1] - rows[a - 1][1]) / 1e3:.1f} us adam-to-adam (under the profiler)")
PY
cat $OUT/sequence.md
