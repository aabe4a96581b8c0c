#!/bin/bash
# Run on the GPU box: kernel trace of tools/profile_front_door.py; prints the launches of the LAST call that are not the
# loop's iterations (preprocess, point clouds, initialisation network, pose set-up), with durations and gaps.
TAG=${1:-frontdoor}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 "$ROOT/tools/profile_front_door.py" > $OUT/cmd.log 2>&1 || echo "trace failed"
python3 - <<PY > $OUT/sequence.md
import csv, glob, re
rows = []
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tails = [i for i, r in enumerate(rows) if "loop_tail_kernel" in r[2]]
# the last call: its 50 tails are the last 50; what precedes the first of its iterations, back to the previous call's last tail
first_tail = tails[-50]
prev_tail = tails[-51]
# the first iteration's launches begin after the call's set-up: walk back from first_tail to the decoder's first conv of that iteration
seq = rows[prev_tail + 1:first_tail + 1]
print("| # | kernel | duration us | gap before us |\n|---|---|---|---|")
tot = 0.0
for i, (s, e, n) in enumerate(seq):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    k = m.group(1) if m else n[:70]
    gp = (s - rows[prev_tail + i][1]) / 1e3
    tot += (e - s) / 1e3
    print(f"| {i + 1} | \`{k}\` | {(e - s) / 1e3:.2f} | {gp:.2f} |")
print(f"\n{len(seq)} launches from the previous call's last tail to this call's first tail: kernels {tot:.1f} us, wall {(seq[-1][1] - rows[prev_tail][1]) / 1e3:.1f} us (under the profiler)")
PY
cat $OUT/sequence.md
