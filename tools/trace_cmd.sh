#!/bin/bash
# Kernel trace of an arbitrary python command (run on the GPU box): tools/trace_cmd.sh <tag> <script.py> [args...]
# Prints per-kernel call counts and average / min durations (libsdfr kernels only).
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 "$ROOT/$1" "${@:2}" > $OUT/cmd.log 2>&1 || echo "trace failed"
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "sdfr" not in n: continue
        m=re.search(r"(\w+_kernel)", n); k=m.group(1) if m else n[:60]
        acc[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2=sorted(v); print(f"$TAG {k:40s} calls {len(v):4d} avg {sum(v)/len(v)/1e3:8.1f} us  median {v2[len(v2)//2]/1e3:8.1f}  min {v2[0]/1e3:8.1f}")
PY
