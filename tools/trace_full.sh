#!/bin/bash
# Kernel trace of a python command with the kernels' FULL names (template arguments kept):
#   tools/trace_full.sh <tag> <script.py> [args...]   (run on the GPU box)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 "$ROOT/$1" "${@:2}" > $OUT/cmd.log 2>&1 || echo "trace failed"
cat $OUT/cmd.log | tail -5
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "sdfr" not in n: continue
        k=n.replace("sdfr::(anonymous namespace)::","").replace("void ","").split("(")[0]
        acc[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2=sorted(v); tail=v[-40:]
    print(f"$TAG {k[:110]:110s} calls {len(v):5d} avg {sum(v)/len(v)/1e3:8.1f} us  last40 {sum(tail)/len(tail)/1e3:8.1f}  median {v2[len(v2)//2]/1e3:8.1f}  min {v2[0]/1e3:8.1f}")
PY
