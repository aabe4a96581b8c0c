#!/bin/bash
# One PMC pass over an arbitrary python command (run on the GPU box):
#   tools/pmc_cmd.sh <tag> "<COUNTERS>" <script.py> [args...]      (environment variables pass through)
# Prints per-kernel averages of every counter (kernels of libsdfr only).
TAG=$1; SET=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc -- python3 "$ROOT/$1" "${@:2}" > $OUT/cmd.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "sdfr" not in n: continue
        import re
        k=n.replace("sdfr::(anonymous namespace)::","").replace("void ","").split("(")[0]  # (template arguments kept)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print("$TAG", k, {c: round(sum(v)/len(v)) for c,v in acc[k].items()}, "launches", max(len(v) for v in acc[k].values()))
PY
