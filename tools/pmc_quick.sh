#!/bin/bash
# One PMC pass (run on the GPU box):  tools/pmc_quick.sh <tag> "<COUNTERS>" [bench args]
TAG=$1; SET=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_1 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-configs --prewarm-ms 0 "$@" > $OUT/pmc_1.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="fwd" if "render_forward" in n else "bwd" if "render_backward" in n else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k, {c: round(sum(v)/len(v)) for c,v in acc[k].items()})
PY
