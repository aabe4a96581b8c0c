#!/bin/bash
# Run ON the GPU box (through gpurun): kernel trace + separate PMC passes of one bench command.
#   tools/profile_gpu.sh <tag> [bench args...]
# (the TA block takes two counters per pass: "TA_TA_BUSY_sum" + one more; a set the hardware cannot collect makes
# rocprofv3 abort and then sit until the timeout -- r04: two such passes cost 300 s of box time)
# Writes gpurun_out/<tag>/{trace,pmc_*}/... ; tools/summarize_pmc.py turns them into profiles/.
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-configs --prewarm-ms 0 $*"
# the kernel trace after a pre-warm (durations at the clocks of a running job; the last 20 steps are the timed ones),
# the counter passes without (counts do not depend on the clock)
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-configs --prewarm-ms 150 $* > "$OUT/trace.log" 2>&1
# the counter passes: the half-grid choice of the steady state (C3: 254 of 256 views close -> hint on), forced, because
# 5 cold steps under a serialising profiler end before the device-side count has reached the host
export SDFR_BENCH_CLOSE_VIEWS=${SDFR_BENCH_CLOSE_VIEWS:-1}
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum" \
  "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA" \
  "TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$i.log" 2>&1 || echo "pass $i failed"
done
ls "$OUT"
