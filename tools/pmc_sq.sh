#!/bin/bash
# SQ counters of the fused path, two separate PMC passes (kernel-trace only, pool rule).
# usage: tools/pmc_sq.sh [bench args...]     outputs gpurun_out/pmc_sqA, pmc_sqB
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
rm -rf "$OUT/pmc_sqA" "$OUT/pmc_sqB"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/pmc_sqA" -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --extra-steps 0 --skip-other-path "$@" > "$OUT/pmc_sqA.json" 2> "$OUT/pmc_sqA.err" || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/pmc_sqB" -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --extra-steps 0 --skip-other-path "$@" > "$OUT/pmc_sqB.json" 2> "$OUT/pmc_sqB.err" || exit 2
echo done
