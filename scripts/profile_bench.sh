#!/bin/bash
# rocprofv3 evidence for bench.py's roofline numbers (run on the GPU box via gpurun).
#   pass 1: --kernel-trace --stats  -> per-kernel durations (must agree with bench.py's HIP-event mean)
#   pass 2: --pmc FETCH_SIZE        -> HBM read traffic   (own pass; never combined with trace domains other than kernel-trace)
#   pass 3: --pmc WRITE_SIZE
#   pass 4: SQ_* wave/issue/LDS counters     pass 5: MFMA busy/insts, VMEM/LDS instruction counts, GRBM_GUI_ACTIVE
# Usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 10 --headline-only $*"   # one attention kernel per run: plain, or --op fused
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python bench.py $ARGS > "$OUT/bench_under_trace.json" 2> "$OUT/trace.stderr"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python bench.py $ARGS > "$OUT/bench_under_pmc_fetch.json" 2> "$OUT/pmc_fetch.stderr"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python bench.py $ARGS > "$OUT/bench_under_pmc_write.json" 2> "$OUT/pmc_write.stderr"
# pass 4 / 5: SQ counters (8 slots per pass) and the matrix-pipe counters (expected 0: MFMA is deliberately unused)
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -o sq -- python bench.py $ARGS > /dev/null 2> "$OUT/pmc_sq.stderr"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -o mfma -- python bench.py $ARGS > /dev/null 2> "$OUT/pmc_mfma.stderr"
python scripts/summarize_prof.py "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.stderr"
cat "$OUT/summary.json"
