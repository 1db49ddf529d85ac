#!/bin/bash
# rocprofv3 kernel time of paged_attention_v1 over the under-filled chip: batches 1 .. 64 x 12 heads x 64 at 256 .. 2048 tokens, the
# default pick against the several-waves-per-head kernels of the menu (run on the GPU box via gpurun; the case list comes from
# scripts/underfilled_sweep_cases.txt).  -> gpurun_out/underfilled_<tag>/rocprof.json
set -u
TAG=${1:-r04}; OUT=gpurun_out/underfilled_$TAG; mkdir -p "$OUT/trace"
export TMPDIR=/tmp
CASES=$(cat ${CASES_FILE:-scripts/underfilled_sweep_cases.txt})
timeout 1500 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python scripts/stage_timeline_probe.py --plain --cases "$CASES" > "$OUT/trace/order.json.tmp" 2> "$OUT/trace.stderr"
tail -1 "$OUT/trace/order.json.tmp" > "$OUT/trace/order.json"
python scripts/stage_timeline_probe.py --summarize "$OUT/trace" "$OUT/rocprof.json" > "$OUT/rocprof.log" 2>&1
tail -3 "$OUT/rocprof.log"
find "$OUT/trace" -name "*kernel_trace.csv" -size +8M -delete
