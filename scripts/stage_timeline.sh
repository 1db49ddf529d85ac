#!/bin/bash
# The under-filled chip, term by term (run on the GPU box via gpurun): stage stamps from the diagnostic library, then the
# same cases on the product library under rocprofv3 --kernel-trace.  Usage: scripts/stage_timeline.sh <tag> [cases]
set -u
TAG=${1:-r04}; CASES=${2:-}
OUT=gpurun_out/stage_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python scripts/stage_timeline_probe.py --stamps "$OUT/stamps.json" ${CASES:+--cases $CASES} > "$OUT/stamps.log" 2>&1
tail -3 "$OUT/stamps.log"
mkdir -p "$OUT/trace"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python scripts/stage_timeline_probe.py --plain ${CASES:+--cases $CASES} > "$OUT/trace/order.json.tmp" 2> "$OUT/trace.stderr"
tail -1 "$OUT/trace/order.json.tmp" > "$OUT/trace/order.json"
python scripts/stage_timeline_probe.py --summarize "$OUT/trace" "$OUT/rocprof.json" 2>&1 | tee "$OUT/rocprof.log"
find "$OUT/trace" -name "*kernel_trace.csv" -size +8M -delete   # (scratch stays small)
if [ -n "${IC_PROBE:-}" ]; then timeout 300 python scripts/ic_bandwidth_probe.py "$OUT/ic_bandwidth.json" > "$OUT/ic_bandwidth.log" 2>&1; tail -50 "$OUT/ic_bandwidth.log"; fi
