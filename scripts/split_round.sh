#!/bin/bash
# Round 5: the split kernels against the plain picks, by rocprofv3 kernel time on the product library (run on the GPU box via
# gpurun).  Usage: scripts/split_round.sh <tag> [cases]   -> gpurun_out/split_<tag>/rocprof.json
set -u
TAG=${1:-r05}; CASES=${2:-$(cat scripts/split_cases.txt)}
OUT=gpurun_out/split_$TAG; mkdir -p "$OUT/trace"
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python scripts/stage_timeline_probe.py --plain --cases "$CASES" > "$OUT/trace/order.json.tmp" 2> "$OUT/trace.stderr"
tail -1 "$OUT/trace/order.json.tmp" > "$OUT/trace/order.json"
python scripts/stage_timeline_probe.py --summarize "$OUT/trace" "$OUT/rocprof.json" > "$OUT/rocprof.log" 2>&1
python - "$OUT/rocprof.json" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    print(f"{r['case']:44s} {r['kernel']:22s} med {r['rocprofv3_us_median']:7.2f} min {r['rocprofv3_us_min']:7.2f} grid {r['grid']}  scatter {r.get('scatter_us_median', 0):5.2f} {r.get('scatter_grid', '')}")
PY
find "$OUT/trace" -name "*kernel_trace.csv" -size +8M -delete
