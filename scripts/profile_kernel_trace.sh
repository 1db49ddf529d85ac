#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration (run on the GPU box via gpurun).
# Usage: scripts/profile_kernel_trace.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/{summary.json, trace/...}
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
  python bench.py --steps 60 --warmup 10 --headline-only "$@" > "$OUT/bench_under_trace.json" 2> "$OUT/trace.stderr"
python scripts/summarize_prof.py "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.stderr"
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/summary.json"))
print(sys.argv[1], json.dumps(d.get("pa_v1_dispatches")))
for k in d.get("kernel_stats", [])[:4]:
    print("   ", k["Name"][:70], k["Calls"], k["AverageNs"])
PY
