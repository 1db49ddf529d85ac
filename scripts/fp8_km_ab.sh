#!/bin/bash
# Same-box A/B of the fp8 balanced kernels with the K pass on the VALU and on the matrix cores ("m" names): bench.py's own
# call pair (reshape_and_cache + paged_attention_v1 over fp8 E4M3 pages), kernel time by HIP events (median of 100) and step
# time, three alternations; equal lengths and U{1..L}.  Usage: scripts/fp8_km_ab.sh [cfg3|cfg4] -> stdout
CFG=${1:-cfg3}
KV=${2:-fp8}
if [ "$KV" = auto ]; then if [ "$CFG" = cfg4 ]; then VARS="q_d128_s1q1 q_d128_s1q1m"; else VARS="q_d64_s1q2 q_d64_s1q2m"; fi
elif [ "$CFG" = cfg4 ]; then VARS="fp8_q_d128_s1q2 fp8_q_d128_s1q2m"; else VARS="fp8_q_d64_s2q4 fp8_q_d64_s2q4m fp8_q_d64_s1q2 fp8_q_d64_s1q2m"; fi
for r in 1 2 3; do
  for v in $VARS; do
    for rag in "" "--ragged"; do
      python bench.py --config $CFG --kv $KV --variant-name $v --headline-only --steps 200 --warmup 20 --kernel-samples 100 $rag 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s %-9s step_us %7.1f  kernel_us median %7.1f mean %7.1f min %7.1f' % ('$v', '$rag' or 'equal', d['ms_per_step']*1e3, d['paged_attention_v1_us_median'], d['paged_attention_v1_us_mean'], d['paged_attention_v1_us_min']))"
    done
  done
done
