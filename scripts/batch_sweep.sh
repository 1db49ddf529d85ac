#!/bin/bash
# rocprofv3 kernel times of paged_attention_v1 over batch sizes: heuristic (variant 0) vs named alternatives
export TMPDIR=/tmp
OUT=gpurun_out/batch_sweep; mkdir -p $OUT
NAMES="${NAMES:-d64_h1_w2_u1_nt1 d64_h1_w2_u2_nt1 d64_h1_w2_u4_nt1 d64_h1_w4_u1_nt1 d64_h1_w4_u2_nt1 d64_h1_w4_u4_nt1 d64_h1_w8_u1_nt1 d64_h1_w8_u2_nt1 d64_h4_w1_u1_nt1}"
echo "batch,variant,kernel,avg_us,min_us" > $OUT/summary.csv
for B in ${BATCHES:-32 64 128 192}; do
  for NAME in auto $NAMES; do
    if [ "$NAME" = auto ]; then V=0; else V=$(python -c "
from vllmini_amd import ops; print(ops.variant_names().index('$NAME')+1)"); fi
    D=$OUT/b${B}_${NAME}
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python bench.py --batch $B ${SEQLEN:+--seq-len $SEQLEN} --variant $V --steps 40 --warmup 5 --headline-only > /dev/null 2>&1
    python - "$D" "$B" "$NAME" >> $OUT/summary.csv <<'PY'
import csv,sys
d,b,v=sys.argv[1:4]
try:
    for r in csv.DictReader(open(f"{d}/t_kernel_stats.csv")):
        if "pa_v1" in r["Name"]:
            print(f'{b},{v},"{r["Name"][10:60]}",{float(r["AverageNs"])/1e3:.2f},{float(r["MinNs"])/1e3:.2f}')
except Exception as e:
    print(f"{b},{v},error {e},,")
PY
  done
done
cat $OUT/summary.csv
