#!/bin/bash
# rocprofv3 kernel time of paged_attention_v1 (heuristic pick) over batch sizes at the cfg3 head shape: the latency end
# (B = 1, the reference scheduler's own case) to the bandwidth end (B = 256).  SEQLEN env overrides 1024.
export TMPDIR=/tmp
OUT=gpurun_out/batch_latency; mkdir -p $OUT
echo "batch,seq_len,kernel,avg_us,min_us,GB_per_s" > $OUT/summary.csv
for B in ${BATCHES:-1 2 4 8 16 32 64 128 256}; do
  D=$OUT/b$B
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python bench.py --batch $B ${SEQLEN:+--seq-len $SEQLEN} --steps 40 --warmup 5 --headline-only > $D.json 2>/dev/null
  python - "$D" "$B" "${SEQLEN:-1024}" >> $OUT/summary.csv <<'PY'
import csv, json, sys
d, b, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
name = json.loads(open(d + ".json").read().strip().splitlines()[-1])["config"]["kernel_variant"]
for r in csv.DictReader(open(f"{d}/t_kernel_stats.csv")):
    if "pa_v1" in r["Name"] or "pa_q_" in r["Name"]:
        us = float(r["AverageNs"]) / 1e3
        print(f'{b},{L},{name},{us:.2f},{float(r["MinNs"]) / 1e3:.2f},{4 * b * 12 * L * 64 / us / 1e3:.0f}')
PY
done
cat $OUT/summary.csv
