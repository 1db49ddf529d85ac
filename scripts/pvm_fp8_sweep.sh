#!/bin/bash
# rocprofv3 kernel time of the cfg4 shape with 8 KV heads over fp8 pages: default pick, opt-in pick, and each _pvm row
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/pvm_fp8; mkdir -p $OUT
run() {  # label, bench flags
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o t -- python bench.py --config cfg4 --kv-heads ${KVH:-8} --kv fp8 $2 --steps 50 --warmup 5 --headline-only > /dev/null 2>&1
  echo "$1: $(grep pa_v1 $OUT/$1/t_kernel_stats.csv | awk -F'",' '{print $2}' | awk -F, '{printf "%.1f us avg (min %.1f)", $3/1000, $5/1000}')  $(grep -o 'pa_v1_kernel<[^>]*>' $OUT/$1/t_kernel_stats.csv | head -1)"
}
run default ""
run optin "--pv-mfma"
for NAME in $(python -c "
from vllmini_amd import ops
print(' '.join(n for n in ops.variant_names() if n.startswith('fp8_d128') and n.endswith('_pvm')))"); do
  V=$(python -c "
from vllmini_amd import ops; print(ops.variant_names().index('$NAME')+1)")
  run $NAME "--variant $V"
done
