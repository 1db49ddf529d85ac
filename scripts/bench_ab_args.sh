#!/bin/bash
# Same-box A/B of two PRODUCT library builds on any bench.py configuration: expects the other build at
# vllmini_amd/_C/libvmi_old.so; alternates the two on the (scratch) GPU box copy, restores the new one.
# Usage: scripts/bench_ab_args.sh "<bench args>" ["<bench args 2>" ...]   -> kernel us (event median) per build and round
cp vllmini_amd/_C/libvmi_paged_attention.so /tmp/new.so
cp vllmini_amd/_C/libvmi_old.so /tmp/old.so
for ARGS in "$@"; do
  for i in 1 2 3; do
    for v in old new; do
      cp /tmp/$v.so vllmini_amd/_C/libvmi_paged_attention.so
      python bench.py --steps 200 --warmup 30 --headline-only --kernel-samples 120 $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '[$ARGS]', d['config']['kernel_variant'], 'step_us %.1f' % (d['ms_per_step']*1e3), 'kernel_us median %.1f min %.1f' % (d['paged_attention_v1_us_median'], d['paged_attention_v1_us_min']))"
    done
  done
done
cp /tmp/new.so vllmini_amd/_C/libvmi_paged_attention.so
