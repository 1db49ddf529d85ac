#!/bin/bash
# Same-box A/B of two library builds in bench.py's own context (the reference's call pair): expects the other build at
# vllmini_amd/_C/libvmi_old.so; swaps the files on the (scratch) GPU box copy, three alternations, restores the new one.
cp vllmini_amd/_C/libvmi_paged_attention.so /tmp/new.so
cp vllmini_amd/_C/libvmi_old.so /tmp/old.so
for i in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so vllmini_amd/_C/libvmi_paged_attention.so
    python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-fused --no-fp8 --no-graph --no-cfg4 --no-e2e --no-cfg2 --no-strong --no-long 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'step_us %.1f' % (d['ms_per_step']*1e3), 'kernel_us %.1f' % d['paged_attention_v1_us_per_step'], 'ragged step_us %.1f kernel_us %.1f' % (d['ragged_step']['ms_per_step']*1e3, d['ragged_step']['kernel_us_median']))"
  done
done
cp /tmp/new.so vllmini_amd/_C/libvmi_paged_attention.so
