#!/bin/bash
# Same-box A/B of two library builds over fp8 pages (equal and U{1..L} lengths): expects the other build at
# vllmini_amd/_C/libvmi_old.so; three alternations; kernel time = median of the event-pair pass.
cp vllmini_amd/_C/libvmi_paged_attention.so /tmp/new.so; cp vllmini_amd/_C/libvmi_old.so /tmp/old.so
for i in 1 2 3; do for v in old new; do cp /tmp/$v.so vllmini_amd/_C/libvmi_paged_attention.so
for r in "" "--ragged"; do
python bench.py --steps 300 --warmup 30 --kv fp8 $r --headline-only --kernel-samples 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fp8 $v ${r:-equal}', 'kernel_us median %.1f min %.1f' % (d['paged_attention_v1_us_median'], d['paged_attention_v1_us_min']), 'step_us %.1f' % (d['ms_per_step']*1e3))"
done; done; done; cp /tmp/new.so vllmini_amd/_C/libvmi_paged_attention.so
