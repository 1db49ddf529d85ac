#!/bin/bash
# bench.py over fp8 E4M3 and E5M2 pages: step time, kernel variant and roofline fraction on cfg3 / cfg4, plus the
# grouped-query shape and the end-to-end harness with E5M2 pages
export PYTHONPATH=.
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['config'].get('kernel_variant'), round(d['ms_per_step']*1e3,1), 'us/step', {k: d['roofline'][k] for k in ('achieved','frac')})"; }
for c in cfg3 cfg4; do for k in fp8 fp8_e5m2; do
  python bench.py --config $c --kv $k --steps 100 --warmup 10 2>/dev/null | tail -1 | show "$c $k"
done; done
python bench.py --config cfg4 --kv-heads 8 --kv fp8 --steps 50 --warmup 5 2>/dev/null | tail -1 | show "cfg4/kv8 fp8"
python bench.py --config cfg4 --kv-heads 8 --kv fp8_e5m2 --steps 50 --warmup 5 2>/dev/null | tail -1 | show "cfg4/kv8 fp8_e5m2"
python bench.py --e2e --kv fp8_e5m2 --steps 20 --warmup 5 2>&1 | grep gpt2_small | cut -c1-260
