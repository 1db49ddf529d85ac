#!/bin/bash
# End-of-round bench evidence (GPU box, via gpurun): the default line, the RCCL path forced on one GPU for both scaling modes,
# the serving record at full trace length.  Usage: scripts/evidence_round.sh <tag>
T=${1:-r06z}
O=gpurun_out
python bench.py > $O/${T}_bench_cfg3.json 2> $O/${T}_bench_cfg3.err
LIGHT="--no-e2e --no-serve --no-deferred --no-long --no-strong --no-cfg4 --no-cfg2 --no-cpu-baseline"
VMI_FORCE_DIST=1 python bench.py --gpus 1 $LIGHT > $O/${T}_bench_forced_dist_n1.json 2> $O/${T}_bench_forced_dist_n1.err
VMI_FORCE_DIST=1 python bench.py --gpus 1 --scaling strong $LIGHT > $O/${T}_bench_forced_dist_n1_strong.json 2> $O/${T}_bench_forced_dist_n1_strong.err
python bench.py --headline-only > $O/${T}_bench_headline_only.json 2> /dev/null
python bench.py --scaling strong --config cfg5_strong --headline-only > $O/${T}_bench_strong_plain_n1.json 2> /dev/null
python bench.py --serve 2> $O/${T}_serve.err; cp $O/serve.json $O/${T}_serve.json
python bench.py --serve --serve-preempt drop 2> /dev/null; cp $O/serve.json $O/${T}_serve_drop.json
python bench.py --serve --serve-no-deferred-scatter 2> /dev/null; cp $O/serve.json $O/${T}_serve_call_pair.json
python bench.py --serve --serve-kv fp8 2> /dev/null; cp $O/serve.json $O/${T}_serve_fp8.json
python bench.py --serve --serve-pool-frac 1.5 2> /dev/null; cp $O/serve.json $O/${T}_serve_roomy_pool.json
for r in 200 400 600 800; do python bench.py --serve --serve-rate $r --serve-admit-every 4 2> /dev/null; cp $O/serve.json $O/${T}_serve_rate_$r.json; done
python - "$T" <<'PY'
import json, sys
t = sys.argv[1]
for n in ("bench_cfg3", "bench_forced_dist_n1", "bench_forced_dist_n1_strong", "bench_headline_only", "bench_strong_plain_n1"):
    l = json.load(open(f"gpurun_out/{t}_{n}.json"))
    print(n, l["config"]["workload"][:40], "ms/step %.4f" % l["ms_per_step"], "regions", l.get("timed_regions"), "%.1f ms" % l.get("timed_region_ms", 0),
          "kernel %.2f" % l["paged_attention_v1_us_median"], "frac %.3f" % l["roofline"]["frac"], "traffic", l["roofline"]["traffic"],
          "exch", l.get("token_exchange_us"), "ranks", l.get("rccl_ranks"))
for n in ("serve", "serve_drop", "serve_call_pair", "serve_fp8", "serve_roomy_pool"):
    s = json.load(open(f"gpurun_out/{t}_{n}.json"))
    print(n, {k: (round(s[k], 1) if isinstance(s[k], float) else s[k]) for k in ("value", "wall_s", "decode_steps", "batch_occupancy", "host_us_per_step", "gpu_wait_us_per_step", "admit_s", "preemptions", "dropped", "swap_out_MB")}, s["token_latency_ms"])
PY
