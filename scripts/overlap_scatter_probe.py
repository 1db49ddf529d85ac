#!/usr/bin/env python3
"""Does reshape_and_cache hide beside paged_attention_v1 when it runs on a second stream?  (round 6)

cfg3 call pair, four schedules of the SAME two launches (timing only — with the plain attention kernel the overlapped
forms race on the newest token; the append-read kernel removes that dependence):
    serial        reshape_and_cache ; paged_attention_v1 on one stream          (the reference's call order, the headline)
    attend_only   paged_attention_v1 alone
    two_streams   per step: fork event, reshape on the side stream, attention on the main stream, join event
    graph_serial / graph_forked   12 pairs in one hipGraph, the scatter as a chain member / as a parallel branch
Writes gpurun_out/overlap_scatter_probe.json.
"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402


def timed(fn, steps, warm, dev):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    dev = torch.device("cuda:0")
    cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
    steps = 200
    wl = make_workload(cfg, dev, seed=1234, table_sets=2)
    out = torch.empty((cfg.batch, cfg.num_heads, cfg.head_size), dtype=torch.float16, device=dev)
    main_s = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    res = {"config": cfg.name, "steps": steps}

    res["serial_us"] = [timed(lambda i: bench.one_step(wl, out, i, 0), steps, 20, dev) for _ in range(3)]
    res["attend_only_us"] = [timed(lambda i: bench.attend(wl, out, i % 2, 0), steps, 20, dev) for _ in range(3)]

    def forked(i):
        t = i % 2
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            bench.scatter(wl, t)
        bench.attend(wl, out, t, 0)
        main_s.wait_stream(side)

    res["two_streams_us"] = [timed(forked, steps, 20, dev) for _ in range(3)]

    def capture(fn, n):
        cap = torch.cuda.Stream(dev)
        cap.wait_stream(main_s)
        with torch.cuda.stream(cap):
            for t in range(2):
                bench.one_step(wl, out, t, 0)
        main_s.wait_stream(cap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            for t in range(n):
                fn(t, cap)
        return g

    def pair_serial(t, cap):
        bench.one_step(wl, out, t, 0)

    side2 = torch.cuda.Stream(dev)

    def pair_forked(t, cap):
        side2.wait_stream(cap)
        with torch.cuda.stream(side2):
            bench.scatter(wl, t % 2)
        bench.attend(wl, out, t % 2, 0)
        cap.wait_stream(side2)

    for name, fn in (("graph_serial_us", pair_serial), ("graph_forked_us", pair_forked)):
        g = capture(fn, 12)
        res[name] = [timed(lambda i: g.replay(), 20, 3, dev) / 12 for _ in range(3)]
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "overlap_scatter_probe.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
