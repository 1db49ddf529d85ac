"""What could splitting the LONG items of a ragged batch gain at most?  (round-4 verdict, item 2)

The balanced kernel (pa_queue.hpp) serves BASELINE configs[2] with lengths U{1..1024} — 3072 (sequence, head) items, one per wave
of its 768 workgroups — in a launch that lasts about as long as its longest items.  Before building partitions of long items that
exchange (max, exp-sum) and merge partial rows through the workspace, this measures the BOUND of that idea with no kernel change:
the same batch with every sequence longer than `thr` tokens replaced by P sequences over 1/P of its blocks each (their own
block-table rows and query rows) — the work of an ideally split batch, with NO exchange, NO merge and NO partner waiting.
If the unchanged kernel is not faster on that batch than the verdict's targets, no implementation of the split can be.

  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python scripts/ragged_split_bound.py > DIR/order.json
  python scripts/stage_timeline_probe.py --summarize DIR out.json
(results are not checked here: the rows of a split sequence are partial attention by construction)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import ops  # noqa: E402
from vllmini_amd.workload import CONFIGS, make_workload  # noqa: E402

WARM, TIMED = 10, 50          # (stage_timeline_probe.summarize drops 1 probing + WARM launches of every case)
dev = torch.device("cuda:0")
order = []
for kv in ("auto", "fp8"):
    wl = make_workload(CONFIGS["cfg3"], dev, seed=0, table_sets=2, ragged=True)
    cfg = wl.cfg
    H, D = cfg.num_heads, cfg.head_size
    if kv == "fp8":
        g8 = torch.Generator(device=dev).manual_seed(9)
        wl.key_cache = torch.randint(0, 64, (cfg.num_blocks, H, D // 16, 16, 16), dtype=torch.uint8, device=dev, generator=g8)
        wl.value_cache = torch.randint(0, 64, (cfg.num_blocks, H, D, 16), dtype=torch.uint8, device=dev, generator=g8)
    lens = wl.seq_lens.cpu()
    for label, thr, parts in (("as is", 1 << 30, 1), ("> 896 in 2", 896, 2), ("> 768 in 2", 768, 2), ("> 640 in 2", 640, 2), ("> 512 in 2", 512, 2),
                              ("> 512 in 2, > 768 in 4", 512, 0), ("> 256 in 4", 256, 4)):
        rows, new_lens, src = [[] for _ in wl.tables], [], []
        for s in range(cfg.batch):
            L = int(lens[s])
            P = 1 if L <= thr else (parts or (4 if L > 768 else 2))
            nb = -(-L // 16)
            cuts = [nb * j // P for j in range(P + 1)]
            for j in range(P):
                toks = min(L, cuts[j + 1] * 16) - cuts[j] * 16
                if toks <= 0:
                    continue
                new_lens.append(toks)
                src.append(s)
                for t, tab in enumerate(wl.tables):
                    row = torch.full((tab.shape[1],), -1, dtype=torch.int32)
                    row[: cuts[j + 1] - cuts[j]] = tab[s, cuts[j]: cuts[j + 1]].cpu()
                    rows[t].append(row)
        tabs = [torch.stack(r).to(dev) for r in rows]
        sl = torch.tensor(new_lens, dtype=torch.int32, device=dev)
        q = wl.query[torch.tensor(src, device=dev)].contiguous()
        out = torch.empty((len(src), H, D), dtype=torch.float16, device=dev)
        names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
        # the balanced kernel the unsplit batch gets, and whatever the default entry picks for the longer batch
        for vid in (names["fp8_q_d64_s2q4m" if kv == "fp8" else "q_d64_s1q2"], 0):
            if vid == 0 and parts == 1:
                continue
            for i in range(1 + WARM + TIMED):
                ops.paged_attention_v1(out, q, wl.key_cache, wl.value_cache, cfg.kv_heads, wl.scale, tabs[i % len(tabs)], sl, 16,
                                       cfg.seq_len, None, kv, 1.0, 0, 0, 1, 1, 0, _variant=vid)
            torch.cuda.synchronize()
            order.append({"case": f"cfg3 ragged {'fp8' if kv == 'fp8' else 'fp16'} pages, {label}" + (" (default pick)" if vid == 0 else ""),
                          "kernel": ops.last_launch_label(), "launches": 1 + WARM + TIMED, "sequences": len(src),
                          "longest": int(max(new_lens)), "tokens": int(sum(new_lens))})
print(json.dumps(order))
