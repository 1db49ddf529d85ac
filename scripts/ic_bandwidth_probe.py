"""What read bandwidth does the chip give a working set that is RESIDENT in the 256 MiB Infinity Cache — the regime of BASELINE
configs[1] (50 MB per launch) — against one that streams from HBM?  The attention kernel's gather pattern with the math removed
(vmi_diag_gather_read: pseudo-randomly ordered contiguous 2-KiB chunks, one chunk stream per wave) and the plain coalesced
stream, over buffers of 25 MB ... 1.6 GB, temporal and non-temporal loads, several depths; 30 launches back to back between one
HIP event pair.  Diagnostic library only.  `python scripts/ic_bandwidth_probe.py [out.json]`"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib  # noqa: E402

lib = _lib.load_diag()
dev = torch.device("cuda:0")
sink = torch.zeros(4, dtype=torch.int32, device=dev)
res = []
for mb in (25, 50, 100, 200, 400, 1600):
    n = mb * 1000 * 1000 // 65536 * 65536
    buf = torch.randint(0, 2 ** 31 - 1, (n // 4,), dtype=torch.int32, device=dev)
    for kind, chunk, inflight, blocks, nt in (("gather", 2, 4, 768, 0), ("gather", 2, 4, 768, 1), ("gather", 2, 8, 768, 0),
                                              ("gather", 2, 2, 1536, 0), ("gather", 4, 8, 768, 0), ("gather", 16, 16, 768, 0),
                                              ("stream", 0, 0, 2048, 0), ("stream", 0, 0, 2048, 1)):
        stream = torch.cuda.current_stream().cuda_stream

        def launch():
            if kind == "gather":
                rc = lib.vmi_diag_gather_read(buf.data_ptr(), n, sink.data_ptr(), chunk, inflight, blocks, nt, 0, stream)
            else:
                rc = lib.vmi_diag_stream_read(buf.data_ptr(), n, sink.data_ptr(), blocks, nt, 0, stream)
            assert rc == 0, _lib.last_error()

        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        reps = 30
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                launch()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps)
        rec = {"MB": mb, "kind": kind, "chunk_kb": chunk, "inflight_kb_per_wave": inflight, "workgroups": blocks, "nt": nt,
               "us_per_launch": round(best * 1e3, 2), "TBps": round(n / (best * 1e-3) / 1e12, 2)}
        res.append(rec)
        print(json.dumps(rec), flush=True)
    del buf
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
