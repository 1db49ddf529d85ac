"""Gather-read ceiling by contiguous chunk size (math removed): what HBM gives 1-KiB (fp8 D=64 tile), 2-KiB (fp16 D=64)
and 4-KiB (D=128) requests at 3072 waves.  -> gpurun_out/gather_chunks.json"""
import json, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import _lib
lib = _lib.use_diag().__enter__()   # the diagnostic build for the whole process (python -m vllmini_amd.build --diag)
dev = torch.device("cuda:0")
bufs = [torch.empty(768 * 1024 * 1024, dtype=torch.uint8, device=dev).random_() for _ in range(2)]
sink = torch.zeros(1, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
res = []
for blocks in (768, 1536):
    for kb in (1, 2, 4):
        for infl in (1, 2, 4, 8):
            evs = []
            for i in range(16):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_ = bufs[i % 2]
                a.record()
                rc = lib.vmi_diag_gather_read(s_.data_ptr(), s_.numel(), sink.data_ptr(), kb, infl, blocks, 1, 0, stream)
                b.record()
                if rc != 0:
                    break
                evs.append((a, b))
            if rc != 0:
                continue
            torch.cuda.synchronize()
            ms = statistics.median(a.elapsed_time(b) for a, b in evs[4:])
            r = {"chunk_kb": kb, "inflight_kb_per_wave": infl, "waves": blocks * 4, "TBps": bufs[0].numel() / (ms * 1e-3) / 1e12}
            res.append(r)
            print(json.dumps(r), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gather_chunks.json", "w"), indent=1)
