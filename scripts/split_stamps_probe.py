"""Stage stamps of the split kernels (vllmini_amd/csrc/pa_split.hpp; diagnostic library, vmi_diag_set_split_stamps): where do the
microseconds of a launch go when one (sequence, head) is spread over several workgroups that meet in a workspace?

  python scripts/split_stamps_probe.py out.json [--cases name:batch:seq_len:variant,...] [--flags N]

Per case: every wave's eight stamps (entry, lengths known, first K group consumed, K pass done, granule published, exchange
complete, V pass done, end) relative to the launch's first wave entry — first / median / last wave — in microseconds."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stage_timeline_probe import pair, setup  # noqa: E402
from vllmini_amd import _lib, cache_ops, ops  # noqa: E402

STAGES = ["entry", "lengths known", "first K group consumed", "K pass done", "granule published", "exchange complete",
          "V pass done", "end"]
CASES = ["cfg2:32:512:d64_x8_u1_nt0", "b1:1:1024:d64_x32_u1_nt0", "b1:1:1024:d64_x16_u2_nt0", "b8:8:1024:d64_x16_u1_nt0",
         "b16:16:1024:d64_x8_u1_nt0", "b2_l4k:2:4096:d64_x32_u2_nt0"]


def main():
    out_path = sys.argv[1]
    cases = CASES
    flags = 0
    for i, a in enumerate(sys.argv):
        if a == "--cases":
            cases = sys.argv[i + 1].split(",")
        if a == "--flags":
            flags = int(sys.argv[i + 1])
    lib = _lib.use_diag().__enter__()
    lib.vmi_debug_set_split_flags(flags)
    dev = torch.device("cuda:0")
    names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
    rec = torch.zeros((65536, 10), dtype=torch.int64, device=dev)
    results = {}
    for case in cases:
        cfg, wl, out, var = setup(case, dev)
        vid = names[var]
        for i in range(20):
            pair(ops, cache_ops, cfg, wl, out, i, vid)
        torch.cuda.synchronize()
        reps = []
        for rep in range(12):
            rec.zero_()
            torch.cuda.synchronize()
            assert lib.vmi_diag_set_split_stamps(rec.data_ptr(), 0) == 0
            pair(ops, cache_ops, cfg, wl, out, rep, vid)
            pair(ops, cache_ops, cfg, wl, out, rep + 1, vid)   # the second launch's records stay
            torch.cuda.synchronize()
            assert lib.vmi_diag_set_split_stamps(None, 0) == 0
            r = rec.cpu().numpy().astype(np.int64)
            reps.append(r[r[:, 0] > 0])
        first, med, last, spans = [], [], [], []
        for r in reps:
            t0 = r[:, 0].min()
            ts = np.where(r[:, :8] > 0, (r[:, :8] - t0) * 0.01, np.nan)
            first.append(np.nanmin(ts, axis=0))
            med.append(np.nanmedian(ts, axis=0))
            last.append(np.nanmax(ts, axis=0))
            spans.append(float(np.nanmax(ts[:, 7])))
        m = lambda a: np.nanmedian(np.array(a), axis=0).round(2).tolist()   # noqa: E731
        r = reps[-1]
        results[case] = {"waves": int(len(r)), "blocks_per_wave_max": int((r[:, 9] >> 8).max()),
                         "cus_used": int(len(np.unique(((r[:, 9] & 0xF) << 16) | ((r[:, 8] >> 8) & 0xFF)))),
                         "span_us": float(np.median(spans)), "stages": STAGES, "stamp_us_first_wave": m(first),
                         "stamp_us_median_wave": m(med), "stamp_us_last_wave": m(last)}
        print(case, json.dumps(results[case]), flush=True)
        del wl, out
    json.dump(results, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
