"""Soak run of the SPLIT kernels (vllmini_amd/csrc/pa_split.hpp) on the PRODUCT library: one (sequence, head) over several
workgroups of one launch that meet in the wrapper's workspace.  What the unit tests do not vary: random batches of 1 .. 48
sequences (the larger ones in rounds), 1 .. 16 heads (grouped KV heads sometimes), head size 64 / 128, contexts up to 20 000 tokens with every length
distribution of soak_auto (equal, uniform, one long among one-token sequences, empty ones), max_seq_len at the longest length
or a capacity far above it, ALiBi sometimes — and, as the hand-off rules demand (MI355X_MICROARCH.md: "test every hand-off
under UNEVEN load, consumer L1-warm"), HALF the cases run while another stream streams through a 1 GiB buffer, and every case
is launched three times back to back on the same workspace.  Each case: an explicit split kernel (random waves per item, U,
nt; four query heads per item where the grouping allows; fp8 E4M3 pages every seventh case) or the default entry; all rows against the CPU kernel model (checker only) at the tight bound, the three launches
bit-identical, the workspace's control words zero and its give-up counter 0 afterwards.
`PYTHONPATH=.:tests python tests/soak/soak_split.py [n_cases] [first_seed]`; exit code 1 if anything failed."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
import oracle  # noqa: E402  (checker)
from helpers import make_case, ulp16  # noqa: E402
from vllmini_amd import _lib, ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
dev = torch.device("cuda:0")
assert _lib.load().vmi_is_diag_build() == 0
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
load_buf = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()     # 1 GiB
load_stream = torch.cuda.Stream()
fails, t0, ran, refused, picked = 0, time.time(), 0, 0, {}
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    D = int(rng.choice([64, 64, 128]))
    qpk = int(rng.choice([1, 1, 2, 4]))
    hkv = int(rng.choice([1, 2, 3, 4]))
    H = hkv * qpk
    B = int(rng.choice([1, 1, 2, 3, 4, 8, 12, 24, 48]))      # (24 / 48: more workgroups than are resident -> in rounds)
    top = int(rng.choice([40, 700, 2048, 4096, 9000, 20000], p=[.1, .15, .2, .25, .2, .1]))
    if B * H * top > 1.2e6:
        top = max(64, int(1.2e6 / (B * H)))
    kind = int(rng.integers(0, 5))
    lens = (np.full(B, top) if kind == 0 else rng.integers(0, top + 1, B) if kind == 1 else
            np.where(np.arange(B) == int(rng.integers(0, B)), top, 1) if kind == 2 else
            np.minimum((rng.exponential(1.0, B) * top / 3).astype(np.int64), top) if kind == 3 else
            rng.choice([0, 1, 15, 16, 17, top], B))
    lens = np.asarray(lens, dtype=np.int64)
    case = make_case(rng, B, H, D, lens.tolist(), num_kv_heads=hkv, q_row_pad=int(rng.integers(0, 3)), poison_tail=True)
    msl = max(int(lens.max()), 1)
    if seed % 3 == 0:
        msl = int(msl * rng.choice([1.5, 4, 16])) + 5          # capacity-style max_seq_len (scheduler.py:97)
    slopes = (rng.uniform(0.01, 0.5, H).astype(np.float32) if seed % 4 == 0 else None)
    x = int(rng.choice([8, 16, 32, 64, 128, 256]))
    vname = f"d{D}_x{x}_u{int(rng.choice([1, 2]))}_nt{int(rng.choice([0, 1]))}"
    if vname not in names:      # (round 6: the one-block-per-group forms live in the diagnostic library only)
        vname = vname.replace("_u1_", "_u2_")
    fp8 = seed % 7 == 3                                       # fp8 E4M3 pages (random codes, kv_scale 1 or 0.7)
    if fp8:
        pool = [n for n in names if n.startswith(f"fp8_d{D}_x") or (qpk % 4 == 0 and n.startswith(f"fp8_d{D}_gq4_x"))]
        vname = pool[int(rng.integers(0, len(pool)))]
    elif qpk % 4 == 0 and seed % 3 == 1:                      # four query heads of a KV head per item
        pool = [n for n in names if n.startswith(f"d{D}_gq4_x")]
        vname = pool[int(rng.integers(0, len(pool)))]
    vid = 0 if seed % 5 == 0 else names[vname]
    kv_scale = 1.0
    if fp8:
        kv_scale = float(rng.choice([1.0, 0.7]))
        NB = case["kc"].shape[0]
        kq = rng.integers(0, 256, (NB, hkv, D // 16, 16, 16), dtype=np.uint8)
        vq = rng.integers(0, 256, (NB, hkv, D, 16), dtype=np.uint8)
        case["kq"] = np.where((kq & 0x7f) >= 0x40, (kq & 0x80) | 0x30 | (kq & 7), kq).astype(np.uint8)   # |x| < 2, no NaN codes
        case["vq"] = np.where((vq & 0x7f) >= 0x40, (vq & 0x80) | 0x30 | (vq & 7), vq).astype(np.uint8)
    S = B
    qbuf = torch.from_numpy(case["qbuf"]).to(dev)
    q = qbuf[:, : H * D].view(S, H, D)
    kc, vc = torch.from_numpy(case["kq" if fp8 else "kc"]).to(dev), torch.from_numpy(case["vq" if fp8 else "vc"]).to(dev)
    tab, ln = torch.from_numpy(case["tables"]).to(dev), torch.from_numpy(case["lens"]).to(dev)
    al = None if slopes is None else torch.from_numpy(slopes).to(dev)
    busy = seed % 2 == 0
    outs = []
    try:
        for rep in range(3):
            if busy:
                with torch.cuda.stream(load_stream):
                    load_buf.mul_(1.0000001)
            out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=dev)
            ops.paged_attention_v1(out, q, kc, vc, hkv, case["scale"], tab, ln, 16, msl, al, "fp8" if fp8 else "auto", kv_scale, 0, 0, 1, 1,
                                   0, _variant=vid)
            outs.append(out)
        torch.cuda.synchronize()
    except RuntimeError as e:
        if "would launch" in str(e) or "LDS" in str(e):      # not all resident / LDS of this shape: refused, never wrong
            refused += 1
            continue
        raise
    ran += 1
    label = ops.last_launch_label()
    picked[label] = picked.get(label, 0) + 1
    if fp8:
        ref = oracle.paged_attention_v1_fp8(case["q"], case["kq"], case["vq"], hkv, case["scale"], case["tables"], case["lens"], 16,
                                            kv_scale=kv_scale, alibi_slopes=slopes, threads=8).astype(np.float64)
    else:
        ref = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], hkv, case["scale"], case["tables"], case["lens"], 16,
                                        alibi_slopes=slopes, threads=8).astype(np.float64)
    got = outs[0].cpu().numpy().astype(np.float64)
    d = np.abs(got - ref)
    ok = np.isfinite(got).all() and bool((d <= np.maximum(2 * ulp16(ref), 5e-4 * max(1.0, 2 * kv_scale))).all())
    same = all(torch.equal(outs[0].view(torch.int16), o.view(torch.int16)) for o in outs[1:])
    ws = ops.workspace_for(0, create=False)
    n_ctl = 256 + 8192 * 4 + 2048 * 4 * 8
    clean = ws is None or int(ws[:n_ctl].view(torch.int64).ne(0).sum().item()) == 0
    if not (ok and same and clean):
        fails += 1
        print(f"FAIL seed {seed}: {label} D{D} H{H}/{hkv} B{B} lens {lens.tolist()} msl {msl} busy {busy}: "
              f"max|d| {d.max() if np.isfinite(d).all() else 'nan'} ok {ok} identical {same} workspace clean {clean}", flush=True)
print(f"{ran} cases run, {refused} refused (not resident), {fails} failures, {time.time() - t0:.0f} s; kernels: "
      + ", ".join(f"{k} x{v}" for k, v in sorted(picked.items(), key=lambda kv: -kv[1])[:12]))
sys.exit(1 if fails else 0)
