"""Soak run for the balanced paged_attention_v1 kernels (pa_queue.hpp): random batch shapes big enough that the hand-out
over persistent workers really runs (several rounds, late and early ranking, teams), random length distributions, every
forced mode and the kernel's own choice.  Each case: all rows bit for bit against the one-wave-per-head kernel (solo
modes) or within the team tolerance, determinism, and a sample of sequences against the CPU kernel model (checker only).
`PYTHONPATH=.:tests python tests/soak/soak_queue.py [n_cases] [first_seed] [auto|fp8|mix]`; exit code 1 if anything failed."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
import oracle  # noqa: E402  (checker)
from vllmini_amd import _lib, ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
KV = sys.argv[3] if len(sys.argv) > 3 else "mix"   # "auto" (16-bit pages), "fp8" (E4M3 pages), "mix" (a third fp8)
dev = torch.device("cuda:0")
lib = _lib.use_diag().__enter__()   # the diagnostic build for the whole process (python -m vllmini_amd.build --diag)
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
BS = 16


def flags(mode=0, wq=0, nosort=0, team=0, early=0, nohybrid=0):
    return mode | (wq << 2) | (nohybrid << 10) | (nosort << 11) | (team << 12) | (early << 15)


MODES = [("auto", flags(), None), ("S", flags(1), True), ("solo", flags(2, 2, 0, 1), True),
         ("solo early sort", flags(2, 2, 0, 1, 1), True), ("solo 1 worker", flags(2, 1, 0, 1), True),
         ("solo unranked", flags(2, 2, 1, 1), True), ("team + solo quads", flags(2, 0, 0, 2), False), ("team only", flags(2, 0, 0, 2, 0, 1), False)]
fails = 0
t0 = time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    f8 = KV == "fp8" or (KV == "mix" and seed % 3 == 0)
    D = 64 if f8 else int(rng.choice([64, 64, 128]))
    H = int(rng.choice([4, 5, 7, 8, 12, 16] if D == 64 else [4, 8, 12]))
    B = int(rng.integers(40, 700 if D == 64 else 300))
    top = int(rng.choice([48, 130, 300, 520]))
    if seed % 23 == 0:      # more sequences than the kernel ranks in LDS (2048): index order, teams take every item
        B, top = int(rng.integers(2049, 2500)), int(rng.choice([48, 130]))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        lens = np.full(B, top)
    elif kind == 1:
        lens = rng.integers(0, top + 1, B)
    elif kind == 2:
        lens = np.minimum((rng.exponential(1.0, B) * top / 4).astype(np.int64) + 1, top)
    elif kind == 3:
        lens = np.where(rng.random(B) < 0.1, top, max(top // 8, 1))
    else:
        lens = np.sort(rng.integers(1, top + 1, B))[::-1].copy()
    lens = lens.astype(np.int32)
    lens[int(rng.integers(0, B))] = top
    nblk = (lens + BS - 1) // BS
    need = int(nblk.sum())
    MB = max(int(nblk.max()), 1)
    g = torch.Generator(device=dev).manual_seed(seed)
    NB = need + 5
    if f8:   # E4M3 codes of magnitude < 2, either sign (no NaN codes)
        kc = (torch.randint(0, 64, (NB, H, D // 16, BS, 16), dtype=torch.uint8, device=dev, generator=g)
              | (torch.randint(0, 2, (NB, H, D // 16, BS, 16), dtype=torch.uint8, device=dev, generator=g) << 7))
        vc = (torch.randint(0, 64, (NB, H, D, BS), dtype=torch.uint8, device=dev, generator=g)
              | (torch.randint(0, 2, (NB, H, D, BS), dtype=torch.uint8, device=dev, generator=g) << 7))
    else:
        kc = (torch.rand((NB, H, D // 8, BS, 8), device=dev, generator=g) * 2 - 1).to(torch.float16)
        vc = (torch.rand((NB, H, D, BS), device=dev, generator=g) * 2 - 1).to(torch.float16)
    q = torch.randn((B, H, D), device=dev, generator=g).to(torch.float16)
    perm = rng.permutation(NB)[:need].astype(np.int32)
    tables = np.full((B, MB), -1, dtype=np.int32)
    pos = 0
    for s in range(B):
        n = int(nblk[s])
        tables[s, :n] = perm[pos:pos + n]
        pos += n
    tab = torch.from_numpy(tables).to(dev)
    lens_d = torch.from_numpy(lens).to(dev)
    scale = float(D) ** -0.5

    def attend(variant, fl):
        lib.vmi_debug_set_queue_flags(fl)
        out = torch.full((B, H, D), float("nan"), dtype=torch.float16, device=dev)
        ops.paged_attention_v1(out, q, kc, vc, H, scale, tab, lens_d, BS, top, None, "fp8" if f8 else "auto", 1.0, 0, 0, 1, 1,
                               0, _variant=variant)
        torch.cuda.synchronize()
        lib.vmi_debug_set_queue_flags(0)
        return out

    plain = attend(names["fp8_d64_bs16_h1_w1_u2_nt1" if f8 else f"d{D}_h1_w1_u1_nt1"], 0)
    # ((seed // 3) % 3 == 0: the default fp8 kernel, K pass on the matrix cores — single-wave modes bit-identical to EACH OTHER, all
    #  modes within the tolerance of the plain kernel; the VALU forms are bit-identical to the plain kernel as well)
    qn = names[("fp8_q_d64_s2q4m", "fp8_q_d64_s2q4", "fp8_q_d64_s1q2")[(seed // 3) % 3] if f8 else (f"q_d{D}_s1q2" if D == 64 else f"q_d{D}_s1q1")]
    km = f8 and (seed // 3) % 3 == 0
    base = None if km else plain
    what = f"seed {seed}: B{B} H{H} D{D} top {top} kind {kind}{' fp8' if f8 else ''}"
    ok = bool(torch.isfinite(plain).all())
    nwaves = torch.cuda.get_device_properties(dev).multi_processor_count * (3 if D == 64 else 2) * 4
    for label, fl, bitwise in MODES:
        if label == "S" and B * H > nwaves:
            bitwise = None  # more items than waves: mode S cannot be forced, the kernel chooses (possibly teams)
        got = attend(qn, fl)
        again = attend(qn, fl)
        if not torch.equal(got.view(torch.int16), again.view(torch.int16)):
            print(f"FAIL {what} [{label}]: not deterministic")
            ok = False
        if bitwise and base is None:
            base = got
        if bitwise:
            if not torch.equal(got.view(torch.int16), base.view(torch.int16)):
                print(f"FAIL {what} [{label}]: differs from the plain kernel, max {(got.float() - plain.float()).abs().max().item():.3e}")
                ok = False
        if not bitwise or km:
            d = (got.float() - plain.float()).abs().max().item()
            if not (d <= 1e-3) or not bool(torch.isfinite(got).all()):
                print(f"FAIL {what} [{label}]: max|d| vs the plain kernel {d:.3e}")
                ok = False
    # a sample of sequences against the kernel model
    idx = np.unique(np.r_[np.argsort(-lens, kind="stable")[:3], rng.integers(0, B, 3), 0, B - 1])
    blocks = np.unique(np.concatenate([tables[i, : nblk[i]] for i in idx])) if need else np.zeros(1, dtype=np.int32)
    remap = {int(b): j for j, b in enumerate(blocks)}
    small = np.full((len(idx), MB), -1, dtype=np.int32)
    for r, i in enumerate(idx):
        small[r, : nblk[i]] = [remap[int(b)] for b in tables[i, : nblk[i]]]
    bsel = torch.from_numpy(blocks.astype(np.int64)).to(dev)
    qs, ks, vs = q[torch.from_numpy(idx).to(dev)].cpu().numpy(), kc[bsel].cpu().numpy(), vc[bsel].cpu().numpy()
    ref = (oracle.paged_attention_v1_fp8(qs, ks, vs, H, scale, small, lens[idx], BS, kv_scale=1.0) if f8 else
           oracle.paged_attention_v1(qs, ks, vs, H, scale, small, lens[idx], BS, threads=8))
    d = np.abs(plain.cpu().numpy()[idx].astype(np.float64) - ref.astype(np.float64)).max()
    if not d <= 1e-3:
        print(f"FAIL {what}: plain kernel vs model {d:.3e}")
        ok = False
    fails += 0 if ok else 1
print(f"soak_queue: {n_cases} cases from seed {first}, {fails} failed, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
