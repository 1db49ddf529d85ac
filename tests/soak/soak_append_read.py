"""Soak run for the append-read form of paged_attention_v1 (vmi_paged_attention_v1_newest_f16; pa_queue.hpp APP and the
pa_v1_kernel twins): random batch shapes — full chip, more items than waves, small batches —, random length distributions
(equal, uniform, exponential, bimodal, sorted, empty rows), grouped-query heads, forced balanced modes.  Each case: `out` of the
append-read entry on caches that do NOT hold this step's rows, bit for bit against reshape_and_cache + paged_attention_v1 with
the same work decomposition; the caches untouched by it; the writing fused entry leaves the call pair's caches.
`PYTHONPATH=.:tests python tests/soak/soak_append_read.py [n_cases] [first_seed]`; exit code 1 if anything failed."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
from vllmini_amd import _lib, cache_ops, ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 61000
dev = torch.device("cuda:0")
lib = _lib.use_diag().__enter__()      # the diagnostic build: the balanced kernels' mode knob (same sources, same kernels)
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
BS = 16


def flags(mode=0, wq=0, nosort=0, team=0, early=0, nohybrid=0):
    return mode | (wq << 2) | (nohybrid << 10) | (nosort << 11) | (team << 12) | (early << 15)


MODES = [("auto", flags()), ("S", flags(1)), ("solo", flags(2, 2, 0, 1)), ("solo early sort", flags(2, 2, 0, 1, 1)),
         ("solo unranked", flags(2, 2, 1, 1)), ("team + solo quads", flags(2, 0, 0, 2)), ("team only", flags(2, 0, 0, 2, 0, 1))]
fails = 0
ran = {}
t0 = time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    D = int(rng.choice([64, 64, 64, 128]))
    Hkv = int(rng.choice([1, 2, 3, 4, 8, 12]))
    qpk = int(rng.choice([1, 1, 1, 2, 4]))
    H = Hkv * qpk
    B = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 400)), int(rng.integers(400, 900))]))
    top = int(rng.choice([17, 48, 130, 300, 520, 1024]))
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
    nblk = (np.maximum(lens, 1) + BS - 1) // BS
    NB = int(nblk.sum()) + 5
    MB = int(nblk.max()) + int(rng.integers(0, 3))
    g = torch.Generator(device=dev).manual_seed(seed)
    kc = (torch.rand((NB, Hkv, D // 8, BS, 8), device=dev, generator=g) * 2 - 1).to(torch.float16)
    vc = (torch.rand((NB, Hkv, D, BS), device=dev, generator=g) * 2 - 1).to(torch.float16)
    pad = int(rng.integers(0, 3)) * 8
    qkv = torch.randn((B, (H + 2 * Hkv) * D + pad), device=dev, generator=g).to(torch.float16)
    q = qkv[:, : H * D].view(B, H, D)
    key = qkv[:, H * D: (H + Hkv) * D].view(B, Hkv, D)
    value = qkv[:, (H + Hkv) * D: (H + 2 * Hkv) * D].view(B, Hkv, D)
    perm = rng.permutation(NB).astype(np.int32)
    tables = np.full((B, MB), -1, dtype=np.int32)
    at = 0
    for s in range(B):
        tables[s, : nblk[s]] = perm[at: at + nblk[s]]
        at += nblk[s]
    p1 = np.maximum(lens.astype(np.int64) - 1, 0)
    slots = tables[np.arange(B), p1 // BS].astype(np.int64) * BS + p1 % BS
    slots[lens <= 0] = -1
    tab, lens_d, slots_d = torch.from_numpy(tables).to(dev), torch.from_numpy(lens).to(dev), torch.from_numpy(slots).to(dev)
    scale, msl = float(D) ** -0.5, max(int(lens.max()), 1)
    i16 = torch.int16
    # which kernels: the library's own pick, and — where the shape allows — the balanced kernel by id under every mode
    qname = f"q_d{D}_s1q{2 if D == 64 else 1}"
    runs = [(0, "default", 0)]
    if D == 64 and B * H >= 8:
        runs += [(names[qname], f"{qname} {m}", f) for m, f in MODES if not (m == "S" and B * H > 3072)]
    for vid, what, qf in runs:
        lib.vmi_debug_set_queue_flags(qf)
        kc_n, vc_n = kc.clone(), vc.clone()
        out_n = torch.full((B, H, D), float("nan"), dtype=torch.float16, device=dev)
        try:
            ops.paged_attention_v1_append(out_n, q, key, value, kc_n, vc_n, Hkv, scale, tab, lens_d, BS, msl, _variant=vid, write_cache=False)
        except RuntimeError as e:
            if vid:
                continue           # (a forced id the shape does not admit)
            raise
        kname = ops.variant_names()[ops.last_variant() - 1]
        ran[kname] = ran.get(kname, 0) + 1
        kc_p, vc_p = kc.clone(), vc.clone()
        out_p = torch.full_like(out_n, float("nan"))
        cache_ops.reshape_and_cache(key, value, kc_p, vc_p, slots_d, "auto", 1.0)
        prev = ops.set_workspace_enabled(False)
        try:
            ops.paged_attention_v1(out_p, q, kc_p, vc_p, Hkv, scale, tab, lens_d, BS, msl, None, "auto", 1.0, _variant=names[kname])
        finally:
            ops.set_workspace_enabled(prev)
        torch.cuda.synchronize()
        ok = torch.equal(kc_n.view(i16), kc.view(i16)) and torch.equal(vc_n.view(i16), vc.view(i16))
        same = torch.equal(out_n.view(i16), out_p.view(i16))
        if not (ok and same):
            fails += 1
            bad = int((out_n.view(i16) != out_p.view(i16)).any(-1).any(-1).sum())
            print(f"FAIL seed {seed} {what} -> {kname}: B{B} H{H}/{Hkv} D{D} top {top} kind {kind}: caches untouched {ok}, out equal {same} "
                  f"({bad} sequences differ, max {float((out_n.float() - out_p.float()).abs().nan_to_num(9).max()):.3e})", flush=True)
    lib.vmi_debug_set_queue_flags(0)
    # the writing fused entry: the call pair's caches, an out within fp32 summation order of the pair's (its own decomposition)
    kc_f, vc_f = kc.clone(), vc.clone()
    out_f = torch.full((B, H, D), float("nan"), dtype=torch.float16, device=dev)
    ops.paged_attention_v1_append(out_f, q, key, value, kc_f, vc_f, Hkv, scale, tab, lens_d, BS, msl)
    torch.cuda.synchronize()
    live = torch.from_numpy(lens > 0).to(dev)
    ref_f = out_p[live].float()
    tol = 2.0 * torch.exp2(torch.floor(torch.log2(ref_f.abs().clamp_min(2.0 ** -14))) - 10.0)      # 2 fp16 ulp at |out|
    if not (torch.equal(kc_f.view(i16), kc_p.view(i16)) and torch.equal(vc_f.view(i16), vc_p.view(i16))
            and bool(((out_f[live].float() - ref_f).abs() <= torch.clamp_min(tol, 1e-3)).all())):
        # (max(1e-3, 2 fp16 ulp at |out|) — this step's value rows are N(0, 1), outputs reach 2 - 3: the fused entry may run another decomposition than the pair — the grouped-query kernels sum
        #  q.K^T on the matrix cores — and one flipped fp16 probability of a 2-3-token context moves an output by that much; every
        #  decomposition is pinned to the oracle at that bound by tests/test_parity_gpu.py)
        fails += 1
        dd = (out_f[live].float() - out_p[live].float()).abs().nan_to_num(99)
        print(f"FAIL seed {seed} fused entry ({ops.variant_names()[ops.last_variant() - 1]}; pair by {kname}): B{B} H{H}/{Hkv} D{D} top {top} kind {kind}: "
              f"kc equal {torch.equal(kc_f.view(i16), kc_p.view(i16))}, vc equal {torch.equal(vc_f.view(i16), vc_p.view(i16))}, "
              f"out max diff {float(dd.max()):.3e} in {int((dd > 1e-3).any(-1).any(-1).sum())} sequences", flush=True)
    if (seed - first + 1) % 50 == 0:
        print(f"{seed - first + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases from seed {first}, {fails} failures, {time.time() - t0:.0f} s; append-read comparisons by kernel: "
      + ", ".join(f"{k} {v}" for k, v in sorted(ran.items(), key=lambda kv: -kv[1])))
sys.exit(1 if fails else 0)
