import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle
from helpers import make_case, ulp16
from vllmini_amd import ops
dev = torch.device("cuda:0")
names = {n: i + 1 for i, n in enumerate(ops.variant_names())}
bad = 0
for D in (64, 128):
    for (S, hkv, qpk, lens) in ((3, 2, 4, [4096, 100, 0]), (2, 1, 8, [1500, 17]), (1, 2, 4, [9000]), (5, 1, 4, [1, 16, 33, 700, 2048])):
        H = hkv * qpk
        rng = np.random.default_rng(D + S + H)
        case = make_case(rng, S, H, D, lens, num_kv_heads=hkv, q_row_pad=1, poison_tail=True)
        slopes = (rng.uniform(0.01, 0.3, H).astype(np.float32) if S == 2 else None)
        ref = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], hkv, case["scale"], case["tables"], case["lens"], 16, alibi_slopes=slopes, threads=8).astype(np.float64)
        qbuf = torch.from_numpy(case["qbuf"]).to(dev); q = qbuf[:, : H * D].view(S, H, D)
        kc, vc = torch.from_numpy(case["kc"]).to(dev), torch.from_numpy(case["vc"]).to(dev)
        tab, ln = torch.from_numpy(case["tables"]).to(dev), torch.from_numpy(case["lens"]).to(dev)
        al = None if slopes is None else torch.from_numpy(slopes).to(dev)
        for n in [k for k in names if k.startswith(f"d{D}_gq4_x")]:
            outs = []
            for rep in range(2):
                out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=dev)
                ops.paged_attention_v1(out, q, kc, vc, hkv, case["scale"], tab, ln, 16, max(lens), al, "auto", 1.0, 0, 0, 1, 1, 0, _variant=names[n])
                outs.append(out)
            torch.cuda.synchronize()
            got = outs[0].cpu().numpy().astype(np.float64)
            d = np.abs(got - ref)
            ok = np.isfinite(got).all() and (d <= np.maximum(2 * ulp16(ref), 5e-4)).all() and torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
            if not ok:
                bad += 1; print("FAIL", n, S, H, hkv, lens, d.max() if np.isfinite(d).all() else "nan")
print("gq4 check done, failures:", bad, "status", ops.workspace_status(0))
