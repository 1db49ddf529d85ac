import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from vllmini_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_case, BS
dev = torch.device("cuda:0")
def run(case, variant):
    S,H,D = case["q"].shape
    q = torch.from_numpy(np.ascontiguousarray(case["q"])).to(dev)
    kc = torch.from_numpy(case["kc"]).to(dev); vc = torch.from_numpy(case["vc"]).to(dev)
    out = torch.zeros((S,H,D), dtype=torch.float16, device=dev)
    ops.paged_attention_v1(out, q, kc, vc, H, case["scale"], torch.from_numpy(case["tables"]).to(dev),
        torch.from_numpy(case["lens"]).to(dev), BS, int(case["lens"].max()), None, "auto", 1.0, 0,0,1,1,0, _variant=variant)
    torch.cuda.synchronize()
    return out.cpu().numpy()
rng = np.random.default_rng(0)
for L in [3, 16, 40]:
    for name in ["random", "V=1", "K=0", "K=0,V=tok", "V=dim"]:
        case = make_case(rng, 1, 1, 64, [L])
        if name == "V=1": case["vc"][:] = 1
        if name.startswith("K=0"): case["kc"][:] = 0
        if name == "K=0,V=tok":
            case["vc"][:] = np.arange(16, dtype=np.float16)[None,None,None,:]
        if name == "V=dim":
            case["vc"][:] = (np.arange(64, dtype=np.float16)/64)[None,None,:,None]
        ref = oracle.paged_attention_v1(case["q"], case["kc"], case["vc"], 1, case["scale"], case["tables"], case["lens"], BS)
        for v in [1, 3]:
            got = run(case, v)
            d = np.abs(got.astype(np.float64)-ref.astype(np.float64))
            print(f"L={L} {name:10s} variant={v} maxdiff={d.max():.4f} got[:6]={got[0,0,:6]} ref[:6]={ref[0,0,:6]}")
