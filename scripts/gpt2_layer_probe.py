"""Times of the GPT-2 block's linear layers at a decode batch: this build's kernels (vllmini_amd/gpt2_layer.py, one launch
each) against the torch module chain they replace (layer_norm + F.linear [+ gelu | + add]), both replayed from hipGraphs of
`reps` chained launches so that the figure is device time per launch, launch gaps included.  The launches of a graph walk
through TWELVE weight sets in turn, as a token's 12 layers do: a layer's weights then come from the Infinity Cache, not from an
L2 that the previous launch of the same weights left warm (one set: 30.9 us per layer at batch 256; twelve: what the step sees).
`python scripts/gpt2_layer_probe.py [out.json] [batch ...]`"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import gpt2_layer as gl  # noqa: E402

dev = torch.device("cuda:0")
E = 768
out_path = sys.argv[1] if len(sys.argv) > 1 else None
batches = [int(a) for a in sys.argv[2:]] or [256]
g = torch.Generator(device=dev).manual_seed(0)
res = []


LAYERS = 12


def graph_us(fn, reps=24, rounds=5):
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i % LAYERS)
    torch.cuda.current_stream(dev).wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for i in range(reps):
            fn(i % LAYERS)
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gr.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


for M in batches:
    x = torch.randn(M, E, dtype=torch.float16, device=dev, generator=g)
    h4 = torch.randn(M, 4 * E, dtype=torch.float16, device=dev, generator=g)
    lnw, lnb = torch.ones(E, dtype=torch.float16, device=dev), torch.zeros(E, dtype=torch.float16, device=dev)
    WL = [{n: (torch.randn(o, i, dtype=torch.float16, device=dev, generator=g) * 0.02,
               torch.randn(o, dtype=torch.float16, device=dev, generator=g) * 0.02)
           for n, (o, i) in {"c_attn": (3 * E, E), "c_proj": (E, E), "c_fc": (4 * E, E), "mlp_proj": (E, 4 * E)}.items()}
          for _ in range(LAYERS)]
    PL = [{n: gl.pack_weight(w) for n, (w, _) in W.items()} for W in WL]
    y3, y1, y4 = (torch.empty(M, k * E, dtype=torch.float16, device=dev) for k in (3, 1, 4))
    cases = {
        "ln_1 + c_attn": (lambda l: gl.linear(x, *WL[l]["c_attn"], ln=(lnw, lnb, 1e-5), out=y3),
                          lambda l: F.linear(F.layer_norm(x, (E,), lnw, lnb, 1e-5), *WL[l]["c_attn"]), (3 * E, E, True, gl.EPI_BIAS)),
        "c_proj + residual": (lambda l: gl.linear(x, *WL[l]["c_proj"], residual=y1, out=y1),
                              lambda l: y1 + F.linear(x, *WL[l]["c_proj"]), (E, E, False, gl.EPI_BIAS_RESIDUAL)),
        "ln_2 + c_fc + gelu": (lambda l: gl.linear(x, *WL[l]["c_fc"], ln=(lnw, lnb, 1e-5), gelu=True, out=y4),
                               lambda l: F.gelu(F.linear(F.layer_norm(x, (E,), lnw, lnb, 1e-5), *WL[l]["c_fc"])),
                               (4 * E, E, True, gl.EPI_BIAS_GELU)),
        "mlp.c_proj + residual": (lambda l: gl.linear(h4, *WL[l]["mlp_proj"], residual=y1, out=y1),
                                  lambda l: y1 + F.linear(h4, *WL[l]["mlp_proj"]), (E, 4 * E, False, gl.EPI_BIAS_RESIDUAL)),
    }
    packed = {
        "ln_1 + c_attn": lambda l: gl.linear(x, PL[l]["c_attn"], WL[l]["c_attn"][1], ln=(lnw, lnb, 1e-5), out=y3),
        "c_proj + residual": lambda l: gl.linear(x, PL[l]["c_proj"], WL[l]["c_proj"][1], residual=y1, out=y1),
        "ln_2 + c_fc + gelu": lambda l: gl.linear(x, PL[l]["c_fc"], WL[l]["c_fc"][1], ln=(lnw, lnb, 1e-5), gelu=True, out=y4),
        "mlp.c_proj + residual": lambda l: gl.linear(h4, PL[l]["mlp_proj"], WL[l]["mlp_proj"][1], residual=y1, out=y1),
    }
    tot_n = tot_t = tot_p = 0.0
    for name, (native, torch_chain, (N, K, ln, epi)) in cases.items():
        n_us, t_us, p_us = graph_us(native), graph_us(torch_chain), graph_us(packed[name])
        tot_p += p_us
        tot_n, tot_t = tot_n + n_us, tot_t + t_us
        rec = {"batch": M, "layer": name, "kernel": gl.kernel_name(M, N, K, ln, epi), "native_us": round(n_us, 2), "native_packed_us": round(p_us, 2),
               "torch_modules_us": round(t_us, 2)}
        res.append(rec)
        print(json.dumps(rec), flush=True)
    rec = {"batch": M, "layer": "all four", "native_us": round(tot_n, 2), "native_packed_us": round(tot_p, 2), "torch_modules_us": round(tot_t, 2)}
    res.append(rec)
    print(json.dumps(rec), flush=True)
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
