"""Can a decode step hide the block's linear layers behind the other half of the batch's attention?  The step is a chain
per layer: [ln_1 + c_attn] -> scatter -> attention (HBM-bound, 125 us at batch 256) -> three more linear launches (latency chains,
~25 us together).  Two micro-batches of 128 sequences on two streams of ONE hipGraph, tied so that their attention launches
alternate (A.attn(l) -> B.attn(l) -> A.attn(l + 1) ...), would run a half's linear layers while the other half's attention
streams its pages.  This probe builds the three graphs over synthetic buffers — (0) one batch of 256, (1) two halves in sequence
on one stream, (2) two halves on two streams with the alternating edges — and times their replays.
`python scripts/microbatch_overlap_probe.py [out.json]`"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllmini_amd import cache_ops, gpt2_layer as gl, ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, D, L, BS, E, LAYERS = 256, 12, 64, 1024, 16, 768, 12
PER = L // BS
NB = 2 * B * PER
g = torch.Generator(device=dev).manual_seed(0)
kc = torch.empty((NB, H, D // 8, BS, 8), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
vc = torch.empty((NB, H, D, BS), dtype=torch.float16, device=dev).uniform_(-1, 1, generator=g)
perm = np.random.default_rng(0).permutation(NB)[: B * PER].reshape(B, PER).astype(np.int32)
scale = D ** -0.5
lnw, lnb = torch.ones(E, dtype=torch.float16, device=dev), torch.zeros(E, dtype=torch.float16, device=dev)
WL = [{n: (gl.pack_weight(torch.randn(o, i, dtype=torch.float16, device=dev, generator=g) * 0.02),
           torch.randn(o, dtype=torch.float16, device=dev, generator=g) * 0.02)
       for n, (o, i) in {"c_attn": (3 * E, E), "c_proj": (E, E), "c_fc": (4 * E, E), "mlp_proj": (E, 4 * E)}.items()}
      for _ in range(LAYERS)]


class Half:
    def __init__(self, rows):
        n = len(rows)
        self.n = n
        self.tab = torch.from_numpy(perm[rows]).to(dev)
        self.lens = torch.full((n,), L, dtype=torch.int32, device=dev)
        self.slots = (self.tab[:, -1].to(torch.int64) * BS + BS - 1).contiguous()
        self.x = torch.randn(n, E, dtype=torch.float16, device=dev, generator=g)
        self.qkv = torch.empty(n, 3 * E, dtype=torch.float16, device=dev)
        self.out = torch.empty(n, H, D, dtype=torch.float16, device=dev)
        self.h = torch.empty(n, 4 * E, dtype=torch.float16, device=dev)

    def pre(self, l):     # ln_1 + c_attn, scatter
        gl.linear(self.x, *WL[l]["c_attn"], ln=(lnw, lnb, 1e-5), out=self.qkv)
        q, k, v = (self.qkv[:, j * E:(j + 1) * E].view(self.n, H, D) for j in range(3))
        cache_ops.reshape_and_cache(k, v, kc, vc, self.slots, "auto", 1.0)
        return q

    variant = 0

    def attn(self, q):
        ops.paged_attention_v1(self.out, q, kc, vc, H, scale, self.tab, self.lens, BS, L, None, "auto", 1.0, 0, 0, 1, 1, 0,
                               _variant=self.variant)

    def post(self, l):    # c_proj + residual, ln_2 + c_fc + GELU, mlp.c_proj + residual
        gl.linear(self.out.view(self.n, E), *WL[l]["c_proj"], residual=self.x, out=self.x)
        gl.linear(self.x, *WL[l]["c_fc"], ln=(lnw, lnb, 1e-5), gelu=True, out=self.h)
        gl.linear(self.h, *WL[l]["mlp_proj"], residual=self.x, out=self.x)


def sequential(parts):
    for l in range(LAYERS):
        for p in parts:
            q = p.pre(l)
            p.attn(q)
            p.post(l)


def overlapped(a, b, s0, s1, alternate=True):
    """a on s0 (the capture stream), b on s1; attention launches alternate: a(l) -> b(l) -> a(l + 1)."""
    fork = torch.cuda.Event()
    fork.record(s0)
    s1.wait_event(fork)
    b_done = None
    for l in range(LAYERS):
        with torch.cuda.stream(s0):
            q = a.pre(l)
            if b_done is not None and alternate:
                s0.wait_event(b_done)
            a.attn(q)
            a_done = torch.cuda.Event()
            a_done.record(s0)
            a.post(l)
        with torch.cuda.stream(s1):
            q = b.pre(l)
            if alternate:
                s1.wait_event(a_done)
            b.attn(q)
            b_done = torch.cuda.Event()
            b_done.record(s1)
            b.post(l)
    join = torch.cuda.Event()
    join.record(s1)
    s0.wait_event(join)


def capture(fn, s0):
    s0.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s0):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s0):
        fn()
    return gr


def time_graph(gr, rounds=7):
    gr.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gr.replay()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) * 1e3)
    return float(np.median(best))


s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
full = Half(np.arange(B))
ha, hb = Half(np.arange(B // 2)), Half(np.arange(B // 2, B))
res = {}
res["one_batch_of_256_us"] = time_graph(capture(lambda: sequential([full]), s0))
res["two_halves_one_stream_us"] = time_graph(capture(lambda: sequential([ha, hb]), s0))
res["two_halves_two_streams_alternating_us"] = time_graph(capture(lambda: overlapped(ha, hb, s0, s1), s0))
res["two_halves_two_streams_free_running_us"] = time_graph(capture(lambda: overlapped(ha, hb, s0, s1, alternate=False), s0))
res["half_batch_default_kernel"] = ops.last_launch_label()
names = ops.variant_names()
for vname in ("q_d64_s1q2", "d64_h1_w4_u2_nt1", "d64_h1_w2_u2_nt1"):   # kernels that leave wave slots free at half a batch
    if vname in names:
        ha.variant = hb.variant = names.index(vname) + 1
        res[f"two_halves_two_streams_alternating_{vname}_us"] = time_graph(capture(lambda: overlapped(ha, hb, s0, s1), s0))
        res[f"two_halves_one_stream_{vname}_us"] = time_graph(capture(lambda: sequential([ha, hb]), s0))
res["note"] = "12 layers x (ln_1 + c_attn, scatter, attention, c_proj + res, ln_2 + c_fc + GELU, mlp.c_proj + res); no embeddings / lm_head"
print(json.dumps(res), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
