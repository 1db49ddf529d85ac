"""Soak run of the randomized parity sweep (tests/test_parity_gpu.py::test_randomized_parity_sweep logic) over many
seeds, plus random fp8 cases: `PYTHONPATH=.:tests python tests/soak/soak_parity.py [n_cases] [first_seed]`.
Prints one line per failure and a summary; exit code 1 if anything failed."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, "tests")
import oracle  # noqa: E402  (checker)
import test_parity_gpu as T  # noqa: E402
from helpers import make_case  # noqa: E402
from vllmini_amd import ops  # noqa: E402
from vllmini_amd import _lib  # noqa: E402
_lib.use_extras().__enter__()   # bfloat16 / float32 / E5M2 / block-sparse live in libvmi_paged_attention_extras.so (build.py --extras)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
names = ops.variant_names()
fails = 0
t0 = time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    D = int(rng.choice([64, 80, 96, 112, 128, 192, 256]))
    bs = int(rng.choice([8, 16, 32]))
    hkv = int(rng.choice([1, 2, 3, 4]))
    H = hkv * int(rng.choice([1, 2, 4]))
    S = int(rng.integers(1, 9))
    top = int(rng.choice([bs, 3 * bs + 1, 200, 700, 1300, 2100]))
    lens = rng.integers(0 if seed % 3 == 0 else 1, top + 1, S).astype(np.int32)
    lens[int(rng.integers(0, S))] = top
    nblk_max = int((lens.max() + bs - 1) // bs)
    what = f"seed {seed}: S{S} H{H}/{hkv} D{D} bs{bs} top {top}"
    try:
        case = make_case(rng, S, H, D, lens, num_kv_heads=hkv, block_size=bs, q_row_pad=int(rng.integers(0, 3)),
                         poison_tail=bool(rng.integers(0, 2)), max_blocks=nblk_max + int(rng.integers(0, 5)),
                         kv="normal" if seed % 2 else "uniform")
        alibi = (rng.uniform(0.0, 0.3, H)).astype(np.float32) if rng.integers(0, 3) == 0 else None
        tag = (f"d{D}_" if bs == 16 else f"d{D}_bs{bs}_")
        cands = [i + 1 for i, n in enumerate(names)
                 if n.startswith(tag) and "LOADSONLY" not in n and (bs != 16 or "_bs" not in n) and
                 "_pvm" not in n and      # opt-in kernels: tests/soak/soak_gqa.py, north-star bound
                 "_x" not in n and        # split kernels (no fused-append twin, need the workspace): tests/soak/soak_split.py
                 ("_gq" not in n or (H // hkv) % int(n.split("_gq")[1].split("_")[0]) == 0)]
        vid = int(rng.choice(cands)) if (cands and rng.integers(0, 2)) else 0
        msl = int(max(lens.max(), 1)) + int(rng.integers(0, 40))
        ref = T.run_model(case, alibi=alibi)
        try:
            got = T.run_hip(case, variant=vid, max_seq_len=msl, alibi=alibi)
        except RuntimeError as e:
            if "needs num_heads" not in str(e):
                raise
            got = T.run_hip(case, variant=0, max_seq_len=msl, alibi=alibi)
        vmax = float(np.nanmax(np.abs(case["vc"].astype(np.float32))))
        T.assert_close(got, ref, what + f" v1 {names[vid - 1] if vid else 'auto'}", vmax=vmax)
        if lens.max() > 0:
            T._check_v2(case, ((msl + 511) // 512) * 512, alibi=alibi, what=what + " v2", vmax=vmax)
        if alibi is None:
            try:
                T._append_vs_two_ops(case, vid if vid and "_bs" not in names[vid - 1] else 0, seed=seed, what=what + " append")
            except RuntimeError as e:           # forced decomposition not applicable to this head count
                if "needs num_heads" not in str(e):
                    raise
                T._append_vs_two_ops(case, 0, seed=seed, what=what + " append")
            T._append_vs_two_ops(case, 0, dtype=torch.bfloat16, seed=seed, what=what + " append bf16")
        if bs >= 16:
            c8 = T._fp8_case(rng, S, H, D, np.maximum(lens, 1), bs, num_kv_heads=hkv)
            kvs = float(rng.choice([1.0, 0.5, 1.7]))
            r8 = oracle.paged_attention_v1_fp8(c8["q"], c8["kq"], c8["vq"], hkv, c8["scale"], c8["tables"], c8["lens"], bs,
                                               kv_scale=kvs, threads=4)
            T.assert_close(T._run_fp8(c8, kvs), r8, what + f" fp8 scale {kvs}", vmax=2 * kvs)
            # bfloat16 query over the same fp8 pages
            qbits = oracle.f32_to_bf16_bits(c8["qbuf"].astype(np.float32))
            qn = np.ascontiguousarray(qbits[:, : H * D].reshape(S, H, D))
            rb = oracle.paged_attention_v1_fp8(qn, c8["kq"], c8["vq"], hkv, c8["scale"], c8["tables"], c8["lens"], bs,
                                               kv_scale=kvs, threads=4, bf16=True)
            dev = T._dev()
            ob = torch.full((S, H, D), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.paged_attention_v1(ob, T._bf16_tensor(qbits, dev)[:, : H * D].view(S, H, D),
                                   torch.from_numpy(c8["kq"]).to(dev), torch.from_numpy(c8["vq"]).to(dev), hkv, c8["scale"],
                                   torch.from_numpy(c8["tables"]).to(dev), torch.from_numpy(c8["lens"]).to(dev), bs,
                                   int(c8["lens"].max()), None, "fp8", kvs)
            torch.cuda.synchronize()
            T.assert_close_bf16(ob.view(torch.int16).cpu().numpy().view(np.uint16), rb, what + f" bf16 x fp8 scale {kvs}", vmax=2 * kvs)
    except AssertionError as e:
        fails += 1
        print("FAIL", what, "|", str(e)[:300], flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("ERROR", what, "|", type(e).__name__, str(e)[:300], flush=True)
print(f"soak: {n_cases} cases from seed {first}, {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
