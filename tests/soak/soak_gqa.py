"""Soak run for the grouped-query kernels, default and opt-in (P.V on the matrix cores): random head groupings, ragged
lengths (empty sequences, partial pairs / quads of blocks, idle waves), ALiBi, poisoned tails, fp16 / bf16 / fp8 pages,
forced and automatic kernels, fused-append twins.  `PYTHONPATH=.:tests python tests/soak/soak_gqa.py [n_cases] [first_seed]`.
Bounds: the tests' tight one for the default kernels, the north-star 1e-3 * max(1, |v|max) for `_pvm` kernels."""
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
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
names = ops.variant_names()
fails, ran_pvm, t0 = 0, 0, time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    H, hkv = [(8, 2), (16, 4), (16, 2), (32, 4), (8, 1), (28, 4), (24, 8), (4, 2), (64, 8)][int(rng.integers(0, 9))]
    D = int(rng.choice([64, 128]))
    kind = ["f16", "bf16", "fp8"][int(rng.integers(0, 3))]
    S = int(rng.integers(1, 7))
    top = int(rng.choice([16, 17, 49, 200, 700, 1300]))
    lens = rng.integers(0 if seed % 3 == 0 else 1, top + 1, S).astype(np.int32)
    lens[int(rng.integers(0, S))] = top
    fast = bool(rng.integers(0, 2))
    qpk = H // hkv
    prefix = {"f16": f"d{D}_gq", "bf16": f"bf16_d{D}_gq", "fp8": f"fp8_d{D}_bs16_gq"}[kind]
    cands = [i + 1 for i, n in enumerate(names) if n.startswith(prefix) and T._gq_ok(n, qpk) and "_x" not in n]   # (split kernels: soak_split.py)
    vid = int(rng.choice(cands)) if (cands and rng.integers(0, 2)) else 0
    alibi = (rng.uniform(0.0, 0.3, H)).astype(np.float32) if rng.integers(0, 3) == 0 else None
    what = f"seed {seed}: {kind} S{S} H{H}/{hkv} D{D} top {top} fast={fast} alibi={alibi is not None}"
    ops.set_pv_mfma(fast)
    try:
        msl = int(max(lens.max(), 1))
        auto = names[ops.pick_variant(S, H, D, msl, 16, bf16=kind == "bf16", fp8=kind == "fp8", num_kv_heads=hkv) - 1]
        name = names[vid - 1] if vid else auto
        loose = "_pvm" in name
        ran_pvm += loose
        what += f" [{name}]"
        if kind == "fp8":
            case = T._fp8_case(rng, S, H, D, lens, 16, num_kv_heads=hkv)
            ks = float(rng.choice([1.0, 0.6, 2.0]))
            ref = oracle.paged_attention_v1_fp8(case["q"], case["kq"], case["vq"], hkv, case["scale"], case["tables"],
                                                case["lens"], 16, kv_scale=ks, alibi_slopes=alibi, threads=8)
            try:
                got = T._run_fp8(case, ks, variant=vid, alibi=alibi)
            except RuntimeError as e:
                if "needs num_heads" not in str(e):
                    raise
                got = T._run_fp8(case, ks, alibi=alibi)
                loose = "_pvm" in auto
            T.assert_close(got, ref, what, vmax=2 * ks, tight=not loose)
            continue
        case = make_case(rng, S, H, D, lens, num_kv_heads=hkv, q_row_pad=int(rng.integers(0, 3)),
                         poison_tail=bool(rng.integers(0, 2)), kv="normal" if seed % 2 else "uniform")
        vmax = float(np.nanmax(np.abs(case["vc"].astype(np.float32))))
        if kind == "bf16":
            cb = T._to_bf16_case(case)
            ref = oracle.paged_attention_v1(cb["q"], cb["kc"], cb["vc"], hkv, cb["scale"], cb["tables"], cb["lens"], 16,
                                            threads=8, bf16=True)
            try:
                bits = T.run_hip_bf16(cb, variant=vid)
            except RuntimeError as e:
                if "needs num_heads" not in str(e):
                    raise
                bits = T.run_hip_bf16(cb)
                loose = "_pvm" in auto
            if loose:
                r64 = oracle.bf16_bits_to_f32(ref).astype(np.float64)
                d = np.abs(oracle.bf16_bits_to_f32(bits).astype(np.float64) - r64)
                ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(r64), 2.0 ** -126))) - 7)       # bf16 ulp of the result
                assert np.isfinite(d).all() and (d <= np.maximum(2 * ulp, max(2.0 ** -7, 2.0 ** -8 * vmax))).all(), \
                    f"{what}: {d.max():.3e}"
            else:
                T.assert_close_bf16(bits, ref, what, vmax=vmax)
            continue
        ref = T.run_model(case, alibi=alibi)
        try:
            got = T.run_hip(case, variant=vid, alibi=alibi)
        except RuntimeError as e:
            if "needs num_heads" not in str(e):
                raise
            vid = 0
            got = T.run_hip(case, alibi=alibi)
            loose = "_pvm" in auto
        T.assert_close(got, ref, what, vmax=vmax, tight=not loose)
        if alibi is None:
            T._append_vs_two_ops(case, vid, seed=seed, what=what + " append")
    except AssertionError as e:
        fails += 1
        print("FAIL", str(e)[:300], flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("ERROR", what, repr(e)[:300], flush=True)
ops.set_pv_mfma(False)
print(f"{n_cases} cases ({ran_pvm} on _pvm kernels), {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
