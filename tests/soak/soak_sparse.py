"""Soak run for block-sparse attention (blocksparse_vert_stride > 1): random patterns (sparse block sizes that do and do not
divide the cache block, zero / large local windows, both sliding directions, tp_rank), every head x block size, fp16 /
bf16, v1 / v2, ragged lengths, grouped KV heads, ALiBi.  PYTHONPATH=.:tests python tests/soak/soak_sparse.py [n] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, "tests")
import oracle  # noqa: E402  (checker)
import test_parity_gpu as T  # noqa: E402
from helpers import make_case  # noqa: E402
from vllmini_amd import _lib  # noqa: E402

_lib.use_extras().__enter__()   # block-sparse attention lives in libvmi_paged_attention_extras.so (build.py --extras)

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
fails, t0 = 0, time.time()
for seed in range(first, first + n_cases):
    rng = np.random.default_rng(seed)
    D = int(rng.choice([64, 80, 96, 112, 128, 192, 256]))
    bs = int(rng.choice([8, 16, 32]))
    hkv = int(rng.choice([1, 2, 3]))
    H = hkv * int(rng.choice([1, 2, 4]))
    S = int(rng.integers(1, 7))
    top = int(rng.choice([bs, 3 * bs + 1, 200, 700, 1300]))
    lens = rng.integers(0 if seed % 3 == 0 else 1, top + 1, S).astype(np.int32)
    lens[int(rng.integers(0, S))] = top
    sparse = (int(rng.integers(0, 6)), int(rng.integers(2, 9)), int(rng.choice([8, 16, 24, 32, 64, 100, 128])),
              int(rng.integers(-3, 4)))
    tp = int(rng.integers(0, 3))
    bf16 = bool(rng.integers(0, 2))
    v2 = bool(rng.integers(0, 2)) and top > 0
    alibi = rng.uniform(0.0, 0.3, H).astype(np.float32) if rng.integers(0, 3) == 0 else None
    what = f"seed {seed}: S{S} H{H}/{hkv} D{D} bs{bs} top {top} sparse {sparse} tp {tp} bf16={bf16} v2={v2} alibi={alibi is not None}"
    try:
        case = make_case(rng, S, H, D, lens, num_kv_heads=hkv, block_size=bs, q_row_pad=int(rng.integers(0, 3)),
                         poison_tail=bool(rng.integers(0, 2)), kv="normal" if seed % 2 else "uniform")
        vmax = float(np.nanmax(np.abs(case["vc"].astype(np.float32))))
        c = T._to_bf16_case(case) if bf16 else case
        a = (c["q"], c["kc"], c["vc"], hkv, c["scale"], c["tables"], c["lens"], bs)
        msl = ((int(max(lens.max(), 1)) + 511) // 512) * 512
        if v2:
            ref = oracle.paged_attention_v2(*a, msl, alibi_slopes=alibi, blocksparse=sparse, tp_rank=tp, bf16=bf16)[0]
        else:
            ref = oracle.paged_attention_v1(*a, alibi_slopes=alibi, blocksparse=sparse, tp_rank=tp, threads=8, bf16=bf16)
        got = T._run_sparse(c, sparse, tp, alibi=alibi, bf16=bf16, v2_msl=msl if v2 else 0)
        if bf16:
            T.assert_close_bf16(got, ref, what, vmax=vmax)
        else:
            T.assert_close(got, ref, what, vmax=vmax)
    except AssertionError as e:
        fails += 1
        print("FAIL", str(e)[:300], flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("ERROR", what, repr(e)[:300], flush=True)
print(f"{n_cases} block-sparse cases, {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
