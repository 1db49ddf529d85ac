"""Every kernel the PRODUCT library can launch is checked against the CPU oracle in this run.

`ops.variant_names()` / `ops.variant_names_v2()` list every (kernel, work decomposition) the library holds; a variant id
reaches each of them through the C-ABI (`vmi_paged_attention_v1_f16_variant` and friends).  This test walks ALL of them —
no filter by name (the split kernels, "_x<waves>", get the wrapper's workspace like any call through ops) — builds a case of the shape the name describes (element type, KV-cache type, head size, block size,
query heads per KV head), runs the variant and compares with the kernel model.  A name the walker cannot parse, a variant
that is refused for every shape tried, or a library that still carries a diagnostic kernel fails the test: nothing that
ships is untested or wrong by design (the diagnostic build, -DVMI_DIAG, is where "loads only" and the LDS-staging
experiment live; tests/test_abi.py checks the symbol table).

Nothing here reads /root/reference.
"""
from __future__ import annotations

import re

import numpy as np
import pytest
import torch

import oracle
from helpers import make_case
from test_parity_gpu import _bf16_tensor, _dev, _e5m2_case, _fp8_case, assert_close, assert_close_bf16

pytestmark = pytest.mark.gpu

NAME = re.compile(r"^(?P<bf>bf16_)?(?P<kv>fp8e5m2_|fp8_)?(?P<v2>v2_)?(?P<q>q_)?d(?P<D>\d+)(?:_bs(?P<bs>\d+))?"
                  r"(?:_mh(?P<mh>\d+))?(?:_gq(?P<gq>\d+))?(?:_h(?P<h>\d+))?(?:_w(?P<w>\d+))?(?:_x(?P<x>\d+))?(?:_s(?P<s>\d+)q(?P<uq>\d+)(?P<km>m)?)?"
                  r"(?:_u(?P<u>\d+)(?:a(?P<a>\d+))?)?(?:_nt(?P<nt>\d))?(?P<pvm>_pvm)?(?P<lock>_lock)?$")


class Cases:
    """One seeded case (+ device tensors + oracle result) per (element type, cache type, D, bs, H, H_kv, v2)."""

    def __init__(self):
        self.cache = {}

    def get(self, bf, kv, D, bs, H, hkv, v2, nseq=9):
        key = (bf, kv, D, bs, H, hkv, v2, nseq)
        if key in self.cache:
            return self.cache[key]
        dev = _dev()
        rng = np.random.default_rng(hash(key) % (2 ** 32))
        # (nseq < 9: the widest split kernels — every workgroup of a launch must be resident — get the first few of a reordering)
        lens = [1, bs, bs + 1, 100, 333, 47, 700, 0, 2 * bs - 1] if nseq == 9 else [700, 1, 333, 0, bs + 1, 100, 47, bs][:nseq]
        msl = 1024 if v2 else max(lens)
        scale = 0.75 if kv else 1.0                   # kv_scale (the balanced fp8 kernels: 1.0, set by the caller)
        if kv == "fp8_":
            case = _fp8_case(rng, len(lens), H, D, lens, bs, num_kv_heads=hkv)
        elif kv == "fp8e5m2_":
            case = _e5m2_case(rng, len(lens), H, D, lens, bs, num_kv_heads=hkv)
        else:
            case = make_case(rng, len(lens), H, D, lens, num_kv_heads=hkv, block_size=bs, q_row_pad=1,
                             poison_tail=not bf)
        S = len(lens)
        t = {"tab": torch.from_numpy(case["tables"]).to(dev), "len": torch.from_numpy(case["lens"]).to(dev)}
        if bf:
            qbits = oracle.f32_to_bf16_bits(np.nan_to_num(case["qbuf"].astype(np.float32)))
            case["q_np"] = np.ascontiguousarray(qbits[:, : H * D].reshape(S, H, D))
            t["q"] = _bf16_tensor(qbits, dev)[:, : H * D].view(S, H, D)
        else:
            case["q_np"] = case["q"]
            t["q"] = torch.from_numpy(case["qbuf"]).to(dev)[:, : H * D].view(S, H, D)
        if kv:
            t["kc"], t["vc"] = torch.from_numpy(case["kq"]).to(dev), torch.from_numpy(case["vq"]).to(dev)
        elif bf:
            case["kc_np"], case["vc_np"] = (oracle.f32_to_bf16_bits(case[k].astype(np.float32)) for k in ("kc", "vc"))
            t["kc"], t["vc"] = _bf16_tensor(case["kc_np"], dev), _bf16_tensor(case["vc_np"], dev)
        else:
            t["kc"], t["vc"] = torch.from_numpy(case["kc"]).to(dev), torch.from_numpy(case["vc"]).to(dev)
        case.update(dev=t, msl=msl, kv_scale=scale, refs={})
        self.cache[key] = case
        return case

    @staticmethod
    def ref(case, bf, kv, bs, hkv, v2, kv_scale):
        k = (v2, kv_scale)
        if k not in case["refs"]:
            a = (case["q_np"],)
            common = (hkv, case["scale"], case["tables"], case["lens"], bs)
            if kv:
                kw = dict(kv_scale=kv_scale, bf16=bool(bf), e5m2=kv == "fp8e5m2_")
                r = (oracle.paged_attention_v2_fp8(*a, case["kq"], case["vq"], *common, case["msl"], **kw)[0] if v2 else
                     oracle.paged_attention_v1_fp8(*a, case["kq"], case["vq"], *common, threads=8, **kw))
            else:
                kc, vc = (case["kc_np"], case["vc_np"]) if bf else (case["kc"], case["vc"])
                r = (oracle.paged_attention_v2(*a, kc, vc, *common, case["msl"], bf16=bool(bf))[0] if v2 else
                     oracle.paged_attention_v1(*a, kc, vc, *common, threads=8, bf16=bool(bf)))
            case["refs"][k] = r
        return case["refs"][k]


def _launch(case, vid, v2, kvd, kv_scale, bf):
    from vllmini_amd import ops

    t = case["dev"]
    S, H, D = t["q"].shape
    out = torch.full((S, H, D), float("nan"), dtype=t["q"].dtype, device=t["q"].device)
    tail = (case["num_kv_heads"], case["scale"], t["tab"], t["len"], case["bs"], case["msl"], None, kvd, kv_scale,
            0, 0, 1, 1, 0)
    if v2:
        P = (case["msl"] + 511) // 512
        es = torch.full((S, H, P), float("nan"), dtype=torch.float32, device=out.device)
        ml = torch.full((S, H, P), float("nan"), dtype=torch.float32, device=out.device)
        tmp = torch.full((S, H, P, D), float("nan"), dtype=out.dtype, device=out.device)
        ops.paged_attention_v2(out, es, ml, tmp, t["q"], t["kc"], t["vc"], *tail, _variant=vid)
    else:
        ops.paged_attention_v1(out, t["q"], t["kc"], t["vc"], *tail, _variant=vid)
    torch.cuda.synchronize()
    return out.view(torch.int16).cpu().numpy().view(np.uint16) if bf else out.cpu().numpy()


def _walk(names, v2, cases, only=None):
    checked, refused_everywhere, unparsed = 0, [], []
    for vid, name in enumerate(names, start=1):
        if only is not None and name not in only:
            continue
        m = NAME.match(name)
        if not m or bool(m["v2"]) != v2:
            unparsed.append(name)
            continue
        bf, kv, D, bs = m["bf"], m["kv"], int(m["D"]), int(m["bs"] or 16)
        g = int(m["gq"] or 1)
        kvd = {"fp8_": "fp8", "fp8e5m2_": "fp8_e5m2", None: "auto"}[kv]
        kv_scale = 1.0 if (not kv or m["q"]) else 0.75      # balanced fp8 kernels are built for kv_scale 1
        done = False
        errors = []
        for hkv in (4, 6, 8, 3):                             # H_kv a multiple of what a workgroup takes
            H = hkv * g
            nseq = 9
            if m["x"]:      # split kernel: sequences x heads x (waves / 4) workgroups, all resident (6 per CU at head size 64, 3 at 128)
                nseq = max(1, min(9, (1536 if D == 64 else 768) * 4 // (int(m["x"]) * H)))
            case = cases.get(bf, kv, D, bs, H, hkv, v2, nseq)
            try:
                got = _launch(case, vid, v2, kvd, kv_scale, bf)
            except RuntimeError as e:
                errors.append(str(e))
                continue
            ref = cases.ref(case, bf, kv, bs, hkv, v2, kv_scale)
            what = f"{name} (H{H}/{hkv})"
            vmax = 2 * kv_scale if kv else 1.0
            if bf:
                if m["pvm"]:
                    d = np.abs(oracle.bf16_bits_to_f32(got).astype(np.float64) - oracle.bf16_bits_to_f32(ref))
                    assert np.isfinite(d).all() and d.max() <= 2.0 ** -6 * vmax, f"{what}: {d.max():.3e}"
                else:
                    assert_close_bf16(got, ref, what, vmax=vmax)
            else:
                assert_close(got, ref, what, vmax=vmax, tight=not m["pvm"])
            done = True
            break
        if done:
            checked += 1
        else:
            refused_everywhere.append((name, errors[-1][:120]))
    return checked, refused_everywhere, unparsed


# (the second run of each walk, marked `extras`, is over libvmi_paged_attention_extras.so — the product's kernels again plus
#  the bfloat16 / E5M2 / block-sparse menus; tests/conftest.py switches the library for the marker)
BOTH_LIBRARIES = pytest.mark.parametrize("extras", [False, pytest.param(True, marks=pytest.mark.extras)], ids=["product", "extras"])


@BOTH_LIBRARIES
def test_every_v1_variant_of_the_product_library_is_oracle_checked(extras):
    from vllmini_amd import _lib, ops

    assert _lib.load().vmi_is_diag_build() == 0 and _lib.load().vmi_has_extras() == int(extras)
    names = ops.variant_names()
    assert len(names) == _lib.load().vmi_paged_attention_v1_variant_count() >= (390 if extras else 195)
    assert not [n for n in names if "LOADSONLY" in n or n.startswith("stage_")]
    assert bool([n for n in names if n.startswith("bf16_") or "e5m2" in n]) == extras
    checked, refused, unparsed = _walk(names, False, Cases())
    assert not unparsed, unparsed[:5]
    assert not refused, refused[:5]
    assert checked == len(names)


@BOTH_LIBRARIES
def test_every_v2_variant_of_the_product_library_is_oracle_checked(extras):
    from vllmini_amd import _lib, ops

    names = ops.variant_names_v2()
    assert len(names) == _lib.load().vmi_paged_attention_v2_variant_count() >= (210 if extras else 88)
    assert bool([n for n in names if n.startswith("bf16_") or "e5m2" in n]) == extras
    checked, refused, unparsed = _walk(names, True, Cases())
    assert not unparsed, unparsed[:5]
    assert not refused, refused[:5]
    assert checked == len(names)


def test_the_diagnostic_library_s_comparison_kernels_are_oracle_checked_too():
    """Round 6 moved the work decompositions no pick rule returns out of the product menu into the diagnostic library
    (pa_table_core.inc / pa_table_fp8.inc / pa_split.hip, #ifdef VMI_DIAG): the offline sweeps' comparison points.  They are
    still kernels of this build — every one of them is run against the oracle here (the "loads only" bandwidth probes and the
    LDS-staging experiment are the diagnostic library's only kernels that are wrong / unproven by design)."""
    from vllmini_amd import _lib, ops

    with _lib.use_extras():
        shipped = set(ops.variant_names())
    with _lib.use_diag():
        names = ops.variant_names()
        only = {n for n in names if n not in shipped and "LOADSONLY" not in n and not n.startswith("stage_")}
        assert 50 <= len(only) <= 90, len(only)
        checked, refused, unparsed = _walk(names, False, Cases(), only=only)
    assert not unparsed and not refused, (unparsed[:5], refused[:5])
    assert checked == len(only)


def test_unknown_variant_ids_are_rejected():
    from vllmini_amd import ops

    case = Cases().get(None, None, 64, 16, 4, 4, False)
    n = len(ops.variant_names())
    for vid in (n + 1, n + 1000, -3):
        with pytest.raises(RuntimeError, match="variant"):
            _launch(case, vid, False, "auto", 1.0, None)


@BOTH_LIBRARIES
def test_name_pattern_covers_every_name_without_a_gpu_assumption(extras):
    """(runs on the GPU box with the others; the pattern itself needs no device)"""
    from vllmini_amd import ops

    for n in ops.variant_names() + ops.variant_names_v2():
        assert NAME.match(n), n
