"""The GPT-2 block's linear layers for the decode harness, on libvmi_gpt2_layer.so (include/vmi_gpt2_layer.h).

Counterpart of the torch modules around the reference's call pair — ln_1 -> c_attn, c_proj + residual,
ln_2 -> c_fc -> GELU -> c_proj + residual (vllmini/model/gpt2.py:14-15, 117-128, 130-135, GPT2Block.forward): one
hand-written gfx950 kernel family (csrc/gpt2_layer.hip) that runs each linear layer of a decode step — LayerNorm in front,
bias / GELU / residual add behind — as ONE launch.  `linear()` is the only operation; it has no torch fallback: a missing
library or an unsupported shape raises.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Tuple

import torch

from . import build as _build

ABI_VERSION = 1
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, EPI_BIAS_KV_CACHE = 0, 1, 2, 3

_vp, _i32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
SIGNATURES = {
    "vmi_gpt2_linear_f16": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _i64, _vp, _i64,
                                           _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vmi_gpt2_linear_qkv_cache_f16": (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _i64, _i32, _i32, _i32,
                                                     _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i32, _vp]),
    "vmi_gpt2_embed_f16": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vmi_gpt2_argmax_f16": (ctypes.c_int, [_vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "vmi_gpt2_sample_top_k_f16": (ctypes.c_int, [_vp, _i64, _i32, _i32, _i32, ctypes.c_float, _vp, _vp, _i32, _vp]),
    "vmi_gpt2_linear_kernel_name": (ctypes.c_char_p, [_i32, _i32, _i32, _i32, _i32]),
    "vmi_gpt2_layer_last_error": (ctypes.c_char_p, []),
    "vmi_gpt2_layer_abi_version": (_i32, []),
    "vmi_gpt2_layer_target_arch": (ctypes.c_char_p, []),
}

_lock = threading.Lock()
_lib = None


class LayerLibraryError(RuntimeError):
    """libvmi_gpt2_layer.so is missing / stale / does not match the header."""


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = _build.LAYER_LIB_PATH
                if not os.path.exists(path):
                    raise LayerLibraryError(f"{path} not found: build it with `python -m vllmini_amd.build` "
                                            "(or __graft_entry__.build()).  The native layers have no torch fallback; "
                                            "GPT2PagedDecoder(native_layers=False) is the torch-module harness.")
                try:
                    lib = ctypes.CDLL(path)
                except OSError as e:
                    raise LayerLibraryError(f"cannot load {path}: {e}") from e
                for name, (restype, argtypes) in SIGNATURES.items():
                    try:
                        fn = getattr(lib, name)
                    except AttributeError as e:
                        raise LayerLibraryError(f"{path} does not export {name}") from e
                    fn.restype, fn.argtypes = restype, argtypes
                if lib.vmi_gpt2_layer_abi_version() != ABI_VERSION:
                    raise LayerLibraryError(f"{path}: ABI {lib.vmi_gpt2_layer_abi_version()}, expected {ABI_VERSION}")
                _lib = lib
    return _lib


def supports(M: int, N: int, K: int) -> bool:
    """Whether linear() takes this shape (K % 32 == 0, N % 16 == 0, K <= 4608)."""
    return load().vmi_gpt2_linear_kernel_name(M, N, K, 0, 0) is not None


def kernel_name(M: int, N: int, K: int, ln: bool = False, epilogue: int = EPI_BIAS) -> Optional[str]:
    s = load().vmi_gpt2_linear_kernel_name(M, N, K, int(ln), epilogue)
    return s.decode() if s else None


class PackedWeight:
    """An nn.Linear weight [N, K] re-laid as the MFMA tiles the kernels read (include/vmi_gpt2_layer.h, w_layout 1):
    tile[s][t][kc * 16 + r][e] = w[16 s + r][32 t + 8 kc + e].  Weights are static, so the harness packs each once."""

    def __init__(self, weight: torch.Tensor):
        N, K = weight.shape
        if N % 16 or K % 32 or weight.dtype != torch.float16:
            raise RuntimeError("pack_weight: half [N, K] with N % 16 == 0 and K % 32 == 0")
        self.shape = (N, K)
        self.tiles = weight.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def pack_weight(weight: torch.Tensor) -> PackedWeight:
    return PackedWeight(weight)


def linear(x: torch.Tensor, weight, bias: Optional[torch.Tensor] = None, *,
           ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None, gelu: bool = False,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = epilogue(LN?(x) @ weight.T + bias): x [M, K] half (last dim contiguous), weight [N, K] half contiguous
    (nn.Linear layout) or its pack_weight() form, ln = (gamma, beta, eps) for a LayerNorm over K in front, gelu / residual [M, N] behind (not both).
    `out` may be `residual` itself.  Launches on torch's current stream; never synchronises."""
    lib = load()
    packed = isinstance(weight, PackedWeight)
    wshape = weight.shape
    if packed:
        weight = weight.tiles
    if x.device.type != "cuda":
        raise RuntimeError("gpt2_layer.linear: there is no CPU path")
    if x.dtype != torch.float16 or weight.dtype != torch.float16:
        raise RuntimeError("gpt2_layer.linear: float16 tensors only")
    if gelu and residual is not None:
        raise RuntimeError("gpt2_layer.linear: gelu and residual are alternative epilogues")
    if x.dim() != 2 or len(wshape) != 2 or x.shape[1] != wshape[1] or x.stride(1) != 1 or not weight.is_contiguous():
        raise RuntimeError("gpt2_layer.linear: x [M, K] with unit stride in K, weight [N, K] contiguous")
    M, K = x.shape
    N = wshape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    if out.shape != (M, N) or out.stride(1) != 1 or out.dtype != torch.float16:
        raise RuntimeError("gpt2_layer.linear: out [M, N] half with unit stride in N")
    if out.data_ptr() == x.data_ptr():
        # a workgroup stores its columns of `out` while others still read the rows of `x`: only the RESIDUAL may alias out
        raise RuntimeError("gpt2_layer.linear: out must not alias x (out may alias the residual)")
    for t, what in ((bias, "bias"), (ln[0] if ln else None, "ln gamma"), (ln[1] if ln else None, "ln beta")):
        if t is not None and (t.dtype != torch.float16 or not t.is_contiguous() or t.device != x.device):
            raise RuntimeError(f"gpt2_layer.linear: {what} must be a contiguous half tensor on x's device")
    if residual is not None and (residual.shape != (M, N) or residual.stride(1) != 1 or residual.dtype != torch.float16):
        raise RuntimeError("gpt2_layer.linear: residual [M, N] half with unit stride in N")
    epi = EPI_BIAS_GELU if gelu else EPI_BIAS_RESIDUAL if residual is not None else EPI_BIAS
    stream = torch.cuda.current_stream(x.device).cuda_stream
    rc = lib.vmi_gpt2_linear_f16(x.data_ptr(), x.stride(0), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                 ln[0].data_ptr() if ln else None, ln[1].data_ptr() if ln else None,
                                 float(ln[2]) if ln else 0.0, residual.data_ptr() if residual is not None else None,
                                 residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0),
                                 M, N, K, epi, int(packed), x.device.index or 0, stream)
    if rc != 0:
        raise RuntimeError(lib.vmi_gpt2_layer_last_error().decode("utf-8", "replace"))
    return out


def linear_qkv_cache(x: torch.Tensor, weight, bias: Optional[torch.Tensor], key_cache: torch.Tensor, value_cache: torch.Tensor,
                     slot_mapping: torch.Tensor, num_heads: int, *, ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The q / k / v projection with cache_ops.reshape_and_cache's copy done by the producing kernel: returns qkv [M, 3E] and
    writes every row's k and v into the float16 paged caches at slot_mapping[m] (rows with a negative slot are skipped) — the
    same bytes the reference's call (gpt2.py:44, 81-89) writes from the k / v views of qkv, one launch earlier."""
    lib = load()
    packed = isinstance(weight, PackedWeight)
    wshape = weight.shape
    if packed:
        weight = weight.tiles
    if x.device.type != "cuda":
        raise RuntimeError("gpt2_layer.linear_qkv_cache: there is no CPU path")
    M, K = x.shape
    N = wshape[0]
    E = N // 3
    if N != 3 * E or E % num_heads or x.dtype != torch.float16 or weight.dtype != torch.float16 or x.stride(1) != 1 \
            or wshape[1] != K or not weight.is_contiguous():
        raise RuntimeError("gpt2_layer.linear_qkv_cache: x [M, K] half, weight [3E, K] half contiguous, E = heads * head_size")
    D = E // num_heads
    if key_cache.dtype != torch.float16 or value_cache.dtype != torch.float16 or key_cache.dim() != 5 or value_cache.dim() != 4 \
            or key_cache.shape[4] != 8 or key_cache.shape[1] != num_heads or key_cache.shape[2] * 8 != D \
            or value_cache.shape[1:3] != (num_heads, D) or key_cache.shape[3] != value_cache.shape[3] \
            or not key_cache[0].is_contiguous() or not value_cache[0].is_contiguous() \
            or key_cache.stride(0) != value_cache.stride(0):
        raise RuntimeError("gpt2_layer.linear_qkv_cache: float16 caches in the reference layout [NB, H, D/8, bs, 8] / [NB, H, D, bs]")
    if slot_mapping.dtype != torch.int64 or slot_mapping.shape != (M,) or not slot_mapping.is_contiguous():
        raise RuntimeError("gpt2_layer.linear_qkv_cache: slot_mapping int64 [M]")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    if out.shape != (M, N) or out.stride(1) != 1 or out.dtype != torch.float16:
        raise RuntimeError("gpt2_layer.linear_qkv_cache: out [M, 3E] half with unit stride in N")
    for t, what in ((bias, "bias"), (ln[0] if ln else None, "ln gamma"), (ln[1] if ln else None, "ln beta")):
        if t is not None and (t.dtype != torch.float16 or not t.is_contiguous() or t.device != x.device):
            raise RuntimeError(f"gpt2_layer.linear_qkv_cache: {what} must be a contiguous half tensor on x's device")
    stream = torch.cuda.current_stream(x.device).cuda_stream
    rc = lib.vmi_gpt2_linear_qkv_cache_f16(x.data_ptr(), x.stride(0), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                           ln[0].data_ptr() if ln else None, ln[1].data_ptr() if ln else None,
                                           float(ln[2]) if ln else 0.0, out.data_ptr(), out.stride(0), M, K, int(packed),
                                           key_cache.data_ptr(), value_cache.data_ptr(), slot_mapping.data_ptr(), num_heads, D,
                                           key_cache.shape[3], key_cache.stride(0), key_cache.stride(1), x.device.index or 0, stream)
    if rc != 0:
        raise RuntimeError(lib.vmi_gpt2_layer_last_error().decode("utf-8", "replace"))
    return out


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(load().vmi_gpt2_layer_last_error().decode("utf-8", "replace"))


def embed(input_ids: torch.Tensor, position_ids: torch.Tensor, wte: torch.Tensor, wpe: torch.Tensor,
          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """wte[input_ids] + wpe[position_ids] in one launch (gpt2.py: inputs_embeds + position_embeds): int64 ids [T], half tables."""
    lib = load()
    T, E = input_ids.shape[0], wte.shape[1]
    if input_ids.device.type != "cuda":
        raise RuntimeError("gpt2_layer.embed: there is no CPU path")
    if input_ids.dtype != torch.int64 or position_ids.dtype != torch.int64 or position_ids.shape != (T,) \
            or not input_ids.is_contiguous() or not position_ids.is_contiguous() or wte.dtype != torch.float16 \
            or wpe.dtype != torch.float16 or wpe.shape[1] != E or not wte.is_contiguous() or not wpe.is_contiguous():
        raise RuntimeError("gpt2_layer.embed: int64 ids / positions [T], contiguous half tables [V, E] / [P, E]")
    if out is None:
        out = torch.empty((T, E), dtype=torch.float16, device=wte.device)
    _check(lib.vmi_gpt2_embed_f16(input_ids.data_ptr(), position_ids.data_ptr(), wte.data_ptr(), wpe.data_ptr(), out.data_ptr(),
                                  T, E, wte.device.index or 0, torch.cuda.current_stream(wte.device).cuda_stream))
    return out


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.argmax(logits, -1) for half logits [B, V] (first maximum on ties), one workgroup per row."""
    lib = load()
    if logits.device.type != "cuda":
        raise RuntimeError("gpt2_layer.argmax: there is no CPU path")
    if logits.dim() != 2 or logits.dtype != torch.float16 or logits.stride(1) != 1:
        raise RuntimeError("gpt2_layer.argmax: half logits [B, V] with unit stride in V")
    B, V = logits.shape
    if out is None:
        out = torch.empty((B,), dtype=torch.int64, device=logits.device)
    _check(lib.vmi_gpt2_argmax_f16(logits.data_ptr(), logits.stride(0), B, V, out.data_ptr(), logits.device.index or 0,
                                   torch.cuda.current_stream(logits.device).cuda_stream))
    return out


def sample_top_k(logits: torch.Tensor, top_k: int = 50, temperature: float = 1.0,
                 generator: Optional[torch.Generator] = None, uniform: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Scheduler.sample_next_token (scheduler.py:144-153) for a batch in one launch: logits / temperature -> the top_k largest ->
    softmax -> one draw per row, by inverse CDF over the top_k in descending order at `uniform` (default: torch.rand from
    `generator`).  Same distribution as topk + softmax + multinomial, not the same stream of draws.  int64 [B]."""
    lib = load()
    if logits.device.type != "cuda":
        raise RuntimeError("gpt2_layer.sample_top_k: there is no CPU path")
    if logits.dim() != 2 or logits.dtype != torch.float16 or logits.stride(1) != 1:
        raise RuntimeError("gpt2_layer.sample_top_k: half logits [B, V] with unit stride in V")
    B, V = logits.shape
    if uniform is None:
        uniform = torch.rand(B, device=logits.device, generator=generator)
    if uniform.dtype != torch.float32 or uniform.shape != (B,) or not uniform.is_contiguous():
        raise RuntimeError("gpt2_layer.sample_top_k: uniform float32 [B]")
    out = torch.empty((B,), dtype=torch.int64, device=logits.device)
    _check(lib.vmi_gpt2_sample_top_k_f16(logits.data_ptr(), logits.stride(0), B, V, min(top_k, V), float(temperature),
                                         uniform.data_ptr(), out.data_ptr(), logits.device.index or 0,
                                         torch.cuda.current_stream(logits.device).cuda_stream))
    return out

