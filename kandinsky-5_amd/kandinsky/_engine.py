"""ctypes binding of libk5.so (C ABI: include/k5.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing a
function from here raises RuntimeError.  torch is used only for device memory and streams; every
call passes raw device pointers (`tensor.data_ptr()`) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("K5_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libk5.so"))

K5_OK = 0
ABI_VERSION = 8          # include/k5.h K5_ABI_VERSION
K5_F32, K5_BF16, K5_F16 = 0, 1, 2
EPI_BIAS, EPI_BIAS_M, EPI_GELU, EPI_GATE = 0, 1, 2, 3

_DT = {torch.float32: K5_F32, torch.bfloat16: K5_BF16, torch.float16: K5_F16}


class DitConfig(C.Structure):
    _fields_ = [("in_visual_dim", C.c_int), ("in_text_dim", C.c_int), ("in_text_dim2", C.c_int),
                ("time_dim", C.c_int), ("out_visual_dim", C.c_int), ("patch_size", C.c_int * 3),
                ("model_dim", C.c_int), ("ff_dim", C.c_int), ("num_text_blocks", C.c_int),
                ("num_visual_blocks", C.c_int), ("axes_dims", C.c_int * 3), ("visual_cond", C.c_int)]


class TextCond(C.Structure):
    _fields_ = [("text_embed", C.c_void_p), ("pooled_embed", C.c_void_p), ("text_dtype", C.c_int),
                ("text_len", C.c_int), ("text_rope_pos", C.POINTER(C.c_int32))]


class ForwardArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int), ("x_channels", C.c_int),
                ("cond", TextCond), ("time", C.c_float), ("pos_t", C.POINTER(C.c_int32)),
                ("pos_h", C.POINTER(C.c_int32)), ("pos_w", C.POINTER(C.c_int32)), ("scale_factor", C.c_float * 3),
                ("attention_type", C.c_int), ("nabla_P", C.c_float), ("nabla_wT", C.c_int), ("nabla_wH", C.c_int),
                ("nabla_wW", C.c_int)]


class SampleArgs(C.Structure):
    _fields_ = [("fwd", ForwardArgs), ("null_cond", TextCond), ("latent", C.c_void_p), ("num_steps", C.c_int),
                ("sigmas", C.POINTER(C.c_float)), ("guidance_weight", C.c_float)]


class VaeConfig(C.Structure):
    _fields_ = [("latent_channels", C.c_int), ("out_channels", C.c_int), ("block_out_channels", C.c_int * 4),
                ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int)]


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); every symbol declared in include/k5.h
_P, _I, _F, _I64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
SYMBOLS = {
    "k5_abi_version": (_I, []),
    "k5_last_error": (C.c_char_p, []),
    "k5_gemm_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "k5_gemm_bf16_variant": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _I]),
    "k5_gemm_bf16_f32out": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "k5_causal_softmax_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "k5_vae_attention512_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "k5_attention_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k5_attention_state_size": (_I64, [_I, _I]),
    "k5_attention_bf16_range": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _I, _P, _I, _P]),
    "k5_gemm_fp8": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "k5_quant_rows_fp8": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "k5_attention_bf16_prescaled": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "k5_attention_bf16_prescaled_auto": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "k5_attention_flags": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "k5_attention_flags_rows": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "k5_attention_bf16_prescaled_rows": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "k5_attention_bf16_prescaled_rows_pass": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _I, _I, _P, _P]),
    "k5_attention_bf16_prescaled_qnorm_pass": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _I, _I, _P, _P]),
    "k5_rmsnorm_rope_centre_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P]),
    "k5_attention_flags_rows_centred": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "k5_attention_bf16_prescaled_rows_centred": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "k5_attention_flags_rows_anchored": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "k5_attention_row_anchor": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "k5_attention_bf16_prescaled_rows_anchored": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "k5_rmsnorm_rope_stats_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "k5_loopback_create": (_I, [_I, C.POINTER(_P)]),
    "k5_loopback_destroy": (None, [_P]),
    "k5_dit_comm_init_loopback": (_I, [_P, _P, _I]),
    "k5_dit_cfg_pair_init": (_I, [_P, C.c_char_p, _I, _P]),
    "k5_dit_cfg_pair_init_loopback": (_I, [_P, _P, _I]),
    "k5_dit_comm_init_ipc": (_I, [_P, C.c_char_p, _I, _I]),
    "k5_dit_cfg_pair_init_ipc": (_I, [_P, C.c_char_p, _I]),
    "k5_dit_cfg_branch": (_I, [_P]),
    "k5_dit_set_option": (_I, [_P, C.c_char_p, _I]),
    "k5_dit_get_option": (_I, [_P, C.c_char_p, C.POINTER(_I)]),
    "k5_dit_sp_schedule": (_I, [_P, C.c_char_p, _I]),
    "k5_sp_pick_schedule": (_I, [C.POINTER(C.c_float), _I, _I, C.POINTER(_I), C.POINTER(C.c_float)]),
    "k5_dit_attn_variant_counts": (_I, [_P, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), _I]),
    "k5_dit_nabla_block_counts": (_I, [_P, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "k5_dit_set_nabla_tap": (_I, [_P, _P, C.c_longlong]),
    "k5_dit_nabla_tap_count": (_I, [_P, C.POINTER(C.c_longlong)]),
    "k5_dit_nabla_executed_blocks": (_I, [_P, C.POINTER(C.c_longlong)]),
    "k5_attention_balance_size": (_I64, [_I, _I]),
    "k5_attention_bf16_balanced": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "k5_nabla_workspace_size": (_I64, [_I, _I]),
    "k5_nabla_select_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "k5_attention_nabla_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "k5_nabla_mask_u8": (_I, [_P, _I, _I, _P, _P]),
    "k5_nabla_select_rect_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "k5_attention_nabla_rect_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _I64, _P]),
    "k5_nabla_select_rect_local_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _I, _I, _P]),
    "k5_attention_nabla_rect_prescaled_pass": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I64, _P, _P, _I, _P, _P]),
    "k5_nabla_mask_rect_u8": (_I, [_P, _I, _I, _I, _P, _P]),
    "k5_attention_bf16_bounded": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "k5_ln_modulate_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "k5_rmsnorm_rope_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "k5_gate_sum_bf16": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "k5_gemv_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "k5_time_features_f32": (_I, [_F, _P, _I, _P]),
    "k5_ln_affine_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "k5_rope_table_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _P, _P]),
    "k5_patchify_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "k5_unpatchify_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "k5_cfg_euler": (_I, [_P, _P, _P, _F, _F, _I64, _P]),
    "k5_dit_create": (_I, [C.POINTER(DitConfig), C.POINTER(_P)]),
    "k5_dit_destroy": (None, [_P]),
    "k5_dit_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_I64), _I]),
    "k5_dit_finalize": (_I, [_P]),
    "k5_dit_missing_keys": (_I, [_P]),
    "k5_dit_forward": (_I, [_P, C.POINTER(ForwardArgs), _P, _P]),
    "k5_sample": (_I, [_P, C.POINTER(SampleArgs), _P]),
    "k5_comm_unique_id": (_I, [C.c_char_p, _P]),
    "k5_dit_comm_init": (_I, [_P, C.c_char_p, _I, _I, _P]),
    "k5_conv3d_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "k5_conv3d_strided_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "k5_groupnorm_workspace_size": (_I64, [_I, _I]),
    "k5_groupnorm_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P]),
    "k5_conv3d_stats_size": (_I64, [_I, _I]),
    "k5_conv3d_bf16_stats": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "k5_groupnorm_bf16_quads": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P, _P]),
    "k5_vae_path_counts": (_I, [_P, C.POINTER(C.c_longlong), _I]),
    "k5_vae_create": (_I, [C.POINTER(VaeConfig), C.POINTER(_P)]),
    "k5_vae_destroy": (None, [_P]),
    "k5_vae_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(_I64), _I]),
    "k5_vae_finalize": (_I, [_P]),
    "k5_vae_decode_tile": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "k5_vae_encode_tile": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "k5_vae_has_encoder": (_I, [_P]),
    "k5_blend_bf16": (_I, [_P, _P, _I64, _I, _I, _I64, _I, _P]),
    "k5_blend_place_bf16": (_I, [_P, _I64, _I, _P, _I64, _P, _I64, _I64, _I64, _I, _I, _P]),
    "k5_frames_to_uint8": (_I, [_P, _P, _I64, _P]),
    "k5_vae_decode_tile_strided": (_I, [_P, _P, _I64, _I, _I, _I, _P, _P]),
    "k5_dit_set_graph": (_I, [_P, _I]),
    "k5_dit_set_fp8": (_I, [_P, _I]),
    "k5_dit_set_magcache": (_I, [_P, C.POINTER(C.c_double), _I, _I, C.c_double, _I, C.c_double]),
    "k5_dit_magcache_calls": (_I, [_P, _I, _I]),
    "k5_dit_magcache_state": (_I, [_P, C.POINTER(_I), C.POINTER(_I64), C.POINTER(_I64)]),
    "k5_dit_set_profiling": (_I, [_P, _I]),
    "k5_dit_get_profile": (_I, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_I64)]),
    "k5_dit_reset_profile": (_I, [_P]),
}


def lib() -> C.CDLL:
    """Load libk5.so once; raise loudly if it is not there (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libk5.so not found at {LIB_PATH}: build it with `python kandinsky-5_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise RuntimeError(f"failed to load {LIB_PATH}: {e}") from e
    try:
        L.k5_abi_version.restype = C.c_int
        have = L.k5_abi_version()
    except AttributeError as e:
        raise RuntimeError(f"{LIB_PATH} is not a libk5.so (no k5_abi_version)") from e
    if have != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has C ABI version {have}, this host binding needs {ABI_VERSION}: rebuild it with "
                           "`python kandinsky-5_amd/build.py`")
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError => symbol missing: loud
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class LoopbackGroup:
    """k5_loopback: `world` engine handles of this process act as the ranks of one sequence-parallel group on one GPU
    (each rank driven by its own host thread).  Test infrastructure for the multi-GPU code path."""

    def __init__(self, world: int):
        h = C.c_void_p()
        check(lib().k5_loopback_create(int(world), C.byref(h)), "k5_loopback_create")
        self.handle, self.world = h, int(world)

    def __del__(self):
        try:
            if self.handle:
                lib().k5_loopback_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def last_error() -> str:
    return (lib().k5_last_error() or b"").decode()


def check(status: int, what: str = "libk5"):
    if status != K5_OK:
        raise RuntimeError(f"{what} failed with status {status}: {last_error()}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def k5_dtype(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}")


def i32_array(values):
    vals = [int(v) for v in values]
    return (C.c_int32 * len(vals))(*vals)


# ------------------------------------------------------------------------------------------
# thin op wrappers over torch device tensors (parity tests, host glue)
# ------------------------------------------------------------------------------------------
def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libk5 kernels run on the GPU only: got a CPU tensor (no CPU fallback)")


def gemm(a, w, bias=None, epilogue=EPI_BIAS, resid=None, gate=None, out=None, kernel=0, token_tile=0):
    """out[M,N] = a[M,K] @ w[N,K]^T (+bias) with fused epilogue. a, w bf16; bias/gate fp32.
    kernel / token_tile != 0: the named kernel / tile height (k5_gemm_bf16_variant; tests and A/B tools)."""
    _need_cuda(a, w, bias, resid, gate)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    if kernel or token_tile:
        check(lib().k5_gemm_bf16_variant(ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K, a.stride(0), w.stride(0), out.stride(0),
                                         epilogue, ptr(resid), 0 if resid is None else resid.stride(0), ptr(gate),
                                         stream_ptr(a.device), int(kernel), int(token_tile)), "k5_gemm_bf16_variant")
        return out
    check(lib().k5_gemm_bf16(ptr(a), ptr(w), ptr(bias), ptr(out), M, N, K, a.stride(0), w.stride(0), out.stride(0),
                             epilogue, ptr(resid), 0 if resid is None else resid.stride(0), ptr(gate),
                             stream_ptr(a.device)), "k5_gemm_bf16")
    return out


def attention(q, k, vt, num_heads, q_len=None, kv_len=None, out=None, score_bound=None):
    """q [Sq, >=H*64] , k [Sk, >=H*64], vt [H*64, >=Sk] bf16 -> out [Sq, H*64]."""
    _need_cuda(q, k, vt)
    q_len = q.shape[0] if q_len is None else q_len
    kv_len = k.shape[0] if kv_len is None else kv_len
    if out is None:
        out = torch.empty(q_len, num_heads * 64, dtype=torch.bfloat16, device=q.device)
    if score_bound is not None:
        check(lib().k5_attention_bf16_bounded(ptr(q), ptr(k), ptr(vt), ptr(out), num_heads, q_len, kv_len, q.stride(0),
                                              k.stride(0), vt.stride(0), out.stride(0), float(score_bound),
                                              stream_ptr(q.device)), "k5_attention_bf16_bounded")
        return out
    check(lib().k5_attention_bf16(ptr(q), ptr(k), ptr(vt), ptr(out), num_heads, q_len, kv_len, q.stride(0), k.stride(0),
                                  vt.stride(0), out.stride(0), stream_ptr(q.device)), "k5_attention_bf16")
    return out


def nabla_select(q, k, num_heads, grid, window, P):
    """q, k [N, >=H*64] bf16 (fractal order); grid = (T, Hb, Wb); window = (wT, wH, wW).  Returns the workspace."""
    _need_cuda(q, k)
    N, nb = q.shape[0], q.shape[0] // 64
    ws = torch.empty(lib().k5_nabla_workspace_size(num_heads, nb), dtype=torch.uint8, device=q.device)
    check(lib().k5_nabla_select_bf16(ptr(q), ptr(k), q.stride(0), k.stride(0), num_heads, N, grid[0], grid[1], grid[2],
                                     window[0], window[1], window[2], float(P), ptr(ws), stream_ptr(q.device)),
          "k5_nabla_select_bf16")
    return ws


def nabla_mask(ws, num_heads, nb):
    out = torch.empty(num_heads, nb, nb, dtype=torch.uint8, device=ws.device)
    check(lib().k5_nabla_mask_u8(ptr(ws), num_heads, nb, ptr(out), stream_ptr(ws.device)), "k5_nabla_mask_u8")
    return out.bool()


def attention_nabla(q, k, vt, num_heads, ws, score_bound=0.0, out=None):
    _need_cuda(q, k, vt, ws)
    N = q.shape[0]
    if out is None:
        out = torch.empty(N, num_heads * 64, dtype=torch.bfloat16, device=q.device)
    check(lib().k5_attention_nabla_bf16(ptr(q), ptr(k), ptr(vt), ptr(out), num_heads, N, q.stride(0), k.stride(0),
                                        vt.stride(0), out.stride(0), float(score_bound), ptr(ws), stream_ptr(q.device)),
          "k5_attention_nabla_bf16")
    return out


def ln_modulate(x, scale, shift):
    _need_cuda(x, scale, shift)
    out = torch.empty_like(x)
    check(lib().k5_ln_modulate_bf16(ptr(x), ptr(scale), ptr(shift), ptr(out), x.shape[0], x.shape[1], x.stride(0),
                                    out.stride(0), stream_ptr(x.device)), "k5_ln_modulate_bf16")
    return out


def rmsnorm_rope_(x, weight, cos=None, sin=None, heads=None, heads_per_weight=None, rope_heads=None):
    _need_cuda(x, weight, cos, sin)
    heads = x.shape[1] // 64 if heads is None else heads
    check(lib().k5_rmsnorm_rope_bf16(ptr(x), ptr(weight), ptr(cos), ptr(sin), x.shape[0], heads, x.stride(0),
                                     heads if heads_per_weight is None else heads_per_weight,
                                     heads if rope_heads is None else rope_heads, stream_ptr(x.device)),
          "k5_rmsnorm_rope_bf16")
    return x


def gate_sum(x, y, gate):
    _need_cuda(x, y, gate)
    out = torch.empty_like(x)
    check(lib().k5_gate_sum_bf16(ptr(x), ptr(y), ptr(gate), ptr(out), x.shape[0], x.shape[1], stream_ptr(x.device)),
          "k5_gate_sum_bf16")
    return out


def gemv_f32(x, w, b=None, silu_in=False, add=None):
    _need_cuda(x, w, b, add)
    y = torch.empty(w.shape[0], dtype=torch.float32, device=x.device)
    check(lib().k5_gemv_f32(ptr(x), ptr(w), ptr(b), ptr(y), w.shape[0], w.shape[1], int(silu_in), ptr(add),
                            stream_ptr(x.device)), "k5_gemv_f32")
    return y


def cfg_euler_(img, v_cond, v_uncond, w, dt):
    _need_cuda(img, v_cond, v_uncond)
    check(lib().k5_cfg_euler(ptr(img), ptr(v_cond), ptr(v_uncond), float(w), float(dt), img.numel(),
                             stream_ptr(img.device)), "k5_cfg_euler")
    return img
