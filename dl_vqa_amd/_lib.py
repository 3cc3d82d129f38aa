"""ctypes binding of libvqa_hip.so (include/vqa_hip.h). No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQA_LIB", os.path.join(_HERE, "libvqa_hip.so"))   # VQA_LIB: diagnostic builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "vqa_hip.h")


def header_abi_version() -> int:
    """VQA_ABI_VERSION as include/vqa_hip.h states it: the one number the loader, the tests and build() compare
    the library's answer with (a stale build_var/*.so or a round-old library then fails here, not in a kernel)."""
    import re
    with open(HEADER_PATH) as f:
        m = re.search(r"^#define\s+VQA_ABI_VERSION\s+(\d+)", f.read(), flags=re.M)
    if not m:
        raise VqaHipError(f"{HEADER_PATH}: no VQA_ABI_VERSION")
    return int(m.group(1))


f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())
u8p = C.c_void_p
i64p = C.c_void_p
i32, i64, f32, u64, vp = C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_void_p

# name -> (restype, argtypes); mirrors include/vqa_hip.h one to one
PROTOTYPES = {
    "vqa_abi_version": (i32, []),
    "vqa_last_error": (C.c_char_p, []),
    "vqa_device_ok": (i32, []),
    "vqa_reload_knobs": (i32, []),
    "vqa_prof_arm": (i32, [i32, i32]),
    "vqa_prof_arm_mask": (i32, [C.c_uint32, i32]),
    "vqa_prof_read": (i32, [C.POINTER(i32), C.POINTER(f32)]),
    "vqa_prof_read_groups": (i32, [C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(f32), i32]),
    "vqa_gemm_workspace_bytes": (i64, [i32, i32, i32]),
    "vqa_gemm": (i32, [f32p, i64, i32, f32p, i64, i32, f32p, i64, i32, i32, i32, f32p, f32p,
                       f32p, i64, i32, i32, i32, i32, f32p, f32p, i64, i32, vp]),
    "vqa_gemm_x3": (i32, [f32p, i64, i32, f32p, i64, i32, f32p, i64, i32, i32, i32, f32p, f32p,
                       f32p, i64, i32, i32, i32, i32, f32p, f32p, i64, i32, vp]),
    "vqa_gemm_x3_workspace_bytes": (i64, [i32, i32, i32]),
    "vqa_nchw_to_nhwc4": (i32, [f32p, f32p, i32, i32, i32, i32, vp]),
    "vqa_conv_pack_weights": (i32, [f32p, f32p, f32p, i32, i32, i32, vp]),
    "vqa_conv3x3_relu_pool_fwd": (i32, [f32p, f32p, f32p, f32p, u8p, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_conv3x3_dgrad": (i32, [f32p, u8p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_conv3x3_wgrad_workspace_bytes": (i64, [i32, i32, i32, i32, i32, i32]),
    "vqa_conv3x3_wgrad": (i32, [f32p, f32p, u8p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32,
                                f32p, i64, i32, vp]),
    "vqa_convk_pack_weights": (i32, [f32p, f32p, i32, i32, i32, i32, vp]),
    "vqa_convk_unpack_wgrad": (i32, [f32p, f32p, i32, i32, i32, i32, vp]),
    "vqa_convk_im2col": (i32, [f32p, f32p, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_convk_relu_pool": (i32, [f32p, f32p, u8p, i32, i32, i32, i32, vp]),
    "vqa_convk_route": (i32, [f32p, u8p, f32p, i32, i32, i32, i32, vp]),
    "vqa_convk_col2im": (i32, [f32p, f32p, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_conv3x3_x3_supported": (i32, [i32, i32, i32, i32, i32]),
    "vqa_x3_split": (i32, [f32p, vp, vp, vp, i64, vp]),
    "vqa_x3_pack": (i32, [f32p, vp, i64, vp]),
    "vqa_x3_pack_pooled_grad_workspace_bytes": (i64, [i32]),
    "vqa_x3_pack_pooled_grad": (i32, [f32p, u8p, vp, f32p, i64, i32, f32p, i64, vp]),
    "vqa_conv3x3_relu_pool_fwd_x3": (i32, [vp, i32, vp, f32p, vp, i32, u8p, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_conv3x3_dgrad_x3": (i32, [vp, i32, u8p, vp, f32p, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_conv3x3_wgrad_x3_workspace_bytes": (i64, [i32, i32, i32, i32, i32, i32]),
    "vqa_conv3x3_wgrad_x3": (i32, [vp, i32, f32p, vp, u8p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32,
                                   f32p, i64, i32, vp]),
    "vqa_conv0_supported": (i32, [i32, i32, i32, i32, i32]),
    "vqa_conv0_relu_pool_fwd": (i32, [vp, i32, f32p, f32p, vp, i32, u8p, i32, i32, i32, i32, i32, vp]),
    "vqa_conv0_wgrad_workspace_bytes": (i64, [i32]),
    "vqa_conv0_wgrad": (i32, [vp, i32, f32p, u8p, f32p, f32p, i32, i32, i32, i32, i32, f32p, i64, vp]),
    "vqa_conv0_wgrad_bf16": (i32, [vp, i32, vp, u8p, f32p, f32p, i32, i32, i32, i32, i32, f32p, i64, vp]),
    "vqa_dropout": (i32, [f32p, f32p, i64, f32, u64, vp]),
    "vqa_dropout_add": (i32, [f32p, f32p, i64, f32, u64, vp]),
    "vqa_l2norm_fwd": (i32, [f32p, f32p, f32p, i64, i32, f32, u64, vp, i32, f32, u64, vp]),
    "vqa_l2norm_bwd": (i32, [f32p, f32p, f32p, vp, i32, i64, i32, i32, f32, u64, vp]),
    "vqa_l2norm_bwd_joined": (i32, [f32p, i64, f32p, i32, f32p, f32, u64, f32p, f32p, vp, i32, i64, i32, i32, f32, u64, vp]),
    "vqa_embed_tanh_fwd": (i32, [i64p, f32p, f32p, i32, i32, i32, i32, f32, u64, vp, vp]),
    "vqa_embed_tanh_bwd_workspace_bytes": (i64, [i32, i32, i32]),
    "vqa_embed_tanh_bwd": (i32, [i64p, f32p, f32p, f32p, i32, i32, i32, i32, f32, u64, vp, i64, vp]),
    "vqa_lstm_cell_fwd": (i32, [f32p, f32p, f32p, f32p, i64p, i32, f32p, f32p, f32p, f32p, i64, i32, i32, vp]),
    "vqa_lstm_cell_bwd": (i32, [f32p, f32p, f32p, i64p, i32, f32p, f32p, f32p, i32, i32, vp]),
    "vqa_lstm_step_supported": (i32, [i32]),
    "vqa_lstm_seq_fwd": (i32, [vp, i32, i64p, i32, i32, i32, i64, i32, vp]),
    "vqa_lstm_seq_bwd": (i32, [vp, i32, i64p, i32, i32, i32, i32, vp]),
    "vqa_lstm_graph_stats": (i32, [C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "vqa_att_score_fwd": (i32, [vp, i32, f32p, i32, f32p, f32p, i32, i32, i32, i32, f32, u64, f32p, vp]),
    "vqa_att_row_splits": (i32, [i32]),
    "vqa_att_score_bwd": (i32, [f32p, f32p, i32, vp, i32, f32p, f32p, i32, i32, i32, i32, f32, u64, i32, f32p, f32p, vp]),
    "vqa_att_apply_fwd": (i32, [f32p, f32p, f32p, f32p, i64, i32, i32, i32, i32, vp]),
    "vqa_att_apply_bwd": (i32, [f32p, i64, f32p, f32p, f32p, f32p, f32p, i32, i32, i32, i32, vp]),
    "vqa_softce_fwd_bwd": (i32, [f32p, i64, i64p, i64p, i32, i32, i32, f32, f32p, f32p, f32p, i64, vp]),
    "vqa_colsum_workspace_bytes": (i64, [i64, i32]),
    "vqa_colsum": (i32, [f32p, i64, u8p, i64, i32, f32p, i32, f32p, i64, vp]),
    "vqa_sum_bgp": (i32, [f32p, f32p, i32, i32, i32, vp]),
    "vqa_sum_parts": (i32, [f32p, f32p, i32, i32, i32, vp]),
    "vqa_relu_drop_bwd": (i32, [f32p, f32p, f32p, i64, f32, u64, vp]),
    "vqa_add2d": (i32, [f32p, i64, f32p, i64, f32p, i64, i64, i32, vp]),
    "vqa_scale_by": (i32, [f32p, i64, f32p, vp]),
    "vqa_half_to_float": (i32, [vp, f32p, i64, vp]),
    "vqa_f32_to_bf16": (i32, [f32p, vp, i64, vp]),
    "vqa_bf16_to_f32": (i32, [vp, f32p, i64, vp]),
    "vqa_dropout_to_bf16": (i32, [f32p, vp, i64, f32, u64, vp]),
    "vqa_f32_to_bf16_transpose": (i32, [f32p, vp, i32, i32, vp]),
    "vqa_gemm_bf16_workspace_bytes": (i64, [i32, i32, i32]),
    "vqa_gemm_bf16": (i32, [vp, i64, i32, vp, i64, i32, vp, i64, i32, i32, i32, i32, f32p, f32p,
                            f32p, i64, i32, i32, i32, i32, f32p, f32p, i64, i32, vp]),
    "vqa_gemm_tall_bf16_supported": (i32, [i32, i32, i32, i32, i32]),
    "vqa_gemm_tall_bf16": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, i32, f32p, i64, i32, i32, i32, i32, vp]),
    "vqa_conv_pack_weights_bf16": (i32, [f32p, vp, vp, i32, i32, i32, vp]),
    "vqa_conv3x3_relu_pool_fwd_bf16": (i32, [vp, vp, f32p, vp, i32, u8p, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_conv3x3_dgrad_bf16": (i32, [vp, u8p, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_conv3x3_wgrad_bf16_workspace_bytes": (i64, [i32, i32, i32, i32, i32, i32]),
    "vqa_conv3x3_wgrad_bf16": (i32, [vp, vp, u8p, f32p, f32p, i32, i32, i32, i32, i32, i32, i32, f32p, i64, i32, vp]),
    "vqa_pconvf_supported": (i32, [i32, i32, i32, i32, i32]),
    "vqa_pconvf_weights_bytes": (i64, [i32, i32]),
    "vqa_pconvf_pack_weights": (i32, [f32p, f32p, i32, i32, vp]),
    "vqa_pconvf_dgrad": (i32, [f32p, u8p, f32p, f32p, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_pconv_supported": (i32, [i32, i32, i32, i32, i32]),
    "vqa_pconv_weights_bytes": (i64, [i32, i32]),
    "vqa_pconv_pack_weights": (i32, [f32p, vp, vp, i32, i32, vp]),
    "vqa_pconv_fwd": (i32, [vp, vp, f32p, vp, i32, u8p, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_pconv_wgrad_supported": (i32, [i32, i32, i32, i32]),
    "vqa_pconv_wgrad_workspace_bytes": (i64, [i32, i32, i32, i32, i32]),
    "vqa_pconv_wgrad": (i32, [vp, u8p, vp, f32p, f32p, i32, i32, i32, i32, i32, f32p, i64, i32, vp]),
    "vqa_pconv_dgrad": (i32, [vp, u8p, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "vqa_adam": (i32, [f32p, f32p, f32p, f32p, i64, f32, f32, f32, f32, i32, f32, vp]),
}

K_GEMM, K_CONV_FWD, K_CONV_DGRAD, K_CONV_WGRAD = 0, 1, 2, 3
(K_L2NORM_FWD, K_L2NORM_BWD, K_ATT_SCORE_FWD, K_ATT_SCORE_BWD, K_ATT_APPLY_FWD, K_ATT_APPLY_BWD, K_ADAM, K_SOFTCE,
 K_DROPOUT, K_LSTM_SEQ, K_COUNT) = range(4, 15)


class LstmDir(C.Structure):
    """vqa_lstm_dir_t (include/vqa_hip.h): one direction of an LSTM sequence, device pointers as integers."""
    _fields_ = [("w_hh", vp), ("xg", vp), ("gates", vp), ("Hs", vp), ("Cs", vp), ("c_final", vp),
                ("dgates", vp), ("dh", vp), ("dc", vp), ("reverse", i32)]


class VqaHipError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load libvqa_hip.so and attach prototypes. Raises if the library is absent (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VqaHipError(
            f"{LIB_PATH} not found: build it with `python -m dl_vqa_amd.build` "
            "(the HIP extension is mandatory; there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    want = header_abi_version()
    if lib.vqa_abi_version() != want:
        raise VqaHipError(f"{LIB_PATH}: ABI version {lib.vqa_abi_version()}, include/vqa_hip.h says {want} "
                          "(stale library: rebuild with `python -m dl_vqa_amd.build`)")
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    """Call a status-returning entry point; raise VqaHipError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise VqaHipError(f"{name} failed ({rc}): {lib.vqa_last_error().decode()}")
