"""ctypes binding of the C ABI in include/dpc_hip.h.

The product path loads ``dpc_amd/libdpc_hip.so`` (gfx950 code objects, built by
``make`` / ``__graft_entry__.build()``) and raises if it is missing -- there is no
CPU or eager-PyTorch fallback.  ``load_emulator()`` exists for the CPU test tier
only (``tests/``): it loads the same kernel sources compiled against the host-side
SIMT simulator in ``tests/simt_emu`` and is never used by the engine on its own.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "libdpc_hip.so")
EMU_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tests", "simt_emu", "libdpc_emu.so")

F32, BF16 = 0, 1
ABI_VERSION = 1


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dt}")


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype_in", "dtype_out", "mode", "N", "RT", "RH", "RW", "ST", "SH", "SW", "Ci", "src_ld",
        "Co", "ldw", "ldo", "KT", "KH", "KW", "st", "sh", "sw", "pt", "ph", "pw")]


class GruChainDesc(C.Structure):
    """struct dpc_gru_chain_desc (include/dpc_hip.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("dtype", "M", "D", "SQ", "P", "n_agg", "n_steps", "reserved")] +
                [("p_drop", C.c_float), ("reserved2", C.c_uint32), ("seed", C.c_uint64)] +
                [(n, C.c_void_p) for n in ("step_dev", "drop_masks", "packed", "bias_u", "bias_r", "bias_o", "bias_1", "bias_2",
                                           "X_all", "H_all", "HR_all", "U_all", "R_all", "O_all", "P1_all", "pred",
                                           "d_pred", "G_all", "dP1", "dP2", "d_x", "ws", "d_hlast")])


class LcHeadDesc(C.Structure):
    """struct dpc_lc_head_desc (include/dpc_hip.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("dtype", "B", "SQ", "D", "num_class", "train")] +
                [("p_drop", C.c_float), ("momentum", C.c_float), ("eps", C.c_float), ("reserved", C.c_uint32), ("seed", C.c_uint64)] +
                [(n, C.c_void_p) for n in ("step_dev", "drop_mask", "h_last", "bn_weight", "bn_bias", "bn_running_mean", "bn_running_var",
                                           "bn_num_batches", "fc_weight", "fc_bias", "target", "ctx", "xhat", "bn_out", "y", "stat",
                                           "logits", "dlogits", "row_ws", "result", "g_fc_weight", "g_fc_bias", "g_bn_weight",
                                           "g_bn_bias", "dctx", "d_hlast")])


class ConvEpilogue(C.Structure):
    """struct dpc_conv_epilogue (include/dpc_hip.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("addend", "addend_mask", "bn_raw", "bn_mask", "bn_mean", "bn_invstd", "stats")]


class Resample(C.Structure):
    """struct dpc_resample (include/dpc_hip.h)"""
    _fields_ = [("xb", C.c_void_p), ("xk", C.c_void_p), ("yb", C.c_void_p), ("yk", C.c_void_p), ("ksx", C.c_int32), ("ksy", C.c_int32)]


class PackEntry(C.Structure):
    """struct dpc_pack_entry (include/dpc_hip.h)"""
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("d0", C.c_int32), ("d1", C.c_int32), ("d2", C.c_int32), ("block0", C.c_int32),
                ("s0", C.c_int64), ("s1", C.c_int64), ("s2", C.c_int64)]


class Copy2dEntry(C.Structure):
    """struct dpc_copy2d_entry (include/dpc_hip.h)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_ld", C.c_int64), ("dst_ld", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("block0", C.c_int32), ("pad", C.c_int32)]


class DpcError(RuntimeError):
    pass


_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
_SIGS = {
    "dpc_abi_version": [],
    "dpc_conv_stats_rows": [C.POINTER(ConvDesc)],
    "dpc_conv_igemm": [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    "dpc_conv_igemm_ex": [C.POINTER(ConvDesc), _vp, _vp, _vp, C.POINTER(ConvEpilogue), _vp],
    "dpc_gemm_nt_splitk": [_i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, C.POINTER(_i32), _vp],
    "dpc_gemm_tn_splitk": [_i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, C.POINTER(_i32), _vp],
    "dpc_conv_wgrad": [C.POINTER(ConvDesc), _vp, _vp, _i32, _vp, C.POINTER(_i32), _vp],
    "dpc_conv_plan": [C.POINTER(ConvDesc), _i32, _i32, _i32, C.c_char_p, _i32],
    "dpc_last_kernel": [C.c_char_p, _i32],
    "dpc_set_reserved_cus": [_i32],
    "dpc_set_f32_matmul": [_i32],
    "dpc_diag_squat": [_i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp],
    "dpc_pack3d": [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp],
    "dpc_pack3d_multi": [_vp, _i32, _i32, _i32, _vp],
    "dpc_reduce_unpack": [_vp, _i32, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp],
    "dpc_transpose2d_bf16x2": [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp],
    "dpc_transpose2d": [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    "dpc_pack_input_s2d": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dpc_pack_stem_weight": [_vp, _vp, _i32, _i32, _vp],
    "dpc_unpack_stem_wgrad": [_vp, _i32, _vp, _i32, _vp],
    "dpc_bn_finalize": [_vp, _i32, _i32, _f64, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp],
    "dpc_bn_apply": [_vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "dpc_bn_bwd_reduce": [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _i32, _vp, C.POINTER(_i32), _vp],
    "dpc_bn_bwd_finalize": [_vp, _i32, _i32, _f64, _vp, _vp, _vp, _vp],
    "dpc_bn_bwd_apply": [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "dpc_bn_relu_maxpool_fwd": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "dpc_maxpool_bwd": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "dpc_tpool_split_fwd": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "dpc_tpool_split_bwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "dpc_colsum": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i64, _vp],
    "dpc_pool_bn_bwd_reduce": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "dpc_pooled_bn_bwd_reduce": [_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "dpc_pool_bn_bwd_apply": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "dpc_mask_gen": [_vp, _i32, _i32, _i32, _vp],
    "dpc_ce_topk": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp],
    "dpc_ce_topk_bf16": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp],
    "dpc_step_advance": [_vp, _vp, _f64, _f64, _vp],
    "dpc_adam_dev": [_vp, _vp, _vp, _vp, _i64, _f32, _f64, _f64, _f32, _f32, _vp, _f32, _vp],
    "dpc_counter_advance": [_vp, _vp],
    "dpc_copy2d_f32": [_vp, _i64, _vp, _i64, _i32, _i32, _vp],
    "dpc_copy2d_multi": [_vp, _i32, _i32, _vp],
    "dpc_dropout_mask": [_vp, _i64, _f32, C.c_uint64, _vp, _vp],
    "dpc_gru_pack": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "dpc_gru_chain_fwd": [C.POINTER(GruChainDesc), _vp],
    "dpc_gru_chain_bwd": [C.POINTER(GruChainDesc), _vp],
    "dpc_score_ws_floats": [_i32, _i32, C.POINTER(_i64), C.POINTER(_i64)],
    "dpc_score_fwd": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "dpc_score_bwd": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp],
    "dpc_ce_finalize": [_vp, _i32, _vp, _vp],
    "dpc_bn_finalize_running": [_vp, _i32, _i32, _f64, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp],
    "dpc_bn_eval_coeffs": [_vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp, _vp],
    "dpc_relu_tpool_fwd": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "dpc_relu_tpool_bwd": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "dpc_lc_head_fwd": [C.POINTER(LcHeadDesc), _vp],
    "dpc_lc_head_bwd": [C.POINTER(LcHeadDesc), _vp],
    "dpc_frames_to_input": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32,
                            C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp, _i32, _vp],
    "dpc_frames_to_input_ex": [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(Resample), _vp, _vp, _vp,
                               C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp, _i32, _vp],
    "dpc_stem_wgrad_fused": [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i32), _vp],
    "dpc_adam": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp],
}


class Lib:
    """Loaded kernel library + thin checked wrappers (return codes -> DpcError)."""

    def __init__(self, path: str, kind: str):
        if not os.path.exists(path):
            raise DpcError(
                f"{path} not found: build it with `make` (or __graft_entry__.build()); "
                "dpc_amd has no CPU/eager fallback")
        self.kind = kind
        self.path = path
        self.c = C.CDLL(path)
        self._bound = {}
        if self.call("dpc_abi_version") != ABI_VERSION:
            raise DpcError("ABI version mismatch")

    def _fn(self, name: str):
        fn = self._bound.get(name)
        if fn is None:
            fn = getattr(self.c, name)  # AttributeError => ABI mismatch, loud by design
            fn.argtypes = _SIGS[name]
            fn.restype = C.c_int
            self._bound[name] = fn
        return fn

    def missing_symbols(self):
        return [n for n in _SIGS if not hasattr(self.c, n)]

    # -- stream handle for the device the tensors live on
    def stream(self) -> C.c_void_p:
        if self.kind == "hip":
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return C.c_void_p(0)

    def call(self, name: str, *args) -> int:
        conv = [C.c_void_p(a.data_ptr()) if isinstance(a, torch.Tensor) else a for a in args]
        rc = self._fn(name)(*conv)
        if rc < 0:
            raise DpcError(f"{name} failed with code {rc}")
        return rc


PLAN_IGEMM, PLAN_WGRAD, PLAN_ADDEND, PLAN_STATS, PLAN_ADDEND_MASK, PLAN_BNRED = 0, 1, 1, 2, 4, 8


def conv_plan(lib: "Lib", desc: ConvDesc, op: int = PLAN_IGEMM, addend: bool = False, stats: bool = False, dy_ld: int = 0,
              addend_mask: bool = False, bnred: bool = False) -> str:
    """name of the kernel dpc_conv_igemm(_ex) / dpc_conv_wgrad would launch for `desc` (include/dpc_hip.h: dpc_conv_plan);
    "" when the combination is not supported (dpc_conv_igemm_ex returns DPC_ERR_UNSUPPORTED)"""
    buf = C.create_string_buffer(192)
    flags = ((PLAN_ADDEND if addend else 0) | (PLAN_STATS if stats else 0) | (PLAN_ADDEND_MASK if addend_mask else 0) |
             (PLAN_BNRED if bnred else 0))
    rc = lib._fn("dpc_conv_plan")(C.byref(desc), op, flags, dy_ld or desc.Co, buf, 192)
    if rc == -3:
        return ""
    if rc < 0:
        raise DpcError(f"dpc_conv_plan failed with code {rc}")
    return buf.value.decode()


def last_kernel(lib: "Lib") -> str:
    buf = C.create_string_buffer(192)
    lib.call("dpc_last_kernel", buf, 192)
    return buf.value.decode()


_HIP: Optional[Lib] = None
_EMU: Optional[Lib] = None


def load_hip() -> Lib:
    global _HIP
    if _HIP is None:
        _HIP = Lib(HIP_LIB_PATH, "hip")
    return _HIP


def load_emulator() -> Lib:
    """CPU functional simulator of the kernels -- tests only."""
    global _EMU
    if _EMU is None:
        _EMU = Lib(EMU_LIB_PATH, "emu")
    return _EMU


def lib_for(device: torch.device, emu: Optional[Lib] = None) -> Lib:
    if device.type == "cuda":
        return load_hip()
    if emu is not None and emu.kind == "emu":
        return emu
    raise DpcError("dpc_amd runs on MI355X only: tensors must live on a cuda (HIP) device")
