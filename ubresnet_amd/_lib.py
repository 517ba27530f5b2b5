"""ctypes binding of libubresnet_hip.so (the C ABI in include/ubresnet_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (callers in the reference catch ``Exception`` and break their loop,
training/train_ubresnet2018_wlarcv2.py:230-239, so errors must be exceptions, never aborts).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UBR_LIB", os.path.join(HERE, "libubresnet_hip.so"))     # UBR_LIB: A/B builds of the same ABI (tools)

F32, BF16, F16 = 0, 1, 2
MAX_TAPS = 64
STAT_SLOTS = 32      # UBR_STAT_SLOTS
RED_SLOTS = 8        # UBR_RED_SLOTS
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_CPU = {F32: 4, BF16: 8, F16: 8}


def dtype_id(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise RuntimeError("ubresnet_amd: unsupported compute dtype %s" % dt)


def chans_per_unit(dt: torch.dtype) -> int:
    return _CPU[dtype_id(dt)]


class Tensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sn", C.c_int64), ("sy", C.c_int64), ("sx", C.c_int64)]


class ChanAffine(C.Structure):
    _fields_ = [("sub", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("lo", C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("x", Tensor),
        ("xf", ChanAffine),
        ("w", C.c_void_p),
        ("Cout", C.c_int32), ("Cout_pad", C.c_int32),
        ("ntaps", C.c_int32),
        ("dy", C.c_int8 * MAX_TAPS), ("dx", C.c_int8 * MAX_TAPS),
        ("wt", C.c_uint8 * MAX_TAPS),
        ("S", C.c_int32), ("iy0", C.c_int32), ("ix0", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32),
        ("y", Tensor),
        ("addend", Tensor),
        ("bias", C.c_void_p),
        ("stats", C.c_void_p),
        ("epilogue", C.c_int32),
        ("tile_hint", C.c_int32),
        ("act", C.c_int32),
        ("pad_", C.c_int32),
        ("addend_mask", C.c_void_p),
        ("bnb_c", Tensor),
        ("bnb_mean", C.c_void_p), ("bnb_scale", C.c_void_p), ("bnb_shift", C.c_void_p), ("bnb_invstd", C.c_void_p),
        ("nphase", C.c_int32), ("phase_tap0", C.c_int32 * 4), ("phase_ntaps", C.c_int32 * 4), ("pad3_", C.c_int32),
        ("phase_yoff", C.c_int64 * 4), ("phase_aoff", C.c_int64 * 4),
        ("stats_slots", C.c_int32), ("pad2_", C.c_int32),
    ]


class BnFwdFin(C.Structure):
    """ubr_bn_fwd_fin"""
    _fields_ = [
        ("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
        ("momentum", C.c_float), ("eps", C.c_float),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
    ]


REDUCE_BATCH = 16         # UBR_REDUCE_BATCH


class WgradReduceItem(C.Structure):
    _fields_ = [
        ("slabs", C.c_void_p), ("dst", C.c_void_p),
        ("nsplit", C.c_int32), ("ntaps", C.c_int32), ("Cout_pad", C.c_int32), ("Cin", C.c_int32),
        ("Cout_valid", C.c_int32), ("Cin_valid", C.c_int32), ("accumulate", C.c_int32), ("pad_", C.c_int32),
        ("sm", C.c_int64), ("sk", C.c_int64),
        ("tapidx", C.c_int32 * 64),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("x", Tensor),
        ("xf", ChanAffine),
        ("GH", C.c_int32), ("GW", C.c_int32), ("Cout", C.c_int32),
        ("g", Tensor),
        ("ntaps", C.c_int32),
        ("dy", C.c_int8 * MAX_TAPS), ("dx", C.c_int8 * MAX_TAPS),
        ("S", C.c_int32), ("iy0", C.c_int32), ("ix0", C.c_int32),
        ("slabs", C.c_void_p),
        ("nsplit", C.c_int32),
        ("exclusive", C.c_int32),
    ]


# every symbol include/ubresnet_hip.h declares (tests check that all of them are exported)
SYMBOLS = [
    "ubr_conv", "ubr_conv_last_config", "ubr_conv_last_kernel", "ubr_pack_weights", "ubr_pack_weights_batched", "ubr_bn_fold_batched", "ubr_wgrad_plan", "ubr_wgrad", "ubr_wgrad_last_config", "ubr_wgrad_last_pc", "ubr_wgrad_reduce", "ubr_wgrad_reduce_batched",
    "ubr_stem_forward", "ubr_stem_wgrad", "ubr_stem_wgrad_workspace", "ubr_stem_expand",
    "ubr_bn_finalize", "ubr_bn_eval_affine", "ubr_bn_bwd_reduce", "ubr_bn_bwd_finalize", "ubr_bn_bwd_apply",
    "ubr_block_tail_fwd", "ubr_block_tail_bwd_reduce", "ubr_block_tail_bwd_apply",
    "ubr_block_tail_fwd_masked", "ubr_block_tail_bwd_reduce_masked", "ubr_block_tail_bwd_apply_masked",
    "ubr_bn_bwd_apply_fin", "ubr_block_tail_bwd_apply_fin", "ubr_block_tail_fwd_fin",
    "ubr_maxpool_fwd", "ubr_maxpool_bwd",
    "ubr_logsoftmax_bwd", "ubr_pixelwise_nll_fwd", "ubr_pixelwise_nll_bwd", "ubr_confusion",
    "ubr_channel_sum", "ubr_cast_f64_to_f32", "ubr_zero", "ubr_adam_step", "ubr_sgd_step", "ubr_crop_tiles", "ubr_stitch_tiles", "ubr_last_error", "ubr_version",
    "ubr_tape_create", "ubr_tape_destroy", "ubr_tape_begin", "ubr_tape_end", "ubr_tape_pause", "ubr_tape_fork", "ubr_tape_mark",
    "ubr_tape_wait_mark", "ubr_tape_size", "ubr_tape_replay", "ubr_tape_set_label", "ubr_tape_replay_timed",
]

_lib = None
_lock = threading.Lock()
vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double


def _declare(lib):
    lib.ubr_last_error.restype = C.c_char_p
    lib.ubr_version.restype = C.c_int
    lib.ubr_stem_wgrad_workspace.restype = C.c_int64
    lib.ubr_stem_wgrad_workspace.argtypes = [i32] * 5
    lib.ubr_conv.argtypes = [C.POINTER(ConvDesc), vp]
    lib.ubr_conv_last_config.argtypes = [C.POINTER(C.c_int)] * 4
    lib.ubr_conv_last_kernel.argtypes = [C.c_char_p, i32]
    lib.ubr_conv_last_config.restype = None
    lib.ubr_wgrad_last_config.argtypes = [C.POINTER(C.c_int)] * 5
    lib.ubr_wgrad_last_config.restype = None
    lib.ubr_wgrad_last_pc.argtypes = []
    lib.ubr_wgrad_last_pc.restype = C.c_int
    lib.ubr_pack_weights.argtypes = [i32, vp, vp, i32, i32, i32, i32, i64, i64, i32, C.POINTER(C.c_int32), vp]
    lib.ubr_pack_weights_batched.argtypes = [i32, vp, i32, vp]
    lib.ubr_bn_fold_batched.argtypes = [vp, i32, vp]
    lib.ubr_wgrad_plan.argtypes = [C.POINTER(WgradDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.ubr_wgrad.argtypes = [C.POINTER(WgradDesc), vp]
    lib.ubr_wgrad_reduce.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i64, i64, C.POINTER(C.c_int32), i32, vp]
    lib.ubr_wgrad_reduce_batched.argtypes = [C.POINTER(WgradReduceItem), i32, vp]
    lib.ubr_stem_forward.argtypes = [i32, vp, i32, i32, i32, i32, vp, vp, i32, Tensor, vp, vp]
    lib.ubr_stem_wgrad.argtypes = [i32, vp, i32, i32, i32, i32, Tensor, i32, vp, i64, vp, vp, i32, vp]
    lib.ubr_stem_expand.argtypes = [i32, vp, i32, i32, i32, i32, vp, i64, vp]
    lib.ubr_bn_finalize.argtypes = [vp, f64, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp]
    lib.ubr_bn_eval_affine.argtypes = [vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp]
    lib.ubr_bn_bwd_reduce.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, vp, vp]
    lib.ubr_bn_bwd_finalize.argtypes = [vp, f64, vp, vp, i32, vp, vp, i32, vp, vp, vp]
    lib.ubr_bn_bwd_apply.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, vp, vp, vp, i64, vp]
    lib.ubr_bn_bwd_apply_fin.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i32, vp, f64, vp, vp, vp, i64, vp]
    lib.ubr_block_tail_bwd_apply_fin.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp,
                                                 vp, i64, vp, vp, vp, vp, vp, vp, f64, vp, i64, vp, i64, vp]
    lib.ubr_block_tail_fwd_fin.argtypes = [i32, i64, i32, vp, i64, C.POINTER(BnFwdFin), vp, i64, C.POINTER(BnFwdFin), f64, vp, i64, vp, vp]
    lib.ubr_block_tail_fwd.argtypes = [i32, i64, i32, vp, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    lib.ubr_block_tail_fwd_masked.argtypes = [i32, i64, i32, vp, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp]
    lib.ubr_block_tail_bwd_reduce_masked.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]
    lib.ubr_block_tail_bwd_apply_masked.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp,
                                                    vp, i64, vp, i64, vp]
    lib.ubr_block_tail_bwd_reduce.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp,
                                              vp, i64, vp, vp, vp, vp, vp]
    lib.ubr_block_tail_bwd_apply.argtypes = [i32, i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp,
                                             vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]
    lib.ubr_maxpool_fwd.argtypes = [i32, i32, i32, i32, i32, i32, vp, i64, ChanAffine, vp, i64, vp, i64, vp, vp]
    lib.ubr_maxpool_bwd.argtypes = [i32, i32, i32, i32, i32, i32, vp, i64, ChanAffine, vp, i64, vp, i64, vp, i64, vp, vp]
    lib.ubr_logsoftmax_bwd.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, i64, vp]
    lib.ubr_pixelwise_nll_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, vp, vp, vp]
    lib.ubr_pixelwise_nll_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, vp, vp]
    lib.ubr_confusion.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.ubr_channel_sum.argtypes = [i32, i64, i32, vp, i64, vp, vp]
    lib.ubr_cast_f64_to_f32.argtypes = [vp, i32, i32, vp, i32, f64, i32, vp]
    lib.ubr_zero.argtypes = [vp, i64, vp]
    lib.ubr_adam_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, f32, vp]
    lib.ubr_sgd_step.argtypes = [vp, vp, vp, i64, f32, f32, f32, f32, i32, i32, f32, vp]
    lib.ubr_crop_tiles.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_int32), i32, i32, i32, vp, vp]
    lib.ubr_stitch_tiles.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_int32), i32, vp, i32, i32, i32, vp]
    lib.ubr_tape_create.argtypes = []
    lib.ubr_tape_create.restype = vp
    lib.ubr_tape_destroy.argtypes = [vp]
    lib.ubr_tape_destroy.restype = None
    lib.ubr_tape_begin.argtypes = [vp, i32, C.POINTER(vp)]
    lib.ubr_tape_end.argtypes = [vp]
    lib.ubr_tape_pause.argtypes = [vp, i32]
    lib.ubr_tape_fork.argtypes = [vp, i32, i32]
    lib.ubr_tape_mark.argtypes = [vp, i32]
    lib.ubr_tape_wait_mark.argtypes = [vp, i32, vp]
    lib.ubr_tape_size.argtypes = [vp]
    lib.ubr_tape_replay.argtypes = [vp, i32, C.POINTER(vp)]
    lib.ubr_tape_set_label.argtypes = [vp, i32]
    lib.ubr_tape_replay_timed.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_float), C.POINTER(C.c_int32), i32]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("ubr_last_error", "ubr_version", "ubr_stem_wgrad_workspace", "ubr_conv_last_config", "ubr_wgrad_last_config",
                        "ubr_tape_create", "ubr_tape_destroy"):
            fn.restype = C.c_int


def lib():
    """Load (once) and return the C-ABI library; raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "ubresnet_amd: HIP extension %s is missing; build it with "
                        "`python -m ubresnet_amd.build` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
                try:
                    l = C.CDLL(LIB_PATH)
                except OSError as e:
                    raise RuntimeError("ubresnet_amd: cannot load %s: %s" % (LIB_PATH, e))
                _declare(l)
                _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().ubr_last_error().decode("utf-8", "replace")
        raise RuntimeError("ubresnet_amd HIP call failed (%d) %s: %s" % (rc, what, msg))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """raw hipStream_t of torch's current stream on the current device (every launch asks: the Python-level
    torch.cuda.current_stream() costs ~8 us per call, the raw query well under one)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError("ubresnet_amd: %s must live on a ROCm device (got %s); the HIP path has no CPU fallback"
                           % (what, t.device))


def ptr(t):
    return None if t is None else t.data_ptr()
