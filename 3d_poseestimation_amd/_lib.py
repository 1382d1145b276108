"""ctypes binding of libposelift.so (include/poselift.h).

The library is the product: there is no CPU or eager-torch fallback.  If the shared
object is missing, or an entry point reports an error, this module raises.
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch's bundled HIP runtime first so the library binds to it)

_HERE = os.path.dirname(os.path.abspath(__file__))
# POSELIFT_LIB: another build of the same library (same-box A/B of two builds, tools/ab_env.py)
LIB_PATH = os.environ.get("POSELIFT_LIB") or os.path.join(_HERE, "csrc", "libposelift.so")

PL_F32, PL_BF16, PL_BF16X6, PL_F16X3 = 0, 1, 2, 3


# int gather(void* user, float* buf, int64_t floats_per_rank, void* stream)   (poselift.h: PLGatherFn)
GATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)


class PLSync(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int32), ("rank", ctypes.c_int32), ("gather", GATHER_FN),
                ("user", ctypes.c_void_p)]


class PLL1Term(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("da", ctypes.c_void_p), ("db", ctypes.c_void_p)]


L1_MAX_TERMS = 8


class PLDesc(ctypes.Structure):
    _fields_ = [
        ("in_dim", ctypes.c_int32), ("hidden", ctypes.c_int32), ("out_dim", ctypes.c_int32),
        ("num_stage", ctypes.c_int32), ("bn", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("p_dropout", ctypes.c_float), ("bn_eps", ctypes.c_float), ("bn_momentum", ctypes.c_float),
        ("reserved", ctypes.c_int32),
        ("params", ctypes.c_void_p), ("bn_running", ctypes.c_void_p), ("bn_batches", ctypes.c_void_p),
        ("sync", ctypes.POINTER(PLSync)),
        ("step_dev", ctypes.c_void_p),
        ("wplanes", ctypes.c_void_p), ("wplanes_valid", ctypes.c_int32), ("reserved2", ctypes.c_int32),
    ]


ADAMW_MAX_SEGS = 8


class PLAdamWSeg(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_int64), ("numel", ctypes.c_int64), ("h", ctypes.c_void_p), ("l", ctypes.c_void_p)]


class PLAdamWPlanes(ctypes.Structure):
    _fields_ = [("nseg", ctypes.c_int32), ("kind", ctypes.c_int32), ("scale", ctypes.c_float),
                ("reserved", ctypes.c_int32), ("seg", PLAdamWSeg * ADAMW_MAX_SEGS)]


class PLAdamWStep(ctypes.Structure):
    _fields_ = [("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("lr", ctypes.c_float), ("lr_dev", ctypes.c_void_p),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float),
                ("t", ctypes.c_int64), ("t_dev", ctypes.c_void_p)]


class PLPlanesEpilogue(ctypes.Structure):
    _fields_ = [("bias", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("resid", ctypes.c_void_p),
                ("relu", ctypes.c_int32), ("reserved", ctypes.c_int32), ("y_planes", ctypes.c_void_p)]


_c = ctypes
_P = ctypes.c_void_p
_D = ctypes.POINTER(PLDesc)
# name -> (restype, argtypes); one entry per declaration in include/poselift.h
SIGNATURES = {
    "pl_version": (_c.c_int, []),
    "pl_last_error": (_c.c_char_p, []),
    "pl_num_hidden": (_c.c_int64, [_D]),
    "pl_param_tensors": (_c.c_int64, [_D]),
    "pl_param_offset": (_c.c_int64, [_D, _c.c_int64]),
    "pl_param_numel": (_c.c_int64, [_D, _c.c_int64]),
    "pl_param_arena_floats": (_c.c_int64, [_D]),
    "pl_workspace_bytes": (_c.c_size_t, [_D, _c.c_int64]),
    "pl_workspace_view": (_c.c_int, [_D, _c.c_int64, _c.c_int, _c.c_int64,
                                     _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_size_t)]),
    "pl_workspace_bitmap_format": (_c.c_int, [_D, _c.c_int64, _c.c_int64]),
    "pl_lifter_fwd_eval": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _P]),
    "pl_lifter_fwd_train": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _c.c_uint64,
                                       _c.c_uint64, _P, _P]),
    "pl_lifter_bwd": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _P, _P, _P]),
    "pl_lifter_fwd_eval_saved": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _P]),
    "pl_lifter_bwd_eval": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _P, _P, _P]),
    "pl_lifter_bwd_layers": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _P, _P, _c.c_int, _c.c_int, _P]),
    "pl_lifter_train_fwd_bwd": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _c.c_uint64, _c.c_uint64,
                                           _P, _P, _P, _c.c_int, _c.c_int, _P]),
    "pl_lifter_step_carries_adamw": (_c.c_int, [_D, _c.c_int64]),
    "pl_lifter_train_step": (_c.c_int, [_D, _P, _P, _c.c_int64, _P, _c.c_size_t, _c.c_uint64, _c.c_uint64,
                                        _P, _P, _P, _c.POINTER(PLAdamWStep), _P]),
    "pl_mse_scratch_bytes": (_c.c_size_t, [_c.c_int64]),
    "pl_mse_fwd_bwd": (_c.c_int, [_P, _P, _c.c_int64, _c.c_float, _P, _P, _P, _P]),
    "pl_l1_scratch_bytes": (_c.c_size_t, [_c.c_int]),
    "pl_l1_terms_fwd_bwd": (_c.c_int, [_c.POINTER(PLL1Term), _c.c_int, _c.c_float, _P, _P, _P]),
    "pl_mpjpe_scratch_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    "pl_mpjpe_accum": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P, _P]),
    "pl_adamw_flat": (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_float, _c.c_float, _c.c_float,
                                 _c.c_float, _c.c_float, _c.c_int64, _c.c_float, _P]),
    "pl_adamw_flat_dev": (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _P, _c.c_float, _c.c_float, _c.c_float,
                                     _c.c_float, _c.c_int64, _P, _c.c_float, _P]),
    "pl_adamw_flat_planes": (_c.c_int, [_P, _P, _P, _P, _c.c_int64, _c.c_float, _P, _c.c_float, _c.c_float, _c.c_float,
                                        _c.c_float, _c.c_int64, _P, _c.c_float, _c.POINTER(PLAdamWPlanes), _P]),
    "pl_wplanes_bytes": (_c.c_size_t, [_D]),
    "pl_wplanes_layer_bytes": (_c.c_size_t, [_D]),
    "pl_weight_plane_scale": (_c.c_float, []),
    "pl_wplanes_refresh": (_c.c_int, [_D, _P]),
    "pl_flip_pose": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P]),
    "pl_counter_add": (_c.c_int, [_P, _c.c_int64, _P]),
    "pl_flip_pose_ex": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_float, _c.c_float, _P]),
    "pl_flip_w_nhwc": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P]),
    "pl_softargmax3d_nhwc_fwd": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _P]),
    "pl_softargmax3d_nhwc_bwd": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_conv2d_nhwc_scratch_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                                  _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "pl_conv2d_nhwc_scratch_bytes_ex": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                                     _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "pl_conv2d_nhwc_fwd": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int,
                                      _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _c.c_int, _P, _P, _c.c_int, _P, _c.c_size_t,
                                      _P]),
    "pl_conv2d_nhwc_wgrad_scratch_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                                        _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    "pl_conv2d_nhwc_wgrad": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int,
                                        _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _P, _c.c_size_t, _P]),
    "pl_bn_train_scratch_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    "pl_bn_train_fwd": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _c.c_float, _c.c_float, _P, _P, _P, _c.c_int,
                                   _P, _P, _P, _P, _P, _P]),
    "pl_bn_train_bwd": (_c.c_int, [_P, _P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _P]),
    "pl_add_relu_fwd": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P, _P]),
    "pl_mask_by_bits": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_conv_act_plane_scale": (_c.c_float, []),
    "pl_planes_split": (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_float, _P, _P]),
    "pl_planes_split_strided": (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_float, _P, _P]),
    "pl_bn_train_fwd_ex": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _c.c_float, _c.c_float, _P, _P, _P, _c.c_int,
                                      _P, _P, _P, _P, _P, _P, _c.c_int, _P, _P, _P]),
    "pl_gemm_stat_groups": (_c.c_int, [_c.c_int64]),
    "pl_bn_train_bwd_ex": (_c.c_int, [_P, _P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _P, _c.c_int, _P, _P]),
    "pl_add_relu_fwd_ex": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_int, _P]),
    "pl_mask_add_by_bits": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_bn_join_bwd": (_c.c_int, [_P, _P, _P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _P, _P, _c.c_int, _P, _P]),
    "pl_maxpool3x3s2_nhwc": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_maxpool3x3s2_nhwc_bwd": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_maxpool3x3s2_nhwc_idx": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _P]),
    "pl_maxpool3x3s2_nhwc_bwd_idx": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_upsample2x_zero_nhwc": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_colsum_scratch_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64]),
    "pl_colsum": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _P]),
    "pl_deconv4x4s2_nhwc_scratch_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64]),
    "pl_deconv4x4s2_nhwc_fwd": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int64, _P, _P,
                                           _c.c_int, _P, _c.c_int, _P, _c.c_size_t, _P]),
    "pl_nhwc_to_nchw": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P]),
    "pl_gather_rows2": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int64, _P, _c.c_int64, _c.c_int64, _P, _P, _P]),
    "pl_flip_tta_pack": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P]),
    "pl_flip_tta_merge": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P]),
    "pl_softargmax_dl_scale": (_c.c_int, [_P, _c.c_int64, _c.c_int, _P, _P]),
    "pl_softargmax_fwd": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _P, _P, _P]),
    "pl_softargmax_bwd": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                     _P, _P]),
    "pl_gemm_f32": (_c.c_int, [_c.c_int, _P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                               _c.c_int, _P, _P]),
    "pl_gemm_arith": (_c.c_int, [_c.c_int, _c.c_int, _P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                 _c.c_int, _P, _P]),
    "pl_gemm_planes_scratch_bytes": (_c.c_size_t, [_c.c_int64, _c.c_int64, _c.c_int64]),
    "pl_gemm_planes": (_c.c_int, [_c.c_int, _c.c_int, _P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                  _c.c_float, _c.c_float, _P, _P]),
    "pl_gemm_planes_splits": (_c.c_int, [_c.c_int64, _c.c_int64, _c.c_int64]),
    "pl_gemm_planes_raw": (_c.c_int, [_c.c_int, _c.c_int, _P, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int64, _P,
                                      _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_float, _P, _P, _P, _P]),
    "pl_conv2d_planes_fwd": (_c.c_int, [_c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                        _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_float, _P, _P, _P]),
    "pl_softargmax3d_nhwc_bwd_ex": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _c.c_int,
                                               _P, _P]),
    "pl_colsum_planes": (_c.c_int, [_P, _c.c_int, _c.c_int64, _c.c_int64, _P, _P, _P, _P]),
    "pl_deconv4x4s2_planes_fwd": (_c.c_int, [_c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                             _c.c_int64, _c.c_int64, _P, _c.c_float, _P, _P]),
    "pl_conv2d_planes_fwd_ep": (_c.c_int, [_c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                           _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_float,
                                           _c.POINTER(PLPlanesEpilogue), _P]),
    "pl_deconv4x4s2_planes_fwd_ep": (_c.c_int, [_c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                                _c.c_int64, _c.c_int64, _P, _c.c_float, _c.POINTER(PLPlanesEpilogue), _P]),
    "pl_conv2d_planes_wgrad": (_c.c_int, [_c.c_int, _P, _c.c_int64, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                          _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_float,
                                          _P, _P, _P]),
    "pl_conv2d_planes_fwd_hw": (_c.c_int, [_c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                           _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int, _P, _c.c_float, _P, _P, _P]),
    "pl_conv2d_planes_fwd_ep_hw": (_c.c_int, [_c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                              _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                              _c.c_int, _P, _c.c_float, _c.POINTER(PLPlanesEpilogue), _P]),
    "pl_conv2d_planes_wgrad_hw": (_c.c_int, [_c.c_int, _P, _c.c_int64, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                             _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                             _c.c_int, _P, _c.c_float, _P, _P, _P]),
    "pl_prof_enable": (_c.c_int, [_c.c_int]),
    "pl_prof_read": (_c.c_int, [_c.c_double, _c.c_double, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64),
                                _c.POINTER(_c.c_double)]),
}

_lib = None


class PoseliftError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PoseliftError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().pl_last_error().decode("utf-8", "replace")
        raise PoseliftError(f"{what} failed (status {rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream().cuda_stream builds a Stream
    object per call: 9 us, 170 times per conv training step.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(dev):
    """`with on_device(t.device):` -- torch.cuda.device(dev), minus its cost when dev already is the current device (the
    one-process-per-GPU case: always)."""
    idx = dev.index if isinstance(dev, torch.device) else dev
    if _cur_device is not None and (idx is None or idx == _cur_device()):
        return _NO_SWITCH
    return torch.cuda.device(dev)


def require_device_tensor(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise PoseliftError(
            f"{name} is on {t.device}: the lifter runs on an MI355X (ROCm) device only; "
            "there is no CPU path")
    if t.dtype != dtype:
        raise PoseliftError(f"{name} has dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise PoseliftError(f"{name} must be contiguous")
