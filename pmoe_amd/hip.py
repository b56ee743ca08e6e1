"""ctypes binding of ``libpmoe_hip.so`` (the C ABI declared in ``include/pmoe_hip.h``).

PyTorch is used for device memory and streams only: every wrapper takes torch tensors, checks
device / dtype / contiguity / shape the way torch would (``ValueError`` / ``RuntimeError``), and
passes raw ``data_ptr()`` + the CURRENT torch stream.  There is no fallback: if the shared
library is missing, or a tensor is not on a ROCm device, this module raises.
"""
import ctypes as C
import os
from pathlib import Path

import torch

# PMOE_HIP_LIB: a differently built library (tools/stamp_conv.py loads its cycle-stamped build this way)
_LIB_PATH = Path(os.environ.get("PMOE_HIP_LIB") or Path(__file__).resolve().parent / "libpmoe_hip.so")
_lib = None

ABI_VERSION = 401            # include/pmoe_hip.h: PMOE_ABI_VERSION
DT_BF16, DT_F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_ELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
RES_NONE, RES_ADD, RES_DRELU, RES_DELU, RES_DTANH, RES_DSIGMOID, RES_DBN, RES_INBN = 0, 1, 2, 3, 4, 5, 6, 7
ERR_ARG, ERR_UNSUPPORTED = -1, -2      # include/pmoe_hip.h: PMOE_ERR_*
# derivative mode that undoes each activation in the data-gradient epilogue (from the layer's saved output)
RES_OF_ACT = {ACT_RELU: RES_DRELU, ACT_ELU: RES_DELU, ACT_TANH: RES_DTANH, ACT_SIGMOID: RES_DSIGMOID}
ACT_BY_NAME = {"relu": ACT_RELU, "elu": ACT_ELU, "tanh": ACT_TANH, "sigmoid": ACT_SIGMOID}

_TORCH_DT = {torch.bfloat16: DT_BF16, torch.float32: DT_F32}


class HipUnavailable(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p), ("res", C.c_void_p),
        ("bias", C.c_void_p), ("stats", C.c_void_p),
        ("n", C.c_int32), ("h", C.c_int32), ("w_", C.c_int32), ("cin", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32), ("cout", C.c_int32), ("coutp", C.c_int32),
        ("in_ld", C.c_int32), ("in_coff", C.c_int32), ("out_ld", C.c_int32), ("out_coff", C.c_int32),
        ("res_ld", C.c_int32), ("res_coff", C.c_int32),
        ("ipe", C.c_int32), ("in_shared", C.c_int32),
        ("ks", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dilate", C.c_int32),
        ("act", C.c_int32), ("res_mode", C.c_int32),
        ("drop_p", C.c_float), ("seed", C.c_uint64), ("dtype", C.c_int32),
        ("w_fp8", C.c_int32), ("in_scale", C.c_float), ("out_scale", C.c_void_p),
        ("in_fp8", C.c_int32), ("bn_coef", C.c_void_p), ("bn_ipe", C.c_int32), ("shuffle_c", C.c_int32),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("dy", C.c_void_p), ("dw_ws", C.c_void_p),
        ("n", C.c_int32), ("h", C.c_int32), ("w_", C.c_int32), ("cin", C.c_int32), ("cinp", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32), ("cout", C.c_int32), ("coutp", C.c_int32),
        ("x_ld", C.c_int32), ("x_coff", C.c_int32), ("dy_ld", C.c_int32), ("dy_coff", C.c_int32),
        ("ipe", C.c_int32), ("x_shared", C.c_int32),
        ("ks", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dtype", C.c_int32),
        ("per_image", C.c_int32),
        ("part_ws", C.c_void_p), ("part_ws_floats", C.c_int64),
        ("grads", C.c_void_p), ("cout_real", C.c_int32), ("cin_real", C.c_int32), ("defer_fold", C.c_int32),
        ("bn_fused", C.c_int32), ("bn_z", C.c_void_p), ("bn_coef", C.c_void_p), ("bn_c1", C.c_void_p), ("bn_c2", C.c_void_p),
        ("bn_z_ld", C.c_int32),
    ]


# name -> argtypes (restype is int unless listed in _RESTYPES).  Kept in one table so
# tests/test_abi.py can check that every symbol of include/pmoe_hip.h is exported and bound.
_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "pmoe_version": [],
    "pmoe_error_string": [C.c_int],
    "pmoe_abi_sizeof": [C.c_int],
    "pmoe_conv2d_igemm": [C.POINTER(ConvDesc), _P],
    "pmoe_conv2d_stat_rows": [C.POINTER(ConvDesc)],
    "pmoe_conv2d_plan": [C.POINTER(ConvDesc)],
    "pmoe_conv2d_wgrad": [C.POINTER(WgradDesc), _P],
    "pmoe_conv2d_wgrad_fold": [C.POINTER(WgradDesc), _P],
    "pmoe_conv2d_wgrad_ws_floats": [C.POINTER(WgradDesc)],
    "pmoe_conv2d_wgrad_plan": [C.POINTER(WgradDesc)],
    "pmoe_mlp_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_bn_apply_pool2": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_pack_conv_weights": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_pack_conv_weights_fp8": [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_pack_conv_weights_scaled": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_pack_conv_weights_gated": [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_unpack_conv_wgrad": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_pack_bias": [_P, _P, _I, _I, _I, _P],
    "pmoe_act_fwd": [_P, _P, _L, _I, _F, C.c_uint64, _I, _P],
    "pmoe_act_bwd": [_P, _P, _P, _L, _I, _F, _I, _P],
    "pmoe_colstats": [_P, _L, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    "pmoe_reduce_partials": [_P, _P, _I, _I, _I, _I, _P],
    "pmoe_bn_finalize": [_P, _I, _L, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _I, _I, _P, _P],
    "pmoe_bn_apply": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _F, _P],
    "pmoe_bn_bwd_reduce": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _I, _P, _I, _P],
    "pmoe_bn_bwd_finalize": [_P, _I, _L, _P, _P, _P, _P, _I, _I, _P],
    "pmoe_bn_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P],
    "pmoe_stem_tail_stats": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_stem_tail_pooled": [_P, _P, _P, C.POINTER(C.c_void_p), _P, _I, _I, _L, _I, _I, _P],
    "pmoe_stem_tail_combine": [_P, _I, _P, _I, C.POINTER(C.c_void_p), _L, _P, _P, _I, _I, _P],
    "pmoe_stem_tail_pool": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_stem_tail_bwd": [_I, _P, _P, _P, _P, C.POINTER(C.c_void_p), _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_maxpool3s2_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pmoe_maxpool3s2_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pmoe_gap_partial": [_P, _P, _P, _I, _L, _I, _I, _I, _I, _P],
    "pmoe_bn_apply_gap": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _I, _I, _I, _P],
    "pmoe_gap_finish": [_P, _P, _I, _I, _I, _L, _I, _I, _I, _P],
    "pmoe_gap_bwd": [_P, _P, _I, _L, _I, _I, _I, _I, _P],
    "pmoe_eca_gate": [_P, _I, _L, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "pmoe_eca_scale": [_P, _P, _P, _I, _L, _I, _I, _I, _P],
    "pmoe_eca_bwd_small": [_P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "pmoe_eca_stem_fold": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_eca_bwd_apply": [_P, _P, _P, _P, _I, _L, _I, _I, _P],
    "pmoe_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_pad_rows": [_P, _P, _I, _I, _I, _I, _P],
    "pmoe_gate_mixture_fwd": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "pmoe_gate_mixture_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_moe_loss": [_P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "pmoe_maxpool2s2_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_pixel_shuffle2": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_copy_window": [_P, _I, _I, _P, _I, _I, _L, _I, _I, _P],
    "pmoe_action_head_fwd": [_P, _I, _P, _I, _P, _P, _I, _I, _P],
    "pmoe_action_head_bwd": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _P],
    "pmoe_action_loss": [_P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P],
    "pmoe_blend_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P],
    "pmoe_blend_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "pmoe_maxpool2s2_bwd": [_P, _I, _I, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P],
    "pmoe_pixel_unshuffle2": [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_add_window": [_P, _I, _I, _P, _I, _I, _L, _I, _I, _P],
    "pmoe_cat_windows": [C.POINTER(C.c_void_p), _I, _I, _I, _I, _P, _I, _I, _L, _I, _P],
    "pmoe_nhwc_to_nchw": [_P, _I, _I, _P, _I, _L, _I, _I, _P],
    "pmoe_seg_loss_rows": [_I, _I, _I],
    "pmoe_seg_loss_cp": [_I],
    "pmoe_seg_loss_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P],
    "pmoe_seg_loss_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "pmoe_resample_u8_horizontal": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "pmoe_resample_u8_vertical_to_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "pmoe_resample_u8_vertical_to_i64": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "pmoe_mt_grad_norm": [_P, _P, _P, _I, _F, _P, _P, _I, _P],
    "pmoe_mt_adam": [_P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _F, _F, _P, _P],
    "pmoe_mt_swa_update": [_P, _P, _P, _I, _L, _P],
}
_RESTYPES = {"pmoe_error_string": C.c_char_p, "pmoe_conv2d_wgrad_ws_floats": C.c_int64}


def lib_path():
    return _LIB_PATH


# ---- launch recording (pmoe_amd/infer.py:PlannedMixture): every library call made while a LaunchRecorder is active is kept as
# (ctypes function, converted arguments) and can be re-issued without any of the Python above it -- the descriptors, pointer
# tables and scalar arguments are exactly the ones of the recorded run.
_recorder = None
_NOT_LAUNCHES = frozenset(("pmoe_error_string", "pmoe_abi_sizeof", "pmoe_conv2d_plan", "pmoe_conv2d_stat_rows",
                           "pmoe_conv2d_wgrad_plan", "pmoe_conv2d_wgrad_ws_floats"))


class LaunchRecorder:
    """``with LaunchRecorder() as plan: ...`` records the launches of the block (they also run); ``plan.replay()`` issues them again
    on the stream they were recorded on.  The caller keeps every buffer of the recorded run alive and in place (PlannedMixture
    runs it inside a private ``torch.cuda.MemPool``)."""

    def __init__(self):
        self.calls = []
        self.stream = None

    def __enter__(self):
        global _recorder
        if _recorder is not None:
            raise RuntimeError("a LaunchRecorder is already active")
        self.stream = torch.cuda.current_stream().cuda_stream
        _recorder = self
        return self

    def __exit__(self, *exc):
        global _recorder
        _recorder = None
        return False

    def replay(self):
        if torch.cuda.current_stream().cuda_stream != self.stream:
            raise RuntimeError("LaunchRecorder.replay: the plan was recorded on another stream")
        for fn, args in self.calls:
            rc = fn(*args)
            if rc:
                check(rc, fn.__name__)


class _RecordingLib:
    def __init__(self, lib, rec):
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in _NOT_LAUNCHES:
            return fn
        rec = self._rec

        def call(*args):
            rc = fn(*args)
            rec.calls.append((fn, args))
            return rc
        return call


def load():
    """Load (once) and return the shared library; raise HipUnavailable if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib if _recorder is None else _RecordingLib(_lib, _recorder)
    if not _LIB_PATH.exists():
        raise HipUnavailable(
            f"{_LIB_PATH} not found: build it with ./build.sh (hipcc --offload-arch=gfx950). "
            "pmoe_amd has no CPU or PyTorch fallback path.")
    lib = C.CDLL(str(_LIB_PATH))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    # the binding above was written against ONE revision of include/pmoe_hip.h: a library of another revision would take
    # misaligned arguments silently (descriptor sizes are checked too, tests/test_abi.py)
    if lib.pmoe_version() != ABI_VERSION:
        raise HipUnavailable(f"{_LIB_PATH}: ABI revision {lib.pmoe_version()}, this binding expects {ABI_VERSION}: rebuild with ./build.sh")
    if lib.pmoe_abi_sizeof(0) != C.sizeof(ConvDesc) or lib.pmoe_abi_sizeof(1) != C.sizeof(WgradDesc):
        raise HipUnavailable(f"{_LIB_PATH}: descriptor layouts differ from pmoe_amd/hip.py (pmoe_abi_sizeof): rebuild with ./build.sh")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pmoe_error_string(rc)
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt(t):
    try:
        return _TORCH_DT[t.dtype]
    except KeyError:
        raise ValueError(f"pmoe_amd kernels take bfloat16 or float32 activations, got {t.dtype}") from None


def ptr(t, name="tensor", dtype=None):
    """Validated device pointer of a dense tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the MI355X (cuda) device, got {t.device}; "
                           "pmoe_amd has no CPU path")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def ptr_table(tensors, device):
    """Device array of raw pointers (one per expert) as an int64 tensor."""
    return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device=device)
