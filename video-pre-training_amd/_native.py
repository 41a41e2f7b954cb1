"""ctypes binding of libvpt_hip.so (the C ABI declared in include/vpt_hip.h).

The library is built in-tree by build.py.  There is NO fallback: if it cannot be loaded every op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("VPT_HIP_LIB") or os.path.join(_HERE, "libvpt_hip.so")   # VPT_HIP_LIB: profiling builds (tools/)
# one library per 16-bit operand format (vpt_operand_format()): same sources, same ABI
_LIB_PATHS = {"bf16": _LIB_PATH, "fp16": os.path.join(_HERE, "libvpt_hip_f16.so")}
_libs = {}

ABI_VERSION = 5      # VPT_HIP_ABI of the include/vpt_hip.h this signature table was written against

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_long
_D = ctypes.c_double

# name -> argtypes, exactly as declared in include/vpt_hip.h
SIGNATURES = {
    "vpt_conv_first_forward": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vpt_conv3d_t5_forward": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_pack_conv3x3": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vpt_pack_linear": [_P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_pack_conv_first": [_P, _P, _P, _I, _P],
    "vpt_pack_conv3d_t5": [_P, _P, _P, _P, _I, _P],
    "vpt_chw_to_blocked": [_P, _P, ctypes.c_int64, _I, _I, _I, _P],
    "vpt_conv3x3_forward": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_conv3x3_forward_tiled": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpt_conv3x3_pool_forward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpt_conv3x3_forward_folded": [_P] * 12 + [_I, _I, _I, _I, _I, _P],
    "vpt_channel_stats": [_P, _P, _I, _I, _I, _P],
    "vpt_nfold_coef": [_P] * 12 + [_I, _I, _I, _I, _P],
    "vpt_maxpool_forward": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vpt_frame_affine_forward": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vpt_linear_forward": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "vpt_linear_forward_tiled": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "vpt_linear_wgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vpt_dense_fold_epilogue": [_P, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "vpt_linear_splitk_epilogue": [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    "vpt_layernorm_forward": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vpt_masked_attention_forward": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vpt_kv_memory_update": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_log_softmax_forward": [_P, _P, _I, _I, _I, _I, _F, _P],
    "vpt_action_head_forward": [_P, _P, _P, _P, ctypes.c_uint32, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "vpt_uniform_noise": [_P, ctypes.c_uint32, _P, _I, _I, _P],
    "vpt_conv_backward_prepare": [_P] * 14 + [_I, _I, _I, _I, _I, _P],
    "vpt_conv3x3_dgrad": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_conv3x3_pool_argmax_forward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpt_conv_backward_prepare_pooled": [_P] * 15 + [_I, _I, _I, _I, _I, _P],
    "vpt_conv3x3_dgrad_gated": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_conv_backward_reduce": [_P] * 10 + [_I, _I, _I, _I, _I, _P],
    "vpt_conv_first_backward": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vpt_conv_first_backward_nfold": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vpt_conv3x3_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_conv3x3_wgrad_scratch_floats": [_I, _I, _I],
    "vpt_camera_discretize": [_P, _P, _L, _D, _D, _D, _I, _P],
    "vpt_camera_undiscretize": [_P, _P, _L, _D, _D, _D, _I, _P],
    "vpt_action_from_factored": [_P, _P, _P, _P, _L, _I, _P],
    "vpt_action_to_factored": [_P, _P, _P, _P, _L, _I, _P],
    "vpt_maxpool_backward": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vpt_frame_affine_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_bc_nll_backward": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "vpt_heads_logprob_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P],
    "vpt_layernorm_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vpt_gate_cast": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_column_sum": [_P, _P, _P, _I, _I, _I, _P],
    "vpt_masked_attention_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vpt_adam_step": [_P, _P, _P, _P, ctypes.c_uint64, _I, _F, _F, _F, _F, _F, _F, _P],
    "vpt_adam_step_multi": [_P, _I, _L, _I, _F, _F, _F, _F, _F, _F, _P, _P],
    "vpt_grads_nonfinite_multi": [_P, _I, _L, _P, _P],
    "vpt_layernorm_linear_forward": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "vpt_masked_attention_step": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_masked_attention_step_inplace": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vpt_act_epilogue": [_P, _P, _P, _P, _P, _I, _I, _F, _F, _P, _P, _P, _I, _P],
    "vpt_clip_frames": [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _P],
    "vpt_debug_poison_lds": [_P],
}


class NativeLibraryError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load(fmt: str = "bf16"):
    """Load (once) and type the shared library for one operand format.  Raises NativeLibraryError if it is missing."""
    if fmt in _libs:
        return _libs[fmt]
    # PyTorch-ROCm ships its own libamdhip64.so; load it FIRST so that libvpt_hip.so binds to the same HIP
    # runtime instance torch uses (two runtimes in one process cannot share streams or device pointers).
    import torch  # noqa: F401
    path = _LIB_PATHS[fmt]
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} is missing: run `python __graft_entry__.py` (build()) first; there is no CPU fallback")
    lib = ctypes.CDLL(path)
    try:
        lib.vpt_abi_version.restype = ctypes.c_int
        abi = int(lib.vpt_abi_version())
    except AttributeError:
        abi = None
    if abi != ABI_VERSION:      # a stale .so would take a stream handle for a flag pointer, etc.: refuse before the first call
        raise NativeLibraryError(f"{path} has C-ABI version {abi}, this package binds version {ABI_VERSION}: rebuild it (`python __graft_entry__.py`)")
    override = fmt == "bf16" and bool(os.environ.get("VPT_HIP_LIB"))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # only an A/B reference build named by VPT_HIP_LIB (tools/build_ref_lib.sh: an OLDER revision of the same ABI) may lack entry points
            # added since -- calling one raises below; the in-tree library must export every symbol
            if override:
                continue
            raise NativeLibraryError(f"{path} does not export {name}: rebuild it (`python __graft_entry__.py`)")
        fn.argtypes = argtypes
        fn.restype = _I
    lib.vpt_version.restype = ctypes.c_char_p
    lib.vpt_operand_format.restype = ctypes.c_char_p
    lib.vpt_conv3x3_wgrad_scratch_floats.restype = ctypes.c_long
    lib.vpt_workspace_bytes.argtypes, lib.vpt_workspace_bytes.restype = [_I] * 6, ctypes.c_int64
    lib.vpt_conv3x3_pool_seam_elems.argtypes, lib.vpt_conv3x3_pool_seam_elems.restype = [_I] * 4, ctypes.c_int64
    for q, at in (("vpt_conv3x3_packed_elems", [_I, _I]), ("vpt_conv3x3_table_floats", [_I]), ("vpt_linear_packed_elems", [_I, _I]),
                  ("vpt_conv_first_packed_elems", [_I]), ("vpt_conv3d_t5_packed_elems", [_I])):
        getattr(lib, q).argtypes, getattr(lib, q).restype = at, ctypes.c_long
    lib.vpt_last_error.restype = ctypes.c_char_p
    if lib.vpt_operand_format().decode() != fmt:
        raise NativeLibraryError(f"{path} reports operand format {lib.vpt_operand_format().decode()}, expected {fmt}")
    _libs[fmt] = lib
    return lib


def call(name, *args, fmt: str = "bf16"):
    lib = load(fmt)
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {lib.vpt_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
