"""ctypes binding of libdlwpmi.so (include/dlwpmi.h).

The library is the product; there is no CPU or PyTorch fallback.  Importing this module
without the built library, or calling into it without a GPU, raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DLWP_LIB_FILE: another build of the SAME library next to this file (measurement builds, e.g. the in-kernel stamps); never a fallback
LIB_PATH = os.path.join(_HERE, os.path.basename(os.environ.get("DLWP_LIB_FILE", "libdlwpmi.so")))

c_float_p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class ChanSrc(C.Structure):
    _fields_ = [("base", C.c_void_p), ("bstride", C.c_longlong), ("cstride", C.c_longlong),
                ("tab", C.c_void_p), ("tab_bstride", C.c_void_p)]


class FnoCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "B", "T", "D", "H", "W", "context_size", "teacher_forcing_steps", "hidden", "lifting",
        "projection", "n_layers", "m1", "m2c", "out_channels", "form", "constant_channels", "prescribed_channels", "m0")]


# parameter kinds of the flat FNO parameter buffer (dlwpmi.h enum)
P_LIFT_W1, P_LIFT_B1, P_LIFT_W2, P_LIFT_B2, P_PROJ_W1, P_PROJ_B1, P_PROJ_W2, P_PROJ_B2, \
    P_SPEC_W, P_SKIP_W, P_SPEC_B = range(11)

# every symbol include/dlwpmi.h declares: name -> (restype, argtypes)
_V, _I, _L, _F = C.c_void_p, C.c_int, C.c_longlong, C.c_float
SIGNATURES = {
    "dlwp_version": (_I, []),
    "dlwp_last_error": (C.c_char_p, []),
    "dlwp_set_tuning": (_I, [C.c_char_p, _I]),
    "dlwp_clear_tuning": (_I, [C.c_char_p]),
    "dlwp_get_tuning": (_I, [C.c_char_p, C.POINTER(_I)]),
    "dlwp_tuning_list": (_I, [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "dlwp_prof_enable": (_I, [_I]),
    "dlwp_prof_collect": (_I, []),
    "dlwp_prof_get": (_I, [_I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dlwp_pwmlp_fwd": (_I, [_V] * 6 + [_I] * 5 + [_V]),
    "dlwp_pwmlp_bwd": (_I, [_V] * 10 + [_I] * 5 + [_V]),
    "dlwp_pwmlp_slab_floats": (_L, [_I] * 5),
    "dlwp_pwmlp_bwd_slab": (_I, [_V] * 7 + [_I] * 6 + [_V]),
    "dlwp_pwmlp_slab_fold": (_I, [_V] * 5 + [_I] * 5 + [_V]),
    "dlwp_fno_plan_create": (_I, [_I] * 5 + [C.POINTER(_V)]),
    "dlwp_fno_plan_create3d": (_I, [_I] * 7 + [C.POINTER(_V)]),
    "dlwp_fno_plan_destroy": (None, [_V]),
    "dlwp_fno_block_workspace_bytes": (C.c_size_t, [_V, _I]),
    "dlwp_fno_block_fwd": (_I, [_V, _V, _I, _V, _V, _V, _V, _V, _I, _V, _V]),
    "dlwp_fno_block_bwd": (_I, [_V, _V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _I, _V, _V]),
    "dlwp_sqerr_sum": (_I, [_V, _V, _L, _F, _V, _V]),
    "dlwp_mse_fwd_bwd": (_I, [_V, _V, _L, _V, _V, _V]),
    "dlwp_error_moments": (_I, [_V, _V, _V, _V, _I, _I, _I, _I, _V, _V]),
    "dlwp_adam_step": (_I, [_V, _V, _V, _V, _V, _L, _F, _F, _F, _F, _F, _I, _V]),
    "dlwp_adam_step_clipped": (_I, [_V, _V, _V, _V, _V, _L, _F, _F, _F, _F, _F, _I, _V, _F, _V]),
    "dlwp_sumsq": (_I, [_V, _L, _V, _V]),
    "dlwp_clip_scale": (_I, [_V, _L, _V, _F, _F, _V]),
    "dlwp_fno_param_offset": (_L, [C.POINTER(FnoCfg), _I, _I, C.POINTER(_L)]),
    "dlwp_fno_trainer_create": (_I, [C.POINTER(FnoCfg), C.POINTER(_V)]),
    "dlwp_fno_trainer_destroy": (None, [_V]),
    "dlwp_fno_trainer_bind_io": (_I, [_V, _V, _V, _V, _V]),
    "dlwp_fno_trainer_bind": (_I, [_V, _V, _V]),
    "dlwp_fno_trainer_bind_aux": (_I, [_V, _V, _V]),
    "dlwp_fno_trainer_forward": (_I, [_V, _I, _V]),
    "dlwp_fno_trainer_backward": (_I, [_V, _V, _V]),
    "dlwp_fno_trainer_fwd_bwd": (_I, [_V, _I, _V]),
    "dlwp_afno2d_save_elems": (_L, [_I, _I, _I, _I, _I, _F]),
    "dlwp_afno2d_fwd": (_I, [_V] * 7 + [_I] * 5 + [_F, _F, _V]),
    "dlwp_afno2d_fwd_res": (_I, [_V] * 8 + [_I] * 5 + [_F, _F, _V]),
    "dlwp_afno2d_bwd": (_I, [_V] * 11 + [_I] * 5 + [_F, _F, _V]),
    "dlwp_afno_wq_expand": (_I, [_V, _V, _I, _I, _I, _V]),
    "dlwp_afno_wq_fold": (_I, [_V, _V, _I, _I, _I, _V]),
    "dlwp_set_gemm_precision": (_I, [_I]),
    "dlwp_get_gemm_precision": (_I, []),
    "dlwp_set_gemm_tile256": (_I, [_I]),
    "dlwp_weight_grad_group": (_I, [_V, _I, _V]),
    "dlwp_wgrad_segments_workspace_bytes": (C.c_size_t, [_V, _I, _I, _I]),
    "dlwp_wgrad_segments": (_I, [_V, _I, _I, _I, _V, C.c_size_t, _V]),
    "dlwp_gemm_group_begin": (_I, []),
    "dlwp_gemm_group_end": (_I, [_V]),
    "dlwp_window_gather": (_I, [_V, _V, _I, _I] + [_V] * 7 + [_V]),
    "dlwp_window_gather_fill": (_I, [_V, _V, _V, _I, _I] + [_V] * 7 + [_V]),
    "dlwp_window_gather_ex": (_I, [_V, _V, _V, _V, _I, _I] + [_V] * 7 + [_I, _V]),
    "dlwp_window_scatter_ex": (_I, [_V, _V, _V, _V, _I, _I] + [_V] * 7 + [_I, _I, _V]),
    "dlwp_window_pad_colsum": (_I, [_V, _V, _I, _I] + [_V] * 7 + [_I, _V]),
    "dlwp_window_scatter": (_I, [_V, _V, _I, _I] + [_V] * 7 + [_I, _V]),
    "dlwp_window_scatter_add": (_I, [_V, _V, _V, _I, _I] + [_V] * 7 + [_I, _V]),
    "dlwp_patch_merge": (_I, [_V, _V, _I, _I, _I, _I, _I, _V]),
    "dlwp_upconv_shuffle": (_I, [_V] * 5 + [_I] * 10 + [_V]),
    "dlwp_window_advance_fwd": (_I, [_V, _L, _V, _V, _V, _I, _I, _L] + [_I] * 6 + [_V]),
    "dlwp_window_advance_bwd": (_I, [_V, _V, _L, _V, _L, _V, _V, _I, _I, _L] + [_I] * 6 + [_V]),
    "dlwp_gemm": (_I, [_V, _V, _V] + [_I] * 8 + [_V, _I, _V, _V, _I, _V, _V]),
    "dlwp_gemm_mixed": (_I, [_V, _V, _V] + [_I] * 8 + [_V, _I, _V, _V, _I, _V, _I, _V]),
    "dlwp_gemm_batched_mixed": (_I, [_V, _V, _V] + [_I] * 10 + [_L] * 6 + [_V, _L, _L, _I, _F, _V, _V, _L, _L, _I, _I, _I, _V]),
    "dlwp_cast_bf16": (_I, [_V, _V, _L, _V]),
    "dlwp_transpose_cast_bf16_many": (_I, [_V, _V, _V, _I, _I, _V]),
    "dlwp_gemm_rowscale": (_I, [_V, _V, _V] + [_I] * 8 + [_V, _V, _V, _I, _I, _V]),
    "dlwp_cast_bf16_scaled": (_I, [_V, _V, _V, _I, _L, _V]),
    "dlwp_gemm_batched": (_I, [_V, _V, _V] + [_I] * 10 + [_L] * 6 + [_V, _L, _L, _I, _F, _V, _V, _L, _L, _I, _I, _V]),
    "dlwp_act_bwd": (_I, [_V, _V, _V, _L, _I, _F, _V]),
    "dlwp_sht_fused_supported": (_I, [_I] * 5),
    "dlwp_sht_analysis": (_I, [_V, _V, _V, _V] + [_I] * 6 + [_V]),
    "dlwp_sht_synthesis": (_I, [_V, _V, _V, _V] + [_I] * 6 + [_V]),
    "dlwp_sht_bf16_supported": (_I, [_I] * 5),
    "dlwp_sht_analysis_bf16": (_I, [_V, _V, _V, _V] + [_I] * 6 + [_V]),
    "dlwp_sht_synthesis_bf16": (_I, [_V, _V, _V, _V, _V] + [_I] * 6 + [_V]),
    "dlwp_sht_synthesis_bf16_ex": (_I, [_V, _V, _V, _V, _V] + [_I] * 7 + [_V]),
    "dlwp_sht_analysis_bf16_ex": (_I, [_V, _V, _V, _V] + [_I] * 7 + [_V]),
    "dlwp_cweight_expand": (_I, [_V, _V, _I, _I, _I, _V]),
    "dlwp_cweight_fold": (_I, [_V, _V, _I, _I, _I, _V]),
    "dlwp_dhconv_supported": (_I, [_I, _I, _I]),
    "dlwp_dhconv_image_elems": (_L, [_I, _I, _I]),
    "dlwp_dhconv_pack": (_I, [_V, _V, _V, _I, _I, _I, _V]),
    "dlwp_dhconv_pack_many": (_I, [_V, _V, _V, _I, _I, _I, _I, _V]),
    "dlwp_dhconv_apply": (_I, [_V, _V, _V, _I, _I, _I, _I, _I, _I, _V]),
    "dlwp_dhconv_wgrad": (_I, [_V, _V, _I, _V, _I, _I, _I, _I, _I, _V]),
    "dlwp_dhconv_fold": (_I, [_V, _V, _I, _I, _I, _V]),
    "dlwp_mlp_chain_supported": (_I, [_I, _I]),
    "dlwp_mlp_chain_pack": (_I, [_V, _I, _I, _I, _V, _V]),
    "dlwp_sfno_tail_pack": (_I, [_V, _V, _V, _I, _I, _V, _V]),
    "dlwp_sfno_tail_pack_many": (_I, [_V, _V, _V, _I, _I, _I, _V, _V]),
    "dlwp_mlp_stream_supported": (_I, [_I, _I]),
    "dlwp_mlp_stream_pack": (_I, [_V, _V, _I, _I, _V, _V]),
    "dlwp_mlp_stream_fwd": (_I, [_V, _I, _V, _V, _V, _V, _V, _V, _V, _V, _V, _I, _I, _I, _V]),
    "dlwp_mlp_stream_bwd": (_I, [_V, _V, _V, _V, _V, _V, _V, _I, _I, _I, _I, _V]),
    "dlwp_sfno_io_supported": (_I, [_I, _I, _I]),
    "dlwp_sfno_io_image_elems": (_L, [_I]),
    "dlwp_sfno_io_pack": (_I, [_V, _V, _V, _V, _I, _I, _I, _I, _V, _V]),
    "dlwp_sfno_encode_fwd": (_I, [_V, _V]),
    "dlwp_sfno_encode_bwd": (_I, [_V, _V]),
    "dlwp_sfno_decode_fwd": (_I, [_V, _V]),
    "dlwp_sfno_decode_bwd": (_I, [_V, _V]),
    "dlwp_sfno_tail_fwd": (_I, [_V, _V]),
    "dlwp_sfno_tail_bwd": (_I, [_V, _V]),
    "dlwp_layernorm_fwd": (_I, [_V] * 6 + [_I, _I, _F, _V]),
    "dlwp_layernorm_fwd_ex": (_I, [_V] * 6 + [_I, _I, _F, _I, _V]),
    "dlwp_layernorm_bwd": (_I, [_V] * 8 + [_I, _I, _V]),
    "dlwp_layernorm_bwd_res": (_I, [_V] * 9 + [_I, _I, _V]),
    "dlwp_layernorm_bwd_ex": (_I, [_V] * 5 + [_I] + [_V] * 4 + [_I, _I, _V]),
    "dlwp_layernorm_bwd_lowp": (_I, [_V] * 5 + [_I] + [_V] * 4 + [_I, _I, _V, _V, _I, _V]),
    "dlwp_instnorm_fwd": (_I, [_V] * 6 + [_I, _I, _I, _F, _V]),
    "dlwp_instnorm_bwd": (_I, [_V] * 8 + [_I, _I, _I, _V]),
    "dlwp_gelu_bwd": (_I, [_V, _V, _V, _L, _V]),
    "dlwp_colsum": (_I, [_V, _V, _I, _I, _V]),
    "dlwp_colsum_ex": (_I, [_V, _V, _I, _I, _I, _V]),
    "dlwp_add2d_many": (_I, [_V, _I, _V]),
    "dlwp_scale_rows_add": (_I, [_V, _V, _V, _V, _I, _L, _V]),
    "dlwp_add_bcast": (_I, [_V, _V, _V, _I, _L, _V]),
    "dlwp_cmode_product": (_I, [_V, _V, _V, _I, _I, _I, _I, _V]),
    "dlwp_cmode_product_bwd": (_I, [_V] * 5 + [_I] * 4 + [_V]),
    "dlwp_window_attn_fwd": (_I, [_V] * 7 + [_I] * 7 + [_F, _V]),
    "dlwp_window_attn_bwd": (_I, [_V] * 12 + [_I] * 7 + [_F, _V]),
    "dlwp_window_attn_pack_table": (_I, [_V, _V, _I, _I, _I, _V]),
    "dlwp_window_attn_fwd_packed": (_I, [_V] * 8 + [_I] * 7 + [_F, _V]),
    "dlwp_window_attn_bwd_packed": (_I, [_V] * 13 + [_I] * 7 + [_F, _V]),
    "dlwp_window_attn_fwd_qrange": (_I, [_V] * 8 + [_I] * 7 + [_F, _I, _I, _V]),
    "dlwp_window_attn_bwd_qrange": (_I, [_V] * 13 + [_I] * 7 + [_F, _I, _I, _V]),
    "dlwp_window_attn_io_bf16_supported": (_I, [_I, _I, _I, C.c_longlong]),
    "dlwp_window_attn_fwd_bf16": (_I, [_V] * 8 + [_I] * 7 + [_F, _V]),
    "dlwp_window_attn_bwd_bf16": (_I, [_V] * 11 + [_I] * 7 + [_F, _V]),
    "dlwp_window_attn_bwd_tokens_supported": (_I, [_I] * 3),
    "dlwp_window_attn_bwd_tokens": (_I, [_V] * 15 + [_I] * 8 + [_F, _I, _I, _I, _V]),
    "dlwp_window_attn_fwd_tokens_supported": (_I, [_I, _I, C.c_longlong]),
    "dlwp_window_attn_fwd_tokens": (_I, [_V] * 11 + [_I] * 8 + [_F, _I, _I, _I, _V]),
    "dlwp_window_attn_bwd_slab_floats": (_L, [_I] * 4),
    "dlwp_window_softmax_fwd": (_I, [_V] * 5 + [_I] * 5 + [_F, _V]),
    "dlwp_window_softmax_bwd": (_I, [_V] * 5 + [_I] * 5 + [_F, _V]),
    "dlwp_fno_spatial_fwd_probe": (_I, [_V, _V, _V, _V, _V, _V, _V, _I, _V]),
    "dlwp_fft_plan_create": (_I, [_I, _I, C.POINTER(_V)]),
    "dlwp_fft_plan_destroy": (None, [_V]),
    "dlwp_rfft2": (_I, [_V, _V, _V, _I, _I, _I, _I, _I, _V]),
    "dlwp_irfft2": (_I, [_V, _V, _V, _V, _I, _I, _I, _I, _I, _V]),
    "dlwp_rfft2_planar": (_I, [_V, _V, _V, _V] + [_I] * 8 + [_V]),
    "dlwp_rfft2_planar_masked": (_I, [_V, _V, _V, _V, _V, _F] + [_I] * 8 + [_V]),
    "dlwp_rfft2_planar_ex": (_I, [_V, _V, _V, _V, _V, _F] + [_I] * 9 + [_V]),
    "dlwp_irfft2_planar_ex": (_I, [_V, _V, _V, _V, _V, _V] + [_I] * 9 + [_V]),
    "dlwp_colsum_bf16": (_I, [_V, _V, _I, _I, _V]),
    "dlwp_irfft2_planar": (_I, [_V, _V, _V, _V, _V] + [_I] * 8 + [_V]),
    "dlwp_irfft2_planar2": (_I, [_V, _V, _V, _V, _V, _V] + [_I] * 8 + [_V]),
    "dlwp_afno_wq_expand_bp": (_I, [_V, _V, _V, _V, _I, _I, _I, _V]),
    "dlwp_afno_wq_fold_bp": (_I, [_V, _V, _V, _V, _I, _I, _I, _V]),
    "dlwp_comm_unique_id": (_I, [_V]),
    "dlwp_comm_create": (_I, [_V, _I, _I, C.POINTER(_V)]),
    "dlwp_comm_destroy": (None, [_V]),
    "dlwp_comm_allreduce": (_I, [_V, _V, _L, _V]),
    "dlwp_comm_broadcast": (_I, [_V, _V, _L, _I, _V]),
    "dlwp_fno_mix_fwd_probe": (_I, [_V, _V, _V, _V, _V, _I, _V]),
    "dlwp_debug_null_kernels": (_I, [_I, _I, _V]),
    "dlwp_debug_spin_kernels": (_I, [_I, _I, _I, _I, _I, _V]),
    "dlwp_debug_clock_probe": (_I, [_V, _I, _I, _V]),
}

_lib = None


class DlwpError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DlwpError(
            f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
            "There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().dlwp_last_error()
        raise DlwpError(f"libdlwpmi error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DlwpError("libdlwpmi needs CUDA/HIP tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise DlwpError("libdlwpmi needs contiguous tensors")
    return t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


class gemm_precision:
    """Context manager / setter for the GEMM operand precision: "fp32" (default, exact fp32 MFMA) or "bf16"
    (bf16 operands, fp32 accumulation: the reference's bf16-autocast arithmetic)."""

    MODES = {"fp32": 0, "bf16": 1}

    def __init__(self, mode):
        self.mode = self.MODES[mode]

    def __enter__(self):
        lib = load()
        self.prev = lib.dlwp_get_gemm_precision()
        check(lib.dlwp_set_gemm_precision(self.mode))
        return self

    def __exit__(self, *exc):
        check(load().dlwp_set_gemm_precision(self.prev))
        return False


class gemm_group:
    """with lib.gemm_group(): ...  -- the small, mutually INDEPENDENT GEMMs issued inside go out as one launch
    (dlwp_gemm_group_begin / _end); large ones launch as usual."""

    def __enter__(self):
        check(load().dlwp_gemm_group_begin())
        return self

    def __exit__(self, *exc):
        check(load().dlwp_gemm_group_end(stream()))
        return False


def set_tuning(name, value=None):
    """Override (value) or release (None) a measurement knob of the library (include/dlwpmi.h: dlwp_set_tuning); names as in
    tuning_knobs().  While no override is set the environment variable DLWP_<NAME> is consulted at the time of use."""
    lib = load()
    check(lib.dlwp_clear_tuning(name.encode()) if value is None else lib.dlwp_set_tuning(name.encode(), int(value)))


def tuning_knobs():
    """{name: meaning} of every knob."""
    lib, out, i = load(), {}, 0
    n, d = C.c_char_p(), C.c_char_p()
    while lib.dlwp_tuning_list(i, C.byref(n), C.byref(d)):
        out[n.value.decode()] = d.value.decode()
        i += 1
    return out


class kernel_accounting:
    """with lib.kernel_accounting() as acc: <eager launches> ; acc.rows -- live per-kernel accounting (dlwp_prof_enable /
    _collect / _get): one row per kernel name, sorted by total time, with the calls, the event-bracketed milliseconds and the
    algorithmic flops / HBM bytes the launches' own arguments imply.  Launches inside a hipGraph capture are not recorded."""

    def __init__(self, shapes=False):
        self.shapes = shapes      # True: GEMM rows carry the product's shape and epilogue (one row per distinct product)

    def __enter__(self):
        check(load().dlwp_prof_enable(2 if self.shapes else 1))
        self.rows = []
        return self

    def __exit__(self, *exc):
        lib = load()
        n = lib.dlwp_prof_collect()
        buf = C.create_string_buffer(160)
        calls, ms, fl, by = _L(), C.c_double(), C.c_double(), C.c_double()
        for i in range(max(n, 0)):
            check(lib.dlwp_prof_get(i, buf, 160, C.byref(calls), C.byref(ms), C.byref(fl), C.byref(by)))
            self.rows.append({"name": buf.value.decode(), "calls": calls.value, "ms": ms.value, "flops": fl.value, "bytes": by.value})
        return False


def set_gemm_tile256(mode):
    """-1: never use the 256 x 256 bf16 GEMM kernel, 0: by shape (default), 1: wherever it applies (measurement / tests)."""
    check(load().dlwp_set_gemm_tile256(int(mode)))


def set_gemm_precision(mode):
    check(load().dlwp_set_gemm_precision(gemm_precision.MODES[mode]))


# ---- bf16 storage (BASELINE configs C3-C5 train under bf16 autocast).  "bf16": the token MLPs keep their hidden activations,
# pre-activations and hidden gradients as bf16 in HBM and read a per-step bf16 copy of the weights (train_engine keeps it
# next to the fp32 master weights); sums, epilogues, LayerNorm / softmax statistics, the residual stream and every parameter
# gradient stay fp32.  Only meaningful with gemm precision "bf16" (the matrix units round the operands to bf16 anyway).
_STORAGE = "fp32"
SHADOW_ACTIVE = False      # set by train_engine while a step runs: the bf16 weight copies are current


def set_storage(mode):
    global _STORAGE
    if mode not in ("fp32", "bf16"):
        raise ValueError("storage mode must be 'fp32' or 'bf16'")
    if mode == "bf16" and load().dlwp_get_gemm_precision() != 1:
        raise DlwpError("bf16 storage goes with bf16 GEMM operands: call set_gemm_precision('bf16') first")
    _STORAGE = mode


def storage_bf16():
    return _STORAGE == "bf16"


def shadow_t(p):
    """The current TRANSPOSED bf16 copy ([in][out]) of a Linear weight `p` ([out][in]), or None (train_engine keeps them for the 2-D
    weights whose input-gradient products read them)."""
    if not (SHADOW_ACTIVE and _STORAGE == "bf16"):
        return None
    return getattr(p, "_dlwp_bf16_t", None)


def shadow(p):
    """The current bf16 copy of parameter `p` (same shape), or None when there is none / it may be stale."""
    if not (SHADOW_ACTIVE and _STORAGE == "bf16"):
        return None
    return getattr(p, "_dlwp_bf16", None)
