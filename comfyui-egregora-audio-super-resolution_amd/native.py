"""ctypes binding of libegregora_amd.so (the C ABI declared in include/egregora_amd.h).

There is no CPU fallback: if the shared library or an MI355X is missing, every compute entry point
raises RuntimeError.  torch is used only to own device memory and streams.
"""
import ctypes as C
import os
from pathlib import Path

_LIB = None
LIB_PATH = Path(__file__).resolve().parent / "libegregora_amd.so"

EGR_OK = 0
FL_NORMALIZE, FL_AUTOSCALE, FL_PCM_IN, FL_NODE_POST = 0x1, 0x2, 0x4, 0x8
FL_THR_RELATIVE, FL_THR_SOFT, FL_NO_INIT_THR, FL_ZERO_STUFF, FL_INTERP_LINSPACE = 0x10, 0x20, 0x40, 0x80, 0x100      # SPEC.md section 3
FL_DEFER_FINALIZE = 0x400    # enhance stops behind out = y + d; egr_fatllama_joint_peak / _finalize complete it (channel-parallel runs)
FL_THR_RECOMPUTE = 0x200     # with FL_THR_RELATIVE: a maximum pass in every iteration instead of the carried maximum (tests, A/B)
FL_INFO_LEN = 48
ABI_VERSION = 5          # include/egregora_amd.h EGR_ABI_VERSION

# name -> (restype, argtypes); must list every symbol of include/egregora_amd.h
_vp, _i, _i64, _f, _u = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint
SIGNATURES = {
    "egr_last_error": (C.c_char_p, []),
    "egr_abi_version": (_i, []),
    "egr_device_count": (_i, []),
    "egr_device_arch": (_i, [_i, C.c_char_p, C.c_size_t]),
    "egr_fatllama_plan_query": (_i, [_i64, _i, _i, C.POINTER(_i64)]),
    "egr_fatllama_plan_create": (_i, [C.POINTER(_vp), _i64, _i, _i, _i, _i]),
    "egr_fatllama_plan_create_n": (_i, [C.POINTER(_vp), _i64, _i64, _i]),
    "egr_fatllama_plan_create_ex": (_i, [C.POINTER(_vp), _i64, _i, _i, _i, _i, _i, _i]),
    "egr_fatllama_plan_create_bluestein": (_i, [C.POINTER(_vp), _i64, _i, _i]),
    "egr_fatllama_plan_create_chirpz": (_i, [C.POINTER(_vp), _i64, _i, _i, _i]),
    "egr_fatllama_plan_destroy": (_i, [_vp]),
    "egr_fatllama_enhance": (_i, [_vp, _vp, _vp, _i, _f, _u, _vp]),
    "egr_fatllama_joint_peak": (_i, [_vp, _u, _vp, _vp]),
    "egr_fatllama_finalize": (_i, [_vp, _vp, _u, _vp, _vp]),
    "egr_spectral_gain": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "egr_fatllama_last_peaks": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), _vp]),
    "egr_fatllama_set_profiling": (_i, [_vp, _i]),
    "egr_fatllama_kernel_times": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64),
                                       C.POINTER(_i64)]),
    "egr_fatllama_kernel_times3": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "egr_lds_canary_failures": (C.c_longlong, []),
    "egr_lds_canary_selftest": (C.c_longlong, [_i]),
    "egr_pcm16_roundtrip": (_i, [_vp, _vp, _i64, _f, _f, _vp]),
    "egr_stft_mag": (_i, [_vp, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "egr_dfn_workspace_bytes": (C.c_size_t, [_i, _i64]),
    "egr_dfn_vad_gains": (_i, [_vp, _i, _i64, C.c_double, _i, C.c_double, C.c_double, C.c_double, _i, _vp, _vp, _vp, _vp]),
    "egr_dfn_mix": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i64, _i, _f, _i, _i, C.c_double, _vp, _vp, _vp]),
    "egr_shift_fir": (_i, [_vp, _i, _i64, _i64, _vp, _i, _vp, _i64, _vp]),
    "egr_gcc_phat": (_i, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    "egr_band_filter": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "egr_kweight": (_i, [_vp, _i, _i64, _f, _f, _vp, _vp]),
    "egr_mono_mean": (_i, [_vp, _i, _i64, _i64, _vp, _vp]),
    "egr_frame_meansq": (_i, [_vp, _i, _i64, _i64, _i64, _i64, _vp, _vp]),
    "egr_pair_stats": (_i, [_vp, _i, _i64, _vp, _i, _i64, _i64, _f, _i, _vp, _vp]),
    "egr_null_mix": (_i, [_vp, _i64, _vp, _i64, _i, _i64, _f, _i, _i, _vp, _vp, _vp]),
    "egr_band_sums": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "egr_resample_linear": (_i, [_vp, _i, _i64, _vp, _i64, _vp]),
    "egr_lsd_frames": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "egr_sum_f64": (_i, [_vp, _i64, _vp, _vp]),
    "egr_order_stats2": (_i, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "egr_si_sdr_terms": (_i, [_vp, _i, _i64, _vp, _i, _i64, _i64, _vp, _vp]),
    "egr_wola_stitch": (_i, [_vp, _i, _i, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "egr_chunk_gather": (_i, [_vp, _i, _i64, _i64, _i64, _i, _i, _vp, _vp]),
    "egr_resample_poly": (_i, [_vp, _i, _i64, _i, _i, _vp, _i, _vp, _i64, _vp]),
    "egr_conv_nhwc": (_i, [_vp] * 6 + [_i] * 15 + [_f, _vp]),
    "egr_conv_nhwc_placed": (_i, [_vp] * 6 + [_i] * 15 + [_f] + [_i] * 6 + [_vp]),
    "egr_split3_pack": (_i, [_vp, _vp, _i64, _i, _vp]),
    "egr_conv_s3": (_i, [_vp] * 6 + [_i] * 15 + [_f] + [_i] * 7 + [_i64] * 3 + [_vp]),
    "egr_split2h_pack": (_i, [_vp, _vp, _i64, _i, _f, _vp]),
    "egr_absmax": (_i, [_vp, _i64, _vp, _vp]),
    "egr_absmax_rows": (_i, [_vp, _i, _i64, _i, _i64, _vp, _vp]),
    "egr_conv_h2_gn": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "egr_gn_operand_bound": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "egr_conv_h2": (_i, [_vp] * 6 + [_i] * 15 + [_f] + [_i] * 7 + [_i64] * 3 + [_f, _vp, _i, _vp, _vp]),
    "egr_winograd_input": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "egr_groupnorm_coeff": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "egr_conv_nhwc_gn": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "egr_gemm_zbatched": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _vp]),
    "egr_winograd_output": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "egr_winograd4_g": (_i, [C.POINTER(C.c_double)]),
    "egr_winograd4_input": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "egr_winograd4_input_ra": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "egr_winograd4_output": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "egr_winograd4_output_stats": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "egr_winograd4_output_ra": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "egr_groupnorm_stats_from_partials": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "egr_groupnorm_coeff_from_stats": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "egr_bgemm_nt_s3": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_i64] * 6 + [_f, _vp]),
    "egr_bgemm_nt_h2": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_i64] * 6 + [_f, _vp, _vp, _vp, _vp]),
    "egr_bgemm": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_i64] * 6 + [_i, _f, _vp]),
    "egr_groupnorm_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "egr_groupnorm_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "egr_groupnorm_nhwc_ra": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "egr_layernorm_rows_ra": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _i, _vp, _vp]),
    "egr_eltwise_ra": (_i, [_vp, _vp, _vp, _i64, _i, _f, _f, _i, _vp, _vp]),
    "egr_geglu_ra": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "egr_concat_channels_ra": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "egr_layernorm_rows": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    "egr_softmax_rows": (_i, [_vp, _i64, _i, _vp]),
    "egr_eltwise": (_i, [_vp, _vp, _vp, _i64, _i, _f, _f, _vp]),
    "egr_geglu": (_i, [_vp, _vp, _i64, _i, _vp]),
    "egr_fatllama_set_side_stream": (_i, [_vp, _vp]),
    "egr_fatllama_set_graph": (_i, [_vp, _i]),
    "egr_streams_overlap_us": (_i, [_vp, _vp, _i, _vp]),
    "egr_tap_gather": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "egr_conv_cin1": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "egr_concat_channels": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "egr_transpose_batched": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "egr_snake_aa": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "egr_snake_aa_ra": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "egr_col2im_convtr1d": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "egr_stft_frames": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "egr_lowpass_gain": (_i, [_vp, _i, _i, _i, _i, _f, _f, _i, _f, _i64, _vp, _vp, _vp]),
    "egr_randn": (_i, [_vp, _i64, _i, C.c_uint64, _vp, _vp]),
    "egr_randn_base": (_i, [_vp, _i64, _i, C.c_uint64, _vp, _i64, _vp]),
    "egr_flashsr_default_config": (_i, [_vp]),
    "egr_flashsr_create": (_i, [C.POINTER(_vp), _vp, _vp, _i, _u, _vp]),
    "egr_flashsr_create_from_file": (_i, [C.POINTER(_vp), C.c_char_p, _u, _vp]),
    "egr_flashsr_destroy": (_i, [_vp]),
    "egr_flashsr_infer": (_i, [_vp, _vp, _i, _i, C.c_uint64, _vp, _vp, _vp]),
    "egr_flashsr_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "egr_flashsr_set_rows_per_pass": (_i, [_vp, _i]),
    "egr_flashsr_set_streams": (_i, [_vp, _i, _i]),
    "egr_flashsr_set_split": (_i, [_vp, _i]),
    "egr_flashsr_split_info": (_i, [_vp, _vp, _vp, _vp]),
    "egr_flashsr_set_arena_cap": (_i, [_vp, C.c_double]),
    "egr_flashsr_set_profiling": (_i, [_vp, _i]),
    "egr_flashsr_profile": (_i, [_vp, _i, C.c_char_p, C.c_size_t, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.POINTER(_i)]),
    "egr_flashsr_flop_count": (_i, [_vp, _i, C.POINTER(C.c_double), _vp]),
    "egr_flashsr_scratch_bytes": (_i64, [_vp]),
    "egr_flashsr_warmup": (_i, [_vp, _i, _vp]),
    "egr_amp_unit_h2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "egr_pack_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "egr_phase_weights": (_i, [_vp, _vp, _i, _i, _vp]),
    "egr_winograd_pack_u": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
}

FSR_MAX = 8
FSR_F32_MFMA, FSR_NO_WINOGRAD, FSR_NO_WINO_F4, FSR_NO_GN_PARTIALS, FSR_NO_THIN_ENDS, FSR_NO_FUSE_GN = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
FSR_SPLIT_BF16X3 = 0x40


class FlashSRConfigC(C.Structure):
    """egr_flashsr_config (include/egregora_amd.h)."""
    _A = _i * FSR_MAX
    _fields_ = [("struct_bytes", _i), ("sr", _i), ("chunk", _i), ("n_fft", _i), ("hop", _i), ("n_mels", _i), ("n_frames", _i),
                ("fmin", _f), ("fmax", _f), ("log_floor", _f),
                ("vae_ch", _i), ("vae_levels", _i), ("vae_mult", _A), ("vae_res", _i), ("z_ch", _i), ("gn_groups", _i),
                ("unet_ch", _i), ("unet_levels", _i), ("unet_mult", _A), ("unet_res", _i), ("unet_n_attn", _i), ("unet_attn_ds", _A),
                ("head_dim", _i), ("t_steps", _i),
                ("voc_ch", _i), ("voc_n_rates", _i), ("voc_rates", _A), ("voc_n_kernels", _i), ("voc_kernels", _A),
                ("voc_n_dils", _i), ("voc_dils", _A), ("aa_taps", _i)]


class TensorDescC(C.Structure):
    """egr_tensor_desc."""
    _fields_ = [("name", C.c_char_p), ("data", _vp), ("ndim", _i), ("shape", _i64 * 4)]


def flashsr_config_c(cfg) -> FlashSRConfigC:
    """flashsr_arch.FlashSRConfig -> egr_flashsr_config."""
    c = FlashSRConfigC()
    c.struct_bytes = C.sizeof(FlashSRConfigC)
    for f in ("sr", "chunk", "n_fft", "hop", "n_mels", "n_frames", "fmin", "fmax", "log_floor", "vae_ch", "vae_res", "z_ch", "gn_groups",
              "unet_ch", "unet_res", "head_dim", "t_steps", "voc_ch", "aa_taps"):
        setattr(c, f, getattr(cfg, f))
    for name, n_name, vals in (("vae_mult", "vae_levels", cfg.vae_mult), ("unet_mult", "unet_levels", cfg.unet_mult),
                               ("unet_attn_ds", "unet_n_attn", cfg.unet_attn_ds), ("voc_rates", "voc_n_rates", cfg.voc_rates),
                               ("voc_kernels", "voc_n_kernels", cfg.voc_kernels), ("voc_dils", "voc_n_dils", cfg.voc_dils)):
        if len(vals) > FSR_MAX:
            raise RuntimeError(f"FlashSRConfig.{name} has {len(vals)} entries; the C ABI holds {FSR_MAX}")
        setattr(c, n_name, len(vals))
        arr = getattr(c, name)
        for i, v in enumerate(vals):
            arr[i] = int(v)
    return c


def lib():
    """Load the shared library (once).  Raises RuntimeError with build instructions when absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = Path(os.environ.get("EGREGORA_AMD_LIB") or str(LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"libegregora_amd.so not found at {path}. Build it with `make -C {LIB_PATH.parent / 'csrc'}` "
            "(hipcc, --offload-arch=gfx950). This pack has no CPU fallback.")
    try:
        L = C.CDLL(str(path))
    except OSError as e:
        raise RuntimeError(f"failed to load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)            # AttributeError here = ABI mismatch, which must be loud
        fn.restype, fn.argtypes = res, args
    if L.egr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libegregora_amd.so ABI {L.egr_abi_version()} != {ABI_VERSION} (stale build: run `make -C csrc`)")
    _LIB = L
    return L


def last_error() -> str:
    return (lib().egr_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != EGR_OK:
        raise RuntimeError(f"{what} failed (status {rc}): {last_error()}")


def require_device():
    """Loud check that an MI355X (gfx950) is visible.  Stands in for _ensure_gpu_stack
    (reference egregora_fat_llama_gpu.py:132-159), minus every CuPy / NVIDIA-wheel concern."""
    import torch
    L = lib()
    n = L.egr_device_count()
    if n <= 0 or not torch.cuda.is_available():
        raise RuntimeError("No AMD GPU detected. This pack's compute nodes require an MI355X (gfx950) with ROCm; "
                           "there is no CPU fallback. " + (last_error() if n < 0 else ""))
    buf = C.create_string_buffer(256)
    check(L.egr_device_arch(torch.cuda.current_device(), buf, 256), "egr_device_arch")
    arch = buf.value.decode()
    if not arch.startswith("gfx950"):
        raise RuntimeError(f"GPU architecture {arch!r} is not gfx950 (MI355X); libegregora_amd.so holds gfx950 code only.")
    return arch


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr())
