"""ctypes binding of libst2_hip.so (the C ABI declared in include/st2.h).

There is no fallback: if the shared library is missing or does not match this binding the import
of any compute entry point raises.  Build it with `python -m styletts2_amd._build`.
"""
import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).  It must be the
# first HIP runtime mapped into the process so that libst2_hip.so binds to the SAME runtime instance torch uses
# (streams and device pointers are shared); loading ours first makes two HSA runtimes fight over the device
# ("no ROCm-capable device is detected").
import torch  # noqa: F401,E402  (deliberately before the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libst2_hip.so")
ABI_VERSION = 22
HEADROOM_COLS, CALIBRATION_COLS = 12, 5  # st2.h ST2_HEADROOM_COLS / ST2_CALIBRATION_COLS

f32p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())

PRO_NONE, PRO_LEAKY, PRO_ADAIN_LEAKY, PRO_ADAIN_SNAKE, PRO_SNAKE, PRO_COLNORM = range(6)
STATUS_F16_RANGE, STATUS_LSTM_TIMEOUT, STATUS_DURATION_SUM, STATUS_LSTM_RECOVERED = 1, 2, 4, 8
ACT_NONE, ACT_GELU, ACT_EXP_SIN, ACT_TANH, ACT_LEAKY, ACT_GELU_TANH = range(6)


class ConvDesc(C.Structure):
    """Mirror of `st2_conv_desc` (include/st2.h); size is cross-checked at load time."""
    _fields_ = [
        ("B", C.c_int32), ("C_in", C.c_int32), ("C_out", C.c_int32), ("L_in", C.c_int32),
        ("L_out", C.c_int32), ("ks", C.c_int32), ("dil", C.c_int32), ("pad_left", C.c_int32),
        ("x", f32p), ("x_bs", C.c_int64), ("x_cs", C.c_int32),
        ("wt", f32p), ("w_ld", C.c_int32),
        ("bias", f32p),
        ("y", f32p), ("y_bs", C.c_int64), ("y_cs", C.c_int32),
        ("pro", C.c_int32),
        ("slope", C.c_float),
        ("stats", f32p),
        ("gamma", f32p), ("beta", f32p),
        ("gb_bs", C.c_int64), ("gamma_plus_one", C.c_int32),
        ("alpha", f32p),
        ("res", f32p), ("res_bs", C.c_int64), ("res_cs", C.c_int32), ("res_shift", C.c_int32),
        ("res2", f32p), ("res2_bs", C.c_int64), ("res2_cs", C.c_int32),
        ("div", C.c_float),
        ("act", C.c_int32), ("act_split", C.c_int32), ("act_slope", C.c_float),
        ("wq", f32p), ("wq_co_pad", C.c_int32), ("wq_cin_pad", C.c_int32),
        ("x_scale", C.c_float), ("out_scale", C.c_float),
        ("w_row_scale", f32p),
        ("xs", f32p), ("xs_cg", C.c_int32), ("xs_lp", C.c_int32), ("xs_halo", C.c_int32),
        ("part", f32p), ("part_nt", C.c_int32), ("part_cols", C.c_int32),
        ("splitk_ws", f32p), ("splitk_ws_bytes", C.c_int64),
    ]


class ModelConfig(C.Structure):
    """Mirror of `st2_model_config` (include/st2.h)."""
    _fields_ = [
        ("decoder_kind", C.c_int32), ("dim_in", C.c_int32), ("style_dim", C.c_int32),
        ("upsample_initial_channel", C.c_int32),
        ("n_upsamples", C.c_int32), ("upsample_rates", C.c_int32 * 4), ("upsample_kernel_sizes", C.c_int32 * 4),
        ("n_resblock_kernels", C.c_int32), ("resblock_kernel_sizes", C.c_int32 * 4),
        ("resblock_dilations", (C.c_int32 * 3) * 4),
        ("gen_istft_n_fft", C.c_int32), ("gen_istft_hop", C.c_int32),
        ("multispeaker", C.c_int32),
        ("dn_layers", C.c_int32), ("dn_heads", C.c_int32), ("dn_head_features", C.c_int32),
        ("dn_multiplier", C.c_int32), ("dn_channels", C.c_int32), ("dn_embedding", C.c_int32),
        ("dn_context_features", C.c_int32), ("dn_max_length", C.c_int32),
        ("pred_hidden", C.c_int32),
        ("bert_layers", C.c_int32), ("bert_ln_eps", C.c_float),
    ]


class FrontArgs(C.Structure):
    """Mirror of `st2_front_args`."""
    _fields_ = [("tokens", C.c_void_p), ("lengths", C.c_void_p), ("noise", C.c_void_p), ("step_noise", C.c_void_p),
                ("ref_s", C.c_void_p), ("s_prev", C.c_void_p),
                ("B", C.c_int32), ("N", C.c_int32), ("steps", C.c_int32), ("tail", C.c_int32),
                ("embedding_scale", C.c_double), ("table", C.POINTER(C.c_double)), ("sigma0", C.c_double),
                ("alpha", C.c_double), ("beta", C.c_double), ("t", C.c_double),
                ("t_en", C.c_void_p), ("d_cm", C.c_void_p), ("s", C.c_void_p), ("ref", C.c_void_p),
                ("s_pred_out", C.c_void_p), ("durations", C.c_void_p), ("carry", C.c_int32)]


class DecoderTaps(C.Structure):
    """Mirror of `st2_decoder_taps`."""
    _fields_ = [("encode", f32p), ("front", f32p), ("har_source", f32p), ("har", f32p), ("stage", f32p * 4),
                ("spec_phase", f32p)]


SAMPLER_TABLE_COLS = 11
BACKEND_SLOTS = ["conv1d_f16s", "conv1d_xs", "act_split", "stats_finalize", "conv1d_direct", "phase_split",
                 "instnorm_stats", "colnorm_stats", "style_fc", "convt_interleave_stats", "adain_leaky_pool",
                 "har_source", "stft_mag_phase", "istft", "attention_keylen", "add_chanvec", "mean_tokens_len",
                 "axpbypcz", "time_features", "tokens_to_channels", "broadcast_cols", "copy_ncl", "expand_by_durations",
                 "lstm_bidir", "colnorm_apply", "duration_head", "mask_tail", "embed_tokens", "dwconv3x3s2",
                 "avgpool2x2", "dev_alloc", "dev_free", "upload"]  # enum st2_backend_slot

_SIGNATURES = {
    # name: (restype, argtypes)
    "st2_abi_version": (C.c_int, []),
    "st2_last_error": (C.c_char_p, []),
    "st2_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "st2_sizeof_conv_desc": (C.c_int, []),
    "st2_status": (C.c_int, [C.c_int]),
    "st2_conv1d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "st2_conv1d_f16s": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "st2_conv1d_f16s_splitk_bytes": (C.c_int64, [C.POINTER(ConvDesc)]),
    "st2_conv1d_f16s_chunk": (C.c_int, [C.c_int]),
    "st2_conv1d_f16s_co_block": (C.c_int, [C.c_int]),
    "st2_conv1d_f16s_set_variant": (None, [C.c_int]),
    "st2_conv1d_xs": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "st2_act_split": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                f32p, f32p, f32p, C.c_int64, C.c_int32, C.c_int32, f32p, C.c_float, f32p, C.c_int32,
                                C.c_int32, C.c_int32, C.c_void_p]),
    "st2_stats_finalize": (C.c_int, [f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, f32p, C.c_int32, C.c_void_p]),
    "st2_conv1d_direct": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, f32p, f32p, C.c_int64, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p]),
    "st2_phase_split": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "st2_instnorm_stats": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, f32p,
                                     C.c_void_p]),
    "st2_colnorm_stats": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, f32p,
                                    C.c_void_p]),
    "st2_style_fc": (C.c_int, [f32p, C.c_int32, C.c_int32, f32p, f32p, C.c_int32, C.c_int32, f32p, C.c_void_p]),
    "st2_convt_interleave": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, f32p, f32p, C.c_int64, C.c_int32,
                                       f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_void_p]),
    "st2_convt_interleave_stats": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, f32p, f32p, C.c_int64, C.c_int32,
                                             f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, f32p, C.c_int32, C.c_void_p]),
    "st2_adain_leaky_pool": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, f32p, f32p, C.c_int64, C.c_float, f32p,
                                       f32p, f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p]),
    "st2_har_source": (C.c_int, [f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, f32p, f32p, C.c_float,
                                 C.c_float, C.c_float, C.c_float, f32p, f32p, C.c_void_p]),
    "st2_stft_mag_phase": (C.c_int, [f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, C.c_int64, C.c_int32,
                                     C.c_void_p]),
    "st2_istft": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, C.c_int64,
                            C.c_void_p]),
    "st2_attention": (C.c_int, [f32p, f32p, f32p, C.c_int64, C.c_int32, f32p, C.c_int64, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "st2_attention_keylen": (C.c_int, [f32p, f32p, f32p, C.c_int64, C.c_int32, f32p, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "st2_colnorm_apply": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, f32p, f32p, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_float, C.c_void_p, f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p]),
    "st2_lstm_bidir": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, f32p,
                                 C.c_int64, C.c_int32, C.c_void_p]),
    "st2_lstm_coop_scratch_bytes": (C.c_int64, [C.c_int32]),
    "st2_lstm_coop_set_exchange": (C.c_int, [C.c_int]),
    "st2_lstm_coop_set_spin_limit": (C.c_int, [C.c_int]),
    "st2_lstm_coop_set_block": (C.c_int, [C.c_int]),
    "st2_lstm_bidir_coop": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      f32p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "st2_lstm_bidir_coop_recovering": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                 f32p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "st2_add_chanvec": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, C.c_int64, f32p, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "st2_mean_tokens": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_void_p]),
    "st2_mean_tokens_len": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p]),
    "st2_time_features": (C.c_int, [C.c_float, f32p, C.c_int32, C.c_int32, f32p, C.c_void_p]),
    "st2_tokens_to_channels": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, f32p, C.c_int64, C.c_int32,
                                         C.c_void_p]),
    "st2_broadcast_cols": (C.c_int, [f32p, C.c_int64, f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    "st2_copy_ncl": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p]),
    "st2_duration_head": (C.c_int, [f32p, C.c_int64, C.c_int32, f32p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_int32, C.c_void_p, f32p, C.c_void_p]),
    "st2_expand_by_durations": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, f32p, C.c_int64, C.c_int32, C.c_void_p]),
    "st2_create": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(C.c_void_p)]),
    "st2_destroy": (C.c_int, [C.c_void_p]),
    "st2_load_weights": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "st2_finalize_weights": (C.c_int, [C.c_void_p, C.c_int32]),
    "st2_decoder_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "st2_decoder_forward": (C.c_int, [C.c_void_p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int32, C.c_int32, f32p,
                                      C.c_void_p, C.c_int64, C.POINTER(DecoderTaps), C.c_void_p]),
    "st2_embed_tokens": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, f32p, C.c_int32, C.c_int32, f32p, f32p, C.c_void_p, f32p,
                                   C.c_int64, C.c_int32, C.c_void_p]),
    "st2_bert_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "st2_bert_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, f32p, C.c_void_p, C.c_int64,
                                   C.c_void_p]),
    "st2_style_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "st2_style_forward": (C.c_int, [C.c_void_p, C.c_int32, f32p, C.c_int32, C.c_int32, C.c_int32, f32p, C.c_void_p, C.c_int64,
                                    C.c_void_p]),
    "st2_sizeof_front_args": (C.c_int, []),
    "st2_front_workspace_bytes": (C.c_int64, [C.c_void_p, C.POINTER(FrontArgs)]),
    "st2_front_forward": (C.c_int, [C.c_void_p, C.POINTER(FrontArgs), C.c_void_p, C.c_int64, C.c_void_p]),
    "st2_text_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "st2_text_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, f32p, C.c_void_p, C.c_int64,
                                   C.c_void_p]),
    "st2_mask_tail": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "st2_duration_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "st2_duration_forward": (C.c_int, [C.c_void_p, f32p, f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, f32p,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "st2_prosody_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "st2_prosody_forward": (C.c_int, [C.c_void_p, f32p, f32p, C.c_void_p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      f32p, f32p, f32p, C.c_void_p, C.c_int64, C.c_void_p]),
    "st2_sampler_table": (C.c_int, [C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double)]),
    "st2_sampler_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
    "st2_sampler_run": (C.c_int, [C.c_void_p, f32p, f32p, f32p, f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_double, C.POINTER(C.c_double), C.c_double, f32p, C.c_void_p, C.c_int64, f32p,
                                  C.c_void_p]),
    "st2_conv_timing": (C.c_int, [C.c_int]),
    "st2_conv_timing_read": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
    "st2_conv_tune": (C.c_int, [C.c_int]),
    "st2_conv_tune_set": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "st2_conv_tune_read": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
    "st2_probe_box": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32]),
    "st2_probe_cu_health": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_int32)]),
    "st2_probe_mfma_stream": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "st2_debug_headroom": (C.c_int, [C.c_int]),
    "st2_debug_headroom_read": (C.c_int, [C.POINTER(C.c_double), C.c_int32]),
    "st2_conv1d_xs_part_cols": (C.c_int, [C.POINTER(ConvDesc)]),
    "st2_conv1d_f16s_set_splitk": (None, [C.c_int, C.c_int]),
    "st2_calibrate": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "st2_calibration_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int32]),
    "st2_calibration_site_name": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]),
    "st2_calibration_write": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int32]),
    "st2_calibration_scale": (C.c_float, [C.c_float, C.c_int32]),
    "st2_stream_create_cu_mask": (C.c_int, [C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_void_p)]),
    "st2_stream_destroy": (C.c_int, [C.c_void_p]),
    "st2_debug_set_backend": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32]),
    "st2_stft_frames": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p,
                                  C.c_int64, C.c_int32, C.c_void_p]),
    "st2_power_spectrum": (C.c_int, [f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, C.c_int64,
                                     C.c_int32, C.c_void_p]),
    "st2_log_norm": (C.c_int, [f32p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "st2_dwconv3x3s2": (C.c_int, [f32p, C.c_int64, C.c_int64, C.c_int32, f32p, f32p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, f32p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "st2_avgpool2x2": (C.c_int, [f32p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p,
                                 C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "st2_axpbypcz": (C.c_int, [f32p, C.c_float, f32p, C.c_float, f32p, C.c_float, f32p, C.c_int64, C.c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


class St2Error(RuntimeError):
    pass


def load():
    """Loads (once) and returns the ctypes handle; raises St2Error if the HIP library is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise St2Error("libst2_hip.so is not built (%s missing). Run `python -m styletts2_amd._build`; "
                       "there is no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise St2Error("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise St2Error("libst2_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    if lib.st2_abi_version() != ABI_VERSION:
        raise St2Error("ABI mismatch: library %d, binding %d" % (lib.st2_abi_version(), ABI_VERSION))
    if lib.st2_sizeof_conv_desc() != C.sizeof(ConvDesc):
        raise St2Error("st2_conv_desc layout mismatch: library %d B, binding %d B"
                       % (lib.st2_sizeof_conv_desc(), C.sizeof(ConvDesc)))
    if lib.st2_sizeof_front_args() != C.sizeof(FrontArgs):
        raise St2Error("st2_front_args layout mismatch: library %d B, binding %d B"
                       % (lib.st2_sizeof_front_args(), C.sizeof(FrontArgs)))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().st2_last_error()
        raise St2Error("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))
