"""ctypes binding of libsvb_hip.so (C ABI declared in include/svb_hip.h).

The HIP library is the only compute path of this package: `get_lib()` raises if it has not been
built (`python -c "import __graft_entry__ as g; g.build()"`); there is no CPU or eager fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsvb_hip.so")

c_fp = C.c_void_p  # device pointers are passed as integers


class SvbConvEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p), ("in_gate", C.c_void_p), ("out_gate", C.c_void_p),
        ("residual", C.c_void_p), ("mask", C.c_void_p),
        ("in_slope", C.c_float), ("out_slope", C.c_float), ("out_gate_slope", C.c_float),
        ("out_act", C.c_int), ("force_cfg", C.c_int), ("x_q", C.c_void_p),
    ]


SVB_WN_MAX_LAYERS = 16


class SvbWnLayer(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("in_a_hi", "in_a_lo", "in_b_hi", "in_b_lo", "rs_a_hi", "rs_a_lo", "rs_b_hi", "rs_b_lo",
                                           "in_bias", "rs_bias", "in_v", "in_g", "rs_v", "rs_g", "d_in_v", "d_in_g", "d_in_b",
                                           "d_rs_v", "d_rs_g", "d_rs_b")] +
                [(n, C.c_int) for n in ("rs_cout", "cfg_in_fwd", "cfg_rs_fwd", "cfg_in_bwd", "cfg_rs_bwd")])


class SvbWnStack(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("B", "C", "T", "n_layers", "k", "dil_rate", "g_channels")] +
                [(n, C.c_void_p) for n in ("x0", "mask", "G", "xbuf", "xin", "acts")] +
                [("layer", SvbWnLayer * SVB_WN_MAX_LAYERS)])


class SvbWnBackward(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("dout", "dx", "dG", "drs", "dxin", "dacts", "dxm", "arena")] +
                [("arena_floats", C.c_size_t), ("need_dx0", C.c_int)])


SVB_CT_MAX_BLOCKS = 4


class SvbCtBlock(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("a_hi", "a_lo", "b_hi", "b_lo", "bias", "keep", "gamma", "beta")] + [("eps", C.c_float)] +
                [(n, C.c_int) for n in ("cout", "cfg_fwd", "cfg_bwd")] +
                [(n, C.c_void_p) for n in ("y4", "out", "stats", "dy4", "dgb", "dx4", "d_weight", "d_bias")])


class SvbCriticTower(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("nb", "N", "C", "H", "W", "has_slope")] + [("slope", C.c_float)] +
                [(n, C.c_void_p) for n in ("x4", "score_w", "score_b", "score")] + [("blk", SvbCtBlock * SVB_CT_MAX_BLOCKS)])


class SvbCtBackward(C.Structure):
    _fields_ = [("ds", C.c_void_p), ("ds_stride", C.c_long), ("dh", C.c_void_p), ("d_score_w", C.c_void_p),
                ("d_score_b", C.c_void_p), ("ws", C.c_void_p), ("ws_floats", C.c_size_t)]


class SvbPackDesc(C.Structure):
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("qa_hi", C.c_void_p), ("qa_lo", C.c_void_p), ("qb_hi", C.c_void_p),
                ("qb_lo", C.c_void_p), ("d0", C.c_int), ("d1", C.c_int), ("k", C.c_int), ("groups", C.c_int),
                ("weight_norm", C.c_int), ("row_start", C.c_int)]


class SvbReduceDesc(C.Structure):
    _fields_ = [("part", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p), ("dv", C.c_void_p), ("dg", C.c_void_p),
                ("bias_part", C.c_void_p), ("db", C.c_void_p), ("nsplit", C.c_int), ("rows", C.c_int), ("rowlen", C.c_int),
                ("weight_norm", C.c_int), ("accumulate", C.c_int), ("row_start", C.c_int)]


class SvbL1Pair(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("da", C.c_void_p), ("db", C.c_void_p), ("n", C.c_long),
                ("scale", C.c_float), ("block0", C.c_int), ("mode", C.c_int), ("target", C.c_float)]


I, F, P, SZ, I64 = C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_int64

# name -> (restype, argtypes)   -- must mirror include/svb_hip.h exactly
SIGNATURES = {
    "svb_abi_version": (I, []),
    "svb_adamw_flat_workspace_floats": (I, []),
    "svb_adamw_flat": (I, [P, P, P, P, SZ, F, F, F, F, F, F, F, F, P, P, P]),
    "svb_gather_segments": (I, [P, P, P, I, P, P]),
    "svb_spectral_norm_workspace_floats": (SZ, [I, I]),
    "svb_spectral_norm_fwd": (I, [P, P, P, P, P, P, P, I, I, I, F, P, P]),
    "svb_spectral_norm_bwd": (I, [P, P, P, P, P, P, I, I, P, P]),
    "svb_batchnorm_nct_fwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, F, P]),
    "svb_batchnorm_nct_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    "svb_weight_pack": (I, [P, P, P, P, I, I, I, I, P]),
    "svb_conv1d_forward": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, C.POINTER(SvbConvEpilogue), P]),
    "svb_conv1d_transposed": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, C.POINTER(SvbConvEpilogue), P]),
    "svb_conv1d_pick_cfg": (I, [I, I, I]),
    "svb_conv1d_taps": (I, [P, P, P, I, I, I, I, I, I, C.POINTER(I), C.POINTER(SvbConvEpilogue), P]),
    "svb_conv1d_taps_bf16x3": (I, [P, P, P, P, I, I, I, I, I, I, C.POINTER(I), C.POINTER(SvbConvEpilogue), P]),
    "svb_weight_pack_bf16x3": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "svb_weight_pack_bf16x3_multi": (I, [P, I, I, P]),
    "svb_conv1d_forward_bf16x3": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, C.POINTER(SvbConvEpilogue), P]),
    "svb_conv1d_transposed_bf16x3": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, C.POINTER(SvbConvEpilogue), P]),
    "svb_conv1d_wgrad_workspace_floats": (SZ, [I, I, I, I, I, I, I, C.POINTER(I)]),
    "svb_conv1d_wgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, P, F, P, F, I, P]),
    "svb_conv1d_wgrad_bf16x3_workspace_floats": (SZ, [I, I, I, I, I, I, I, I, I, C.POINTER(I)]),
    "svb_conv1d_wgrad_bf16x3": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, P, F, P, F, I, P, P]),
    "svb_wgrad_reduce": (I, [P, I, P, P, P, P, I, I, I, I, P, P, P]),
    "svb_wgrad_reduce_multi": (I, [P, I, P]),
    "svb_l1_pairs_blocks": (I, [P, I]),
    "svb_l1_pairs_fwd": (I, [P, I, P, P, I, P]),
    "svb_l1_pairs_bwd": (I, [P, I, P, P]),
    "svb_sum_scale": (I, [P, P, P, F, P, C.c_long, P]),
    "svb_bias_grad": (I, [P, P, F, P, I, I, I, P]),
    "svb_wn_gate_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "svb_wn_gate_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "svb_wn_res_skip_bwd": (I, [P, P, P, P, P, P, I, I, I, P]),
    "svb_wn_res_skip": (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    "svb_wn_stack_forward": (I, [P, P, P, P]),
    "svb_wn_stack_backward": (I, [P, P, P, P]),
    "svb_critic_tower_forward": (I, [P, P]),
    "svb_critic_tower_backward": (I, [P, P, P, P]),
    "svb_split_q": (I, [P, P, P, I, I, I, P]),
    "svb_layernorm_fwd": (I, [P, P, P, P, P, P, I, I, F, P]),
    "svb_relpos_softmax": (I, [P, P, P, P, I, I, I, F, P]),
    "svb_relpos_attn_fwd": (I, [P, P, P, C.c_long, P, P, C.c_long, C.c_long, C.c_long, P, P, I, I, I, I, F, P]),
    "svb_relpos_attn_pos_fwd": (I, [P, P, P, C.c_long, P, P, P, P, P, P, P, I, I, I, I, F, P]),
    "svb_glu_dwconv_bn_swish": (I, [P, P, P, P, P, P, P, F, P, I, I, I, I, P]),
    "svb_layernorm_nct_fwd": (I, [P, P, P, P, I, I, I, F, P]),
    "svb_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "svb_im2col": (I, [P, P] + [I] * 12 + [C.c_long] * 4 + [P]),
    "svb_col2im": (I, [P, P] + [I] * 12 + [C.c_long] * 2 + [P]),
    "svb_s2d_pad": (I, [P, P, I, I, I, I, C.c_long, C.c_long, C.c_long, C.c_long, P]),
    "svb_s2d_pad_bwd": (I, [P, P, I, I, I, I, P]),
    "svb_win_s2d": (I, [P, P, P, P, I, I, I, I, P]),
    "svb_win_s2d_bwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "svb_s2_weight": (I, [P, P, I, I, P]),
    "svb_s2_weight_bwd": (I, [P, P, P, I, I, I, P]),
    "svb_crop_drop_inorm_fwd": (I, [P, P, P, P, C.c_float, P, P, I, I, I, I, I, P]),
    "svb_crop_drop_inorm_bwd": (I, [P, C.c_long, C.c_long, C.c_long, C.c_long, P, P, P, P, P, P, I, I, I, I, I, P]),
    "svb_plane_score_fwd": (I, [P, C.c_long, C.c_long, P, P, P, I, I, I, P]),
    "svb_plane_score_bwd": (I, [P, C.c_long, P, C.c_long, C.c_long, P, P, P, P, I, I, I, P]),
    "svb_ssim_fwd": (I, [P, C.c_long, C.c_long, C.c_long, P, C.c_long, C.c_long, C.c_long, P, I, I, I, F, P]),
    "svb_ssim_bwd": (I, [P, C.c_long, C.c_long, C.c_long, P, C.c_long, C.c_long, C.c_long, P, P, P, I, I, I, F, P]),
    "svb_vae_head_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "svb_vae_head_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "svb_gn_relu_fwd": (I, [P, P, P, P, P, P, I, I, I, I, F, P]),
    "svb_gn_relu_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, P]),
    "svb_mel_loss_fwd": (I, [P, C.c_long, C.c_long, C.c_long, P, C.c_long, C.c_long, C.c_long, P, P, I, I, I, F, I, P]),
    "svb_mel_loss_bwd": (I, [P, C.c_long, C.c_long, C.c_long, P, C.c_long, C.c_long, C.c_long, P, P, P, P, I, I, I, F, I, P]),
    "svb_stft_mel": (I, [P, P, P, P, I, I, I, I, I, I, I, F, P]),
    "svb_nsf_source": (I, [P, P, P, P, P, P, P, P, I, I, I, I, F, F, F, P]),
    "svb_f0_to_coarse_f64": (I, [P, P, I64, P]),
    "svb_f0_to_coarse_f32": (I, [P, P, I64, P]),
    "svb_collate_pad_f32": (I, [P, P, P, P, I, I, I, F, P]),
    "svb_collate_pad_i64": (I, [P, P, P, P, P, I, I, I64, P]),
    "svb_mel_energy": (I, [P, P, P, I, I, I, P]),
    "svb_norm_interp_f0": (I, [P, P, P, P, P, I, I, I, C.c_double, C.c_double, I, P]),
    "svb_f0_shape_hist": (I, [P, P, P, P, I, I, P]),
    "svb_hist_cost": (I, [P, P, P, P, P, I, I, I, P]),
    "svb_dtw_align": (I, [P, P, P, P, P, P, I, I, I, P]),
    "svb_embed_nct_fwd": (I, [P, P, P, I, I, I, I, P]),
    "svb_embed_nct_bwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "svb_period_s2d": (I, [P, P, C.c_long, I, I, I, I, I, I, P]),
    "svb_period_weight": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "svb_upsample_nearest_nct": (I, [P, P, C.c_long, I, I, I, P]),
    "svb_conv_set_single_product": (None, [I]),
    "svb_attn_set_split3": (None, [I]),
    "svb_attn_get_split3": (I, []),
    "svb_conv_get_single_product": (I, []),
}

_LIB = None
_LIB_IS_EMU = False  # only ever set by tests/emu (CPU lane emulator, test infrastructure)


class SvbLibraryMissing(RuntimeError):
    pass


# instrumentation build only (libsvb_hip_instr.so, loaded by tools/): bound when present
INSTR_SIGNATURES = {
    "svb_debug_set_timing_buffer": (None, [P]),
    "svb_debug_set_tw": (None, [P, I, I]),
}


def bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in INSTR_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    return lib


def get_lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SvbLibraryMissing(
                f"{LIB_PATH} not built: the HIP extension is the only compute path of neuralsvb_amd "
                f"(run __graft_entry__.build()).")
        _LIB = bind(LIB_PATH)
    return _LIB


def lib_is_emulator():
    return _LIB_IS_EMU


def device_type():
    """Memory space the bound library's pointers refer to: "cuda" for libsvb_hip.so (the product; nothing else is ever
    loaded by this package).  A harness that binds another build of the same sources says so through _LIB_IS_EMU."""
    return "cpu" if _LIB_IS_EMU else "cuda"


class SvbError(RuntimeError):
    pass


_ERR = {-1: "bad argument", -2: "kernel launch failure", -3: "unsupported shape"}


def check(rc, what):
    if rc != 0:
        raise SvbError(f"{what} failed: {_ERR.get(rc, rc)}")
