"""ctypes binding of ``csrc/libtorchsde_amd.so`` -- the C ABI declared in ``include/torchsde_amd.h``.

This is the only way the Python host code reaches the HIP kernels. There is no CPU fallback: if the
shared library is missing, or a tensor is not on a ROCm device, the call fails loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtorchsde_amd.so")

F32, F64 = 0, 1
KID_STEP_DIAG, KID_STEP_GENERAL, KID_MILSTEIN_DIAG, KID_SRK_STAGE, KID_AUG_UPDATE, KID_BROWNIAN_QUERY = 1, 2, 3, 4, 5, 6
KID_RHEUN, KID_TRAJECTORY, KID_MLP_BACKWARD, KID_MLP_ADJOINT = 7, 8, 9, 10
KID_RHEUN_MLP = 13
ERROR_NORM_WORKSPACE = 1024
TRAJ_SENS = 5
ACT_TANH, ACT_SOFTPLUS, ACT_SILU = 0, 1, 2
DIFF_AFFINE, DIFF_SIGMOID = 0, 1
FINAL_NONE, FINAL_SIGMOID, FINAL_TANH = 0, 1, 2
PRECISION_F32, PRECISION_BF16X3 = 0, 1
NOISE_DIAGONAL, NOISE_SCALAR, NOISE_GENERAL, NOISE_ADDITIVE = 0, 1, 2, 3
TRAJ_EULER, TRAJ_MILSTEIN_ITO, TRAJ_MILSTEIN_STRAT, TRAJ_MIDPOINT, TRAJ_SRK, TRAJ_HEUN, TRAJ_EULER_HEUN = 0, 1, 2, 3, 4, 5, 6
TRAJ_REVERSIBLE_HEUN = 7
FN_CODES = {"identity": 0, "exp": 1, "sigmoid": 2, "tanh": 3, "softplus": 4, "sin": 5, "cos": 6, "poly3": 7}

# include/torchsde_amd.h: the device tables of adaptive stepping (TSDE_CTL_*, TSDE_SUB_*, TSDE_SCAL_*)
ADAPTIVE_MAX_STAGES = 6
CTL_CURR_T, CTL_PREV_T, CTL_STEP_SIZE, CTL_PREV_ERROR_RATIO, CTL_OUT_T, CTL_T_END, CTL_DT_MIN = range(7)
CTL_ATTEMPTS, CTL_ACCEPTED, CTL_DT_MIN_HITS, CTL_NAN_SEEN, CTL_ACTIVE = 7, 8, 9, 10, 11
CTL_BOUNDS_A, CTL_BOUNDS_B, CTL_WIDTHS = 12, 14, 16
CTL_OUT_IDX, CTL_N_OUT, CTL_EMIT_FIRST, CTL_EMIT_COUNT, CTL_SIZE = 18, 19, 20, 21, 22
SUB_DT, SUB_HALF_DT, SUB_SQRT_DT, SUB_RDT, SUB_TIMES = 0, 1, 2, 3, 4
SUB_STRIDE = SUB_TIMES + ADAPTIVE_MAX_STAGES
SCAL_W0, SCAL_W1, SCAL_ACCEPT, SCAL_SIZE = 3 * SUB_STRIDE, 3 * SUB_STRIDE + 1, 3 * SUB_STRIDE + 2, 3 * SUB_STRIDE + 3
DEV_SCALAR_TAG = 0x7FFC

_c_i64 = ctypes.c_int64
_c_u64 = ctypes.c_uint64
_c_u32 = ctypes.c_uint32
_c_i32 = ctypes.c_int32
_c_dbl = ctypes.c_double
_c_ptr = ctypes.c_void_p
_c_int = ctypes.c_int


class NativeLibraryError(RuntimeError):
    pass


class Noise(ctypes.Structure):
    """``tsde_noise_t``."""
    _fields_ = [("dW", _c_ptr), ("dU", _c_ptr), ("entropy", _c_u64), ("elem0", _c_u64), ("cell", _c_u32),
                ("reserved", _c_u32), ("h", _c_dbl), ("bcast_d", _c_i64), ("entropy_dev", _c_ptr)]


class Seg(ctypes.Structure):
    """``tsde_seg_t``."""
    _fields_ = [("out", _c_ptr), ("s", _c_ptr), ("F", _c_ptr), ("G", _c_ptr), ("D", _c_ptr), ("n", _c_i64),
                ("sF", _c_dbl), ("sG", _c_dbl), ("sD", _c_dbl)]


class Traj(ctypes.Structure):
    """``tsde_traj_t``."""
    _fields_ = [("step_rows", _c_ptr), ("cells", _c_ptr), ("out_step", _c_ptr), ("out_w", _c_ptr),
                ("n_steps", ctypes.c_int32), ("n_out", ctypes.c_int32)]


class Mlp(ctypes.Structure):
    """``tsde_mlp_t``."""
    _fields_ = [("w1", _c_ptr), ("w1t", _c_ptr), ("b1", _c_ptr), ("w2", _c_ptr), ("b2", _c_ptr), ("hidden", _c_i32),
                ("out", _c_i32), ("activation", _c_i32), ("final", _c_i32), ("scale", _c_dbl), ("precision", _c_i32),
                ("reserved", _c_i32)]


class DeepMlp(ctypes.Structure):
    """``tsde_deep_mlp_t``."""
    _fields_ = [("w1", _c_ptr), ("w1t", _c_ptr), ("b1", _c_ptr), ("wm", _c_ptr * 2), ("bm", _c_ptr * 2), ("w2", _c_ptr),
                ("b2", _c_ptr), ("hidden", _c_i32), ("out", _c_i32), ("activation", _c_i32), ("final", _c_i32),
                ("n_mid", _c_i32), ("reserved", _c_i32), ("scale", _c_dbl), ("act_scale", _c_dbl)]


class RheunState(ctypes.Structure):
    """``tsde_rheun_state_t``."""
    _fields_ = [("y", _c_ptr), ("z", _c_ptr), ("a_y", _c_ptr), ("a_z", _c_ptr), ("a_f", _c_ptr), ("p", _c_ptr)]


class RheunStash(ctypes.Structure):
    """``tsde_rheun_stash_t``."""
    _fields_ = [("z", _c_ptr), ("cf", _c_ptr), ("p", _c_ptr), ("q", _c_ptr), ("wa", _c_ptr), ("wb", _c_ptr),
                ("hf", _c_ptr * 3), ("df", _c_ptr * 3), ("hg", _c_ptr * 3), ("dg", _c_ptr * 3),
                ("stride_d", _c_i32), ("stride_m", _c_i32), ("stride_hf", _c_i32), ("stride_hg", _c_i32)]


_PTR4 = _c_ptr * 4
_PTR3 = _c_ptr * 3
_PTR5 = _c_ptr * 5

# name -> (restype, argtypes); mirrors include/torchsde_amd.h one to one.
SIGNATURES = {
    "tsde_abi_version": (_c_int, []),
    "tsde_last_error": (ctypes.c_char_p, []),
    "tsde_philox4x32_10": (None, [ctypes.POINTER(_c_u32), ctypes.POINTER(_c_u32), ctypes.POINTER(_c_u32)]),
    "tsde_noise_counter": (None, [_c_u64, _c_u32, _c_u64, _c_u32, ctypes.POINTER(_c_u32)]),
    "tsde_brownian_normals": (_c_int, [_c_ptr, _c_i64, _c_u64, _c_u64, _c_u32, _c_u64, _c_u32, _c_int, _c_ptr]),
    "tsde_brownian_query": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_u64, _c_u64, _c_ptr, _c_i64, _c_i64, _c_dbl,
                                     _c_dbl, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_ptr, _c_int, _c_ptr]),
    "tsde_cell_increment": (_c_int, [_c_ptr, _c_ptr, _c_i64, ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_step_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl, ctypes.POINTER(Noise), _c_int,
                                _c_ptr]),
    "tsde_step_prod": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl, _c_int, _c_ptr]),
    "tsde_step_general": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl,
                                   ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_step_general_w": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _c_dbl,
                                     _c_int, _c_dbl, _c_dbl, _c_dbl, ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_step_shared": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _c_dbl,
                                  _c_int, _c_dbl, _c_dbl, _c_dbl, ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_milstein_v": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_dbl, _c_int, _c_dbl, ctypes.POINTER(Noise), _c_int,
                                 _c_ptr]),
    "tsde_milstein_weight": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_dbl, _c_int, _c_dbl, ctypes.POINTER(Noise), _c_int,
                                      _c_ptr]),
    "tsde_milstein_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, ctypes.POINTER(Noise),
                                    _c_int, _c_ptr]),
    "tsde_milstein_gf_prime": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl, _c_int, _c_int,
                                        _c_ptr]),
    "tsde_milstein_gf_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl, _c_int,
                                       ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_srk_diag_stage": (_c_int, [_c_int, _PTR3, _PTR5, _c_i64, _c_dbl, _c_dbl, _c_dbl, ctypes.POINTER(Noise), _c_int,
                                     _c_ptr]),
    "tsde_milstein_gf_general_support": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_dbl,
                                                  _c_int, _c_int, _c_ptr]),
    "tsde_milstein_gf_general_correction": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_dbl,
                                                     _c_int, _c_ptr]),
    "tsde_heun_final": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_int, _c_int,
                                 ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_iterated_integrals": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_dbl, _c_int, _c_int, _c_ptr]),
    "tsde_levy_area": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_dbl, _c_int, _c_u64, _c_u64, _c_u32, _c_u64,
                                _c_ptr, _c_int, _c_ptr]),
    "tsde_levy_iterated_integrals": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_dbl, _c_int, _c_u64, _c_u64,
                                              _c_u32, _c_u64, _c_ptr, _c_dbl, _c_int, _c_int, _c_ptr]),
    "tsde_rheun_z_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl,
                                   ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_rheun_y_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl,
                                   ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_lincomb2": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl, _c_int, _c_ptr]),
    "tsde_rheun_adj_a_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl,
                                       ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_rheun_adj_b_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl,
                                       ctypes.POINTER(Noise), _c_int, _c_ptr]),
    "tsde_aug_update": (_c_int, [ctypes.POINTER(Seg), _c_int, _c_dbl, _c_dbl, _c_int, _c_ptr]),
    "tsde_linear_interp": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl, _c_int, _c_ptr]),
    "tsde_error_norm": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_dbl, _c_dbl, _c_dbl, _c_int, _c_ptr]),
    "tsde_trajectory_affine_diag": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int,
                                             ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_trajectory_affine_diag_timed": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                                   _c_int, ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_trajectory_expr_diag_timed": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr * 8, _c_i64, _c_int, _c_int, _c_int,
                                                 ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_trajectory_affine_diag_sens": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                                  _c_ptr, _c_int, ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int,
                                                  _c_ptr]),
    "tsde_trajectory_mlp_diag": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                          _c_ptr, _c_int, _c_dbl, _c_int, _c_int, ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int,
                                          _c_ptr]),
    "tsde_trajectory_prog_diag": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i32, _c_i32, _c_i32, _c_ptr, _c_i32, _c_int,
                                           _c_int, ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_trajectory_prog_diag_sens": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i32, _c_i32, _c_i32, _c_ptr, _c_i32,
                                                _c_ptr, _c_int, _c_int, ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int,
                                                _c_ptr]),
    "tsde_trajectory_prog_additive": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i32, _c_ptr, _c_i32, _c_ptr,
                                               _c_int, _c_int, ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_trajectory_mlp_additive": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, ctypes.POINTER(Mlp), _c_ptr, _c_int, _c_int,
                                              ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_trajectory_mlp_general": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, ctypes.POINTER(Mlp),
                                             ctypes.POINTER(Mlp), _c_int, ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int,
                                             _c_ptr]),
    "tsde_trajectory_mlp_general_lds": (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int]),
    "tsde_trajectory_mlp_diag_backward": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i32, _c_ptr,
                                                   _c_ptr, _c_i32, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                                   _c_int, _c_dbl, _c_int, _c_int, ctypes.POINTER(Traj), _c_i32, _c_i32, _c_u64, _c_u64,
                                                   _c_ptr, _c_int, _c_ptr]),
    "tsde_trajectory_expr_diag": (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr * 8, _c_int, _c_int, _c_int,
                                           ctypes.POINTER(Traj), _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_adjoint_mlp_diag": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64,
                                       _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_dbl, _c_int, _c_int,
                                       ctypes.POINTER(Traj), _c_i32, _c_i32, _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_rheun_mlp_forward": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, ctypes.POINTER(DeepMlp),
                                        ctypes.POINTER(DeepMlp), ctypes.POINTER(Traj), _c_ptr, _c_u64, _c_u64, _c_ptr, _c_int,
                                        _c_ptr]),
    "tsde_deep_mlp_forward": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, ctypes.POINTER(DeepMlp),
                                       ctypes.POINTER(DeepMlp), _c_int, ctypes.POINTER(Traj), _c_ptr, _c_u64, _c_u64, _c_ptr, _c_int,
                                       _c_ptr]),
    "tsde_rheun_mlp_lds": (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int]),
    "tsde_rheun_mlp_backward": (_c_int, [ctypes.POINTER(RheunState), ctypes.POINTER(RheunStash), _c_ptr, _c_ptr, _c_i64, _c_i64,
                                         _c_i64, _c_int, ctypes.POINTER(DeepMlp), ctypes.POINTER(DeepMlp),
                                         ctypes.POINTER(Traj), _c_ptr, _c_i32, _c_i32, _c_u64, _c_u64, _c_ptr, _c_int, _c_ptr]),
    "tsde_rheun_last_layer_grad": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64,
                                            ctypes.POINTER(DeepMlp), _c_i32, _c_i32, _c_i32, _c_i32, _c_int, _c_ptr]),
    "tsde_gram_partials": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_i32, _c_int, _c_ptr]),
    "tsde_adaptive_begin": (_c_int, [_c_ptr, _c_ptr, _c_dbl, ctypes.POINTER(_c_dbl), _c_int, _c_int, _c_ptr]),
    "tsde_adaptive_control": (_c_int, [_c_ptr, _c_ptr, _c_ptr, ctypes.POINTER(_c_dbl), _c_int, _c_int, _c_ptr]),
    "tsde_adaptive_begin_outputs": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i32, ctypes.POINTER(_c_dbl), _c_int, _c_int, _c_ptr]),
    "tsde_adaptive_control_outputs": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i32, ctypes.POINTER(_c_dbl), _c_int,
                                               _c_int, _c_ptr]),
    "tsde_adaptive_emit": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_int, _c_ptr]),
    "tsde_adaptive_commit": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int, _c_ptr]),
    "tsde_merge_halves": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_dbl, _c_dbl, _c_int,
                                   _c_ptr]),
    "tsde_brownian_query_dev": (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_u64, _c_u64, _c_ptr, _c_i64, _c_ptr, _c_int,
                                         _c_int, _c_ptr, _c_int, _c_ptr]),
    "tsde_prof_begin": (_c_int, [_c_int, _c_int]),
    "tsde_prof_bracket_open": (_c_int, [_c_int, _c_ptr]),
    "tsde_prof_bracket_close": (_c_int, [_c_int, _c_ptr]),
    "tsde_delay_us": (_c_int, [_c_dbl, _c_ptr]),
    "tsde_graph_memset_nodes_to_kernels": (_c_int, [_c_ptr, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "tsde_prof_bracket_overhead": (_c_int, [_c_int, _c_dbl, ctypes.POINTER(_c_dbl), _c_ptr]),
    "tsde_prof_read": (_c_int, [ctypes.POINTER(_c_dbl), _c_int, ctypes.POINTER(_c_int)]),
    "tsde_prof_end": (_c_int, [ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_i64)]),
}

_lib = None


def load():
    """Load the shared library (once) and type every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"torchsde_amd: HIP extension not built ({LIB_PATH} is missing). Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C torchsde_amd/csrc`. "
            f"There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 not found
        raise NativeLibraryError(f"torchsde_amd: cannot load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"torchsde_amd: {LIB_PATH} does not export `{name}`; rebuild it.") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.tsde_abi_version() != 2:
        raise NativeLibraryError("torchsde_amd: ABI version mismatch between the Python host code and the .so")
    _lib = lib
    return lib


def is_built():
    return os.path.exists(LIB_PATH)


def dtype_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.float64:
        return F64
    raise ValueError(f"torchsde_amd supports float32 and float64 state, got {dtype}.")


def require_device(*tensors):
    """The product path runs on the GPU only."""
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise NativeLibraryError(
                "torchsde_amd: tensors must live on a ROCm device (got device "
                f"'{t.device}'). This package is the MI355X hot path of torchsde and has no CPU fallback.")


def dev_scalar(tensor, index=0):
    """TSDE_DEV_SCALAR: the `double` that tells a step kernel to read its coefficient from element `index` of the
    device tensor `tensor` (of the launch's dtype) instead of taking a value: a quiet NaN whose low 48 bits are the
    address. It is an ordinary Python float, so it passes through every wrapper that takes a coefficient."""
    import struct
    address = tensor.data_ptr() + index * tensor.element_size()
    if not 0 < address < (1 << 48):
        raise NativeLibraryError(f"torchsde_amd: device address {address:#x} does not fit a boxed scalar")
    return struct.unpack("<d", struct.pack("<Q", (DEV_SCALAR_TAG << 48) | address))[0]


class on_device_of:
    """Make the device of `tensor_or_device` current for the duration of the block. Launches go to that device's
    current stream (`stream_ptr`, kernels._launch_env); HIP wants the launching thread's current device to be the
    stream's, so every public entry point (sdeint, sdeint_adjoint, BrownianInterval queries) runs inside this guard.
    Costs two cheap calls when the device is already current."""

    def __init__(self, tensor_or_device):
        dev = tensor_or_device.device if torch.is_tensor(tensor_or_device) else torch.device(tensor_or_device)
        self._index = dev.index if dev.type == "cuda" else None
        self._prev = None

    def __enter__(self):
        if self._index is not None:
            prev = torch.cuda.current_device()
            if prev != self._index:
                self._prev = prev
                torch.cuda.set_device(self._index)
        return self

    def __exit__(self, *exc):
        if self._prev is not None:
            torch.cuda.set_device(self._prev)
        return False


def stream_ptr(device=None):
    """The HIP stream torch is currently issuing work on (so launches order with the user's f/g ops
    and are captured by an enclosing HIP graph)."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(code, what):
    if code != 0:
        lib = load()
        msg = lib.tsde_last_error().decode("utf-8", "replace")
        raise NativeLibraryError(f"torchsde_amd: {what} failed with hipError {code}: {msg}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def contiguous(t):
    return t if t.is_contiguous() else t.contiguous()
