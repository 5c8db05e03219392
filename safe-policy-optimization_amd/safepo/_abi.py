"""ctypes binding of libsafepo_hip.so (C ABI declared in include/safepo_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950) into
safepo/_lib/.  Loading fails LOUDLY: there is no Python/CPU implementation behind these calls.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPO_LIB") or os.path.join(_HERE, "_lib", "libsafepo_hip.so")   # SPO_LIB: debug override


class SpoError(RuntimeError):
    pass


class PpoCfg(Structure):
    """spo_ppo_cfg (include/safepo_hip.h)."""
    _fields_ = [("obs_dim", c_int), ("act_dim", c_int), ("batch", c_int), ("use_critic_norm", c_int),
                ("use_value_coefficient", c_int), ("clip", c_float), ("max_grad_norm", c_float),
                ("lr_actor", c_float), ("lr_critic", c_float), ("beta1", c_float), ("beta2", c_float),
                ("adam_eps", c_float), ("l2_coef", c_float)]


class MaNet(Structure):
    """spo_ma_net (include/safepo_hip.h)."""
    _fields_ = [("in_dim", c_int32), ("hidden", c_int32), ("n_blocks", c_int32), ("out_dim", c_int32),
                ("is_actor", c_int32)]


class MaCollectNet(Structure):
    """spo_ma_collect_net (include/safepo_hip.h): one network of a multi-network collect launch."""
    _fields_ = [("theta", c_void_p), ("net", MaNet), ("deterministic", c_int32), ("x", c_void_p), ("out", c_void_p),
                ("eps", c_void_p), ("act", c_void_p), ("logp", c_void_p), ("std_x_coef", c_float), ("std_y_coef", c_float)]


MA_COLLECT_MAX_NETS = 16
MA_COLLECT_UNSUPPORTED = 1


class MlpNet(Structure):
    """spo_mlp_net (include/safepo_hip.h): n_layers Linear layers, dims[0] = input ... dims[n_layers] = output."""
    _fields_ = [("n_layers", c_int32), ("dims", c_int32 * 6)]

    @classmethod
    def of(cls, sizes):
        sizes = [int(x) for x in sizes]
        if not (2 <= len(sizes) <= 6):
            raise SpoError(f"an MLP here has 1..5 Linear layers, got sizes {sizes}")
        return cls(len(sizes) - 1, (c_int32 * 6)(*(sizes + [0] * (6 - len(sizes)))))


class MaLossCfg(Structure):
    """spo_ma_loss_cfg (include/safepo_hip.h)."""
    _fields_ = [("clip_param", c_float), ("entropy_coef", c_float), ("std_x_coef", c_float), ("std_y_coef", c_float),
                ("use_policy_active_masks", c_int32), ("per_dim_ratio", c_int32)]


P = c_void_p
# name -> (restype, argtypes); must list every symbol declared in include/safepo_hip.h
ACTOR_LOSS_CLIP, ACTOR_LOSS_KL_PENALTY = 0, 1      # include/safepo_hip.h SPO_ACTOR_LOSS_*
MAX_OBS, MAX_ACT, CPO_MAX_OBS, WIDE_MAX_ACT = 128, 16, 64, 64   # SPO_MAX_OBS, SPO_MAX_ACT, cpo.hip's limit, SPO_WIDE_MAX_ACT
WIDE_ACTOR_CLIP, WIDE_ACTOR_SURR, WIDE_ACTOR_KLPEN = 0, 1, 2     # spo_wide_actor_loss modes
GAE_PARTIAL_STRIDE = 16                            # include/safepo_hip.h SPO_GAE_PARTIAL_STRIDE (doubles per workgroup)

ABI_VERSION = 2          # include/safepo_hip.h SPO_ABI_VERSION

PROTOTYPES = {
    "spo_abi_version": (c_int, []),
    "spo_last_error": (c_char_p, []),
    "spo_gae_num_blocks": (c_int, [c_int64, c_int64]),
    "spo_gae_fused": (c_int, [P] * 12 + [c_int64, c_int64, c_double, c_double, c_double, P]),
    "spo_debug_gae_variant": (c_int, [c_int]),
    "spo_gae_fused_timed": (c_int, [P] * 12 + [c_int64, c_int64, c_double, c_double, c_double, c_int, P, P]),
    "spo_adv_reduce": (c_int, [P, c_int, P, P]),
    "spo_adv_apply": (c_int, [P, P, P, P, c_int64, c_double, c_int, c_int, P, P]),
    "spo_policy_step": (c_int, [P] * 12 + [c_int64, c_int64, c_int64, c_int, c_int, P]),
    "spo_obs_normalize": (c_int, [P, P, c_int64, c_int, c_int, P]),
    "spo_policy_step_norm": (c_int, [P, P, P, c_int] + [P] * 10 + [c_int64, c_int64, c_int64, c_int, c_int, P]),
    "spo_values": (c_int, [P, P, P, P, c_int64, c_int, c_int, P]),
    "spo_boundary_step": (c_int, [P] * 18 + [c_int, c_int64, c_int64, c_int64, c_int, P]),
    "spo_boundary_step_fold": (c_int, [P] * 18 + [c_int, c_int64, c_int64, c_int64, c_int, P, P, c_double, P]),
    "spo_boundary_step_fold_mb": (c_int, [P] * 18 + [c_int, c_int64, c_int64, c_int64, c_int, P, P, c_double, P]),
    "spo_mlp_forward_multi": (c_int, [c_int, P, P, P, c_int64, P, P]),
    "spo_mlp_backward_multi": (c_int, [c_int, P, P, P, c_int64, P, P, P, P, P]),
    "spo_wide_grad_rows_supported": (c_int, [POINTER(MlpNet), POINTER(MlpNet), c_int64]),
    "spo_wide_grad_rows_part_floats": (c_int64, [c_int64, c_int64]),
    "spo_wide_ppo_grad_rows": (c_int, [P, POINTER(MlpNet), POINTER(MlpNet)] + [P] * 8 + [c_int64, c_float, P, P]),
    "spo_wide_reduce_parts": (c_int, [P, c_int64, c_int64, c_int, P, P, P]),
    "spo_wide_rows_clip_adam_dev_log": (c_int, [P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_int64, POINTER(PpoCfg), P, P, P, P,
                                                c_int, P, P, c_int64, P]),
    "spo_gather_rows": (c_int, [c_int, P, P, P, P, c_int64, P]),
    "spo_gather_rows_at": (c_int, [c_int, P, P, P, P, P, c_int64, P]),
    "spo_values_boundary_step_fold": (c_int, [P] * 4 + [c_int, c_int] + [P] * 16 + [c_int, c_int64, c_int64, c_int64, c_int, P, P,
                                              c_double, P]),
    "spo_ppo_lag_update_iter": (c_int, [P, P, P, c_int64] + [P] * 7 + [c_int64, POINTER(PpoCfg), P, P, P]),
    "spo_ks_supported": (c_int, [c_int, c_int, c_int]),
    "spo_update_rs_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "spo_ppo_lag_update_iter_ks": (c_int, [P, P, P, c_int64] + [P] * 7 + [c_int64, POINTER(PpoCfg), P, P, P]),
    "spo_ppo_lag_grad_ks": (c_int, [P] * 8 + [c_int, POINTER(PpoCfg), P, P, P, P]),
    "spo_update_iter_ex_ks": (c_int, [P, P, P, c_int64, c_int64, P, P, P, P, P, P, P, c_int64, POINTER(PpoCfg), c_int, P, P,
                                   c_float, c_float, c_int, P, P, P]),
    "spo_critic_fit_ks_supported": (c_int, [c_int, c_int]),
    "spo_critic_fit_iter_ks": (c_int, [P, P, P, c_int64, P, P, P, P, c_int64, POINTER(PpoCfg), P, P, P, P]),
    "spo_update_scratch_release": (c_int, [P, c_int]),
    "spo_debug_update_counters": (c_int, [P, c_int]),
    "spo_debug_crosslane_selftest": (c_int, [P, P, P]),
    "spo_debug_ma_gemm": (c_int, [c_int, c_int, P, P, P, c_int64, c_int, c_int, P]),
    "spo_debug_set_update_profile": (c_int, [P]),
    "spo_ppo_lag_grad": (c_int, [P] * 8 + [c_int, c_int64, POINTER(PpoCfg), P, P, P]),
    "spo_clip_adam": (c_int, [P, P, P, P, c_int64, c_float, POINTER(PpoCfg), P]),
    "spo_clip_adam_then_grad": (c_int, [P, P, P, P, c_int64, c_float] + [P] * 7 + [c_int, POINTER(PpoCfg), P, P]),
    "spo_actor_mean": (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    "spo_actor_kl": (c_int, [P, P, P, P, P, c_int, P, c_int64, c_int, c_int, P]),
    "spo_cpo_num_partials": (c_int, [c_int64]),
    "spo_cpo_surrogate_grad": (c_int, [P] * 5 + [c_float, c_int64, c_int, c_int, P, P, P, P, P]),
    "spo_cpo_fvp": (c_int, [P, P, P, c_int64, c_int, c_int, P, P, P, P]),
    "spo_cpo_linesearch_eval": (c_int, [P] * 8 + [c_int64, c_int, c_int, P, c_int, P, P]),
    "spo_critic_fit_iter": (c_int, [P, P, P, c_int64, P, P, P, P, c_int64, POINTER(PpoCfg), P, P, P, P]),
    "spo_update_iter_ex": (c_int, [P, P, P, c_int64, c_int64, P, P, P, P, P, P, P, c_int64, POINTER(PpoCfg), c_int, P, P,
                                   c_float, c_float, c_int, P, P, P]),
    "spo_p2p_region_bytes": (c_int64, []),
    "spo_debug_xr_profile": (c_int, [P, c_int]),
    "spo_p2p_alloc": (c_int, [POINTER(c_void_p), P]),
    "spo_p2p_open": (c_int, [P, POINTER(c_void_p)]),
    "spo_p2p_close": (c_int, [P]),
    "spo_p2p_free": (c_int, [P]),
    "spo_p2p_selftest": (c_int, [c_int, c_int, POINTER(c_void_p), c_uint32, c_int, P, P]),
    "spo_p2p_select_form": (c_int, [c_int]),
    "spo_p2p_form_valid": (c_int, [c_int, c_int]),
    "spo_p2p_current_form": (c_int, [c_int]),
    "spo_p2p_selftest_one_grid": (c_int, [POINTER(c_void_p), c_uint32, c_int, P, P]),
    "spo_critic_fit_iter_split": (c_int, [P] * 6 + [c_int64, P, P, P, P, P, c_int64, POINTER(PpoCfg)] + [P] * 6
                                  + [POINTER(c_void_p), c_uint32, P]),
    "spo_critic_fit_iter_dp": (c_int, [P, P, P, c_int64, P, P, P, P, c_int64, POINTER(PpoCfg), P, P, P, c_int, c_int,
                                       POINTER(c_void_p), c_uint32, P]),
    "spo_ppo_lag_update_iter_dp": (c_int, [P, P, P, c_int64, P, P, P, P, P, P, P, c_int64, POINTER(PpoCfg), P, P,
                                           c_int, c_int, POINTER(c_void_p), c_uint32, P]),
    "spo_ma_param_count": (c_int64, [POINTER(MaNet)]),
    "spo_ma_param_offset": (c_int64, [POINTER(MaNet), c_int, c_int]),
    "spo_ma_workspace_floats": (c_int64, [POINTER(MaNet), c_int64]),
    "spo_ma_backward_scratch_floats": (c_int64, [POINTER(MaNet), c_int64]),
    "spo_ma_forward": (c_int, [P, POINTER(MaNet), P, c_int64, P, P, P]),
    "spo_ma_backward": (c_int, [P, POINTER(MaNet), P, c_int64, P, P, P, P, P]),
    "spo_ma_insert_step": (c_int, [P, P, P, P, P, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int64,
                                  c_int32, c_int32, c_int32, P]),
    "spo_ma_collect_scratch_floats": (c_int64, [c_int32]),
    "spo_ma_collect_forward": (c_int, [c_int32, POINTER(MaCollectNet), c_int64, P, P]),
    "spo_ma_sample": (c_int, [P, P, P, c_float, c_float, c_int, P, P, c_int64, c_int, P]),
    "spo_ma_log_probs": (c_int, [P, P, P, c_float, c_float, P, c_int64, c_int, P]),
    "spo_ma_actor_loss": (c_int, [P] * 9 + [POINTER(MaLossCfg), c_int64, c_int, c_float, c_int64, P, P, P, P, P]),
    "spo_ma_lamda_update": (c_int, [P, P, c_float, c_float, c_float, c_float, P]),
    "spo_ma_popart_stats": (c_int, [P, c_int64, P, P, P]),
    "spo_ma_popart_forward": (c_int, [P, c_int64, P, c_double, c_float, c_int, P, c_int64, P, P]),
    "spo_ma_jvp_scratch_floats": (c_int64, [POINTER(MaNet), c_int64]),
    "spo_ma_jvp": (c_int, [P, POINTER(MaNet), P, c_int64, P, P, P, P]),
    "spo_ma_value_loss": (c_int, [P, P, P, P, P, c_float, c_float, c_float, c_float, c_int64, c_int64, P, P, P, P]),
    "spo_ma_clip_adam": (c_int, [P, P, P, P, c_int64, c_int64, c_float, c_float, c_float, c_float, c_int, P, P, P]),
    "spo_ma_gae": (c_int, [P] * 7 + [c_int64, c_int64, c_double, c_double, c_float, c_float, c_float, c_float, P]),
    "spo_mlp_param_count": (c_int64, [POINTER(MlpNet)]),
    "spo_mlp_workspace_floats": (c_int64, [POINTER(MlpNet), c_int64]),
    "spo_mlp_backward_scratch_floats": (c_int64, [POINTER(MlpNet), c_int64]),
    "spo_mlp_forward": (c_int, [P, POINTER(MlpNet), P, c_int64, P, P]),
    "spo_mlp_backward": (c_int, [P, POINTER(MlpNet), P, c_int64, P, P, P, P, P]),
    "spo_gauss_sample": (c_int, [P, P, P, P, P, c_int64, c_int, P]),
    "spo_gauss_kl_sum": (c_int, [P, P, P, P, c_int64, c_int, P, c_int, P, c_int, P]),
    "spo_wide_ppo_loss": (c_int, [P] * 9 + [c_int64, c_int, c_float] + [P] * 6 + [c_int, P]),
    "spo_wide_clip_adam": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, POINTER(PpoCfg), c_int64, P, P, P, c_int, P]),
    "spo_wide_actor_loss": (c_int, [c_int] + [P] * 7 + [c_int64, c_int64, c_int, c_float, c_float, P, P, c_int, P, P, P, c_int, P]),
    "spo_wide_critic_loss": (c_int, [P, P, P, P, c_int64, P, P, P, P, c_int, P]),
    "spo_mlp_jvp_scratch_floats": (c_int64, [POINTER(MlpNet), c_int64]),
    "spo_mlp_jvp": (c_int, [P, POINTER(MlpNet), P, P, c_int64, P, P, P, P]),
    "spo_wide_fvp_cotangent": (c_int, [P, P, c_int64, c_int64, c_int, P, P]),
    "spo_wide_linesearch_sums": (c_int, [P] * 8 + [c_int64, c_int, P, c_int, P, c_int, P]),
    "spo_wide_clip_adam_ex": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, POINTER(PpoCfg), c_int64, c_int64, c_int64,
                                      c_int64, c_int64, c_int, P, P, P, c_int, P]),
    "spo_wide_clip_adam_dev": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, POINTER(PpoCfg), P, c_int64, c_int64, c_int64, c_int,
                                       P, P, P, c_int, P]),
    "spo_wide_clip_adam_dev_log": (c_int, [P, P, P, P, c_int64, c_int64, c_int64, c_int64, POINTER(PpoCfg), P, c_int64, c_int64, c_int64,
                                           c_int, P, P, P, c_int, P, P, P, c_int64, P]),
    "spo_param_count": (c_int64, [c_int, c_int]),
    "spo_param_offset": (c_int64, [c_int, c_int, c_int]),
    "spo_synth_env_step": (c_int, [P] * 7 + [c_int64, c_int, c_uint64, c_uint64, c_float, c_float, c_int, P]),
    "spo_synth_env_step_rel": (c_int, [P] * 7 + [c_int64, c_int, c_uint64, c_uint64, P, c_float, c_float, c_int, c_int, c_float,
                                       c_float, P]),
}

_lib = None


def load(path: str | None = None):
    """Load the shared library and bind every prototype.  Needs no GPU (symbol check only)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # SPO_LIB_PATH (development aid: A/B of kernel builds) is honoured only together with SPO_LIB_OVERRIDE=1 and is
    # announced on stderr, so a stale variable cannot silently swap the kernels under a training run or a parity test;
    # loaded_library() reports what was loaded (bench.py puts it in its JSON line when it is not the in-tree build)
    override = os.environ.get("SPO_LIB_PATH")
    if override and path is None:
        if os.environ.get("SPO_LIB_OVERRIDE", "0") != "1":
            raise SpoError(f"SPO_LIB_PATH={override!r} is set without SPO_LIB_OVERRIDE=1: refusing to load kernels from "
                           "outside the tree (unset it, or set SPO_LIB_OVERRIDE=1 for a deliberate A/B run)")
        import sys
        print(f"[safepo] WARNING: kernels loaded from SPO_LIB_PATH={override} instead of the in-tree {LIB_PATH}", file=sys.stderr)
    p = path or override or LIB_PATH
    if not os.path.exists(p):
        raise SpoError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "at the repo root (hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.spo_abi_version() != ABI_VERSION:
        raise SpoError(f"ABI version mismatch: library {lib.spo_abi_version()} != binding {ABI_VERSION} (stale "
                       "libsafepo_hip.so: rebuild with `python -c 'import __graft_entry__ as g; g.build()'`)")
    if path is None:
        _lib = lib
        global _lib_loaded_from
        _lib_loaded_from = os.path.abspath(p)
    return lib


_lib_loaded_from = None


def loaded_library() -> str | None:
    """Absolute path of the shared object the process-wide binding was loaded from (None before load())."""
    return _lib_loaded_from


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().spo_last_error()
        raise SpoError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu_tensor(t, name: str, dtype=None):
    import torch
    if not torch.is_tensor(t) or not t.is_cuda:
        raise SpoError(f"{name} must be a GPU tensor (this path runs only on the HIP device; no CPU fallback)")
    if not t.is_contiguous():
        raise SpoError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise SpoError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t
