"""ctypes binding of lib/libmctsq_hip.so.  Fails loudly when the library is missing: there is no
CPU fallback for the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCTSQ_HIP_LIB") or os.path.join(_HERE, "lib", "libmctsq_hip.so")


class MctsqError(RuntimeError):
    pass


class EncoderDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("seq_len", C.c_int32), ("data_dim", C.c_int32),
                ("time_feat", C.c_int32)]


class NetDesc(C.Structure):
    _fields_ = [("enc", EncoderDesc * 3), ("embed_dim", C.c_int32), ("hidden_layers", C.c_int32),
                ("hidden_units", C.c_int32), ("action_dim", C.c_int32), ("activation", C.c_int32),
                ("reserved", C.c_int32)]


_vp, _i64, _i32, _f32, _f64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double
_ND = C.POINTER(NetDesc)


class StepArgs(C.Structure):
    """mctsq_step_args (include/mctsq_hip.h)."""
    _fields_ = ([(n, _vp) for n in ("d_tree", "d_prio", "d_stamp", "d_states", "d_next_states", "d_actions",
                                    "d_rewards", "d_dones")]
                + [(n, _i64) for n in ("capacity", "head", "size", "mass_capacity")]
                + [("alpha", _f64), ("beta", _f64), ("d_per_ws", _vp), ("per_ws_bytes", _i64)]
                + [(n, _vp) for n in ("d_params", "d_target_params", "d_grad", "d_m", "d_v", "d_prep_policy",
                                      "d_prep_target", "d_prep_bwd")]
                + [(n, _f64) for n in ("gamma", "lr", "beta1", "beta2", "eps", "max_norm", "tau")]
                + [("adam_step", _i64), ("d_u", _vp), ("B", _i64), ("d_loss", _vp), ("d_gnorm", _vp), ("d_wmax", _vp),
                   ("d_ws", _vp), ("ws_bytes", _i64), ("resplit_first", _i64)]
                + [(n, _i64) for n in ("world", "rank", "bcap")]
                + [(n, _vp) for n in ("d_shard_mass", "d_shard_size", "d_count", "d_draw", "d_exchange")])

# name -> (restype, argtypes); every symbol declared in include/mctsq_hip.h
SIGNATURES = {
    "mctsq_abi_version": (_i32, []),
    "mctsq_last_error": (C.c_char_p, []),
    "mctsq_param_layout": (_i32, [_ND, _vp, _vp, _i32, _vp]),
    "mctsq_time_features": (_i32, [_i32, _i32, _vp]),
    "mctsq_q_forward_workspace_bytes": (_i64, [_ND, _i64]),
    "mctsq_q_forward": (_i32, [_ND, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mctsq_q_forward_ex": (_i32, [_ND, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "mctsq_q_prepared_bytes": (_i64, [_ND]),
    "mctsq_q_prepare": (_i32, [_ND, _vp, _vp, _i64, _vp]),
    "mctsq_q_forward_prepared": (_i32, [_ND, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "mctsq_encoder_param_layout": (_i32, [C.POINTER(EncoderDesc), _vp, _vp, _i32, _vp]),
    "mctsq_encoder_workspace_bytes": (_i64, [C.POINTER(EncoderDesc), _i64]),
    "mctsq_encoder_forward": (_i32, [C.POINTER(EncoderDesc), _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "mctsq_dqn_grad_workspace_bytes": (_i64, [_ND, _i64]),
    "mctsq_dqn_grad": (_i32, [_ND, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32,
                              _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mctsq_dqn_prepared_bwd_bytes": (_i64, [_ND]),
    "mctsq_dqn_prepare": (_i32, [_ND, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mctsq_dqn_grad_sumsq_parts": (_i64, [_ND]),
    "mctsq_dqn_grad_prepared": (_i32, [_ND, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _f32, _f32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mctsq_per_dqn_step_workspace_bytes": (_i64, [_ND, _i64]),
    "mctsq_per_dqn_step": (_i32, [_ND, C.POINTER(StepArgs), _vp]),
    "mctsq_per_dqn_step_begin": (_i32, [_ND, C.POINTER(StepArgs), _vp]),
    "mctsq_per_dqn_step_finish": (_i32, [_ND, C.POINTER(StepArgs), _vp]),
    "mctsq_clip_adam_polyak_ex": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _f64, _f64, _f64,
                                         _f64, _vp, _vp, _i64, _vp, _i64, _vp]),
    "mctsq_optim_workspace_bytes": (_i64, []),
    "mctsq_clip_adam_polyak": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _f64, _f64, _f64,
                                      _f64, _vp, _vp, _i64, _vp]),
    "mctsq_polyak": (_i32, [_vp, _vp, _i64, _f64, _vp]),
    "mctsq_per_tree_words": (_i64, [_i64]),
    "mctsq_per_mass_shift": (_i32, [_i64]),
    "mctsq_per_workspace_bytes": (_i64, [_i64, _i64]),
    "mctsq_per_push": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f64, _i64, _vp]),
    "mctsq_per_refresh_max": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp]),
    "mctsq_per_sample": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _f64, _f64, _i64, _vp, _vp, _vp, _vp, _vp,
                                _i64, _vp]),
    "mctsq_per_sample_batch": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _f64, _f64, _i64, _vp, _vp, _vp, _vp]
                               + [_vp] * 14 + [_vp, _i64, _vp]),
    "mctsq_per_shard_select": (_i32, [_vp, _i64, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp]),
    "mctsq_per_sample_shard": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _f64, _f64, _i64, _vp, _vp, _i32,
                                      _vp, _vp, _vp, _vp] + [_vp] * 14 + [_vp, _i64, _vp]),
    "mctsq_per_sample_noalloc_max_draws": (_i64, []),
    "mctsq_per_sample_noalloc": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _f64, _f64, _i64, _vp, _vp, _vp, _vp,
                                        _vp]),
    "mctsq_per_update": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _f64, _i64, _vp]),
    "mctsq_per_total_mass": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "mctsq_per_shard_targets": (_i32, [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp]),
    "mctsq_per_sample_at": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _f64, _f64, _i64, _f64, _f64, _vp, _vp,
                                   _vp, _vp, _vp]),
    "mctsq_per_normalize_weights": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "mctsq_per_gather": (_i32, [_vp] * 6 + [_i64] + [_vp] * 10),
}


class _Lib:
    """Lazy loader so that importing the package on a box without the .so gives a clear error at
    first use, not an obscure one at import time."""

    def __init__(self):
        self._dll = None

    def _load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise MctsqError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
            dll = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(dll, name)      # AttributeError if the .so lacks a declared symbol
                fn.restype, fn.argtypes = res, args
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        return getattr(self._load(), name)

    def check(self, rc):
        if rc < 0:
            raise MctsqError(self._load().mctsq_last_error().decode() or f"libmctsq_hip error {rc}")
        return rc


lib = _Lib()


def stream_ptr(device):
    """hipStream_t of torch's current stream on `device` as an integer.  The raw getter costs ~0.3 us against ~4.5 us
    for torch.cuda.current_stream(device).cuda_stream (a Stream object per call): the data-parallel step looks the
    stream up nine times and is host-bound."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is None:
        return torch.cuda.current_stream(device).cuda_stream
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    return raw(torch.cuda.current_device() if idx is None else idx)


def ptr(t):
    """Raw device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    try:
        return t.data_ptr()
    except AttributeError:
        return t.ctypes.data
