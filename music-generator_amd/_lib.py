"""ctypes binding of libdeepj_hip.so (include/deepj_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails,
this module raises.  The product never routes through the CPU oracle.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEEPJ_LIB") or os.path.join(HERE, "lib", "libdeepj_hip.so")   # DEEPJ_LIB: A/B kernel builds

DTYPE_F32, DTYPE_BF16 = 0, 1


class DjConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "time_steps", "num_notes", "num_styles", "notes_per_bar", "octave", "octave_units",
        "style_units", "note_units", "time_axis_units", "note_axis_units", "time_axis_layers",
        "note_axis_layers", "dtype", "recurrent_sigmoid")] + [("input_dropout", C.c_float), ("dropout", C.c_float),
                                                              ("kernel_flags", C.c_int32), ("fuse_xw_min_tiles", C.c_int32)]


# dj_config.kernel_flags (include/deepj_hip.h DJ_KF_*)
KF_NO_CLUSTER, KF_NO_CLUSTER_PAIR, KF_NO_CLUSTER_F32, KF_NO_CLUSTER_COOP = 1, 2, 4, 8
KF_NO_FUSE_DX, KF_NO_GEN_KSPLIT, KF_DEBUG_CLUSTER_FAULT, KF_DEBUG_CLUSTER_LATE, KF_NO_STEP_EPILOGUE = 16, 32, 64, 128, 256
KF_COUNTED_EXCHANGE, KF_DEBUG_CLUSTER_MUTE, KF_BWD_PLAIN, KF_NO_GEN_MFMA = 512, 1024, 2048, 4096
FAULT_REPORT_WORDS = 32                      # DJ_FAULT_REPORT_WORDS


class DeepJError(RuntimeError):
    pass


_lib = None

_P = C.c_void_p
_SIGS = {
    "dj_abi_version": (C.c_int32, []),
    "dj_config_size": (C.c_int32, []),
    "dj_env_reload": (C.c_int32, []),
    "dj_style_embedding": (C.c_int32, [C.POINTER(DjConfig), _P, _P, C.c_int32, _P, _P]),
    "dj_workspace_cluster_fault_report": (C.c_int32, [C.POINTER(DjConfig), _P, C.c_int64, C.POINTER(C.c_int32), _P]),
    "dj_workspace_faults_async": (C.c_int32, [C.POINTER(DjConfig), _P, C.c_int64, _P, _P]),
    "dj_workspace_cluster_faults_take": (C.c_int32, [C.POINTER(DjConfig), _P, C.c_int64, C.POINTER(C.c_int32), _P]),
    "dj_param_count": (C.c_int64, [C.POINTER(DjConfig)]),
    "dj_param_info": (C.c_int32, [C.POINTER(DjConfig), C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dj_workspace_bytes": (C.c_int64, [C.POINTER(DjConfig)]),
    "dj_workspace_init": (C.c_int32, [C.POINTER(DjConfig), _P, C.c_int64, _P]),
    "dj_train_fwd_bwd": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64,
                                     C.c_uint64, _P]),
    "dj_train_fwd_bwd_acc": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64,
                                     C.c_uint64, C.c_int32, _P]),
    "dj_train_fwd_bwd_mb": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64,
                                        C.c_uint64, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "dj_pitch_bins": (C.c_int32, [C.POINTER(DjConfig), _P, _P, C.c_uint64, C.c_int32, C.c_int32, _P]),
    "dj_nadam_step": (C.c_int32, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "dj_predict": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "dj_time_model_predict": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "dj_note_model_predict": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "dj_gemm_nt": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P,
                               C.c_int32, C.c_int32, _P, _P]),
    "dj_gemm_tn": (C.c_int32, [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32,
                               _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "dj_lstm_wgrad": (C.c_int32, [C.c_int32, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P,
                                  C.c_int32, C.c_int64, _P, _P, _P, _P]),
    "dj_gemm_nt_tiled_a": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P, C.c_int32, _P,
                                       C.c_int32, C.c_int32, _P, _P]),
    "dj_lstm_pack": (C.c_int32, [C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "dj_lstm_fwd": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "dj_lstm_stash_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int64]),
    "dj_lstm_pack_w": (C.c_int32, [C.c_int32, C.c_int32, _P, C.c_int32, _P, _P]),
    "dj_lstm_fwd_fused": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P, _P,
                                      _P, _P, C.c_int32, _P, _P]),
    "dj_lstm_bwd": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int32,
                                _P]),
    "dj_lstm_bwd_dx": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int32,
                                   _P, C.c_int32, _P, C.c_int32, _P]),
    "dj_lstm_pack_wt": (C.c_int32, [C.c_int32, C.c_int32, _P, C.c_int32, _P, _P]),
    "dj_lstm_cluster_scratch_bytes": (C.c_int64, []),
    "dj_lstm_cluster_faults": (C.c_int32, [_P, _P]),
    "dj_workspace_cluster_faults": (C.c_int32, [C.POINTER(DjConfig), _P, C.c_int64, _P]),
    "dj_dropout_mask": (C.c_int32, [C.c_uint64, C.c_int32, C.c_float, C.c_int64, C.c_int32, _P, _P]),
    "dj_profile_enable": (C.c_int32, [C.c_int32]),
    "dj_profile_category_count": (C.c_int32, []),
    "dj_profile_category_name": (C.c_char_p, [C.c_int32]),
    "dj_profile_read": (C.c_int32, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dj_gen_state_size": (C.c_int32, []),
    "dj_generate_step_resident": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "dj_generate_step": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "dj_generate_prepare": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, C.c_int64, _P]),
    "dj_generate_step_prepared": (C.c_int32, [C.POINTER(DjConfig), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
}
OPTIONAL = set()


def load():
    """Load the library (once) and declare every prototype of include/deepj_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  Import torch first
    # so that our library binds to the HIP runtime torch's tensors/streams live in; loaded the
    # other way round the process ends up with two runtimes (hipErrorNoDevice on first use).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DeepJError(
            f"{LIB_PATH} is missing: build it with `python -m music_generator_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name in OPTIONAL:
                continue
            raise DeepJError(f"{LIB_PATH} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    if lib.dj_abi_version() != 5:
        raise DeepJError("libdeepj_hip.so ABI version mismatch")
    if lib.dj_config_size() != C.sizeof(DjConfig):
        raise DeepJError("dj_config layout mismatch: library %d bytes, binding %d" % (lib.dj_config_size(), C.sizeof(DjConfig)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = "hipError_t" if rc < 1000 else "argument error"
        raise DeepJError(f"{what} failed: code {rc} ({kind})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
