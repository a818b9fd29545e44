"""ctypes binding of libdragposer_hip.so (the C ABI declared in include/dragposer.h).

There is no fallback: if the shared library is missing or cannot be loaded this module raises,
and every product entry point built on it fails loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libdragposer_hip.so")  # (the product reads no environment variable; diagnostic tools pass their build's
                                                              #  path explicitly: bench.py --lib, tools/_diaglib.py)

DP_OK = 0
DP_ERR_INVALID = -1
DP_ERR_DEVICE = -2
DP_ERR_UNSUPPORTED = -3
DP_ERR_LAUNCH = -4
DP_ERR_TIMEOUT = -5
DP_WEIGHTS_FP32 = 0
DP_WEIGHTS_BF16 = 1
DP_MAX_ITERS = 1000000
DP_KERNEL_AUTO, DP_KERNEL_W4, DP_KERNEL_W16 = 0, 1, 2

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)
_u8 = C.POINTER(C.c_ubyte)


class DpModel(C.Structure):
    _fields_ = [
        ("f_latent_w", _f), ("f_latent_b", _f),
        ("unpool_w", _f * 3), ("conv_w", _f * 3), ("conv_mask", _f * 3), ("conv_b", _f * 3),
        ("mean_q", _f), ("std_q", _f), ("mean_disp", _f), ("std_disp", _f),
        ("parents", _i), ("offsets", _f), ("weight_dtype", C.c_int),
    ]


class DpFolded(C.Structure):
    _fields_ = [
        ("A0", C.c_float * (40 * 24)), ("c0", C.c_float * 40),
        ("A1", C.c_float * (60 * 40)), ("b1", C.c_float * 60),
        ("A2", C.c_float * (92 * 60)), ("b2", C.c_float * 92),
    ]


class DpBatch(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int),
        ("z0", C.c_void_p), ("z_tgt", C.c_void_p), ("cur_rot", C.c_void_p), ("tgt_pos", C.c_void_p),
        ("tgt_rot", C.c_void_p), ("w", C.c_void_p), ("tracked", C.c_void_p),
    ]


class _Sized(C.Structure):
    """dp_params / dp_result start with struct_size = sizeof the struct as THIS binding declares it (include/dragposer.h, 0.5.0)"""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(type(self))


class DpParams(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint), ("n_iter", C.c_int), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("lambda_rot", C.c_float), ("lambda_tmp", C.c_float), ("early_stop", C.c_int),
        ("stop_eps_pos", C.c_float), ("stop_eps_rot", C.c_float), ("min_loss_incr", C.c_float), ("max_trackers", C.c_int),
        ("kernel", C.c_int),
    ]


class DpSeqState(C.Structure):
    _fields_ = [
        ("global_pos", C.c_void_p), ("global_rot", C.c_void_p), ("latent_buf", C.c_void_p), ("disp_buf", C.c_void_p),
        ("heights_buf", C.c_void_p), ("history", C.c_int), ("n_heights", C.c_int), ("height_joints", C.c_int * 8),
    ]


class DpSeqStep(C.Structure):
    _fields_ = [
        ("adjust_joint", C.c_int), ("adjust_target_joint", C.c_int), ("adjust_weight", C.c_float),
        ("tgt_pos", C.c_void_p), ("pose_ret", C.c_void_p), ("pos_ret", C.c_void_p),
    ]


class DpSeqFrames(C.Structure):
    _fields_ = [
        ("n_steps", C.c_int), ("tgt_pos", C.c_void_p), ("tgt_rot", C.c_void_p), ("tgt_root", C.c_void_p), ("w", C.c_void_p),
        ("tracked", C.c_void_p), ("z_tgt", C.c_void_p), ("z_tgt_step", C.c_int), ("z_tgt_seq", C.c_int),
    ]


class DpSeqResults(_Sized):
    _fields_ = [("struct_size", C.c_uint), ("reserved0", C.c_uint),
                ("pose_ret", C.c_void_p), ("pos_ret", C.c_void_p), ("world_rot", C.c_void_p), ("iters", C.c_void_p), ("loss", C.c_void_p),
                ("hist_scratch", C.c_void_p), ("status", C.c_void_p)]


class DpTemporalLayer(C.Structure):
    _fields_ = [(n, _f) for n in (
        "sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ca_in_w", "ca_in_b", "ca_out_w", "ca_out_b",
        "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "norm3_w", "norm3_b")]


class DpTemporalModel(C.Structure):
    _fields_ = [
        ("n_heights", C.c_int), ("dim_feedforward", C.c_int), ("n_encoder_layers", C.c_int), ("n_decoder_layers", C.c_int),
        ("max_len", C.c_int), ("sample_step", C.c_int),
        ("in_proj_encoder_w", _f), ("in_proj_encoder_b", _f), ("in_proj_decoder_w", _f), ("in_proj_decoder_b", _f),
        ("out_proj_w", _f), ("out_proj_b", _f), ("pos_encoding", _f),
        ("enc_norm_w", _f), ("enc_norm_b", _f), ("dec_norm_w", _f), ("dec_norm_b", _f),
        ("means_latent", _f), ("stds_latent", _f),
        ("enc", C.POINTER(DpTemporalLayer)), ("dec", C.POINTER(DpTemporalLayer)),
    ]


class DpResult(_Sized):
    _fields_ = [
        ("struct_size", C.c_uint), ("reserved0", C.c_uint), ("z", C.c_void_p), ("z_pre", C.c_void_p), ("pose", C.c_void_p), ("disp", C.c_void_p),
        ("world_disp", C.c_void_p), ("world_rot", C.c_void_p), ("pos", C.c_void_p), ("rot", C.c_void_p),
        ("loss", C.c_void_p), ("iters", C.c_void_p), ("status", C.c_void_p), ("clock", C.c_void_p),
    ]


DP_STATUS_NONFINITE_RESULT, DP_STATUS_BAD_STATE, DP_STATUS_BAD_TARGETS, DP_STATUS_TARGET_NOT_ROTATION = 1, 2, 4, 8
DP_TEMPORAL_TEAM_TIMEOUT = 1
DP_INPUT_LIMIT = 1.0e4


# every symbol include/dragposer.h declares (checked by tests/test_abi.py)
PUBLIC_SYMBOLS = (
    "dp_version", "dp_last_error", "dp_fold_decoder", "dp_create", "dp_destroy", "dp_optimize",
    "dp_forward", "dp_sequence_advance", "dp_optimize_sequence", "dp_kernel_geometry", "dp_auto_kernel", "dp_io_alloc", "dp_io_free", "dp_io_upload", "dp_io_download", "dp_stream_sync", "dp_io_alloc_host", "dp_io_free_host",
    "dp_temporal_create", "dp_temporal_destroy", "dp_temporal_last_error", "dp_temporal_predict", "dp_temporal_status",
)

_libs = {}


def load(path=None):
    """Load the shared library (once per path; the product never passes one -- tests do, for the test-only second
    implementation libdragposer_hip_ref8.so).  Raises RuntimeError with the build hint when absent."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  dragposer_amd has no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and the device pointers / streams
    # it hands us are only meaningful to THAT runtime.  Import torch (and pin its copy) before our library is
    # resolved, so that the shared SONAME binds libdragposer_hip.so to torch's runtime, never to /opt/rocm's.
    import torch

    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(path)
    lib.dp_version.restype = C.c_int
    lib.dp_last_error.restype = C.c_char_p
    lib.dp_last_error.argtypes = [C.c_void_p]
    lib.dp_fold_decoder.argtypes = [C.POINTER(DpModel), C.POINTER(DpFolded)]
    lib.dp_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(DpModel), C.c_int]
    lib.dp_destroy.argtypes = [C.c_void_p]
    lib.dp_optimize.argtypes = [C.c_void_p, C.POINTER(DpBatch), C.POINTER(DpParams), C.POINTER(DpResult), C.c_void_p]
    lib.dp_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(DpResult), C.c_void_p]
    lib.dp_sequence_advance.argtypes = [C.c_void_p, C.c_int, C.POINTER(DpResult), C.POINTER(DpSeqState), C.POINTER(DpSeqStep), C.c_void_p]
    lib.dp_optimize_sequence.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(DpSeqFrames), C.POINTER(DpParams), C.POINTER(DpSeqState),
                                         C.POINTER(DpSeqStep), C.POINTER(DpSeqResults), C.c_void_p]
    lib.dp_kernel_geometry.argtypes = [C.c_void_p, _i, _i, _i]
    lib.dp_auto_kernel.argtypes = [C.c_void_p, C.c_int]
    lib.dp_auto_kernel.restype = C.c_int
    lib.dp_io_alloc_host.argtypes = [C.c_void_p, C.c_ulonglong, C.POINTER(C.c_void_p)]
    lib.dp_io_free_host.argtypes = [C.c_void_p, C.c_void_p]
    lib.dp_temporal_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(DpTemporalModel), C.c_int]
    lib.dp_temporal_destroy.argtypes = [C.c_void_p]
    lib.dp_temporal_last_error.restype = C.c_char_p
    lib.dp_temporal_last_error.argtypes = [C.c_void_p]
    lib.dp_temporal_predict.argtypes = [C.c_void_p, C.c_int, C.POINTER(DpSeqState), C.c_int, C.c_void_p, C.c_void_p]
    # private test hooks (not part of include/dragposer.h)
    lib.dp_optimize_debug.argtypes = [C.c_void_p, C.POINTER(DpBatch), C.POINTER(DpParams), C.POINTER(DpResult),
                                      C.c_void_p, C.c_void_p]
    lib.dp_temporal_debug_force_variant.argtypes = [C.c_void_p, C.c_int]
    lib.dp_temporal_debug_team_status.argtypes = [C.c_void_p]
    lib.dp_temporal_status.argtypes = [C.c_void_p]
    lib.dp_temporal_debug_team_fault.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.dp_temporal_debug_team_size.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.dp_temporal_debug_split3.argtypes = [C.c_float, C.POINTER(C.c_ushort)]
    lib.dp_temporal_debug_split3.restype = None
    lib.dp_debug_pack.argtypes = [C.POINTER(DpFolded), _i, _f, _f, C.POINTER(C.c_uint)]
    lib.dp_debug_items.argtypes = [C.POINTER(DpModel), C.c_void_p]
    lib.dp_debug_pack_w16.argtypes = [C.POINTER(DpFolded), C.POINTER(DpModel), C.c_void_p, C.c_void_p, C.c_void_p]
    _libs[path] = lib
    return lib


def last_error(ctx=None):
    msg = load().dp_last_error(ctx)
    return msg.decode() if msg else ""


class DragPoserError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dragposer error {code}: {msg}")
        self.code = code
