"""ctypes loader for libnncf_hip.so (the C ABI declared in include/nncf_hip.h).

The product has no CPU fallback: if the shared library is missing, or a device entry
point is called without a HIP device, this raises instead of degrading.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnncf_hip.so")

_lib = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_u64p = ctypes.POINTER(ctypes.c_uint64)


class NncfError(RuntimeError):
    pass


class ModelCfg(ctypes.Structure):
    """nncf_model_cfg (include/nncf_hip.h)."""
    _fields_ = [
        ("batch_size_p", ctypes.c_int32),
        ("num_negatives", ctypes.c_int32),
        ("dim", ctypes.c_int32),
        ("row_stride", ctypes.c_int32),
        ("loss", ctypes.c_int32),
        ("normalize_user", ctypes.c_int32),
        ("normalize_item", ctypes.c_int32),
        ("loss_gamma", ctypes.c_float),
        ("neg_loss_weight", ctypes.c_float),
        ("u_reg", ctypes.c_float),
        ("learn_rate", ctypes.c_float),
        ("user_count", ctypes.c_int64),
        ("item_count", ctypes.c_int64),
    ]


class TowerCfg(ctypes.Structure):
    """nncf_tower_cfg (include/nncf_hip.h)."""
    _fields_ = [
        ("word_count", ctypes.c_int32),
        ("word_dim", ctypes.c_int32),
        ("item_dim", ctypes.c_int32),
        ("content_len", ctypes.c_int32),
        ("max_rows", ctypes.c_int32),
        ("item_count", ctypes.c_int64),
        ("bn_eps", ctypes.c_float),
        ("bn_momentum", ctypes.c_float),
        ("learn_rate", ctypes.c_float),
        ("conv_filters", ctypes.c_int32),
        ("filter_len", ctypes.c_int32),
        ("conv_dropout", ctypes.c_float),
        ("n_filter_lens", ctypes.c_int32),
        ("filter_lens", ctypes.c_int32 * 4),
    ]


class TowerParams(ctypes.Structure):
    """nncf_tower_params (include/nncf_hip.h): borrowed device pointers."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "words", "dense_w", "dense_b", "bn_gamma", "bn_beta", "bn_mean", "bn_var",
        "content", "item_ptr", "item_word", "item_cnt",
        "conv_w", "conv_b", "bn1_gamma", "bn1_beta", "bn1_mean", "bn1_var", "item_spos", "item_start")]


LOSS_CODES = {"skip-gram": 0, "mse": 1, "log-loss": 2, "max-margin": 3}


# nncf_xfer_fn (include/nncf_hip.h): the callback transport of the multi-GPU tests
XFER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p,
                           ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p)


def _declare(lib):
    vp, i32, i64, u64, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double
    lib.nncf_last_error.restype = ctypes.c_char_p
    lib.nncf_version.restype = ctypes.c_char_p
    lib.nncf_device_count.restype = i32
    lib.nncf_lcg_jump.restype = u64
    lib.nncf_lcg_jump.argtypes = [u64, u64]
    sig = {
        "nncf_sampler_create": [c_f64p, i64, f64, u64, i32, ctypes.POINTER(vp)],
        "nncf_sampler_destroy": [vp],
        "nncf_sampler_build_bounds": [c_f64p, i64, f64, c_i32p],
        "nncf_sampler_bounds": [vp, c_i32p],
        "nncf_sampler_sample_dev": [vp, i64, vp, vp],
        "nncf_sampler_sample_slice_dev": [vp, i64, i64, vp, vp],
        "nncf_sampler_skip": [vp, i64],
        "nncf_sampler_sample_batch": [vp, i64, c_i32p, vp],
        "nncf_sampler_get_state": [vp, c_u64p],
        "nncf_sampler_set_state": [vp, u64],
    }
    f32 = ctypes.c_float
    sig.update({
        "nncf_model_create": [ctypes.POINTER(ModelCfg), vp, vp, i32, i64, ctypes.POINTER(vp)],
        "nncf_model_destroy": [vp],
        "nncf_model_set_hparams": [vp, f32, f32, f32, f32],
        "nncf_model_set_optimizer": [vp, i32, f32, f32, f32, vp, vp, vp, vp],
        "nncf_model_get_step": [vp, ctypes.POINTER(ctypes.c_int64)],
        "nncf_model_set_shared_device": [vp, i32],
        "nncf_model_check": [vp, vp],
        "nncf_model_clear_error": [vp, vp],
        "nncf_model_set_sparse_adam_duplicates": [vp, i32],
        "nncf_model_set_bias": [vp, vp, vp, i64, vp, vp, vp, vp],
        "nncf_model_debug_stamps": [vp, ctypes.POINTER(ctypes.c_longlong), i64],
        "nncf_tower_debug_stamps": [vp, ctypes.POINTER(ctypes.c_longlong), i64],
        "nncf_profile_begin": [ctypes.c_char_p, i64],
        "nncf_profile_end": [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64),
                             ctypes.POINTER(ctypes.c_double)],
        "nncf_step_original": [vp, vp, vp, vp, i32, vp],
        "nncf_step_neg_shared": [vp, vp, vp, vp, i32, vp],
        "nncf_step_rowwise": [vp, vp, vp, vp, vp, i32, vp],
        "nncf_epoch_rowwise": [vp, vp, vp, vp, i64, vp, vp],
        "nncf_step_group_neg_shared": [vp, vp, vp, vp, i32, vp, vp, vp, vp],
        "nncf_epoch_original": [vp, vp, vp, vp, i64, vp, vp],
        "nncf_epoch_neg_shared": [vp, vp, vp, i64, vp, vp],
        "nncf_epoch_group_neg_shared": [vp, vp, vp, i64, vp, vp, vp],
        "nncf_predict_scores": [vp, vp, i64, vp, i64, vp, vp],
        "nncf_step_neg_shared_export": [vp, vp, vp, vp, vp, vp],
        "nncf_shard_route": [vp, i32, i32, i32, i32, vp, vp, vp, vp],
        "nncf_gather_rows": [vp, i32, i32, vp, i64, i32, vp, vp],
        "nncf_shard_create": [vp, vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, ctypes.POINTER(vp)],
        "nncf_shard_destroy": [vp],
        "nncf_shard_counts": [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int32)],
        "nncf_shard_buffers": [vp, i32] + [ctypes.POINTER(vp)] * 6 + [ctypes.POINTER(ctypes.c_int32)],
        "nncf_shard_plan": [vp, vp, vp, i64, vp],
        "nncf_shard_plan_owner": [vp, i64, vp, vp],
        "nncf_shard_step": [vp, i64, vp, vp, vp],
        "nncf_shard_apply": [vp, i64, i64, f32, vp],
        "nncf_shard_set_optimizer": [vp, i32, vp, vp, vp],
        "nncf_shard_set_comm": [vp, vp],
        "nncf_shard_set_bias": [vp, vp, vp],
        "nncf_shard_bias_buffers": [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp)],
        "nncf_shard_bias_grad_buffers": [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp)],
        "nncf_shard_moment_buffers": [vp, i32] + [ctypes.POINTER(vp)] * 4,
        "nncf_shard_run_chunk": [vp, vp, vp, i64, vp, f32, vp],
        "nncf_comm_unique_id": [vp],
        "nncf_comm_create": [vp, i32, i32, i32, ctypes.POINTER(vp)],
        "nncf_comm_create_callback": [XFER_FN, vp, i32, i32, ctypes.POINTER(vp)],
        "nncf_comm_destroy": [vp],
        "nncf_comm_all_to_all": [vp, vp, ctypes.POINTER(ctypes.c_int64), vp, ctypes.POINTER(ctypes.c_int64), i32, vp],
        "nncf_comm_all_reduce_f32": [vp, vp, i64, vp],
        "nncf_comm_wire_bytes": [vp, ctypes.POINTER(ctypes.c_int64), i32],
        "nncf_scatter_apply": [vp, i32, i32, vp, i32, i32, vp, f32, vp, vp],
        "nncf_scatter_merge": [vp, i32, i32, vp, i32, i32, i32, vp, f32, vp],
        "nncf_tower_create": [ctypes.POINTER(TowerCfg), ctypes.POINTER(TowerParams), i32, ctypes.POINTER(vp)],
        "nncf_tower_destroy": [vp],
        "nncf_tower_set_learn_rate": [vp, f32],
        "nncf_tower_set_dropout": [vp, f32, u64],
        "nncf_tower_set_pooling": [vp, i32],
        "nncf_tower_set_hot_words": [vp, c_i32p, i32],
        "nncf_tower_set_postings": [vp, vp, i32, vp, vp],
        "nncf_tower_set_content_id": [vp, vp, f32, vp, vp, vp],
        "nncf_tower_set_c_reg": [vp, f32],
        "nncf_tower_set_dense_tail": [vp, i32, f32],
        "nncf_tower_set_no_dense": [vp, i32],
        "nncf_tower_set_no_bn": [vp, i32],
        "nncf_tower_dense_mask": [vp, u64, i64, vp, vp],
        "nncf_tower_sync_params": [vp, vp],
        "nncf_tower_conv_mask": [vp, u64, i64, vp, vp],
        "nncf_tower_set_grad_export": [vp, vp],
        "nncf_tower_apply_grads": [vp, vp, f32, i64, vp],
        "nncf_tower_set_optimizer": [vp, i32, f32, f32, f32, vp, vp, vp, vp],
        "nncf_tower_dropout_scale": [vp, u64, vp, vp],
        "nncf_tower_forward": [vp, vp, i64, vp, i32, i32, vp, vp, vp],
        "nncf_tower_backward": [vp, vp, vp],
        "nncf_tower_step": [vp, vp, i32, vp, vp, vp, i32, vp, vp],
        "nncf_tower_epoch": [vp, vp, i32, vp, vp, vp, i64, vp, vp, vp],
        "nncf_tower_epoch_dp": [vp, vp, i32, vp, vp, vp, vp, i64, vp, f32, vp],
        "nncf_tower_step_rowwise": [vp, vp, vp, vp, vp, vp, i32, vp, vp],
        "nncf_tower_plan_steps": [vp, vp, i32, vp, vp, vp, i64, vp],
        "nncf_tower_step_at": [vp, vp, i32, i64, vp, vp, vp],
        "nncf_tower_epoch_rowwise": [vp, vp, vp, vp, vp, i64, vp, vp],
        "nncf_model_set_item_grad_export": [vp, vp],
        "nncf_plan_item_block": [vp, i32, vp, vp, vp, vp, vp],
        "nncf_step_planned": [vp, i32, vp, vp, i32, vp],
        "nncf_group_shuffle": [vp, vp, i64, i32, i32, u64, vp, vp, vp, vp, i64, vp],
        "nncf_group_sample": [vp, i64, i32, vp, vp, u64, i32, vp, vp, vp],
        "nncf_group_sample_with_negs": [vp, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, u64, i32, i32, vp, vp, vp, vp, vp],
        "nncf_eval_topk": [vp, i64, i64, i32, vp, vp, i32, vp, vp, vp],
        "nncf_eval_whole": [vp, vp, i64, vp, i32, vp, vp, i32, vp, vp, vp],
        "nncf_plan_ids": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    })
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = i32
        fn.argtypes = args
    lib.nncf_tower_grad_floats.restype = i64
    lib.nncf_tower_grad_floats.argtypes = [vp]
    lib.nncf_group_shuffle_ws_bytes.restype = i64
    lib.nncf_group_shuffle_ws_bytes.argtypes = [i64]
    lib.nncf_sampler_size.restype = i64
    lib.nncf_sampler_size.argtypes = [vp]
    lib.nncf_scatter_apply_ws_ints.restype = i64
    lib.nncf_scatter_apply_ws_ints.argtypes = [i32]


def lib():
    """The loaded library; raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NncfError(
                "libnncf_hip.so is not built (%s). Run `make -C nncf_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
        # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Importing torch first
        # makes the loader bind our NEEDED entry to that copy; loading ours first would put two
        # HIP runtimes in one process, and the second one to initialise sees no GPU.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        _declare(l)
        _lib = l
    return _lib


_EXC = {-1: ValueError, -2: MemoryError, -3: NncfError, -4: NncfError}


def check(rc):
    """Translate an nncf_status into the exception the reference's Cython shim would raise
    (sampler.pyx:13 `except +`: bad_alloc -> MemoryError, others -> RuntimeError; buffer
    validation -> ValueError)."""
    if rc != 0:
        msg = lib().nncf_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, NncfError)(msg)


def stream_ptr(stream=None):
    """hipStream_t of `stream` (a torch.cuda.Stream) or of torch's current stream."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream)
