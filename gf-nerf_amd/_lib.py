"""ctypes binding of include/gfnerf_amd.h.  No fallback: if the shared library cannot be
loaded (or built), every op raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgfnerf_amd.so")

_vp, _i64, _i32, _f32, _u64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64



class WgradLayer(C.Structure):
    """gfn_wgrad_layer (include/gfnerf_amd.h)."""
    _fields_ = [("dz_hi", _vp), ("dz_lo", _vp), ("a_hi", _vp), ("a_lo", _vp), ("feat_dz", C.c_int32), ("feat_a", C.c_int32),
                ("m_valid", C.c_int32), ("k_valid", C.c_int32), ("col_map", _vp), ("dW", _vp), ("ldw", C.c_int32), ("db", _vp)]


class LaunchOpts(C.Structure):
    """gfn_launch_opts (include/gfnerf_amd.h): the launch policy of ONE call (ops.launch_opts builds it)."""
    _fields_ = [("decoder_cus", C.c_int32), ("plain_rounding", C.c_int32), ("record_merge", C.c_int32), ("march_lanes", C.c_int32),
                ("merge_front", _vp), ("merge_n", _vp), ("merge_frac", C.c_float), ("colour_x2", C.c_int32)]


MERGE_OFF, MERGE_ON, MERGE_DEVICE = 0, 1, 2  # GFN_MERGE_*
_opts = C.POINTER(LaunchOpts)

ABI_VERSION = 108  # = GFN_ABI_VERSION of the include/gfnerf_amd.h the SIGNATURES below were written for; lib() refuses any other library

# name -> (restype, argtypes); must list every function include/gfnerf_amd.h declares
SIGNATURES = {
    "gfn_last_error": (C.c_char_p, []),
    "gfn_version": (_i32, []),
    "gfn_device_count": (_i32, []),
    "gfn_hash_level_scales_host": (None, [_vp]),
    "gfn_hash_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "gfn_hash_fwd_f16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _i64, _i32, _vp, _vp]),
    "gfn_table_to_f16": (_i32, [_vp, _vp, _i64, _vp]),
    "gfn_hash_bwd": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp]),
    "gfn_normalize_dirs": (_i32, [_vp, _vp, _i64, _vp]),
    "gfn_trace": (_i32, [_vp, _vp, _i64, _vp, _i64, _f32, _f32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gfn_march": (_i32, [_vp, _vp, _vp, _i64, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                         _vp, _vp, _i64, _vp, _vp]),
    "gfn_tree_compact": (_i32, [_vp, _i64, _vp, _vp]),
    "gfn_sample_rays": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _f32, _f32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _opts, _vp]),
    "gfn_sample_rays_record": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _f32, _f32, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _opts, _vp]),
    "gfn_samples_to_dense": (_i32, [_vp, _i64] + [_vp] * 14),
    "gfn_update_nodes": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_update_nodes_mark": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gfn_update_nodes_apply": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "gfn_query_tree_nodes_centers": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp]),
    "gfn_mark_invisible_nodes": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "gfn_octree_proc_host": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_vp),
                                    C.POINTER(_vp)]),
    "gfn_octree_proc_workspace": (_i64, [_i64, _i32]),
    "gfn_octree_proc": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "gfn_linear_fwd": (_i32, [_vp, _i32, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "gfn_linear_bwd": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i64, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp,
                              _vp]),
    "gfn_field_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_field_bwd_dx": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gfn_field_pack_x3": (_i32, [_vp, _vp, _vp, _vp]),
    "gfn_field_fwd_x3": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _opts, _vp]),
    "gfn_field_bwd_x3": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _opts, _vp]),
    "gfn_field_bwd_scatter_x3": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _i64,
                                        _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _opts, _vp]),
    "gfn_grad_record_ranges": (_i32, [_i32]),
    "gfn_grad_records_reduce": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "gfn_grad_max": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "gfn_grad_scale_host": (_f32, [_f32]),
    "gfn_field_bwd_dx_x3": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _opts, _vp]),
    "gfn_wgrad_x3_parts": (_i32, []),
    "gfn_wgrad_x3_layers": (_i32, [C.POINTER(WgradLayer), _i32, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "gfn_wgrad_reduce": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp]),
    "gfn_wgrad_x3": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp]),
    "gfn_fused_render": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _i32, _i32,
                         _vp, _vp, _vp, _vp, _vp, _opts, _vp]),
    "gfn_fused_render_rec": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _i32, _i32,
                             _vp, _vp, _vp, _vp, _vp, _opts, _vp]),
    "gfn_fused_forward": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _opts, _vp]),
    "gfn_sh4": (_i32, [_vp, _i64, _vp, _vp]),
    "gfn_head_pack": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "gfn_head_unpack_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "gfn_generate_rays": (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_hash_fwd_lm": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _opts, _vp]),
    "gfn_pixel_uniform": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "gfn_pixel_collate": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_errmap_block_sums": (_i32, [_vp, _i64, _vp, _vp]),
    "gfn_errmap_update": (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "gfn_errmap_table_slots": (_i64, [_i64]),
    "gfn_errmap_sample": (_i32, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_composite_fwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_depth_finalize": (_i32, [_vp, _i64, _vp, _f32, _vp]),
    "gfn_phase_list": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_composite_fwd_phased": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gfn_two_phase_plan": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _i32, _vp]),
    "gfn_composite_bwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp]),
    "gfn_charbonnier": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "gfn_s3im": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp]),
    "gfn_train_losses": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "gfn_loss_grad_combine": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "gfn_adam_hyper_host": (None, [_i32, _f32, _f32, _f32, _vp]),
    "gfn_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _vp, _vp]),
    "gfn_adam_step_ex": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _vp, _vp, _vp]),
    "gfn_step_draws": (_i32, [_u64, _vp, _vp, _i64, _f32, _f32, _vp, _i64, _vp]),
    "gfn_copy_list": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "gfn_fill_zero": (_i32, [_vp, _i64, _vp]),
    "gfn_reduce_shards": (_i32, [_vp, _i32, _i64, _i32, _f32, _vp, _vp]),
    "gfn_octree_build_host": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _f32, _u64, _i32, C.POINTER(_vp),
                                     C.POINTER(_i64), C.POINTER(_vp), C.POINTER(_i64)]),
    "gfn_free_host": (None, [_vp]),
    "gfn_octree_distance_summary_host": (_f32, [_vp, _i64]),
    "gfn_octree_pca12_host": (_i32, [_vp, _i64, _vp, _vp]),
    "gfn_octree_visible_cams_host": (_i32, [_vp, _vp, _vp, _i64, _vp, _f32, _vp]),
    "gfn_octree_construct_trans_host": (_i32, [_vp, _vp, _i64, _vp, _f32, _i64, _vp, _i64, _u64, _vp, _vp]),
    "gfn_octree_fit_weight_host": (_i32, [_vp, _vp, _i64, _vp]),
}

_lib = None


class GfnError(RuntimeError):
    pass


def lib():
    """Load and return the CDLL.  When hipcc is on the machine the (incremental, lock-protected) build runs first, so an edited
    csrc/ file can never be measured through a stale binary; without hipcc the shipped library is loaded as it is, and its
    absence is an error (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime; it must be the HIP runtime of this process (loading the
    # system libamdhip64 first leaves torch with "No HIP GPUs are available"), so load torch first.
    import torch  # noqa: F401

    from . import build as _build

    have_hipcc = True
    try:
        _build._hipcc()
    except RuntimeError:
        have_hipcc = False
    if have_hipcc and os.environ.get("GFN_NO_REBUILD") != "1":
        _build.build_extension()
    elif not os.path.exists(LIB_PATH):
        raise GfnError(f"the HIP extension {LIB_PATH} is missing and there is no hipcc to build it")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # no fallback
        raise GfnError(f"cannot load the HIP extension {LIB_PATH}: {e}") from e
    handle.gfn_version.restype, handle.gfn_version.argtypes = _i32, []
    got = handle.gfn_version()
    if got != ABI_VERSION:  # a .so and a binding from different revisions: calls would go through with shifted arguments
        raise GfnError(f"{LIB_PATH} implements ABI revision {got}, this binding was written for {ABI_VERSION} (include/gfnerf_amd.h GFN_ABI_VERSION): "
                       "rebuild the extension (python gf-nerf_amd/build.py) or use the _lib.py that ships with it")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().gfn_last_error().decode("utf-8", "replace")
        raise GfnError(f"{what} failed (status {rc}): {msg}")


def require_device():
    n = lib().gfn_device_count()
    if n <= 0:
        raise GfnError("no gfx950 (MI355X) device visible: the HIP path has no CPU fallback")
    return n
