"""ctypes binding of libsdfr_hip.so (the C ABI declared in include/sdfr.h).

The shared library is built in-tree by ``sdfest_amd/csrc/Makefile`` (see
``__graft_entry__.build``).  There is NO fallback: if the library is missing
or a call fails, the error is raised -- the product never computes on the CPU.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDFR_LIB: another build of the same library (timing experiments: tools/microbench/build_variant.sh)
LIB_PATH = os.environ.get("SDFR_LIB") or os.path.join(_HERE, "libsdfr_hip.so")
_lib = None

c_fp = ctypes.c_void_p
c_int = ctypes.c_int
c_ll = ctypes.c_longlong
c_f = ctypes.c_float
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/sdfr.h declaration by declaration
SIGNATURES = {
    "sdfr_version": (c_int, []),
    "sdfr_last_error": (ctypes.c_char_p, []),
    "sdfr_render_forward_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "sdfr_render_sync_offset": (c_sz, [c_int]),
    "sdfr_decoder_set_option": (c_int, [c_fp, c_int, c_int]),
    "sdfr_render_partials_offset": (c_sz, [c_int, c_int, c_int, c_int, c_int]),
    "sdfr_render_step_forward_l1": (c_int, [c_fp, c_int, c_ll, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                                            c_f, c_f, c_f, c_f, c_f, c_fp, c_fp, c_fp, c_fp, c_fp, c_ll, c_fp, c_sz,
                                            c_fp, c_int, c_fp]),
    "sdfr_render_step_backward_l1_pc": (c_int, [c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_int, c_ll, c_fp, c_fp, c_fp,
                                                c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_fp, c_ll, c_fp, c_sz,
                                                c_f, c_fp, c_fp, c_int, c_fp, c_fp, c_sz, c_fp, c_fp, c_int, c_fp]),
    "sdfr_render_step_fused_l1_pc": (c_int, [c_fp, c_int, c_ll, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_f, c_f, c_f,
                                             c_f, c_f, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_sz, c_f, c_fp, c_fp, c_int,
                                             c_fp, c_sz, c_int, c_fp]),
    "sdfr_render_fused_view_count_offset": (c_sz, [c_int, c_int]),
    "sdfr_render_fused_tile_loss_offset": (c_sz, [c_int, c_int, c_int, c_int]),
    "sdfr_loop_tail_fused": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_f, c_f, c_f, c_f, c_int, c_fp, c_fp, c_int,
                                     c_fp, c_sz, c_sz, c_sz, c_f, c_fp, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp,
                                     c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "sdfr_render_step_backward_l1": (c_int, [c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_int, c_ll, c_int, c_int, c_int,
                                             c_f, c_f, c_f, c_f, c_int, c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_sz,
                                             c_fp, c_fp, c_int, c_fp]),
    "sdfr_loop_tail": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_f, c_f, c_f, c_f, c_int, c_fp, c_fp, c_int,
                               c_fp, c_sz, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                               c_f, c_fp, c_fp, c_fp, c_int, c_fp]),
    "sdfr_pose_to_views_objects": (c_int, [c_fp, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int,
                                           c_fp]),
    "sdfr_loop_tail_objects": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_fp, c_fp,
                                       c_int, c_fp, c_sz, c_int, c_int, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp,
                                       c_fp, c_fp, c_fp, c_int, c_fp]),
    "sdfr_loop_view_records": (c_int, [c_fp, c_sz, c_int, c_int, c_int, c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_int,
                                       c_int, c_int, c_fp, c_int, c_fp]),
    "sdfr_loop_tail_records": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_f, c_f, c_f, c_f, c_int, c_fp, c_fp,
                                       c_int, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_fp, c_fp,
                                       c_fp, c_int, c_fp]),
    "sdfr_inlier_counts_record": (c_int, [c_fp, c_fp, c_int, c_int, c_f, c_fp, c_fp, c_int, c_fp]),
    "sdfr_inlier_update_record": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int, c_fp, c_int, c_fp]),
    "sdfr_render_fixed_volume_offset": (c_sz, [c_int, c_int, c_int, c_int, c_int]),
    "sdfr_fixed_to_float": (c_int, [c_fp, c_sz, c_fp, c_int, c_fp]),
    "sdfr_render_forward": (c_int, [c_fp, c_int, c_ll, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                                    c_f, c_f, c_f, c_f, c_f, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_render_backward_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "sdfr_render_backward": (c_int, [c_fp, c_fp, c_fp, c_int, c_ll, c_fp, c_fp, c_fp, c_int,
                                     c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_fp, c_ll, c_fp,
                                     c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_render_step_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "sdfr_render_step_forward": (c_int, [c_fp, c_int, c_ll, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                                         c_f, c_f, c_f, c_f, c_f, c_fp, c_fp, c_ll, c_fp, c_sz, c_int, c_fp]),
    "sdfr_render_step_forward_counted": (c_int, [c_fp, c_int, c_ll, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                                                 c_f, c_f, c_f, c_f, c_f, c_fp, c_fp, c_ll, c_fp, c_sz, c_fp, c_int,
                                                 c_fp]),
    "sdfr_render_step_backward": (c_int, [c_fp, c_fp, c_fp, c_int, c_ll, c_int, c_int, c_int,
                                          c_f, c_f, c_f, c_f, c_int, c_fp, c_ll, c_fp, c_fp, c_fp,
                                          c_fp, c_sz, c_int, c_fp]),
    "sdfr_render_forward_l1_workspace_bytes": (c_sz, [c_int, c_int, c_int, c_int]),
    "sdfr_render_forward_l1": (c_int, [c_fp, c_int, c_ll, c_fp, c_fp, c_fp, c_int, c_int, c_int,
                                       c_f, c_f, c_f, c_f, c_f, c_fp, c_fp, c_fp, c_fp, c_fp, c_sz,
                                       c_int, c_fp]),
    "sdfr_render_backward_l1": (c_int, [c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_int, c_ll, c_fp, c_fp,
                                        c_fp, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_fp,
                                        c_ll, c_fp, c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_pc_loss_forward": (c_int, [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_ll,
                                     c_fp, c_int, c_fp]),
    "sdfr_pc_loss_backward_workspace_bytes": (c_sz, [c_int, c_int]),
    "sdfr_pc_loss_backward": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int,
                                      c_ll, c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_pc_l1_backward": (c_int, [c_f, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_ll,
                                    c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_pc_l1_backward_accumulate": (c_int, [c_f, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_ll,
                                    c_fp, c_ll, c_fp, c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_decoder_create": (c_int, [c_fp, c_sz, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp,
                                    c_fp, c_int, c_f, c_int, c_fp]),
    "sdfr_decoder_destroy": (None, [c_fp]),
    "sdfr_decoder_workspace_bytes": (c_sz, [c_fp, c_int]),
    "sdfr_decoder_tape_bytes": (c_sz, [c_fp, c_int]),
    "sdfr_decoder_forward": (c_int, [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "sdfr_decoder_forward_stage": (c_int, [c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_sz, c_fp, c_int]),
    "sdfr_decoder_backward_workspace_bytes": (c_sz, [c_fp, c_int]),
    "sdfr_decoder_backward_latent": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_sz, c_fp]),
    "sdfr_decoder_backward_latent_deferred": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_fp, c_fp]),
    "sdfr_decoder_backward_latent_deferred_scaled": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_f, c_fp, c_sz,
                                                             c_fp, c_fp]),
    "sdfr_decoder_backward_latent_deferred_batch": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_sz, c_fp, c_fp]),
    "sdfr_pose_to_views": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "sdfr_views_to_pose_grad": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                        c_fp, c_fp, c_fp, c_int, c_fp]),
    "sdfr_views_to_pose_grad_deferred": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_int, c_fp, c_fp, c_int,
                                                 c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "sdfr_render_backward_l1_pc": (c_int, [c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_int, c_ll, c_fp, c_fp, c_fp,
                                           c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_fp, c_ll, c_fp, c_sz,
                                           c_f, c_fp, c_fp, c_int, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_depth_l1_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "sdfr_depth_l1_loss": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_f, c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_pc_l1_loss": (c_int, [c_fp, c_fp, c_int, c_int, c_f, c_fp, c_fp, c_int, c_fp]),
    "sdfr_depth_points_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "sdfr_depth_count": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_depth_to_points": (c_int, [c_fp, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_fp, c_fp, c_fp, c_int,
                                     c_fp]),
    "sdfr_depth_count_ordered": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_depth_to_points_ordered": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_fp, c_fp, c_fp,
                                             c_int, c_fp]),
    "sdfr_depth_centroid_workspace_bytes": (c_sz, [c_int, c_int, c_int]),
    "sdfr_depth_count_centroid": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_fp, c_fp, c_fp, c_fp,
                                          c_sz, c_int, c_fp]),
    "sdfr_depth_to_points_shifted": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_fp, c_fp, c_fp,
                                             c_fp, c_fp, c_int, c_fp]),
    "sdfr_depth_to_points_resident": (c_int, [c_fp, c_int, c_int, c_int, c_int, c_f, c_f, c_f, c_f, c_fp, c_fp, c_fp,
                                              c_sz, c_fp, c_int, c_fp]),
    "sdfr_preprocess_depth": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_f, c_int, c_fp, c_int, c_fp]),
    "sdfr_add_inplace": (c_int, [c_fp, c_fp, c_sz, c_int, c_fp]),
    "sdfr_adam_step": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_f, c_f, c_f, c_f, c_int, c_int, c_fp]),
    "sdfr_point_constraint": (c_int, [c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_int, c_fp]),
    "sdfr_inlier_ratio": (c_int, [c_fp, c_fp, c_int, c_int, c_f, c_fp, c_fp, c_fp, c_int, c_fp, c_fp, c_int,
                                  c_fp, c_int, c_fp]),
    "sdfr_pointnet_layer": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int,
                                    c_fp, c_int, c_fp]),
    "sdfr_pointnet_layer_counted": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp,
                                            c_int, c_int, c_fp, c_int, c_fp]),
    "sdfr_init_estimate": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_fp, c_fp, c_fp, c_int,
                                   c_fp]),
    "sdfr_linear_vec": (c_int, [c_fp, c_int, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_int, c_fp, c_int, c_int, c_fp]),
    "sdfr_orientation_posterior": (c_int, [c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "sdfr_affine_mask": (c_int, [c_fp, c_int, c_int, c_int, c_fp, c_fp, c_int, c_fp]),
    "sdfr_nn_loss_forward": (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_int, c_fp]),
    "sdfr_nn_loss_backward": (c_int, [c_fp, c_fp, c_int, c_fp, c_int, c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
}


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C sdfest_amd/csrc` "
                "(or __graft_entry__.build()); sdfest_amd has no CPU fallback")
        import torch  # noqa: F401  loads torch's libamdhip64 first so both share one HIP runtime
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().sdfr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
