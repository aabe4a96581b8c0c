"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/sdfr.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from helpers import ROOT


@pytest.fixture(scope="module")
def libpath():
    from sdfest_amd import _lib
    _lib.build()
    assert os.path.exists(_lib.LIB_PATH)
    return _lib.LIB_PATH


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sdfr.h")).read()
    return sorted(set(re.findall(r"SDFR_API\s+[\w\s\*]+?\b(sdfr_\w+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for s in ("sdfr_version", "sdfr_last_error", "sdfr_render_forward", "sdfr_render_backward"):
        assert s in syms


def test_library_exports_every_declared_symbol(libpath):
    h = ctypes.CDLL(libpath)
    for s in declared_symbols():
        assert hasattr(h, s), f"{s} declared in include/sdfr.h but not exported"


def test_shipped_library_exports_no_test_switches(libpath):
    """The production ABI has no process-global test hooks: kernel-form selection is per decoder handle
    (sdfr_decoder_set_option), the prologue's poll bound per workspace (sync header words 6 / 7), the timing stamps
    exist only in -DSDFR_TAIL_STAMPS builds.  Every exported symbol is one the header declares."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if " T " in line)
    ours = [s for s in exported if s.startswith("sdfr_")]
    assert ours, "no sdfr_ symbols exported?"
    assert not [s for s in exported if "debug" in s.lower()], "a debug switch is exported"
    assert sorted(ours) == declared_symbols()
    header = open(os.path.join(ROOT, "include", "sdfr.h")).read()
    assert "TEST HOOK" not in header and "sdfr_debug" not in header


def test_binding_table_matches_header(libpath):
    from sdfest_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    text = open(os.path.join(ROOT, "include", "sdfr.h")).read()
    # argument counts of the ctypes signatures equal the header's
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"SDFR_API[^;]*?\b%s\s*\(([^;]*?)\);" % name, text, re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(args), (name, n, len(args))


def test_version_and_error_string(libpath):
    from sdfest_amd import _lib
    L = _lib.lib()
    assert L.sdfr_version() == 100
    assert isinstance(L.sdfr_last_error(), bytes)


def test_argument_errors_without_gpu(libpath):
    """Argument validation happens before any HIP call, so it is testable on CPU."""
    from sdfest_amd import _lib
    L = _lib.lib()
    rc = L.sdfr_render_forward(None, 1, 0, None, None, None, 1, 8, 8, 4.0, 4.0, 4.0, 4.0, 0.01,
                               None, None, 0, 0, None)
    assert rc == -1 and b"R=1" in L.sdfr_last_error()
    buf0 = (ctypes.c_float * 64)()
    q0 = ctypes.cast(buf0, ctypes.c_void_p)   # a non-NULL pointer that is never dereferenced
    rc = L.sdfr_render_forward(None, 64, 0, None, None, None, 1, 8, 8, 4.0, 4.0, 4.0, 4.0, 0.01,
                               None, None, 0, 0, None)
    assert rc == -2
    # every layout: views, then the sync region (128-byte header + 6 x 128 plane-minimum entries of 16 bytes)
    # and the band spans (one word per band of 8 rows, rows of 128 bytes per view)
    sync = 128 + 6 * 128 * 16
    spans = lambda B: B * 256      # 480 rows = 60 bands -> 64 words
    assert L.sdfr_render_sync_offset(3) == 3 * 256
    assert L.sdfr_render_forward_workspace_bytes(64, 3, 640, 480) == 3 * 256 + sync + spans(3) + 64 ** 3 * 16
    # a step keeps face records, the backward's tile partials and the deterministic mode's int64 volume side by side
    # (+ the (sum, count) tile records of the loss-fused step)
    assert L.sdfr_render_step_workspace_bytes(64, 2, 640, 480) == (2 * 256 + sync + spans(2) + 64 ** 3 * 16
                                                                    + 2 * 20 * 60 * 32 + 64 ** 3 * 8 + 2 * 20 * 60 * 32)
    assert L.sdfr_render_partials_offset(64, 2, 640, 480, 1) == 2 * 256 + sync + spans(2) + 64 ** 3 * 16
    assert L.sdfr_render_partials_offset(64, 2, 640, 480, 0) == 2 * 256 + sync + spans(2)
    assert L.sdfr_render_fixed_volume_offset(64, 2, 640, 480, 1) == (2 * 256 + sync + spans(2) + 64 ** 3 * 16
                                                                     + 2 * 20 * 60 * 32)
    assert L.sdfr_render_fixed_volume_offset(64, 2, 640, 480, 0) == 2 * 256 + sync + spans(2) + 2 * 20 * 60 * 32
    assert L.sdfr_render_step_forward(q0, 64, 0, q0, q0, q0, 2, 8, 8, 4.0, 4.0, 4.0, 4.0, 0.01, q0, None, 0,
                                      q0, 1 << 30, 0, None) == -2 and b"g_sdf" in L.sdfr_last_error()
    assert L.sdfr_render_step_forward(q0, 64, 0, q0, q0, q0, 2, 8, 8, 4.0, 4.0, 4.0, 4.0, 0.01, q0, q0, 5,
                                      q0, 1 << 30, 0, None) == -1    # g_sdf_view_stride must be 0 or R^3
    assert L.sdfr_render_step_forward(q0, 64, 0, q0, q0, q0, 2, 8, 8, 4.0, 4.0, 4.0, 4.0, 0.01, q0, q0, 0,
                                      q0, 64, 0, None) == -3 and b"workspace" in L.sdfr_last_error()
    assert L.sdfr_render_step_backward(q0, q0, q0, 64, 0, 2, 8, 8, 4.0, 4.0, 4.0, 4.0, 7, q0, 0, q0, q0, q0,
                                       q0, 1 << 30, 0, None) == -1 and b"sdf_grad_mode" in L.sdfr_last_error()
    assert L.sdfr_render_backward_workspace_bytes(64, 2, 640, 480) == (2 * 256 + sync + spans(2) + 2 * 20 * 60 * 32
                                                                        + 64 ** 3 * 8)
    assert L.sdfr_render_step_backward(q0, q0, q0, 64, 0, 2, 8, 8, 4.0, 4.0, 4.0, 4.0, 0x100, q0, 64 ** 3, q0, q0, q0,
                                       q0, 1 << 30, 0, None) == -1 and b"DETERMINISTIC" in L.sdfr_last_error()
    # the merged launches of the loop: the deferred gradient chain and the two backward passes side by side
    buf = (ctypes.c_float * 64)()
    q = ctypes.cast(buf, ctypes.c_void_p)   # a non-NULL pointer that is never dereferenced
    rc = L.sdfr_views_to_pose_grad_deferred(q, q, q, 65, None, 0, 0, None, None, 0, None, None, q, q, q, 0, None)
    assert rc != 0 and b"V=65" in L.sdfr_last_error()
    rc = L.sdfr_views_to_pose_grad_deferred(q, q, q, 2, None, 0, 0, None, None, 0, None, None, None, q, q, 0, None)
    assert rc != 0 and b"NULL" in L.sdfr_last_error()
    rc = L.sdfr_views_to_pose_grad_deferred(q, q, q, 2, None, 0, 0, q, None, 100, q, None, q, q, q, 0, None)
    assert rc != 0 and b"sampler" in L.sdfr_last_error()      # two views need offsets
    rc = L.sdfr_render_backward_l1_pc(None, 1.0, q, q, q, q, 64, 0, q, q, q, 0, 32, 24, 16.0, 12.0, 30.0, 30.0, 0,
                                      q, 0, q, 1 << 20, 1.0, q, None, 100, q, q, 1 << 20, 0, None)
    assert rc != 0 and b"B=0" in L.sdfr_last_error()
    rc = L.sdfr_render_backward_l1_pc(None, 1.0, q, q, q, q, 64, 0, q, q, q, 2, 32, 24, 16.0, 12.0, 30.0, 30.0, 0,
                                      q, 0, q, 1 << 20, 1.0, q, None, 100, q, q, 1 << 20, 0, None)
    assert rc != 0 and b"offsets" in L.sdfr_last_error()
    rc = L.sdfr_render_backward_l1_pc(None, 1.0, q, q, q, q, 64, 0, q, q, q, 1, 32, 24, 16.0, 12.0, 30.0, 30.0, 0,
                                      q, 0, q, 1 << 20, 1.0, q, None, 100, q, q, 16, 0, None)
    assert rc != 0 and b"workspace" in L.sdfr_last_error()


def test_product_does_not_import_oracle():
    """The product package must never route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "sdfest_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "sdfr_oracle" not in src, f


def test_round5_entry_points_validate_their_arguments_without_gpu(libpath):
    """the entry points added in round 5 (the re-bindable loop, the front door, the loss-fused step, K objects side by
    side, per-handle options): argument errors are reported before any HIP call"""
    from sdfest_amd import _lib
    L = _lib.lib()
    buf = (ctypes.c_float * 64)()
    q = ctypes.cast(buf, ctypes.c_void_p)   # a non-NULL pointer that is never dereferenced
    err = lambda: L.sdfr_last_error()
    # SDFPipeline._preprocess_depth
    assert L.sdfr_preprocess_depth(q, q, -1, 8, 8, 1.0, 1, None, 0, None) == -1 and b"negative" in err()
    assert L.sdfr_preprocess_depth(q, None, 1, 8, 8, 1.0, 1, None, 0, None) == -2
    assert L.sdfr_preprocess_depth(None, None, 0, 8, 8, 1.0, 1, None, 0, None) == 0        # nothing to do
    # depth -> points without a host round trip
    assert L.sdfr_depth_to_points_resident(q, 1, 8, 8, 7, 1.0, 1.0, 4.0, 4.0, q, q, q, 1 << 20, q, 0, None) == -1
    assert b"point order" in err()
    assert L.sdfr_depth_to_points_resident(q, 70000, 8, 8, 0, 1.0, 1.0, 4.0, 4.0, q, q, q, 1 << 20, q, 0, None) == -1
    assert L.sdfr_depth_to_points_resident(q, 1, 8, 8, 0, 1.0, 1.0, 4.0, 4.0, None, q, q, 1 << 20, q, 0, None) == -2
    assert L.sdfr_depth_to_points_resident(None, 0, 8, 8, 0, 1.0, 1.0, 4.0, 4.0, None, None, None, 0, None, 0, None) == 0
    # the loss-fused step
    rc = L.sdfr_render_step_backward_l1(None, 1.0, None, q, q, q, 64, 0, 2, 8, 8, 4.0, 4.0, 4.0, 4.0, 0, q, 0, q, q, q,
                                        q, 1 << 30, None, None, 0, None)
    assert rc == -2 and b"loss_stats" in err()
    rc = L.sdfr_render_step_forward_l1(q, 64, 0, q, q, q, 1, 8, 8, 4.0, 4.0, 8.0, 8.0, 0.0, q, q, q, None, q, 0, q,
                                       1 << 30, None, 0, None)
    assert rc == -2 and b"both, or neither" in err()                                        # loss without loss_stats
    rc = L.sdfr_render_step_forward_l1(q, 64, 0, q, q, q, 1, 8, 8, 4.0, 4.0, 8.0, 8.0, 0.0, q, q, None, None, q, 0, q,
                                       1 << 30, ctypes.c_void_p(4), 0, None)
    assert rc == -1 and b"8-byte aligned" in err()
    # per-handle kernel forms
    assert L.sdfr_decoder_set_option(None, 0, 1) == -2
    # K estimates side by side
    rc = L.sdfr_loop_tail_objects(q, q, q, q, q, 16, 0, 1e-3, 1e-2, 1e-3, 1e-2, 1, q, q, 1, None, 0, 0, 0, None, None, 0,
                                  q, q, q, q, None, None, None, None, 0, None)
    assert rc == -1 and b"n_objects" in err()
    rc = L.sdfr_loop_tail_objects(q, q, q, q, q, 16, 4, 1e-3, 1e-2, 1e-3, 1e-2, 1, q, q, 65, None, 0, 0, 0, None, None, 0,
                                  q, q, q, q, None, None, None, None, 0, None)
    assert rc == -1 and b"V=65" in err()
    rc = L.sdfr_loop_tail_objects(q, q, q, q, q, 300, 4, 1e-3, 1e-2, 1e-3, 1e-2, 1, q, q, 1, None, 0, 0, 0, None, None, 0,
                                  q, q, q, q, None, None, None, None, 0, None)
    assert rc == -1 and b"n_params" in err()
    rc = L.sdfr_loop_tail_objects(q, q, q, q, None, 16, 4, 1e-3, 1e-2, 1e-3, 1e-2, 1, q, q, 1, None, 0, 0, 0, None, None,
                                  0, q, q, q, q, None, None, None, None, 0, None)
    assert rc == -2
    # (a decoder stage needs both the handle and the pointer the batched VJP left)
    rc = L.sdfr_loop_tail_objects(q, q, q, q, q, 16, 4, 1e-3, 1e-2, 1e-3, 1e-2, 1, q, q, 1, None, 0, 0, 0, None, None, 0,
                                  q, q, q, q, None, None, q, None, 0, None)
    assert rc == -2 and b"go together" in err()
    assert L.sdfr_decoder_backward_latent_deferred_batch(None, q, q, q, 4, q, 1 << 20, None, None) == -2
    assert L.sdfr_pose_to_views_objects(q, 4, 2, q, q, 1, q, q, q, q, None, 0, None) == -1        # n_params < 8
    assert L.sdfr_pose_to_views_objects(None, 16, 2, q, q, 1, q, q, q, q, None, 0, None) == -2
    assert L.sdfr_pose_to_views_objects(None, 16, 0, q, q, 1, q, q, q, q, None, 0, None) == 0
    # the initialisation network with the point count on the device
    assert L.sdfr_pointnet_layer_counted(q, None, 64, 3, 3, q, 3, q, q, q, None, q, 8, 8, q, 0, None) == -2
    assert b"row_count" in err()
    assert L.sdfr_pointnet_layer_counted(q, q, 0, 3, 3, q, 3, q, q, q, None, q, 8, 8, q, 0, None) == -1
    assert L.sdfr_init_estimate(None, 8, None, None, None, q, q, 0, 0, None, None, q, 0, None) == -2
    assert L.sdfr_init_estimate(q, 8, q, None, None, q, q, 0, 0, None, None, q, 0, None) == -2     # a table needs an index
    assert L.sdfr_init_estimate(q, 8, None, None, None, q, q, 0, 1, None, None, q, 0, None) == -2  # "best" needs its state
    assert L.sdfr_init_estimate(q, 5000, None, None, None, q, q, 0, 0, None, None, q, 0, None) == -1
