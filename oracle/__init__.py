"""CPU oracle for the sdfest hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy front-end over ``oracle/libsdfr_oracle.so`` (built by ``oracle/Makefile``
from ``sdfr_oracle.c`` / ``sdfr_oracle_impl.h``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``sdfest_amd/`` does.

Parity status: pinned against golden vectors captured from the reference's own
importable code (``tools/make_goldens.py``; see ``tests/test_oracle_golden.py``).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsdfr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no GPU, no reference sources involved)."""
    src = [os.path.join(_HERE, f) for f in ("sdfr_oracle.c", "sdfr_oracle_impl.h")]
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsdfr_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def set_threads(n: int) -> None:
    """OpenMP threads used by the oracle (for the cpu_baseline timing)."""
    lib().sdfo_set_threads(ctypes.c_int(int(n)))


def set_margin_mode(hit_tests_only: bool) -> None:
    """What ``render_forward(..., with_aux=True)``'s margin covers: every branch decision of the ray (default), or
    the hit tests alone -- the decisions that can change a pixel's depth (oracle/sdfr_oracle.c)."""
    lib().sdfo_set_margin_mode(ctypes.c_int(1 if hit_tests_only else 0))


def max_threads() -> int:
    return int(lib().sdfo_max_threads())


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", ctypes.c_float
    if dtype == np.float64:
        return "_f64", ctypes.c_double
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _pose(position, orientation, inv_scale, dtype):
    pos = _c(position, dtype).reshape(-1, 3)
    quat = _c(orientation, dtype).reshape(-1, 4)
    isc = _c(inv_scale, dtype).reshape(-1)
    B = pos.shape[0]
    assert quat.shape[0] == B and isc.shape[0] == B
    return pos, quat, isc, B


def render_forward(sdf, position, orientation, inv_scale, width, height, cx, cy, fx, fy,
                   threshold, dtype=np.float32, max_steps=0, with_aux=False):
    """Depth render of B poses of one SDF.  cx, cy are pixel-centre-0.5 intrinsics.

    Returns depth (B,H,W); with_aux=True also returns (steps int32, margin).
    """
    sfx, _ = _sfx(dtype)
    sdf = _c(sdf, dtype)
    R = sdf.shape[0]
    pos, quat, isc, B = _pose(position, orientation, inv_scale, dtype)
    depth = np.empty((B, height, width), dtype=dtype)
    steps = np.empty((B, height, width), dtype=np.int32) if with_aux else None
    margin = np.empty((B, height, width), dtype=dtype) if with_aux else None
    fn = getattr(lib(), "sdfo_render_forward" + sfx)
    fn.restype = None
    fn(_p(sdf), ctypes.c_int(R), _p(pos), _p(quat), _p(isc), ctypes.c_int(B),
       ctypes.c_int(width), ctypes.c_int(height), ctypes.c_double(cx), ctypes.c_double(cy),
       ctypes.c_double(fx), ctypes.c_double(fy), ctypes.c_double(threshold), _p(depth),
       _p(steps), _p(margin), ctypes.c_int(max_steps))
    if with_aux:
        return depth, steps, margin
    return depth


def render_backward(grad_depth, depth, sdf, position, orientation, inv_scale, cx, cy, fx, fy,
                    dtype=np.float32, sdf_grad_mode=0):
    """Returns (g_sdf (R,R,R) summed over views, g_pos (B,3), g_quat (B,4), g_inv_scale (B,))."""
    sfx, _ = _sfx(dtype)
    sdf = _c(sdf, dtype)
    R = sdf.shape[0]
    pos, quat, isc, B = _pose(position, orientation, inv_scale, dtype)
    depth = _c(depth, dtype).reshape(B, *np.shape(depth)[-2:])
    H, W = depth.shape[-2:]
    grad_depth = _c(grad_depth, dtype).reshape(B, H, W)
    g_sdf = np.empty((R, R, R), dtype=dtype)
    g_pos = np.empty((B, 3), dtype=dtype)
    g_quat = np.empty((B, 4), dtype=dtype)
    g_isc = np.empty((B,), dtype=dtype)
    fn = getattr(lib(), "sdfo_render_backward" + sfx)
    fn.restype = None
    fn(_p(grad_depth), _p(depth), _p(sdf), ctypes.c_int(R), _p(pos), _p(quat), _p(isc),
       ctypes.c_int(B), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_double(cx),
       ctypes.c_double(cy), ctypes.c_double(fx), ctypes.c_double(fy),
       ctypes.c_int(sdf_grad_mode), _p(g_sdf), _p(g_pos), _p(g_quat), _p(g_isc))
    return g_sdf, g_pos, g_quat, g_isc


def render_derivative_images(depth, sdf, position, orientation, inv_scale, cx, cy, fx, fy,
                             dtype=np.float64):
    """(B,H,W,8) per-pixel d depth / d (px,py,pz,qx,qy,qz,qw,inv_scale)."""
    sfx, _ = _sfx(dtype)
    sdf = _c(sdf, dtype)
    R = sdf.shape[0]
    pos, quat, isc, B = _pose(position, orientation, inv_scale, dtype)
    depth = _c(depth, dtype).reshape(B, *np.shape(depth)[-2:])
    H, W = depth.shape[-2:]
    out = np.empty((B, H, W, 8), dtype=dtype)
    fn = getattr(lib(), "sdfo_render_derivative_images" + sfx)
    fn.restype = None
    fn(_p(depth), _p(sdf), ctypes.c_int(R), _p(pos), _p(quat), _p(isc), ctypes.c_int(B),
       ctypes.c_int(W), ctypes.c_int(H), ctypes.c_double(cx), ctypes.c_double(cy),
       ctypes.c_double(fx), ctypes.c_double(fy), _p(out))
    return out


def pc_loss_forward(points, position, orientation, scale, sdf, dtype=np.float32):
    sfx, _ = _sfx(dtype)
    sdf = _c(sdf, dtype)
    pts = _c(points, dtype).reshape(-1, 3)
    pos, quat, sc = _c(position, dtype), _c(orientation, dtype), _c(scale, dtype).reshape(1)
    out = np.empty((pts.shape[0],), dtype=dtype)
    fn = getattr(lib(), "sdfo_pc_loss_forward" + sfx)
    fn.restype = None
    fn(_p(pts), ctypes.c_int(pts.shape[0]), _p(pos), _p(quat), _p(sc), _p(sdf),
       ctypes.c_int(sdf.shape[0]), _p(out))
    return out


def pc_loss_backward(grad_out, points, position, orientation, scale, sdf, dtype=np.float32):
    """Returns (g_sdf, g_pos (3,), g_quat (4,), g_scale ())."""
    sfx, _ = _sfx(dtype)
    sdf = _c(sdf, dtype)
    R = sdf.shape[0]
    pts = _c(points, dtype).reshape(-1, 3)
    go = _c(grad_out, dtype).reshape(-1)
    pos, quat, sc = _c(position, dtype), _c(orientation, dtype), _c(scale, dtype).reshape(1)
    g_sdf = np.empty((R, R, R), dtype=dtype)
    g_pos = np.empty(3, dtype=dtype)
    g_quat = np.empty(4, dtype=dtype)
    g_scale = np.empty(1, dtype=dtype)
    fn = getattr(lib(), "sdfo_pc_loss_backward" + sfx)
    fn.restype = None
    fn(_p(go), _p(pts), ctypes.c_int(pts.shape[0]), _p(pos), _p(quat), _p(sc), _p(sdf),
       ctypes.c_int(R), _p(g_sdf), _p(g_pos), _p(g_quat), _p(g_scale))
    return g_sdf, g_pos, g_quat, g_scale[0]


def decoder_forward(params, config, z, dtype=np.float32, enforce_tsdf=False):
    """config: the reference's vae yaml dict (keys latent_size, decoder{fc_layers,conv_layers}, tsdf).

    params: flat array in state_dict order (see pack_decoder_params)."""
    sfx, _ = _sfx(dtype)
    dec = config["decoder"]
    fc_out = np.array([l["out"] for l in dec["fc_layers"]], dtype=np.int32)
    conv = dec["conv_layers"]
    ins = np.array([l["in_size"] for l in conv], dtype=np.int32)
    cin = np.array([l["in_channels"] for l in conv], dtype=np.int32)
    cout = np.array([l["out_channels"] for l in conv], dtype=np.int32)
    ks = np.array([l["kernel_size"] for l in conv], dtype=np.int32)
    relu = np.array([1 if l["relu"] else 0 for l in conv], dtype=np.int32)
    volume = int(config.get("sdf_size", 64))
    tsdf = config.get("tsdf", False)
    tsdf = float(tsdf) if tsdf is not False else 0.0
    z = _c(z, dtype).reshape(-1, config["latent_size"])
    params = _c(params, dtype)
    out = np.empty((z.shape[0], 1, volume, volume, volume), dtype=dtype)
    fn = getattr(lib(), "sdfo_decoder_forward" + sfx)
    fn.restype = ctypes.c_int
    rc = fn(_p(params), ctypes.c_int(config["latent_size"]), ctypes.c_int(len(fc_out)),
            _p(fc_out), ctypes.c_int(len(conv)), _p(ins), _p(cin), _p(cout), _p(ks), _p(relu),
            ctypes.c_int(volume), ctypes.c_double(tsdf), ctypes.c_int(int(enforce_tsdf)), _p(z),
            ctypes.c_int(z.shape[0]), _p(out))
    if rc != 0:
        raise ValueError(f"decoder layer description inconsistent (code {rc})")
    return out


def pack_decoder_params(state, n_fc, n_conv, prefix="decoder."):
    """Flatten a state-dict-like mapping (name -> array) into the oracle/product order."""
    parts = []
    for i in range(n_fc):
        parts += [state[f"{prefix}_fc_layers.{i}.weight"], state[f"{prefix}_fc_layers.{i}.bias"]]
    for i in range(n_conv):
        parts += [state[f"{prefix}_conv_layers.{i}.weight"], state[f"{prefix}_conv_layers.{i}.bias"]]
    return np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1) for p in parts])


def depth_to_pointcloud(depth, fx, fy, cx0, cy0, dtype=np.float32):
    """cx0, cy0: pixel-centre-0 intrinsics (Camera.get_pinhole_camera_parameters(0.0))."""
    sfx, _ = _sfx(dtype)
    depth = _c(depth, dtype)
    H, W = depth.shape
    pts = np.empty((H * W, 3), dtype=dtype)
    fn = getattr(lib(), "sdfo_depth_to_pointcloud" + sfx)
    fn.restype = ctypes.c_int
    n = fn(_p(depth), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_double(fx), ctypes.c_double(fy),
           ctypes.c_double(cx0), ctypes.c_double(cy0), _p(pts))
    return pts[:n].copy()


def depth_l1(estimate, target, weight=1.0):
    """Masked depth-L1 of SDFPipeline._compute_view_losses (estimation/simple_setup.py:129-135) and
    its gradient w.r.t. the estimate, float64: overlap = (target > 0) & (estimate > 0),
    loss = mean |estimate - target| over overlap (nan if empty; torch.mean of an empty selection),
    grad = weight * sign(estimate - target) / count on the overlap (torch.abs has gradient 0 at 0).
    Works on one image (H,W) or a stack (B,H,W) -> per-view loss."""
    e = np.asarray(estimate, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    single = e.ndim == 2
    e3, t3 = (e[None], t[None]) if single else (e, t)
    mask = (t3 > 0) & (e3 > 0)
    cnt = mask.sum(axis=(1, 2)).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        loss = np.where(mask, np.abs(e3 - t3), 0.0).sum(axis=(1, 2)) / cnt
        k = np.where(cnt > 0, weight / cnt, 0.0)
    grad = np.where(mask, np.sign(e3 - t3), 0.0) * k[:, None, None]
    return (loss[0], grad[0]) if single else (loss, grad)


# synthetic inputs live with the product; re-exported for the tests
from sdfest_amd.synthetic import blobs_sdf, random_poses, sphere_sdf  # noqa: E402,F401
