"""GPU: the reference-side binding printed in INTEGRATION.md section A is executed as written --
the fenced block is extracted from the document, exec'd, and its ``sdf_renderer_cpp.forward`` /
``.backward`` are called with the reference's exact positional order
(sdfest/differentiable_renderer/sdf_renderer.py:311-323 and :347-356) and compared with the
oracle at BASELINE configs[0] (C1).  The boundary document cannot rot unnoticed."""
import os
import re

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## A."):text.index("## B.")]
    block = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    from sdfest_amd import _lib
    _lib.lib()   # fail loudly if the library is missing
    block = block.replace('ctypes.CDLL("libsdfr_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(block, "INTEGRATION.md#A", "exec"), ns)
    return ns


def test_integration_stub_matches_oracle_at_c1():
    cpp = _stub_namespace()["sdf_renderer_cpp"]
    dev = torch.device("cuda", 0)
    W, H, f, thr = 160, 120, 80.0, 0.005
    cx, cy = 80.0, 60.0        # Camera.get_pinhole_camera_parameters(0.5) of the C1 camera
    sdf_np = oracle.blobs_sdf(0)
    sdf = torch.tensor(sdf_np, device=dev)
    for pos_shape, quat_shape, isc_shape in (((3,), (4,), ()), ((3,), (1, 4), (1,))):   # shapes seen in the wild
        position = torch.tensor([0.0, 0.0, -1.5], device=dev).reshape(pos_shape)
        orientation = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev).reshape(quat_shape)
        inv_scale = torch.tensor(2.0, device=dev).reshape(isc_shape)
        # sdf_renderer.py:311-323
        (depth,) = cpp.forward(sdf, position, orientation, inv_scale, W, H, cx, cy, f, f, thr)
        assert depth.shape == (H, W)
        g = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (H, W)).astype(np.float32), device=dev)
        # sdf_renderer.py:347-356
        g_sdf, g_p, g_q, g_s = cpp.backward(g, depth, sdf, position, orientation, inv_scale, W, H, cx, cy, f, f)
        torch.cuda.synchronize()
        assert g_p.shape == position.shape and g_q.shape == orientation.shape and g_s.shape == inv_scale.shape

        d_ref, _, margin = oracle.render_forward(sdf_np, [0, 0, -1.5], [0, 0, 0, 1], [2.0], W, H, cx, cy, f, f,
                                                 thr, dtype=np.float32, with_aux=True)
        d = depth.cpu().numpy()
        robust = margin[0] > 1e-5
        assert np.array_equal((d > 0)[robust], (d_ref[0] > 0)[robust])
        both = (d > 0) & (d_ref[0] > 0)
        assert both.sum() == 948 and np.max(np.abs(d[both] / d_ref[0][both] - 1)) < 1e-4
        o = oracle.render_backward(g.cpu().numpy(), d, sdf_np, [0, 0, -1.5], [0, 0, 0, 1], [2.0], cx, cy, f, f,
                                   dtype=np.float32)
        assert np.max(np.abs(g_sdf.cpu().numpy() - o[0])) <= 1e-4 * np.max(np.abs(o[0]))
        dimg = oracle.render_derivative_images(d, sdf_np, [0, 0, -1.5], [0, 0, 0, 1], [2.0], cx, cy, f, f,
                                               dtype=np.float64)[0]
        l1 = np.array([np.abs(dimg[..., k] * g.cpu().numpy()).sum() for k in range(8)])
        pose = np.concatenate([g_p.cpu().numpy().ravel(), g_q.cpu().numpy().ravel(), g_s.cpu().numpy().ravel()])
        ref = np.concatenate([o[1][0], o[2][0], o[3]])
        assert np.all(np.abs(pose - ref) <= 1e-4 * l1), (pose, ref)
