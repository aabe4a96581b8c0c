"""GPU: error behaviour of the C ABI and of the Python mirror (reference: TORCH_CHECK -> RuntimeError,
sdf_renderer.cpp:9-13; ValueError for a double camera spec, sdf_renderer.py:416-417)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def test_c_abi_rejects_bad_arguments_without_touching_memory():
    from sdfest_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    sdf = torch.zeros((64, 64, 64), device=dev)
    pos = torch.zeros((2, 3), device=dev); quat = torch.zeros((2, 4), device=dev); isc = torch.ones(2, device=dev)
    depth = torch.full((2, 48, 64), 7.0, device=dev)
    ws = torch.empty(1 << 24, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = lambda **kw: [kw.get("sdf", sdf.data_ptr()), kw.get("R", 64), kw.get("stride", 0), pos.data_ptr(),
                         quat.data_ptr(), isc.data_ptr(), kw.get("B", 2), 64, 48, 32.0, 24.0, kw.get("fx", 40.0), 40.0,
                         0.01, depth.data_ptr(), kw.get("ws", ws.data_ptr()), kw.get("wsn", ws.numel()), 0, st]
    assert L.sdfr_render_forward(*args(R=1)) == -1
    assert L.sdfr_render_forward(*args(B=-1)) == -1
    assert L.sdfr_render_forward(*args(fx=0.0)) == -1
    assert L.sdfr_render_forward(*args(stride=5)) == -1
    assert L.sdfr_render_forward(*args(sdf=None)) == -2
    assert L.sdfr_render_forward(*args(wsn=64)) == -3 and b"workspace" in L.sdfr_last_error()
    assert L.sdfr_render_forward(*args(ws=ws.data_ptr() + 4)) == -1   # misaligned workspace
    assert L.sdfr_render_forward(*args(B=0)) == 0
    torch.cuda.synchronize()
    assert torch.all(depth == 7.0)                 # none of the rejected calls launched anything
    assert L.sdfr_render_forward(*args()) == 0
    torch.cuda.synchronize()
    assert torch.all(depth == 0.0)                 # zero quaternion / position: nothing hit, all written


def test_python_mirror_raises_like_the_reference():
    from sdfest_amd import Camera, SDFDecoder, pc_loss, render_depth_gpu
    dev = "cuda"
    sdf = torch.tensor(oracle.sphere_sdf(0.5), device=dev)
    p = torch.tensor([0.0, 0.0, -2.0], device=dev); q = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev)
    s = torch.tensor(1.0, device=dev)
    cam = Camera(32, 24, 16.0, 16.0, 16.0, 12.0, pixel_center=0.5)
    with pytest.raises(ValueError):
        render_depth_gpu(sdf, p, q, s, 32, 24, 90.0, 0.01, cam)
    with pytest.raises(RuntimeError, match="CUDA"):
        render_depth_gpu(sdf.cpu(), p, q, s, None, None, None, 0.01, cam)
    with pytest.raises(RuntimeError, match="contiguous"):
        render_depth_gpu(sdf.permute(2, 1, 0), p, q, s, None, None, None, 0.01, cam)
    with pytest.raises(RuntimeError, match="float32"):
        render_depth_gpu(sdf.double(), p, q, s, None, None, None, 0.01, cam)
    with pytest.raises(RuntimeError):
        pc_loss(torch.zeros((4, 3)), p, q, s, sdf)            # CPU points
    with pytest.raises(ValueError):
        SDFDecoder(64, 8, [{"out": 8}], [dict(in_size=2, in_channels=1, out_channels=1, kernel_size=1, relu=False)])
    with pytest.raises(AssertionError):                       # SDFDecoder.sanity_check (sdf_vae.py:207-215)
        SDFDecoder(64, 8, [{"out": 9}], [dict(in_size=2, in_channels=1, out_channels=1, kernel_size=1, relu=False)],
                   state_dict={})


def test_calls_from_other_threads_and_streams():
    """backward runs on an autograd worker thread in the reference; the library must not depend on
    thread-local HIP state.  Also a non-default stream."""
    import threading
    from sdfest_amd import Camera, render_depth_gpu
    dev = "cuda"
    sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
    cam = Camera(160, 120, 80.0, 80.0, 80.0, 60.0, pixel_center=0.5)
    results = {}

    def work(i):
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            p = torch.tensor([0.0, 0.0, -1.5], device=dev, requires_grad=True)
            q = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev)
            d = render_depth_gpu(sdf, p, q, torch.tensor(2.0, device=dev), None, None, None, 0.005, cam)
            d.sum().backward()
            st.synchronize()
            results[i] = (int((d > 0).sum()), p.grad.clone())

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(results[i][0] == 948 for i in range(4))
    for i in range(1, 4):
        assert torch.equal(results[i][1], results[0][1])      # pose gradients are order-independent


def test_march_step_cap_terminates_where_the_reference_hangs():
    """threshold = 0 on an exactly-zero field: the reference loop (cu:283-293) never terminates
    (dist == 0 is not < 0 and t does not advance); the step cap returns a miss instead."""
    import sdfest_amd.differentiable_renderer as r
    dev = "cuda"
    sdf = torch.zeros((64, 64, 64), device=dev)
    pos = torch.tensor([[0.0, 0.0, -2.0]] * 5, device=dev)
    quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]] * 5, device=dev)
    isc = torch.ones(5, device=dev)
    for batch in (1, 5):     # plain-grid and face-record march
        d = r.forward_raw(sdf, pos[:batch].contiguous(), quat[:batch].contiguous(), isc[:batch].contiguous(),
                          64, 48, 32.0, 24.0, 40.0, 40.0, 0.0)
        torch.cuda.synchronize()
        assert torch.all(d == 0)
    # with a positive threshold the same field is a hit at the cube face
    d = r.forward_raw(sdf, pos[:1].contiguous(), quat[:1].contiguous(), isc[:1].contiguous(), 64, 48, 32.0, 24.0,
                      40.0, 40.0, 0.01)
    assert (d > 0).sum() > 100 and abs(d[d > 0].min().item() - 1.0) < 1e-5


def test_batch_render_plan_validates_its_tensors():
    """BatchRenderPlan hands raw device pointers to the C ABI: wrong dtype / shape / device / layout
    must raise before anything is launched (round-1 advice)."""
    from sdfest_amd import BatchRenderPlan, Camera
    dev = torch.device("cuda", 0)
    cam = Camera(64, 48, 40.0, 40.0, 32.0, 24.0, pixel_center=0.5)
    B = 3
    plan = BatchRenderPlan(64, B, cam, device=dev)
    sdf = torch.tensor(oracle.sphere_sdf(0.5), device=dev)
    pos = torch.tensor([[0.0, 0.0, -2.0]] * B, device=dev)
    quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]] * B, device=dev)
    isc = torch.ones(B, device=dev)
    g = torch.ones((B, 48, 64), device=dev)
    plan.forward(sdf, pos, quat, isc, 0.01)
    plan.backward(g, sdf, pos, quat, isc)
    torch.cuda.synchronize()
    bad = [
        (dict(sdf=sdf.double()), "float32"), (dict(pos=pos.double()), "float32"),
        (dict(sdf=sdf[:32]), "shape"), (dict(pos=pos[:2]), "shape"), (dict(quat=quat[:, :3].contiguous()), "shape"),
        (dict(isc=isc.view(B, 1)), "shape"), (dict(sdf=sdf.permute(2, 1, 0)), "contiguous"),
        (dict(pos=pos.cpu()), "CUDA"), (dict(sdf=sdf.expand(B, 64, 64, 64).contiguous()), "shape"),
    ]
    for kw, what in bad:
        a = dict(sdf=sdf, pos=pos, quat=quat, isc=isc)
        a.update(kw)
        with pytest.raises(RuntimeError, match=what):
            plan.forward(a["sdf"], a["pos"], a["quat"], a["isc"], 0.01)
    with pytest.raises(RuntimeError, match="shape"):
        plan.backward(g[:, :24].contiguous(), sdf, pos, quat, isc)
    with pytest.raises(RuntimeError, match="float32"):
        plan.forward_l1(sdf, pos, quat, isc, 0.01, g.double())
    per_view = BatchRenderPlan(64, B, cam, device=dev, per_view_sdf=True)
    with pytest.raises(RuntimeError, match="shape"):          # one grid for a per-view plan: out of bounds before
        per_view.forward(sdf, pos, quat, isc, 0.01)


def test_nan_upstream_gradient_reaches_the_outputs():
    """The reference adds grad * weight with atomicAdd (cu:373-388, :459-466): a NaN upstream gradient
    poisons the touched voxels and the pose gradients.  The fixed-point LDS sums must not turn it into
    finite garbage (round-1 advice): such pixels take the float path."""
    import sdfest_amd.differentiable_renderer as r
    dev = "cuda"
    sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
    for B in (1, 6):
        pos = torch.tensor([[0.0, 0.0, -1.5]] * B, device=dev)
        quat = torch.tensor([[0.0, 0.0, 0.0, 1.0]] * B, device=dev)
        isc = torch.full((B,), 2.0, device=dev)
        d = r.forward_raw(sdf, pos, quat, isc, 160, 120, 80.0, 60.0, 80.0, 80.0, 0.005)
        g = torch.ones_like(d)
        ref = r.backward_raw(g, d, sdf, pos, quat, isc, 160, 120, 80.0, 60.0, 80.0, 80.0)
        rows, cols = torch.nonzero(d[0] > 0, as_tuple=True)
        k = len(rows) // 2
        g[0, rows[k], cols[k]] = float("nan")
        out = r.backward_raw(g, d, sdf, pos, quat, isc, 160, 120, 80.0, 60.0, 80.0, 80.0)
        torch.cuda.synchronize()
        nan_vox = torch.isnan(out[0])
        assert 1 <= int(nan_vox.sum()) <= 8                       # the 8 corners of one cell (weights may be 0)
        assert torch.allclose(out[0][~nan_vox], ref[0][~nan_vox], rtol=1e-5, atol=1e-6)
        assert torch.isnan(out[1][0]).all() and torch.isnan(out[2][0]).all() and torch.isnan(out[3][0])
        if B > 1:                                                 # the other views are untouched
            assert torch.equal(out[1][1:], ref[1][1:]) and torch.equal(out[3][1:], ref[3][1:])


def test_loop_tail_and_deferred_decoder_vjp_reject_bad_arguments():
    """Round-3 entry points: sdfr_decoder_backward_latent_deferred needs somewhere to leave its pointer;
    sdfr_loop_tail takes the decoder and that pointer together or not at all, and checks the latent's size."""
    import ctypes
    import os
    from sdfest_amd import SDFDecoder, _lib
    L = _lib.lib()
    g = os.path.join(os.path.dirname(__file__), "golden")
    d = np.load(os.path.join(g, "decoder_mug.npz"))
    w = np.load(os.path.join(g, "mug_decoder_weights.npz"))
    cfg = {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
        "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
        "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c), "kernel_size": int(k),
                         "relu": bool(r)}
                        for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"], d["conv_k"],
                                                 d["conv_relu"])]}}
    dec = SDFDecoder.from_config(cfg, {k: w[k] for k in w.files})
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    z = torch.zeros(1, 8, device=dev)
    tape = torch.zeros(L.sdfr_decoder_tape_bytes(dec._h, 1), dtype=torch.uint8, device=dev)
    gout = torch.zeros(64 ** 3, device=dev)
    ws = torch.zeros(L.sdfr_decoder_backward_workspace_bytes(dec._h, 1), dtype=torch.uint8, device=dev)
    assert L.sdfr_decoder_backward_latent_deferred(dec._h, z.data_ptr(), tape.data_ptr(), gout.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), st, None) == -2
    t_mid = ctypes.c_void_p()
    assert L.sdfr_decoder_backward_latent_deferred(dec._h, z.data_ptr(), tape.data_ptr(), gout.data_ptr(), ws.data_ptr(),
                                                   16, st, ctypes.byref(t_mid)) == -3
    # the tail: 8 pose parameters + a latent of 8
    params = torch.zeros(16, device=dev); params[6] = 1.0; params[7] = 0.1
    grads, m, v = torch.zeros(16, device=dev), torch.zeros(16, device=dev), torch.zeros(16, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    cam_pos, cam_quat = torch.zeros(1, 3, device=dev), torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=dev)
    pos_c, quat_c = torch.zeros(1, 3, device=dev), torch.zeros(1, 4, device=dev)
    isc, sc = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    some = torch.zeros(64, device=dev)

    def tail(n_params, decoder, tm):
        return L.sdfr_loop_tail(params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(), step.data_ptr(),
                                n_params, 1e-3, 1e-2, 1e-3, 1e-2, 1, cam_pos.data_ptr(), cam_quat.data_ptr(), 1, None, 0,
                                0, 0, None, None, 0, pos_c.data_ptr(), quat_c.data_ptr(), isc.data_ptr(), sc.data_ptr(),
                                None, None, None, 0.0, None, decoder, tm, 0, st)
    assert tail(16, dec._h, None) == -2            # decoder without its intermediate
    assert tail(16, None, some.data_ptr()) == -2   # ... and the other way round
    assert tail(12, dec._h, some.data_ptr()) == -1 and b"latent" in L.sdfr_last_error()
    torch.cuda.synchronize()
    assert int(step.item()) == 0                   # none of them launched anything
    assert tail(16, None, None) == 0               # the plain tail still runs: one Adam step on zero gradients
    torch.cuda.synchronize()
    assert int(step.item()) == 1
