#!/usr/bin/env python3
"""Side benchmarks of the other ops of the path (sampler, decoder, per-view-SDF render)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    import oracle
    from sdfest_amd import BatchRenderPlan, Camera, SDFDecoder
    from sdfest_amd.losses import _backward_raw, _forward_raw
    dev = torch.device("cuda", 0)
    out = {}
    # sampler: V views x M points each (M = hit pixels of a typical mug view .. every pixel)
    sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
    rng = np.random.default_rng(0)
    for V, M in ((1, 5000), (8, 5000), (1, 307200), (64, 20000)):
        pts = torch.tensor(rng.uniform(-0.6, 0.6, (V * M, 3)).astype(np.float32) + np.array([0, 0, -1.0], np.float32), device=dev)
        offs = torch.arange(0, V * M + 1, M, dtype=torch.int32, device=dev)
        pos = torch.tensor([[0.0, 0.0, -1.0]], device=dev).repeat(V, 1)
        quat = torch.tensor([[0.1, -0.2, 0.3, 0.9]], device=dev).repeat(V, 1)
        sc = torch.full((V,), 0.5, device=dev)
        go = torch.rand(V * M, device=dev) * 2 - 1
        tf = timeit(lambda: _forward_raw(pts, offs, M, pos, quat, sc, sdf))
        tb = timeit(lambda: _backward_raw(go, pts, offs, M, pos, quat, sc, sdf))
        out[f"pc_loss V={V} M={M}"] = {"forward_us": round(tf, 1), "backward_us": round(tb, 1),
                                       "Mpoints_per_s_fwd+bwd": round(V * M / (tf + tb), 1)}
    # ... and on real point sets: the back-projected depth images of 64 rendered views (coherent
    # points, what the loop feeds it) instead of uniformly random ones
    from sdfest_amd import Camera as _Cam
    from sdfest_amd.generated_views import depth_to_pointsets
    cam_r = _Cam(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
    Vr = 64
    pr, qr, ir = (torch.tensor(a, device=dev) for a in oracle.random_poses(Vr, seed=1))
    plan_r = BatchRenderPlan(64, Vr, cam_r, device=dev)
    depth_r = plan_r.forward(sdf, pr, qr, ir, 0.005)
    for tiled in (False, True):   # row-major (the reference's order) and SDFR_POINT_ORDER_TILED (what the loop feeds)
        pts_r, counts_r = depth_to_pointsets(depth_r, cam_r, tiled=tiled)
        offs_r = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), counts_r.cumsum(0)]).to(torch.int32)
        Mr = int(counts_r.max())
        sc_r = 1.0 / ir
        go_r = torch.rand(pts_r.shape[0], device=dev) * 2 - 1
        tf = timeit(lambda: _forward_raw(pts_r, offs_r, Mr, pr, qr, sc_r, sdf))
        tb = timeit(lambda: _backward_raw(go_r, pts_r, offs_r, Mr, pr, qr, sc_r, sdf))
        out[f"pc_loss on {Vr} back-projected views ({pts_r.shape[0]} points)" + (", tiled order" if tiled else "")] = {
            "forward_us": round(tf, 1), "backward_us": round(tb, 1),
            "Mpoints_per_s_fwd+bwd": round(pts_r.shape[0] / (tf + tb), 1)}
    # decoder
    g = os.path.join(ROOT, "tests", "golden")
    d = np.load(os.path.join(g, "decoder_mug.npz"))
    w = np.load(os.path.join(g, "mug_decoder_weights.npz"))
    cfg = {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
        "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
        "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c), "kernel_size": int(k),
                         "relu": bool(r)} for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"],
                                                                   d["conv_k"], d["conv_relu"])]}}
    dec = SDFDecoder.from_config(cfg, {k: w[k] for k in w.files})
    for N in (1, 16, 256):
        z = torch.randn(N, 8, device=dev)
        with torch.no_grad():
            t = timeit(lambda: dec.decode(z), 20)
        zg = z.clone().requires_grad_()
        G = torch.randn(N, 1, 64, 64, 64, device=dev)

        def fb():
            zg.grad = None
            dec.decode(zg).backward(G)
        t2 = timeit(fb, 10)
        out[f"decoder N={N}"] = {"forward_us": round(t, 1), "forward+vjp_us": round(t2, 1),
                                 "decodes_per_s": round(N / t * 1e6, 1)}
    # render with one SDF per view (the dataset-generation caller: generated_dataset.py:247-342)
    B = 64
    with torch.no_grad():
        sdfs = dec.decode(torch.randn(B, 8, device=dev) * 0.5)[:, 0].contiguous()
    pos, quat, isc = (torch.tensor(a, device=dev) for a in oracle.random_poses(B, seed=3))
    pos = pos * torch.tensor([0.3, 0.3, 0.3], device=dev)   # mug-sized scene: z in [0.36, 0.6]
    isc = torch.full((B,), 1 / 0.08, device=dev)
    cam = Camera(640, 480, 320.0, 320.0, 320.0, 240.0, pixel_center=0.5)
    plan = BatchRenderPlan(64, B, cam, device=dev, per_view_sdf=True)
    t = timeit(lambda: plan.forward(sdfs, pos, quat, isc, 0.005), 20)
    out[f"render forward, one SDF per view, B={B}"] = {"us": round(t, 1), "views_per_s": round(B / t * 1e6, 1),
                                                         "hit_pixels": int((plan.depth > 0).sum())}
    # the whole synthetic-view generator (decode -> render -> back-projection), SURVEY 8f-3
    from sdfest_amd.generated_views import SDFVAEViewGenerator
    gcfg = {"width": 640, "height": 480, "fov_deg": 90, "z_min": 0.3, "z_max": 0.7, "extent_mean": 0.15,
            "extent_std": 0.02, "render_threshold": 0.004, "pointcloud": True, "normalize_pose": True}
    for gb in (64, 256):
        gen = SDFVAEViewGenerator(gcfg, dec, batch_size=gb, device=dev, seed=0)
        zl = torch.randn(gb, 8)
        gp, gq, gs = (t.to(dev) for t in __import__("sdfest_amd.generated_views", fromlist=["x"]).sample_poses(
            gb, gen.camera, 0.3, 0.7, 0.15, 0.02, gen.gen))
        zl = zl.to(dev)
        t_r = timeit(lambda: gen.render(zl, gp, gq, gs), 10)
        t_g = timeit(lambda: gen.generate(), 10)
        gen.prefetch_draws = True      # what iterating over the generator does: the next batch is drawn while the GPU works
        t_p = timeit(lambda: gen.generate(), 10)
        gen.decode_ahead = True        # ... and decoded, on a second stream
        t_d = timeit(lambda: gen.generate(), 10)
        out[f"view generator B={gb} 640x480"] = {"decode+render_us": round(t_r, 1),
                                                 "decode+render_views_per_s": round(gb / t_r * 1e6, 1),
                                                 "full_sample_us": round(t_p, 1),
                                                 "full_sample_us_draws_not_prefetched": round(t_g, 1),
                                                 "full_sample_us_decoded_ahead": round(t_d, 1),
                                                 "samples_per_s": round(gb / t_p * 1e6, 1),
                                                 "samples_per_s_decoded_ahead": round(gb / t_d * 1e6, 1)}
    # batched render-and-compare step on the depth term (C3 poses): forward -> masked L1 -> backward,
    # with the loss as its own kernel vs folded into the render kernels
    from sdfest_amd import _lib
    L = _lib.lib()
    B = 256
    sdf64 = torch.tensor(oracle.blobs_sdf(0), device=dev)
    pos, quat, isc = (torch.tensor(a, device=dev) for a in oracle.random_poses(B, seed=1))
    plan = BatchRenderPlan(64, B, cam, device=dev)
    tpos = pos + 0.01 * torch.randn_like(pos)
    target = plan.forward(sdf64, tpos, quat, isc, 0.005).clone()
    loss = torch.empty(B, device=dev)
    grad = torch.empty_like(target)
    ws = torch.empty(max(L.sdfr_depth_l1_workspace_bytes(B, 640, 480), 256), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def separate():
        est = plan.forward(sdf64, pos, quat, isc, 0.005)
        _lib.check(L.sdfr_depth_l1_loss(est.data_ptr(), target.data_ptr(), B, 640, 480, 1.0, loss.data_ptr(),
                                        grad.data_ptr(), ws.data_ptr(), ws.numel(), 0, st), "l1")
        plan.backward(grad, sdf64, pos, quat, isc)

    def folded():
        plan.forward_l1(sdf64, pos, quat, isc, 0.005, target)
        plan.backward_l1(target, sdf64, pos, quat, isc)

    def render_only():
        plan.forward(sdf64, pos, quat, isc, 0.005)
        plan.backward(grad, sdf64, pos, quat, isc)

    # the same three as STEPS (forward prepares the backward: no prologue launch, the forward's rectangles)
    def separate_step():
        est = plan.forward(sdf64, pos, quat, isc, 0.005, prepare_backward=True)
        _lib.check(L.sdfr_depth_l1_loss(est.data_ptr(), target.data_ptr(), B, 640, 480, 1.0, loss.data_ptr(),
                                        grad.data_ptr(), ws.data_ptr(), ws.numel(), 0, st), "l1")
        plan.backward(grad, sdf64, pos, quat, isc)

    def folded_step(defer):
        plan.forward_l1(sdf64, pos, quat, isc, 0.005, target, prepare_backward=True, defer_loss=defer)
        plan.backward_l1(target, sdf64, pos, quat, isc)

    def render_only_step():
        plan.forward(sdf64, pos, quat, isc, 0.005, prepare_backward=True)
        plan.backward(grad, sdf64, pos, quat, isc)
    t_sep, t_fold, t_r = timeit(separate, 20), timeit(folded, 20), timeit(render_only, 20)
    ts_sep, ts_r = timeit(separate_step, 20), timeit(render_only_step, 20)
    ts_fold, ts_fold_d = timeit(lambda: folded_step(False), 20), timeit(lambda: folded_step(True), 20)
    out[f"render-and-compare step (depth term), B={B} 640x480"] = {
        "separate_loss_kernel_us": round(t_sep, 1), "loss_folded_into_renderer_us": round(t_fold, 1),
        "render_fwd+bwd_only_us": round(t_r, 1), "views_per_s_folded": round(B / t_fold * 1e6, 1),
        "views_per_s_separate": round(B / t_sep * 1e6, 1),
        "as_steps": {"separate_loss_kernel_us": round(ts_sep, 1), "loss_folded_us": round(ts_fold, 1),
                     "loss_folded_deferred_reduce_us": round(ts_fold_d, 1), "render_fwd+bwd_only_us": round(ts_r, 1),
                     "views_per_s_folded_deferred": round(B / ts_fold_d * 1e6, 1)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
