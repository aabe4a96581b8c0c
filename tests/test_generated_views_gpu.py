"""GPU: the batched view generator against per-sample calls of the single-view API,
i.e. against what SDFVAEViewDataset._generate_sample (generated_dataset.py:247-342) does per sample."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLDEN

pytestmark = pytest.mark.gpu

CFG = {"width": 320, "height": 240, "fov_deg": 90, "z_min": 0.3, "z_max": 0.7, "extent_mean": 0.15,
       "extent_std": 0.02, "render_threshold": 0.004, "pointcloud": True}


def torch_points(depth, cam):
    """pointset_utils.depth_to_pointcloud (:57-77, "opengl") in torch ops on the tensor's device -- the
    reference expression the batched kernel must reproduce bit for bit."""
    fx, fy, cx, cy, _ = cam.get_pinhole_camera_parameters(0.0)
    rows, cols = torch.nonzero(depth, as_tuple=True)
    z = depth[rows, cols]
    return torch.stack(((cols.float() - cx) * z / fx, -(rows.float() - cy) * z / fy, -z), dim=1)


@pytest.fixture(scope="module")
def mug_decoder():
    from sdfest_amd import SDFDecoder
    from test_decoder_gpu import mug_config
    d = np.load(os.path.join(GOLDEN, "decoder_mug.npz"))
    w = np.load(os.path.join(GOLDEN, "mug_decoder_weights.npz"))
    return SDFDecoder.from_config(mug_config(d), {k: w[k] for k in w.files}), d


def test_batch_equals_per_sample_pipeline(mug_decoder):
    from sdfest_amd import render_depth_gpu
    from sdfest_amd.generated_views import SDFVAEViewGenerator
    from sdfest_amd.pipeline import depth_to_pointcloud
    dec, d = mug_decoder
    B = 6
    gen = SDFVAEViewGenerator(CFG, dec, batch_size=B, seed=3)
    out = gen.generate()
    assert out["depth"].shape == (B, 240, 320) and out["latent_shape"].shape == (B, 8)
    assert int(out["valid"].sum()) >= B - 1
    for b in range(B):   # generated_dataset.py:258-280, :312-315 one sample at a time
        with torch.no_grad():
            sdf = dec.decode(out["latent_shape"][b:b + 1])[0, 0]
            depth = render_depth_gpu(sdf, out["position"][b], out["quaternion"][b], 1.0 / out["scale"][b],
                                     threshold=CFG["render_threshold"], camera=gen.camera)
        assert torch.equal(depth, out["depth"][b])
        assert torch.equal(torch_points(depth, gen.camera), out["pointset"][b])
        assert torch.equal(depth_to_pointcloud(depth, gen.camera), out["pointset"][b])
    # the first view also against the CPU oracle (fp32 march on the decoded grid)
    with torch.no_grad():
        sdf0 = dec.decode(out["latent_shape"][0:1])[0, 0].cpu().numpy()
    cam = gen.camera
    fx, fy, cx, cy, _ = cam.get_pinhole_camera_parameters(0.5)
    d_or = oracle.render_forward(sdf0, out["position"][0:1].cpu().numpy(), out["quaternion"][0:1].cpu().numpy(),
                                 (1.0 / out["scale"][0:1]).cpu().numpy(), 320, 240, cx, cy, fx, fy,
                                 CFG["render_threshold"], dtype=np.float32)[0]
    d_h = out["depth"][0].cpu().numpy()
    both = (d_or > 0) & (d_h > 0)
    assert ((d_or > 0) != (d_h > 0)).sum() <= max(2, 1e-3 * both.sum())
    assert both.sum() > 200 and np.max(np.abs(d_h[both] / d_or[both] - 1)) < 1e-4
    samples = gen.samples(out)
    assert len(samples) == int(out["valid"].sum()) and set(samples[0]) == {
        "depth", "pointset", "latent_shape", "position", "orientation", "quaternion", "scale"}


def test_normalisation_noise_and_smoothing_options(mug_decoder):
    from sdfest_amd.generated_views import SDFVAEViewGenerator, gaussian_kernel
    from sdfest_amd.pipeline import depth_to_pointcloud
    dec, d = mug_decoder
    B = 4
    plain = SDFVAEViewGenerator(CFG, dec, batch_size=B, seed=5).generate()
    cfg = {**CFG, "normalize_pose": True, "scale_to_unit_ball": True, "gaussian_noise_probability": 1.0}
    gen = SDFVAEViewGenerator(cfg, dec, batch_size=B, seed=99)
    out = gen.generate(latent=plain["latent_shape"].cpu(), position=plain["position"].cpu(),
                       quaternion=plain["quaternion"].cpu(), scale=plain["scale"].cpu())
    k = gaussian_kernel(1, 5)
    for b in range(B):   # generated_dataset.py:296-334 one sample at a time (NaN trick: CPU conv)
        depth = plain["depth"][b].cpu().clone()
        depth[depth == 0] = torch.nan
        f = torch.nn.functional.conv2d(depth[None, None], k, padding="same")[0, 0]
        m = torch.logical_or(f.isnan(), f.isinf())
        depth[~m] = f[~m]
        depth[depth.isnan()] = 0.0
        depth = depth.cuda()
        assert torch.equal(out["depth"][b] == 0, depth == 0)
        assert torch.allclose(out["depth"][b], depth, rtol=0, atol=2e-6)   # GPU vs CPU conv2d
        pts = depth_to_pointcloud(out["depth"][b], gen.camera)
        centroid = pts.mean(0)
        pts = pts - centroid
        pos = plain["position"][b] - centroid
        md = torch.max(torch.linalg.norm(pts))
        # (the batch centroid is an index_add_ of float atomics: its last bit depends on their order)
        assert torch.allclose(out["pointset"][b], pts / md, rtol=1e-5, atol=1e-6)
        assert torch.allclose(out["position"][b], pos, rtol=1e-5, atol=1e-6)
        assert torch.allclose(out["scale"][b], plain["scale"][b] / md, rtol=1e-5)
        assert abs(torch.linalg.norm(out["pointset"][b]).item() - 1.0) < 1e-4
    noisy = SDFVAEViewGenerator({**CFG, "normalize_pose": True, "norm_noise": True}, dec, batch_size=B, seed=1)
    o2 = noisy.generate(latent=plain["latent_shape"].cpu(), position=plain["position"].cpu(),
                        quaternion=plain["quaternion"].cpu(), scale=plain["scale"].cpu())
    for b in range(B):
        c = o2["pointset"][b].mean(0)                    # = the noise that was added to both
        assert c.abs().max() <= 0.2 + 1e-5 and c.abs().max() > 0
        ref_c = depth_to_pointcloud(plain["depth"][b], gen.camera).mean(0)
        assert torch.allclose(o2["position"][b], plain["position"][b] - ref_c + c, atol=1e-5)


def test_centred_point_sets_come_out_of_the_two_image_passes(mug_decoder):
    """normalize_pose without the unit ball (the generator's fast path, generated_dataset.py:318-326): the centroid
    is formed in the count pass and subtracted in the compaction -- against the per-sample torch expressions, and
    the kernels alone on ragged images with an empty view, with and without the noise vector."""
    from sdfest_amd import Camera
    from sdfest_amd.generated_views import PointSets, SDFVAEViewGenerator, depth_to_centred_pointsets
    from sdfest_amd.pipeline import depth_to_pointcloud
    dec, d = mug_decoder
    B = 5
    plain = SDFVAEViewGenerator(CFG, dec, batch_size=B, seed=7).generate()
    gen = SDFVAEViewGenerator({**CFG, "normalize_pose": True}, dec, batch_size=B, seed=8)
    out = gen.generate(latent=plain["latent_shape"].cpu(), position=plain["position"].cpu(),
                       quaternion=plain["quaternion"].cpu(), scale=plain["scale"].cpu())
    assert isinstance(out["pointset"], PointSets) and len(out["pointset"]) == B
    assert torch.equal(out["depth"], plain["depth"])
    for b in range(B):
        pts = depth_to_pointcloud(plain["depth"][b], gen.camera)
        c = pts.double().mean(0).float()
        assert out["pointset"][b].shape == pts.shape
        assert torch.allclose(out["pointset"][b], pts - c, rtol=1e-5, atol=2e-6)
        assert torch.allclose(out["position"][b], plain["position"][b] - c, rtol=1e-5, atol=2e-6)
    assert sum(p.shape[0] for p in out["pointset"]) == out["points"].shape[0] == int(out["counts"].sum())
    assert len(gen.samples(out)) == int(out["valid"].sum())
    rng = np.random.default_rng(9)
    for W, H, V in ((37, 29, 4), (640, 480, 3), (1030, 3, 3)):
        cam = Camera(W, H, 0.9 * W + 3.3, 1.1 * W + 1.7, 0.47 * W, 0.55 * H, pixel_center=0.5)
        dd = rng.uniform(0.3, 2.0, (V, H, W)).astype(np.float32)
        dd[rng.uniform(size=dd.shape) < 0.7] = 0
        dd[1] = 0
        depth = torch.tensor(dd, device="cuda")
        for noise in (None, torch.tensor(rng.uniform(-0.2, 0.2, (V, 3)).astype(np.float32), device="cuda")):
            pts, counts, counts_host, centroid = depth_to_centred_pointsets(depth, cam, noise)
            assert counts.tolist() == counts_host.tolist() == [(dd[v] != 0).sum() for v in range(V)]
            assert torch.equal(centroid[1], torch.zeros(3, device="cuda"))          # the empty view
            for v, part in enumerate(torch.split(pts, counts.tolist())):
                ref = torch_points(depth[v], cam)
                if ref.shape[0] == 0:
                    continue
                c = ref.double().mean(0)
                assert torch.allclose(centroid[v].double(), c, rtol=2e-6, atol=1e-7), (W, H, v)
                want = ref.double() - centroid[v].double() + (0 if noise is None else noise[v].double())
                assert torch.allclose(part.double(), want, rtol=0, atol=3e-7 * float(ref.abs().max()))
                # and bit for bit the reference's two float32 operations on the kernel's centroid: `pointset -=
                # centroid`, then `pointset += noise` (generated_dataset.py:314-326)
                want32 = ref - centroid[v]
                if noise is not None:
                    want32 = want32 + noise[v]
                assert torch.equal(part, want32), (W, H, v, (part - want32).abs().max())


def test_prefetched_draws_are_the_same_numbers(mug_decoder):
    """prefetch_draws: batch k + 1's latents and poses are drawn while the GPU works on batch k -- the same numbers
    from the same stream as back-to-back calls draw."""
    from sdfest_amd.generated_views import SDFVAEViewGenerator
    dec, d = mug_decoder
    cfg = {**CFG, "normalize_pose": True, "norm_noise": True}
    a = SDFVAEViewGenerator(cfg, dec, batch_size=4, seed=21)
    b = SDFVAEViewGenerator(cfg, dec, batch_size=4, seed=21, prefetch_draws=True)
    for _ in range(3):
        oa, ob = a.generate(), b.generate()
        for k in ("latent_shape", "position", "quaternion", "scale", "depth", "points"):
            assert torch.equal(oa[k], ob[k]), k
    assert b._ahead is not None and a._ahead is None


def test_decode_ahead_on_a_second_stream_gives_the_same_samples(mug_decoder):
    """decode_ahead: batch k + 1 is uploaded and decoded on a second HIP stream while batch k's render / noise /
    point-set kernels run -- the same kernels on the same numbers, so every sample is bit for bit the plain
    generator's; a call with a caller's latent in between discards what was decoded ahead and goes on correctly."""
    from sdfest_amd.generated_views import SDFVAEViewGenerator
    dec, d = mug_decoder
    cfg = {**CFG, "normalize_pose": True, "norm_noise": True}
    a = SDFVAEViewGenerator(cfg, dec, batch_size=5, seed=33, prefetch_draws=True)   # (same draws: see the test above)
    b = SDFVAEViewGenerator(cfg, dec, batch_size=5, seed=33, prefetch_draws=True, decode_ahead=True)
    keys = ("latent_shape", "position", "quaternion", "scale", "depth", "points", "counts")
    for it in range(4):
        oa, ob = a.generate(), b.generate()
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(oa[k], ob[k]), (it, k)
        assert b._decoded is not None and (it == 0 or b._side is not None)
    z = torch.tensor(d["z"][:5]) * 0.4
    oa, ob = a.generate(latent=z), b.generate(latent=z)     # (the batch decoded ahead is dropped, its draws too)
    for k in keys:
        assert torch.equal(oa[k], ob[k]), ("given latent", k)
    for it in range(2):
        oa, ob = a.generate(), b.generate()
        for k in keys:
            assert torch.equal(oa[k], ob[k]), ("after", it, k)


def test_empty_views_are_flagged_not_returned(mug_decoder):
    from sdfest_amd.generated_views import SDFVAEViewGenerator
    dec, _ = mug_decoder
    gen = SDFVAEViewGenerator({**CFG, "normalize_pose": True, "scale_to_unit_ball": True}, dec, batch_size=3, seed=0)
    pos = torch.tensor([[0.0, 0.0, -0.5], [0.0, 0.0, 3.0], [0.0, 0.0, -0.5]])   # view 1: behind the camera
    out = gen.generate(position=pos)
    assert out["valid"].tolist() == [True, False, True] and out["counts"][1] == 0
    assert len(gen.samples(out)) == 2 and torch.isfinite(out["scale"]).all()
    it = iter(gen)
    s = next(it)
    assert s["depth"].max() > 0


def test_reference_dataset_test_shapes(mug_decoder):
    """The assertions of the reference's tests/initilization/test_generated_dataset.py:58-74, same config."""
    from sdfest_amd.generated_views import SDFVAEViewGenerator
    dec, _ = mug_decoder
    cfg = {"extent_mean": 0.1, "extent_std": 0, "pointcloud": False, "normalize_pose": True, "z_min": 0.15,
           "z_max": 1.0}
    gen = SDFVAEViewGenerator(cfg, dec, batch_size=4, seed=0)
    sample = next(iter(gen))
    assert sample["depth"].shape == (480, 640) and "pointset" not in sample
    gen.cfg["pointcloud"] = True
    sample = next(iter(gen))
    assert sample["pointset"].shape[1] == 3 and sample["pointset"].shape[0] == int((sample["depth"] != 0).sum())
    assert sample["scale"].item() == pytest.approx(0.05)


def test_batched_back_projection_kernel_equals_torch_expression():
    """sdfr_depth_count / sdfr_depth_to_points against pointset_utils.depth_to_pointcloud's expression
    (torch ops on the same device): same points, same order, bit for bit -- ragged sizes, an empty
    view, a view that is all points, image sizes that are no multiple of the block."""
    from sdfest_amd import Camera
    from sdfest_amd.generated_views import depth_to_pointsets
    from sdfest_amd.pipeline import depth_to_pointcloud
    rng = np.random.default_rng(5)
    for W, H, V in ((640, 480, 5), (37, 29, 4), (1, 1, 2), (1030, 3, 3)):
        cam = Camera(W, H, 0.9 * W + 3.3, 1.1 * W + 1.7, 0.47 * W, 0.55 * H, pixel_center=0.5)
        d = rng.uniform(0.3, 2.0, (V, H, W)).astype(np.float32)
        d[rng.uniform(size=d.shape) < 0.7] = 0
        d[1] = 0
        if V > 2:
            d[2] = np.abs(d[2]) + 0.5
        depth = torch.tensor(d, device="cuda")
        pts, counts = depth_to_pointsets(depth, cam)
        parts = torch.split(pts, counts.tolist())
        assert counts.tolist() == [(d[v] != 0).sum() for v in range(V)]
        for v in range(V):
            assert torch.equal(parts[v], torch_points(depth[v], cam)), (W, H, v)
            assert torch.equal(parts[v], depth_to_pointcloud(depth[v], cam))


def test_tiled_point_order_is_a_permutation_in_16x16_patches():
    """SDFR_POINT_ORDER_TILED: every view's points are the row-major ones, permuted -- bit-identical coordinates,
    enumerated tile by tile (64 x 16), sub-tile by sub-tile (16 x 16), row by row -- for ragged image sizes too."""
    from sdfest_amd import Camera
    from sdfest_amd.generated_views import depth_to_pointsets
    rng = np.random.default_rng(6)
    for W, H, V in ((640, 480, 3), (37, 29, 4), (1, 1, 2), (1030, 3, 3), (64, 16, 2), (65, 17, 2)):
        cam = Camera(W, H, 0.9 * W + 3.3, 1.1 * W + 1.7, 0.47 * W, 0.55 * H, pixel_center=0.5)
        d = rng.uniform(0.3, 2.0, (V, H, W)).astype(np.float32)
        d[rng.uniform(size=d.shape) < 0.6] = 0
        d[1] = 0
        depth = torch.tensor(d, device="cuda")
        pts, counts = depth_to_pointsets(depth, cam)
        tpts, tcounts = depth_to_pointsets(depth, cam, tiled=True)
        assert torch.equal(counts, tcounts)
        fx, fy, cx, cy, _ = cam.get_pinhole_camera_parameters(0.0)
        for v, (a, b) in enumerate(zip(torch.split(pts, counts.tolist()), torch.split(tpts, counts.tolist()))):
            rows, cols = np.nonzero(d[v])
            key = (((rows // 16) * ((W + 63) // 64) + cols // 64) * 4 + (cols % 64) // 16) * 256 + (rows % 16) * 16 + cols % 16
            order = np.argsort(key, kind="stable")
            assert torch.equal(b, a[torch.tensor(order, device="cuda")]), (W, H, v)


def test_point_cloud_loss_does_not_depend_on_the_point_order():
    """the loss-fused sampler backward on row-major and on tiled points of the same depth images: loss and pose
    gradients agree to rounding (other summation order), d/dSDF to the order of its atomics."""
    from sdfest_amd import BatchRenderPlan, Camera, _lib
    from sdfest_amd.generated_views import depth_to_pointsets
    import oracle
    dev = torch.device("cuda")
    V, W, H = 6, 320, 240
    cam = Camera(W, H, 160.0, 160.0, 160.0, 120.0, pixel_center=0.5)
    sdf = torch.tensor(oracle.blobs_sdf(0), device=dev)
    pos, quat, isc = (torch.tensor(a, device=dev) for a in oracle.random_poses(V, seed=4, width=W, height=H, f=160.0))
    depth = BatchRenderPlan(64, V, cam).forward(sdf, pos, quat, isc, 0.005).clone()
    pos2 = (pos + 0.01).contiguous()          # sample the observed clouds at a perturbed pose: non-zero values
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for tiled in (False, True):
        pts, counts = depth_to_pointsets(depth, cam, tiled=tiled)
        offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), counts.cumsum(0)]).to(torch.int32)
        M = int(counts.max())
        loss = torch.zeros(V, device=dev)
        gs = torch.zeros(64, 64, 64, device=dev)
        gp, gq, gsc = torch.zeros(V, 3, device=dev), torch.zeros(V, 4, device=dev), torch.zeros(V, device=dev)
        ws = torch.empty(max(L.sdfr_pc_loss_backward_workspace_bytes(V, M), 256), dtype=torch.uint8, device=dev)
        _lib.check(L.sdfr_pc_l1_backward(3.0, loss.data_ptr(), pts.data_ptr(), offs.data_ptr(), V, M, pos2.data_ptr(),
                                         quat.data_ptr(), (1.0 / isc).contiguous().data_ptr(), sdf.data_ptr(), 64, 0,
                                         gs.data_ptr(), 0, gp.data_ptr(), gq.data_ptr(), gsc.data_ptr(), ws.data_ptr(),
                                         ws.numel(), 0, st), "sdfr_pc_l1_backward")
        torch.cuda.synchronize()
        res.append([t.cpu().numpy().astype(np.float64) for t in (loss, gs, gp, gq, gsc)])
    a, b = res
    assert np.all(a[0] > 0) and np.allclose(a[0], b[0], rtol=2e-6)
    assert np.max(np.abs(a[1] - b[1])) <= 1e-5 * np.max(np.abs(a[1]))
    for k in (2, 3, 4):
        assert np.max(np.abs(a[k] - b[k])) <= 2e-5 * max(np.max(np.abs(a[k])), 1e-12), k


def _torchvision_affine_nearest(mask_f, matrix):
    """What torchvision 0.12's functional_tensor.affine does with an (H,W) float image and the 6 numbers
    of _get_inverse_affine_matrix, written with the torch calls it makes (_gen_affine_grid: pixel-centre
    grid in pixels, theta rescaled by the half sizes; then grid_sample nearest / zeros / align_corners=False)."""
    H, W = mask_f.shape
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    base = torch.empty(1, H, W, 3)
    base[..., 0].copy_(torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W))
    base[..., 1].copy_(torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H])
    grid = base.view(1, H * W, 3).bmm(rescaled).view(1, H, W, 2)
    return torch.nn.functional.grid_sample(mask_f[None, None], grid, mode="nearest", padding_mode="zeros",
                                           align_corners=False)[0, 0]


def test_mask_perturbation_equals_grid_sample():
    from sdfest_amd.generated_views import inverse_affine_matrices, perturb_masks, sample_mask_affine
    g = torch.Generator().manual_seed(5)
    B, H, W = 7, 120, 160
    depth = torch.rand((B, H, W), generator=g) * (torch.rand((B, H, W), generator=g) < 0.4)
    depth[0, 40:80, 50:110] = 1.0                                   # a solid blob with straight edges
    angle, translate, scale = sample_mask_affine(B, W, H, g)
    angle[1], translate[1], scale[1] = 0.0, torch.tensor([0.0, 0.0]), 1.0            # identity
    angle[2], translate[2], scale[2] = 0.0, torch.tensor([3.0, -2.0]), 1.0           # pure shift
    angle[3], scale[3] = 25.0, 0.8                                                   # far outside the dataset's range
    assert torch.all((angle >= 0) & (angle <= 25)) and torch.all(translate[:, 0] == 0 + 3.0 * (torch.arange(B) == 2))
    assert torch.all(translate[:, 1].abs() <= 2) and torch.all(translate == translate.round())
    m = inverse_affine_matrices(angle, translate, scale)
    got = perturb_masks(depth.cuda(), m).cpu()
    for b in range(B):
        ref = _torchvision_affine_nearest((depth[b] != 0).float(), m[b].tolist()) > 0.5
        diff = int((got[b] != ref).sum())
        assert diff <= 2, (b, diff)       # (a source coordinate within float rounding of a pixel border)
    assert torch.equal(got[1], depth[1] != 0)
    # RandomAffine translates the image content by (+3, -2): output (i, j) shows input (i + 2, j - 3)
    assert torch.equal(got[2][:-2, 3:], (depth[2] != 0)[2:, :-3]) and not got[2][:, :3].any() and not got[2][-2:].any()


def test_mask_noise_and_discretized_orientation(mug_decoder):
    """generated_dataset.py:286-291, :310 and :358-360 on a batch, against the per-sample statements."""
    from sdfest_amd.generated_views import SDFVAEViewGenerator, inverse_affine_matrices, perturb_masks
    from sdfest_amd.so3grid import SO3Grid
    dec, _ = mug_decoder
    B = 5
    cfg = dict(CFG, mask_noise=True, mask_noise_min=0.2, mask_noise_max=1.5, orientation_repr="discretized",
               orientation_grid_resolution=1)
    gen = SDFVAEViewGenerator(cfg, dec, batch_size=B, seed=11)
    plain = SDFVAEViewGenerator(dict(CFG), dec, batch_size=B, seed=11)
    base = plain.generate()
    aff = (torch.tensor([0.3, 0.9, 0.0, 0.5, 0.7], dtype=torch.float64),
           torch.tensor([[0.0, 2.0], [0.0, -1.0], [0.0, 0.0], [0.0, 1.0], [0.0, -2.0]], dtype=torch.float64),
           torch.tensor([1.0005, 0.9992, 1.0, 1.001, 0.999], dtype=torch.float64))
    noise = torch.tensor([0.5, 0.6, 0.7, 0.8, 0.9])
    out = gen.generate(latent=base["latent_shape"].cpu(), position=base["position"].cpu(),
                       quaternion=base["quaternion"].cpu(), scale=base["scale"].cpu(), mask_affine=aff,
                       mask_noise_value=noise)
    exact = base["depth"] != 0
    final = perturb_masks(base["depth"], inverse_affine_matrices(*aff))
    for b in range(B):
        d = base["depth"][b].clone()                 # the reference's statements, one sample at a time
        d[~exact[b]] = noise[b].item()
        d[~final[b]] = 0
        assert torch.equal(out["depth"][b], d)
    assert (final & ~exact).any() and (exact & ~final).any()     # the perturbation did move the outline
    assert torch.equal(out["depth"][2], base["depth"][2])        # identity map: the exact mask
    grid = SO3Grid(1)
    assert out["orientation"].dtype == torch.long and out["orientation"].shape == (B,)
    for b in range(B):
        assert int(out["orientation"][b]) == grid.quat_to_index(base["quaternion"][b].cpu().numpy().astype(np.float64))
    assert torch.equal(out["quaternion"], base["quaternion"])
    # drawn, not given: distributions of RandomAffine.get_params and of the background depth
    out2 = gen.generate()
    bg = out2["depth"][(out2["depth"] != 0)]
    assert out2["depth"].shape == (B, 240, 320) and bg.numel() > 0
