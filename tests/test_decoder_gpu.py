"""GPU: the HIP VAE decoder against the golden outputs of the reference's SDFDecoder (mug weights)
and against the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def mug_config(d):
    return {"latent_size": int(d["latent_size"]), "tsdf": False, "decoder": {
        "fc_layers": [{"out": int(o)} for o in d["fc_out"]],
        "conv_layers": [{"in_size": int(a), "in_channels": int(b), "out_channels": int(c),
                         "kernel_size": int(k), "relu": bool(r)}
                        for a, b, c, k, r in zip(d["conv_in_size"], d["conv_cin"], d["conv_cout"],
                                                 d["conv_k"], d["conv_relu"])]}}


@pytest.fixture(scope="module")
def mug():
    d = np.load(os.path.join(GOLDEN, "decoder_mug.npz"))
    w = np.load(os.path.join(GOLDEN, "mug_decoder_weights.npz"))
    return d, {k: w[k] for k in w.files}


def test_mug_decoder_matches_reference_golden(mug):
    from sdfest_amd import SDFDecoder
    d, w = mug
    dec = SDFDecoder.from_config(mug_config(d), w)
    z = torch.tensor(d["z"], device="cuda")
    with torch.no_grad():
        out = dec.decode(z)
    assert out.shape == (12, 1, 64, 64, 64)
    o = out.cpu().numpy()
    ref0 = d["z0_full"]
    assert np.max(np.abs(o[0, 0] - ref0)) <= 1e-4 * np.max(np.abs(ref0))
    for i in range(12):
        sub = d["sub16"][i]
        assert np.max(np.abs(o[i, 0, ::4, ::4, ::4] - sub)) <= 1e-4 * max(1.0, np.abs(sub).max())
        s = d["stats"][i]
        assert abs(o[i].sum(dtype=np.float64) - s[0]) <= 1e-4 * s[1]
    # batched == one at a time, bit for bit
    with torch.no_grad():
        one = dec.decode(z[5:6])
    assert torch.equal(one[0], out[5])


def test_decoder_matches_oracle_on_other_architectures():
    """Random weights, shapes that exercise: Cout > 16 (two channel tiles), k=1 mid-network,
    no final resize, a final resize, tsdf clamping, K not a multiple of 4."""
    from sdfest_amd import SDFDecoder
    rng = np.random.default_rng(0)
    cases = [
        dict(volume=16, latent=5, tsdf=0.1,
             fc=[{"out": 12}, {"out": 3 * 4 ** 3}],
             conv=[dict(in_size=4, in_channels=3, out_channels=20, kernel_size=3, relu=True),
                   dict(in_size=9, in_channels=20, out_channels=6, kernel_size=1, relu=True),
                   dict(in_size=12, in_channels=6, out_channels=1, kernel_size=3, relu=False)]),
        dict(volume=8, latent=3, tsdf=False,
             fc=[{"out": 2 * 5 ** 3}],
             conv=[dict(in_size=5, in_channels=2, out_channels=5, kernel_size=3, relu=True),
                   dict(in_size=10, in_channels=5, out_channels=1, kernel_size=3, relu=False)]),
    ]
    for case in cases:
        state = {}
        width = case["latent"]
        for i, l in enumerate(case["fc"]):
            state[f"decoder._fc_layers.{i}.weight"] = rng.normal(size=(l["out"], width)).astype(np.float32) / np.sqrt(width)
            state[f"decoder._fc_layers.{i}.bias"] = rng.normal(size=l["out"]).astype(np.float32) * 0.1
            width = l["out"]
        for i, l in enumerate(case["conv"]):
            k = l["kernel_size"]
            fan = l["in_channels"] * k ** 3
            state[f"decoder._conv_layers.{i}.weight"] = rng.normal(
                size=(l["out_channels"], l["in_channels"], k, k, k)).astype(np.float32) / np.sqrt(fan)
            state[f"decoder._conv_layers.{i}.bias"] = rng.normal(size=l["out_channels"]).astype(np.float32) * 0.1
        cfg = {"latent_size": case["latent"], "tsdf": case["tsdf"], "sdf_size": case["volume"],
               "decoder": {"fc_layers": case["fc"], "conv_layers": case["conv"]}}
        dec = SDFDecoder.from_config(cfg, state, sdf_size=case["volume"])
        z = rng.normal(size=(3, case["latent"])).astype(np.float32)
        params = oracle.pack_decoder_params(state, len(case["fc"]), len(case["conv"]))
        for enforce in (False, True):
            ref = oracle.decoder_forward(params, cfg, z, dtype=np.float32, enforce_tsdf=enforce)
            with torch.no_grad():
                out = dec.forward(torch.tensor(z, device="cuda"), enforce_tsdf=enforce).cpu().numpy()
            assert out.shape == ref.shape
            assert np.max(np.abs(out - ref)) <= 1e-4 * max(np.abs(ref).max(), 1e-3), case["volume"]
            if enforce and case["tsdf"]:
                assert np.abs(out).max() <= case["tsdf"] + 1e-7


def test_decoder_feeds_renderer(mug):
    """decode(z) -> render: the SDFPipeline hand-off (simple_setup.py:413-434), no host copy."""
    from sdfest_amd import Camera, SDFDecoder, render_depth_gpu
    d, w = mug
    dec = SDFDecoder.from_config(mug_config(d), w)
    with torch.no_grad():
        sdf = dec.decode(torch.zeros(1, 8, device="cuda"))
    cam = Camera(160, 120, 80.0, 80.0, 80.0, 60.0, pixel_center=0.5)
    p = torch.tensor([0.0, 0.0, -0.3], device="cuda")
    q = torch.tensor([0.0, 0.0, 0.0, 1.0], device="cuda")
    depth = render_depth_gpu(sdf[0, 0], p, q, torch.tensor(1 / 0.1, device="cuda"), None, None, None,
                             0.005, cam)
    ref = oracle.render_forward(d["z0_full"], [0, 0, -0.3], [0, 0, 0, 1], [10.0], 160, 120, 80.0, 60.0,
                                80.0, 80.0, 0.005, dtype=np.float32)[0]
    dd = depth.cpu().numpy()
    assert (ref > 0).sum() > 1000
    agree = (dd > 0) == (ref > 0)
    assert agree.mean() > 0.999
    both = (dd > 0) & (ref > 0)
    assert np.max(np.abs(dd[both] / ref[both] - 1)) < 1e-3   # the two SDFs differ by fp32 rounding
