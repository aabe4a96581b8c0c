"""GPU: seeded random decoder architectures -- Linear stacks of 1 ... 4 layers, 2 ... 4 Conv3d layers with channel
counts that are no multiple of the MFMA tile (1 ... 24), kernel sizes 1 / 3 / 5, resize targets above and below
the incoming size, volume sizes 9 ... 40, batches of 1 ... 48 latents, tsdf clamps -- forward and latent VJP against
the layers written out in torch (float64 on the CPU, test_decoder_gpu.torch_decoder: the layer sequence of
sdf_vae.py:171-259).  The fixed tests pin the mug architecture and two hand-picked ones; which kernel a layer
takes (MFMA im2col, split-K, z-grouped columns, direct convolution, fused resizes, swapped 1x1) depends on its
channel counts, sizes and the batch, and this file walks those choices."""
import os

import numpy as np
import pytest
import torch

import test_decoder_gpu as D
from helpers import rel_err

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(9000 + seed)
    latent = int(rng.integers(1, 13))
    n_fc = int(rng.integers(1, 5))
    n_conv = int(rng.integers(2, 5))
    chans = [int(rng.integers(1, 25)) for _ in range(n_conv)] + [1]
    size = int(rng.integers(2, 7))                      # spatial size entering the first convolution's resize
    fc, cin = [], latent
    for i in range(n_fc - 1):
        fc.append({"out": int(rng.integers(3, 70))})
        cin = fc[-1]["out"]
    conv, cur = [], size
    for i in range(n_conv):
        k = int(rng.choice([1, 3, 3, 5])) if (i > 0 or size >= 5) else int(rng.choice([1, 3] if size >= 3 else [1]))
        # the Linear stack's output IS the first convolution's input (sdf_vae.py:207-215); later layers resize up,
        # down or not at all
        in_size = max(size, k) if i == 0 else int(np.clip(cur + rng.integers(-1, 14), k, 34))
        if k == 5:     # the library keeps a layer's K x 16 weight tile in LDS: Cin k^3 and Cout k^3 <= 963 (it says so)
            chans[i] = min(chans[i], 7)
            if i + 1 < n_conv:
                chans[i + 1] = min(chans[i + 1], 7)
            if i > 0:
                conv[i - 1]["out_channels"] = chans[i]
        conv.append({"in_size": in_size, "in_channels": chans[i], "out_channels": chans[i + 1], "kernel_size": k,
                     "relu": bool(i < n_conv - 1 and rng.uniform() < 0.8)})
        cur = in_size - k + 1
    fc.append({"out": chans[0] * conv[0]["in_size"] ** 3})
    # a ReLU on the LAST layer too, now and then (sdf_vae.py:194-202 allows it; a taped forward then has to keep that
    # layer's output for the VJP's mask -- round 6 found it going straight to the result instead); its own generator, so
    # that the other draws of a seed stay what they were
    if np.random.default_rng(10_000 + seed).uniform() < 0.3:
        conv[-1]["relu"] = True
    volume = int(np.clip(cur + rng.integers(-2, 12), 2, 40))
    tsdf = [False, False, 0.1, True][int(rng.integers(0, 4))]
    N = int(rng.choice([1, 1, 2, 5, 20, 48]))   # 48: the batch forms of the small launches (Linear stack, 1x1, resident MFMA)
    state = {}
    cin = latent
    for i, l in enumerate(fc):
        state[f"decoder._fc_layers.{i}.weight"] = (rng.normal(size=(l["out"], cin)) / np.sqrt(cin)).astype(np.float32)
        state[f"decoder._fc_layers.{i}.bias"] = rng.normal(scale=0.1, size=l["out"]).astype(np.float32)
        cin = l["out"]
    for i, l in enumerate(conv):
        fan = l["in_channels"] * l["kernel_size"] ** 3
        state[f"decoder._conv_layers.{i}.weight"] = (rng.normal(
            size=(l["out_channels"], l["in_channels"]) + (l["kernel_size"],) * 3) / np.sqrt(fan)).astype(np.float32)
        state[f"decoder._conv_layers.{i}.bias"] = rng.normal(scale=0.1, size=l["out_channels"]).astype(np.float32)
    z = rng.normal(size=(N, latent)).astype(np.float32)
    G = rng.normal(size=(N, 1, volume, volume, volume)).astype(np.float32)
    return dict(latent=latent, fc=fc, conv=conv, volume=volume, tsdf=tsdf, N=N, state=state, z=z, G=G)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SDFR_FUZZ_SEEDS", "16"))))
def test_random_architecture(seed):
    from sdfest_amd.vae import SDFDecoder
    c = draw(seed)
    name = (f"seed {seed}: N={c['N']} latent={c['latent']} fc={[l['out'] for l in c['fc']]} "
            f"conv={[(l['in_size'], l['in_channels'], l['out_channels'], l['kernel_size'], l['relu']) for l in c['conv']]}"
            f" volume={c['volume']} tsdf={c['tsdf']}")
    dec = SDFDecoder(c["volume"], c["latent"], c["fc"], c["conv"], tsdf=c["tsdf"], state_dict=c["state"])
    if "SDFR_FUZZ_FUSED_SINGLE" in os.environ:     # a one-off hunt with other fused layer pairs than the default's
        dec.set_option("fused_single", int(os.environ["SDFR_FUZZ_FUSED_SINGLE"]))
    z = torch.tensor(c["z"], device="cuda", requires_grad=True)
    out = dec(z)
    assert out.shape == (c["N"], 1, c["volume"], c["volume"], c["volume"]), name
    z64 = torch.tensor(c["z"], dtype=torch.float64, requires_grad=True)
    ref = D.torch_decoder(c["state"], c["fc"], c["conv"], c["volume"], z64)
    r = ref.detach().numpy()
    floor = None
    if c["conv"][-1]["relu"]:
        # a ReLU'd last layer leaves an output that is mostly zeros and whose largest entry may be tiny against the
        # values the ReLU saw: the yardstick is the size of THOSE (the same decoder without the last ReLU)
        pre = D.torch_decoder(c["state"], c["fc"], c["conv"][:-1] + [dict(c["conv"][-1], relu=False)], c["volume"],
                              z64.detach())
        floor = float(pre.abs().max())
    assert rel_err(out.detach().cpu().numpy(), r, floor) <= 2e-4, name
    if c["tsdf"] is not False:                           # SDFDecoder.forward's clamp (sdf_vae.py:254-257), forward only
        t = 1.0 if c["tsdf"] is True else float(c["tsdf"])
        clamped = dec(z.detach(), enforce_tsdf=True).cpu().numpy()
        assert np.max(np.abs(clamped - np.clip(r, -t, t))) <= 2e-4 * max(np.abs(r).max(), 1e-6), name
    (out * torch.tensor(c["G"], device="cuda")).sum().backward()
    (ref * torch.tensor(c["G"], dtype=torch.float64)).sum().backward()
    gz, gz_ref = z.grad.cpu().numpy(), z64.grad.numpy()
    # a ReLU (or clamp) whose input is within fp32 rounding of 0 switches differently in fp32 and fp64: with random
    # weights that moves single elements of the gradient; the vector as a whole is held to 1e-3 of its largest entry
    assert rel_err(gz, gz_ref) <= 1e-3, name
