"""VAE decoder: host-side mirror of ``sdfest/vae/sdf_vae.py::SDFDecoder`` (:171-259).

Built from the reference's own config dict (``decoder: {fc_layers, conv_layers}``,
``latent_size``, ``tsdf``) and a state dict with the reference's parameter names
(``decoder._fc_layers.{i}.weight`` ...).  ``forward`` / ``decode`` run in ``libsdfr_hip.so``
(decoder.hip): one launch for the Linear stack, one MFMA launch per Conv3d, one per resize.
"""
import ctypes
from typing import Mapping, Optional, Union

import numpy as np
import torch

from . import _lib
from .differentiable_renderer import _stream


def _flatten_state(state: Mapping, n_fc: int, n_conv: int, prefix: str) -> np.ndarray:
    def get(name):
        t = state[name]
        if isinstance(t, torch.Tensor):
            t = t.detach().cpu().numpy()
        return np.asarray(t, dtype=np.float32).reshape(-1)

    parts = []
    for i in range(n_fc):
        parts += [get(f"{prefix}_fc_layers.{i}.weight"), get(f"{prefix}_fc_layers.{i}.bias")]
    for i in range(n_conv):
        parts += [get(f"{prefix}_conv_layers.{i}.weight"), get(f"{prefix}_conv_layers.{i}.bias")]
    return np.ascontiguousarray(np.concatenate(parts))


class _DecodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, dec):
        zc = z.detach().contiguous()
        nbytes = dec._L.sdfr_decoder_tape_bytes(dec._h, zc.shape[0])
        tape = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=dec.device)
        out = dec._forward_raw(zc, False, tape)
        ctx.save_for_backward(zc, tape)
        ctx.dec = dec
        return out

    @staticmethod
    def backward(ctx, grad_out):
        zc, tape = ctx.saved_tensors
        return ctx.dec._backward_raw(zc, tape, grad_out.contiguous()), None


class SDFDecoder:
    """Decoder of the SDF VAE (reference: SDFDecoder.__init__ sdf_vae.py:171-205)."""

    def __init__(self, volume_size: int, latent_size: int, fc_layers: list, conv_layers: list,
                 tsdf: Optional[Union[bool, float]] = False, state_dict: Optional[Mapping] = None,
                 device="cuda", prefix: str = "decoder."):
        # reference: SDFDecoder.sanity_check (sdf_vae.py:207-215)
        assert fc_layers[-1]["out"] == conv_layers[0]["in_channels"] * conv_layers[0]["in_size"] ** 3
        for a, b in zip(conv_layers[:-1], conv_layers[1:]):
            assert a["out_channels"] == b["in_channels"]
        assert conv_layers[-1]["out_channels"] == 1
        if state_dict is None:
            raise ValueError("state_dict with the decoder parameters is required")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._volume_size = volume_size
        self.latent_size = latent_size
        self._tsdf = tsdf
        if not any(k.startswith(prefix) for k in state_dict):
            prefix = ""
        params = _flatten_state(state_dict, len(fc_layers), len(conv_layers), prefix)
        arr = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        fc_out = arr([l["out"] for l in fc_layers])
        ins = arr([l["in_size"] for l in conv_layers])
        cin = arr([l["in_channels"] for l in conv_layers])
        cout = arr([l["out_channels"] for l in conv_layers])
        ks = arr([l["kernel_size"] for l in conv_layers])
        relu = arr([1 if l["relu"] else 0 for l in conv_layers])
        L = _lib.lib()
        handle = ctypes.c_void_p()
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = L.sdfr_decoder_create(P(params), params.size, latent_size, len(fc_layers), P(fc_out),
                                   len(conv_layers), P(ins), P(cin), P(cout), P(ks), P(relu),
                                   volume_size, float(tsdf) if tsdf is not False else 0.0,
                                   self.device.index, ctypes.byref(handle))
        _lib.check(rc, "sdfr_decoder_create")
        self._L, self._h = L, handle
        self._ws = {}     # stream handle -> scratch buffer
        self._fc_widths = [int(latent_size)] + [int(o) for o in fc_out]

    def narrow_linear_stack(self) -> bool:
        """Whether the Linear stack's leading layers run as ONE wave out of LDS (csrc/decoder_fc.hpp, fc_one_wave_ok: every
        layer input at most 64 wide, at most 8 layers, their parameters within 6144 floats -- the mug decoder's
        8 -> 20 -> 50 -> 8192 does) and that form is switched on for this handle: what lets the captured loop's tail
        launch run the next iteration's Linear stack (``FusedRenderAndCompare(fc_in_tail=)``)."""
        w = self._fc_widths
        span = sum(w[l] * w[l + 1] + w[l + 1] for l in range(len(w) - 2))
        if len(w) - 1 > 8 or max(w[:-1]) > 64 or span > 6144:
            return False
        on = self.set_option("fc_one_wave", 1)
        if on != 1:
            self.set_option("fc_one_wave", on)
        return on == 1

    @classmethod
    def from_config(cls, config: Mapping, state_dict: Mapping, device="cuda", sdf_size: int = 64):
        """config: the reference's vae config (keys latent_size, decoder, tsdf)."""
        return cls(sdf_size, config["latent_size"], config["decoder"]["fc_layers"],
                   config["decoder"]["conv_layers"], config.get("tsdf", False), state_dict, device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.sdfr_decoder_destroy(h)
            self._h = None

    OPTIONS = {"fused_resize": 0, "tiled_vjp": 1, "fc_one_wave": 2, "fused_single": 3}     # SDFR_DECODER_OPT_* (include/sdfr.h)

    def set_option(self, name: str, value: int) -> int:
        """Select one of two equivalent kernel forms for THIS decoder (``sdfr_decoder_set_option``: same results bit
        for bit, the defaults are the faster forms); returns the old value."""
        old = self._L.sdfr_decoder_set_option(self._h, self.OPTIONS[name], int(value))
        if old < 0:
            _lib.check(old, "sdfr_decoder_set_option")
        return old

    def _scratch(self, need: int) -> torch.Tensor:
        """Scratch of the calls on the CURRENT stream, grown on demand -- one buffer per stream, like the renderer's
        (``differentiable_renderer._workspace``): two streams driving the same decoder (the view generator decodes
        the next batch on a side stream while its consumer decodes or differentiates on its own) never share
        intermediate tensors without an ordering between them.  A buffer that is outgrown goes back to the caching
        allocator, which keeps it out of circulation until the work queued on ITS stream has finished."""
        key = _stream(self.device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _forward_raw(self, zc: torch.Tensor, enforce_tsdf: bool, tape: Optional[torch.Tensor]):
        N, D = zc.shape[0], self._volume_size
        out = torch.empty((N, 1, D, D, D), dtype=torch.float32, device=self.device)
        ws = self._scratch(self._L.sdfr_decoder_workspace_bytes(self._h, N))
        rc = self._L.sdfr_decoder_forward(self._h, zc.data_ptr(), N, int(bool(enforce_tsdf)),
                                          out.data_ptr(), None if tape is None else tape.data_ptr(),
                                          ws.data_ptr(), ws.numel(),
                                          _stream(self.device))
        _lib.check(rc, "sdfr_decoder_forward")
        return out

    def _backward_raw(self, zc: torch.Tensor, tape: torch.Tensor, grad_out: torch.Tensor):
        N = zc.shape[0]
        g_z = torch.empty_like(zc)
        ws = self._scratch(self._L.sdfr_decoder_backward_workspace_bytes(self._h, N))
        rc = self._L.sdfr_decoder_backward_latent(self._h, zc.data_ptr(), tape.data_ptr(),
                                                  grad_out.data_ptr(), N, g_z.data_ptr(),
                                                  ws.data_ptr(), ws.numel(),
                                                  _stream(self.device))
        _lib.check(rc, "sdfr_decoder_backward_latent")
        return g_z

    def forward(self, z: torch.Tensor, enforce_tsdf: bool = False) -> torch.Tensor:
        """z (N, latent_size) -> (N, 1, D, D, D), like SDFDecoder.forward (sdf_vae.py:217-259).

        Differentiable w.r.t. z (the weights are constants, as in the estimation loop where only
        the latent is optimised, simple_setup.py:400-406)."""
        if not z.is_cuda or z.dtype != torch.float32:
            raise RuntimeError("z must be a float32 CUDA tensor")
        if z.dim() != 2 or z.shape[1] != self.latent_size:
            raise RuntimeError(f"z must have shape (N, {self.latent_size})")
        if z.requires_grad and torch.is_grad_enabled():
            if enforce_tsdf and self._tsdf is not False:
                raise NotImplementedError("gradient through the tsdf clamp is not supported")
            return _DecodeFn.apply(z, self)
        return self._forward_raw(z.detach().contiguous(), enforce_tsdf, None)

    __call__ = forward

    def decode(self, z: torch.Tensor, enforce_tsdf: bool = False) -> torch.Tensor:
        """Same as SDFVAE.decode (sdf_vae.py:79-87)."""
        return self.forward(z, enforce_tsdf)
