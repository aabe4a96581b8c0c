"""GPU: the HIP trilinear sampler (pc_loss) against the oracle and the torch-reference goldens."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLDEN, dense_from_sparse, rel_err

pytestmark = pytest.mark.gpu
REL = 1e-4


def dev(a, dtype=np.float32):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype), device="cuda")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "pc_loss.npz"))


def test_single_view_matches_torch_reference_goldens(golden):
    from sdfest_amd import pc_loss
    d = golden
    sdf_np = oracle.blobs_sdf(0)
    for i in range(int(d["n_cases"])):
        pts = dev(d[f"c{i}_points"])
        pos = dev(d[f"c{i}_pos"]).requires_grad_()
        quat = dev(d[f"c{i}_quat"]).requires_grad_()
        scale = dev(d[f"c{i}_scale"]).requires_grad_()
        sdf = dev(sdf_np).requires_grad_()
        val = pc_loss(pts, pos, quat, scale, sdf)
        ref = d[f"c{i}_f64_value"]
        v = val.detach().cpu().numpy()
        same = (v != 0) == (ref != 0)
        assert same.mean() > 0.995                       # fp32 may move a point across the face
        assert np.max(np.abs(v - ref)[same]) <= REL * np.abs(ref).max()
        go = d[f"c{i}_gout"].astype(np.float32).copy()
        go[~same] = 0                                    # compare gradients on agreed points
        val.backward(dev(go))
        o = oracle.pc_loss_backward(go, d[f"c{i}_points"], d[f"c{i}_pos"], d[f"c{i}_quat"],
                                    d[f"c{i}_scale"], sdf_np, dtype=np.float64)
        assert rel_err(sdf.grad.cpu().numpy(), o[0]) <= REL
        assert rel_err(pos.grad.cpu().numpy(), o[1]) <= REL
        assert rel_err(quat.grad.cpu().numpy(), o[2]) <= REL
        assert abs(scale.grad.item() - o[3]) <= REL * abs(o[3])
        if same.all():
            # then the comparison is directly against torch autograd of the reference
            assert rel_err(pos.grad.cpu().numpy(), d[f"c{i}_f64_gpos"]) <= REL
            assert rel_err(quat.grad.cpu().numpy(), d[f"c{i}_f64_gquat"]) <= REL
            assert rel_err(sdf.grad.cpu().numpy(),
                           dense_from_sparse(d[f"c{i}_f64_gsdf_idx"], d[f"c{i}_f64_gsdf_val"])) <= REL


def test_batched_views_equal_single_view_calls_and_oracle():
    from sdfest_amd import pc_loss, pc_loss_batch
    rng = np.random.default_rng(3)
    sdf_np = oracle.blobs_sdf(0)
    B = 5
    lens = [1000, 1, 0, 777, 2500]
    pos = rng.uniform(-0.1, 0.1, (B, 3)) + np.array([0, 0, -0.8])
    quat = rng.normal(size=(B, 4))
    scale = rng.uniform(0.2, 0.4, B)
    pts = [pos[b] + rng.uniform(-1.2, 1.2, (n, 3)) * scale[b] for b, n in enumerate(lens)]
    allp = np.concatenate(pts).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    go = rng.uniform(-1, 1, allp.shape[0]).astype(np.float32)
    tp, tq, ts = dev(pos).requires_grad_(), dev(quat).requires_grad_(), dev(scale).requires_grad_()
    tsdf = dev(sdf_np).requires_grad_()
    out = pc_loss_batch(dev(allp), dev(offs, np.int32), max(lens), tp, tq, ts, tsdf)
    out.backward(dev(go))
    acc = np.zeros((64, 64, 64))
    for b in range(B):
        sl = slice(offs[b], offs[b + 1])
        ref = oracle.pc_loss_forward(allp[sl], pos[b], quat[b], scale[b], sdf_np, dtype=np.float32)
        v = out.detach().cpu().numpy()[sl]
        same = (v != 0) == (ref != 0)
        assert same.mean() >= 0.995 if len(v) > 10 else True
        if len(v):
            assert np.max(np.abs(v - ref)[same], initial=0) <= REL * max(np.abs(ref).max(), 1e-6)
        g = go[sl].copy()
        g[~same] = 0
        if not same.all():
            continue
        o = oracle.pc_loss_backward(g, allp[sl], pos[b], quat[b], scale[b], sdf_np, dtype=np.float64)
        acc += o[0]
        if len(v):
            assert rel_err(tp.grad[b].cpu().numpy(), o[1]) <= REL
            assert rel_err(tq.grad[b].cpu().numpy(), o[2]) <= REL
            assert abs(ts.grad[b].item() - o[3]) <= REL * max(abs(o[3]), 1e-6)
        else:
            assert not tp.grad[b].any() and not tq.grad[b].any() and ts.grad[b].item() == 0
        # the single-view drop-in gives the same values bit for bit
        if len(v):
            p1 = dev(pos[b]); q1 = dev(quat[b]); s1 = dev(scale[b])
            v1 = pc_loss(dev(allp[sl]), p1, q1, s1, dev(sdf_np)).cpu().numpy()
            assert np.array_equal(v1, v)
    assert rel_err(tsdf.grad.cpu().numpy(), acc) <= REL


def test_full_size_point_cloud_properties():
    """307200 points (every pixel of a 640x480 depth image): linearity in grad_out, zero grad
    for outside points, generic resolution path."""
    from sdfest_amd import pc_loss
    rng = np.random.default_rng(0)
    for Rn in (64, 40):
        sdf_np = oracle.sphere_sdf(0.5, R=Rn)
        M = 307200
        pts = dev(rng.uniform(-0.4, 0.4, (M, 3)) + np.array([0, 0, -1.0]))
        pos = dev([0.0, 0.0, -1.0]).requires_grad_()
        quat = dev([0.1, -0.2, 0.3, 0.9]).requires_grad_()
        scale = dev(0.3).requires_grad_()
        sdf = dev(sdf_np).requires_grad_()
        g1 = dev(rng.uniform(-1, 1, M)); g2 = dev(rng.uniform(-1, 1, M))
        grads = []
        for g in (g1, g2, g1 + 2 * g2):
            for t in (pos, quat, scale, sdf):
                t.grad = None
            val = pc_loss(pts, pos, quat, scale, sdf)
            val.backward(g)
            grads.append([t.grad.clone() for t in (pos, quat, scale, sdf)])
        for a, b, c in zip(*grads):
            s = (a.abs().max() + 2 * b.abs().max()).item()
            assert (a + 2 * b - c).abs().max().item() <= 3e-5 * s
        v = val.detach().cpu().numpy()
        ref = oracle.pc_loss_forward(pts.cpu().numpy(), [0, 0, -1.0], [0.1, -0.2, 0.3, 0.9], 0.3, sdf_np)
        same = (v != 0) == (ref != 0)
        assert same.mean() > 0.9999
        assert np.max(np.abs(v - ref)[same]) <= REL * np.abs(ref).max()
        assert 0.3 < (v != 0).mean() < 0.5   # a third of the points lie inside the volume


def test_l1_backward_in_one_pass_equals_the_three_call_sequence():
    """sdfr_pc_l1_backward against sdfr_pc_loss_forward -> sdfr_pc_l1_loss -> sdfr_pc_loss_backward:
    same loss, bit-identical pose gradients (fixed-order sums of identical terms), d/dSDF up to the
    order of float atomics; ragged views, an empty view, points outside the volume."""
    from sdfest_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    lens = [700, 0, 1500, 257]
    V = len(lens)
    pts = np.concatenate([rng.uniform(-0.9, 0.9, (n, 3)) for n in lens]).astype(np.float32) + np.array([0, 0, -1.0], np.float32)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    pos = np.tile(np.array([[0.02, -0.03, -1.0]], np.float32), (V, 1)) + rng.normal(0, 0.02, (V, 3)).astype(np.float32)
    quat = rng.normal(size=(V, 4)).astype(np.float32)
    scale = rng.uniform(0.5, 0.8, V).astype(np.float32)
    sdf = oracle.blobs_sdf(0)
    t = lambda a, dt=torch.float32: torch.tensor(a, dtype=dt, device=dev)
    P, O, p, q, s, S = t(pts), t(offs, torch.int32), t(pos), t(quat), t(scale), t(sdf)
    N, M, w = len(pts), max(lens), 3.0
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(max(L.sdfr_pc_loss_backward_workspace_bytes(V, M), 256), dtype=torch.uint8, device=dev)

    def outs():
        return (torch.empty(64, 64, 64, device=dev), torch.empty(V, 3, device=dev), torch.empty(V, 4, device=dev),
                torch.empty(V, device=dev))
    vals, gv, loss_a = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(V, device=dev)
    _lib.check(L.sdfr_pc_loss_forward(P.data_ptr(), O.data_ptr(), V, M, p.data_ptr(), q.data_ptr(), s.data_ptr(),
                                      S.data_ptr(), 64, 0, vals.data_ptr(), 0, st), "fwd")
    _lib.check(L.sdfr_pc_l1_loss(vals.data_ptr(), O.data_ptr(), V, M, w, loss_a.data_ptr(), gv.data_ptr(), 0, st), "l1")
    a = outs()
    _lib.check(L.sdfr_pc_loss_backward(gv.data_ptr(), P.data_ptr(), O.data_ptr(), V, M, p.data_ptr(), q.data_ptr(),
                                       s.data_ptr(), S.data_ptr(), 64, 0, a[0].data_ptr(), 0, a[1].data_ptr(),
                                       a[2].data_ptr(), a[3].data_ptr(), ws.data_ptr(), ws.numel(), 0, st), "bwd")
    b, loss_b = outs(), torch.empty(V, device=dev)
    _lib.check(L.sdfr_pc_l1_backward(w, loss_b.data_ptr(), P.data_ptr(), O.data_ptr(), V, M, p.data_ptr(),
                                     q.data_ptr(), s.data_ptr(), S.data_ptr(), 64, 0, b[0].data_ptr(), 0,
                                     b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), ws.data_ptr(), ws.numel(),
                                     0, st), "l1 bwd")
    la, lb = loss_a.cpu().numpy(), loss_b.cpu().numpy()
    assert np.isnan(la[1]) and np.isnan(lb[1])
    np.testing.assert_allclose(lb[[0, 2, 3]], la[[0, 2, 3]], rtol=2e-6)
    for k in (1, 2, 3):
        assert torch.equal(a[k], b[k]), k
    ga, gb = a[0].cpu().numpy(), b[0].cpu().numpy()
    assert np.abs(ga).max() > 0 and np.max(np.abs(ga - gb)) <= 1e-5 * np.abs(ga).max()
    assert (vals == 0).sum() > 10          # some points fall outside the volume
