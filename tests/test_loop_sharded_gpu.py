"""GPU: the render-and-compare loop sharded over ranks (simple_setup.py:408-470 with the views of :420-446 spread
over processes, ONE exchange per iteration) against the same loop in a single process.  Two ranks share the test
box's one GPU (gloo on device tensors; on a node the same code path runs over RCCL).
  * deterministic mode (SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES, exchange "sdf", integer bucket): the trajectory
    is BITWISE the single-process one;
  * default mode: within 2e-5 of an Adam step's scale per iteration (float atomics, rounding of the summed bucket);
  * every rank ends with the same bits (replicated state, nothing broadcast)."""
import os
import sys

import numpy as np
import pytest
import torch

import _loop_scenes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _spawn(tmp_path, scene, flavour, exchange, graph, form="fused", world=2, backend="gloo"):
    from sdfest_amd.parallel import spawn_ranks
    out = str(tmp_path / f"loop_{scene}_{flavour}_{exchange}_{graph}_{form}.npz")
    rc = spawn_ranks([sys.executable, os.path.join(HERE, "_loop_worker.py"), out, scene, flavour, exchange, graph, form,
                      backend], world, timeout=400)
    assert rc == 0
    r = np.load(out)
    assert bool(r["ranks_identical"]), "the ranks' replicated states differ"
    return r


def _single(scene, mode=0, graph=False, **kw):
    from sdfest_amd.pipeline import FusedRenderAndCompare
    sc = _loop_scenes.build(scene)
    loop = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], camera_positions=sc["cam_pos"],
                                 camera_orientations=sc["cam_quat"], shape_optimization=True, sdf_grad_mode=mode,
                                 track_inliers=True, **kw)
    hist = []
    loop(*sc["init"], use_graph=graph, history=hist)
    torch.cuda.synchronize()
    return (_loop_scenes.history_array(hist), np.array([float(h["loss"]) for h in hist]),
            loop.inlier_history.cpu().numpy())


def _steps(traj_a, traj_b):
    """largest parameter difference per iteration in units of that parameter group's Adam step (its learning rate)"""
    n = traj_a.shape[1]
    lr = np.array([1e-3] * 3 + [1e-2] * 4 + [1e-3] + [1e-2] * (n - 8))
    return (np.abs(traj_a - traj_b) / lr).max(axis=1)


@pytest.mark.parametrize("scene", ["g7a", "seven"])
@pytest.mark.parametrize("graph", ["eager", "graph"])
def test_sharded_loop_is_bitwise_the_single_process_loop_in_the_deterministic_mode(tmp_path, scene, graph):
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    r = _spawn(tmp_path, scene, "det", "sdf", graph)
    V = 2 if scene == "g7a" else 7
    assert r["shards"].tolist() == [[0, (V + 1) // 2], [(V + 1) // 2, V]]
    traj, loss, inl = _single(scene, SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES, graph == "graph")
    assert np.isfinite(traj).all() and np.abs(traj[-1] - traj[0]).max() > 1e-4      # the loop moved
    assert np.array_equal(traj, r["traj"]), _steps(traj, r["traj"])
    assert np.array_equal(inl, r["inlier"])
    # the loss VALUES are fixed-order sums over the forward's tiles, whose shape follows the batch size: they agree
    # to rounding, not to the bit (they do not enter the gradients)
    assert np.allclose(loss, r["loss"], rtol=2e-6, atol=0)
    # and against the default single-process loop (float atomics, batch tiles, one-launch tail)
    traj0, loss0, _ = _single(scene)
    assert _steps(traj0, r["traj"]).max() < 2e-3 and np.allclose(loss0, r["loss"], rtol=1e-4)


@pytest.mark.parametrize("scene", ["g7a", "seven"])
@pytest.mark.parametrize("exchange", ["sdf", "latent"])
def test_sharded_loop_matches_the_single_process_loop(tmp_path, scene, exchange):
    r = _spawn(tmp_path, scene, "float", exchange, "graph")
    traj, loss, inl = _single(scene, graph=True)
    # per iteration: 2e-5 of the parameter's scale ~ 2e-3 of one Adam step at these learning rates; differences
    # come from the order of float atomics and of the bucket's sum, and Adam's normalisation keeps them from growing
    steps = _steps(traj, r["traj"])
    assert steps.max() < 5e-3, steps
    scale = np.maximum(np.abs(traj), 1e-2)
    assert np.max(np.abs(traj - r["traj"]) / scale) < 2e-5 * traj.shape[0], np.max(np.abs(traj - r["traj"]) / scale)
    assert np.allclose(loss, r["loss"], rtol=1e-4) and np.max(np.abs(inl - r["inlier"])) < 2.5 / 300


def test_records_form_multi_iteration_graph_replays(tmp_path, monkeypatch):
    """The records form's replay paths that run only WITHOUT a history (the default for 8 or more views, and the only
    form a process group can use): two ranks replaying `tail + next head` between the all-reduces
    (``_run_records``' graph_many), and a single process replaying graphs of 3 and of 5 iterations with a remainder
    (``_run_records_single``) -- the plan's ring_reset and volume bookkeeping inside a multi-iteration capture, the
    head / tail interleaving around the collective.  Deterministic mode: final parameters, step count and inlier
    history are BITWISE the eager run's."""
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    from sdfest_amd.pipeline import FusedRenderAndCompare
    det = SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES
    n_iter = 7
    sc = _loop_scenes.build("seven", iterations=n_iter)

    def single(use_graph, graph_iterations=5, history=False):
        loop = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], camera_positions=sc["cam_pos"],
                                     camera_orientations=sc["cam_quat"], shape_optimization=True, sdf_grad_mode=det,
                                     track_inliers=True, graph_iterations=graph_iterations)
        assert loop.records_form
        res = []
        for _ in range(2):        # the second call replays what the first captured
            out = loop(*sc["init"], use_graph=use_graph, history=[] if history else None)
            torch.cuda.synchronize()
            res.append((np.concatenate([o.cpu().numpy().ravel() for o in out]), int(loop.step.item()),
                        loop.inlier_history.cpu().numpy().copy(), loop.best_state.cpu().numpy().copy()))
        assert loop.graph_many is not None or not use_graph or graph_iterations == 1
        return res
    ref = single(False, history=True)[0]
    assert ref[1] == n_iter and np.abs(ref[0][8:]).max() > 1e-4 and ref[2][:n_iter].max() > 0
    for gi in (5, 3, 1):
        for got in single(True, gi):
            assert np.array_equal(got[0], ref[0]), (gi, got[0], ref[0])
            assert got[1] == ref[1] and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), gi
    # two ranks, no history: graph_many = this iteration's tail and the next one's head between two all-reduces
    monkeypatch.setenv("SDFR_TEST_ITERATIONS", str(n_iter))
    r = _spawn(tmp_path, "seven", "det", "sdf", "graph_nohist")
    assert int(r["steps_taken"]) == n_iter
    assert np.array_equal(r["final"], ref[0]), (r["final"], ref[0])
    assert np.array_equal(r["inlier"], ref[2])


def test_deterministic_backward_with_the_latent_exchange(tmp_path):
    """SDF_GRAD_DETERMINISTIC with exchange="latent": every rank's d/d latent is a float contribution, so this bucket
    is summed as floats (summing its bits as integers would be no sum at all) -- the trajectory follows the
    single-process one to rounding, not to the bit"""
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    r = _spawn(tmp_path, "seven", "det", "latent", "graph")
    traj, loss, _ = _single("seven", SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES, True)
    assert _steps(traj, r["traj"]).max() < 2e-3 and np.allclose(loss, r["loss"], rtol=1e-4)
    assert np.isfinite(r["traj"]).all() and np.abs(r["traj"][-1, 8:]).max() > 1e-4      # the latent moved


def test_sharded_loop_matches_g7_run_a(tmp_path):
    """the sharded loop against the reference-derived golden itself (same bounds as the single-process G7 test)"""
    g7 = np.load(os.path.join(_loop_scenes.GOLDEN, "loop_g7.npz"))
    r = _spawn(tmp_path, "g7a", "float", "sdf", "eager")
    traj = g7["a_traj"]
    lr = np.array([1e-3] * 3 + [1e-2] * 4 + [1e-3] + [1e-2] * (traj.shape[1] - 8))
    for it in range(traj.shape[0]):
        assert (np.abs(r["traj"][it] - traj[it]) / lr).max() < 0.02 * (it + 1)
        assert abs(r["loss"][it] - g7["a_terms"][it, 3]) < 2e-4 * abs(g7["a_terms"][it, 3]) + 1e-6
    assert np.max(np.abs(r["inlier"][:len(g7["a_inlier"])] - g7["a_inlier"])) < 2.5 / 300.0


def test_sharded_autograd_loop_matches_the_single_process_one(tmp_path):
    from sdfest_amd.pipeline import RenderAndCompare
    r = _spawn(tmp_path, "seven", "float", "latent", "eager", form="autograd")
    sc = _loop_scenes.build("seven")
    hist = []
    RenderAndCompare(sc["decoder"], sc["camera"], sc["config"])(
        sc["depth"], *sc["init"], camera_positions=sc["cam_pos"], camera_orientations=sc["cam_quat"],
        shape_optimization=True, history=hist)
    traj = _loop_scenes.history_array(hist)
    assert _steps(traj, r["traj"]).max() < 5e-3
    assert np.allclose([float(h["loss"]) for h in hist], r["loss"], rtol=1e-4)


def test_sharded_pose_only_loop_with_a_point_constraint(tmp_path):
    """shape optimisation off (the SDF is decoded once, the exchange is the view records alone, d/dSDF is computed and
    dropped as in simple_setup.py:413-414) with the point constraint of :164-175: bitwise in the deterministic mode"""
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    from sdfest_amd.pipeline import FusedRenderAndCompare
    r = _spawn(tmp_path, "seven", "det", "sdf", "graph", form="fused_pose_only")
    sc = _loop_scenes.build("seven")
    con = (torch.tensor([0.0, 1.0, 0.0]), torch.tensor([0.1, 0.9, -0.2]), 0.05)
    loop = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], camera_positions=sc["cam_pos"],
                                 camera_orientations=sc["cam_quat"], shape_optimization=False,
                                 sdf_grad_mode=SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES, track_inliers=True,
                                 point_constraint=con)
    hist = []
    loop(*sc["init"], use_graph=True, history=hist)
    torch.cuda.synchronize()
    traj = _loop_scenes.history_array(hist)
    assert np.array_equal(traj, r["traj"]) and np.array_equal(traj[:, 8:], np.zeros_like(traj[:, 8:]))   # the latent stays
    # and the default single-process loop with the same constraint agrees to rounding
    ref = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], camera_positions=sc["cam_pos"],
                                camera_orientations=sc["cam_quat"], shape_optimization=False, point_constraint=con)
    h2 = []
    ref(*sc["init"], use_graph=True, history=h2)
    assert _steps(_loop_scenes.history_array(h2), r["traj"]).max() < 2e-3


def test_three_ranks_on_seven_views(tmp_path):
    """an uneven split (3 + 2 + 2 views) in the deterministic mode: still the single-process bits"""
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    r = _spawn(tmp_path, "seven", "det", "sdf", "eager", world=3)
    assert r["shards"].tolist() == [[0, 3], [3, 5], [5, 7]]
    traj, _, inl = _single("seven", SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES)
    assert np.array_equal(traj, r["traj"]) and np.array_equal(inl, r["inlier"])


def test_eighty_views_on_two_ranks(tmp_path):
    """a view list longer than the single-process tail takes (64) and than one round of the records tail's chain:
    2 x 40 views, bitwise the single-process records form in the deterministic mode, and the autograd loop to rounding"""
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    from sdfest_amd.pipeline import RenderAndCompare
    r = _spawn(tmp_path, "many", "det", "sdf", "graph")
    assert r["shards"].tolist() == [[0, 40], [40, 80]]
    traj, _, inl = _single("many", SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES, True)
    assert np.array_equal(traj, r["traj"]) and np.array_equal(inl, r["inlier"])
    sc = _loop_scenes.build("many")
    hist = []
    RenderAndCompare(sc["decoder"], sc["camera"], sc["config"])(
        sc["depth"], *sc["init"], camera_positions=sc["cam_pos"], camera_orientations=sc["cam_quat"],
        shape_optimization=True, history=hist)
    assert _steps(_loop_scenes.history_array(hist), r["traj"]).max() < 5e-3


def test_more_ranks_than_views_is_refused():
    from sdfest_amd.parallel import shard_views
    assert shard_views(2, 2, 3) == (2, 2)          # an empty shard: the loop refuses it (every rank needs a view)


@pytest.mark.parametrize("graph", ["eager", "graph"])
def test_sharded_loop_over_rccl_with_one_rank(tmp_path, graph):
    """The same worker over the backend the multi-GPU node uses -- "nccl" is RCCL on ROCm -- with the one rank a
    one-GPU box allows: process-group start-up, the bucket's all-reduce on device memory, the captured head / tail
    graphs around the collective (thread-local capture beside the process group's watchdog thread).  A one-rank sum
    changes nothing, so the trajectory is the single-process one."""
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    r = _spawn(tmp_path, "g7a", "det", "sdf", graph, world=1, backend="nccl")
    traj, _, inl = _single("g7a", SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES, graph == "graph")
    assert np.array_equal(traj, r["traj"]) and np.array_equal(inl, r["inlier"])
    r = _spawn(tmp_path, "g7a", "float", "latent", graph, world=1, backend="nccl")
    traj, _, _ = _single("g7a", graph=graph == "graph")
    assert _steps(traj, r["traj"]).max() < 5e-3


def test_collective_inside_the_graph_over_rccl_with_one_rank(tmp_path):
    """graph_collective=True: head | all-reduce | tail captured as ONE graph (c10d's NCCL backend is capturable).
    Whether or not this platform captures RCCL's launch, the trajectory must be the two-graph form's bit for bit:
    captured -> the whole-iteration graphs replay; refused -> ``graph_collective_error`` says why and the loop runs
    the two-graph form."""
    from sdfest_amd.differentiable_renderer import BWD_SMALL_TILES, SDF_GRAD_DETERMINISTIC
    r = _spawn(tmp_path, "g7a", "det", "sdf", "graph_one", world=1, backend="nccl")
    note = open(str(tmp_path / "loop_g7a_det_sdf_graph_one_fused.npz") + ".graph_one.txt").read().split("\n")
    print("collective captured inside the graph:", note[0], "| error:", note[1])
    traj, _, inl = _single("g7a", SDF_GRAD_DETERMINISTIC | BWD_SMALL_TILES, True)
    assert np.array_equal(traj, r["traj"]) and np.array_equal(inl, r["inlier"])
    assert note[0] == "True" or note[1] != "None"


def test_single_process_forms_agree_and_auto_picks_by_view_count():
    """form="tail" (one workgroup reduces every view) and form="records" (one wave per view; what a process group
    uses) are the same loop: trajectories agree to rounding, eager and captured; "auto" takes the records form from 8
    views on (the faster one there) and the tail below."""
    from sdfest_amd.pipeline import FusedRenderAndCompare
    sc = _loop_scenes.build("seven")
    kw = dict(camera_positions=sc["cam_pos"], camera_orientations=sc["cam_quat"], shape_optimization=True,
              track_inliers=True)
    trajs = {}
    for form in ("tail", "records"):
        for graph in (False, True):
            loop = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], form=form, **kw)
            assert loop.records_form == (form == "records")
            hist = []
            loop(*sc["init"], use_graph=graph, history=hist)
            torch.cuda.synchronize()
            trajs[form, graph] = (_loop_scenes.history_array(hist), loop.inlier_history.cpu().numpy())
    base, inl = trajs["tail", False]
    for key, (t, i) in trajs.items():
        assert _steps(base, t).max() < 2e-3, key
        assert np.max(np.abs(i - inl)) < 2.5 / 300, key
    # captured == eager within a form (the same launches)
    assert np.array_equal(trajs["records", False][0], trajs["records", True][0]) or \
        _steps(trajs["records", False][0], trajs["records", True][0]).max() < 1e-3
    auto7 = FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], **kw)
    many = _loop_scenes.build("many")
    auto80 = FusedRenderAndCompare(many["decoder"], many["camera"], many["config"], many["depth"],
                                   camera_positions=many["cam_pos"], camera_orientations=many["cam_quat"])
    assert not auto7.records_form and auto80.records_form
    with pytest.raises(ValueError):
        FusedRenderAndCompare(sc["decoder"], sc["camera"], sc["config"], sc["depth"], form="sideways", **kw)
