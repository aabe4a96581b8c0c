"""Synthetic inputs of the benchmark configurations: the blobs(seed) SDF,
an analytic sphere, and the seeded random pose list of C3/C4.  Plain numpy; shared by bench.py,
the tools, __graft_entry__.smoke() and the tests (the oracle re-exports them)."""
import numpy as np


def blobs_sdf(seed=0, R=64, K=8):
    """Union of K seeded spheres on linspace(-1,1,R)^3, indexing 'ij', float32."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-0.45, 0.45, (K, 3))
    radii = rng.uniform(0.15, 0.35, K)
    g = np.linspace(-1, 1, R)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    d = np.full((R, R, R), np.inf)
    for c, r in zip(centers, radii):
        d = np.minimum(d, np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r)
    return d.astype(np.float32)


def sphere_sdf(radius=0.5, R=64):
    g = np.linspace(-1, 1, R)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    return (np.sqrt(X * X + Y * Y + Z * Z) - radius).astype(np.float32)


def random_poses(B, seed=1, width=640, height=480, f=320.0):
    """C3/C4 pose generator: Shoemake-uniform q, z~U(1.2,2), centre pixel in the
    central 50% of the image, scale~U(0.4,0.6).  Returns pos(B,3), quat(B,4), inv_scale(B)."""
    rng = np.random.default_rng(seed)
    u1, u2, u3 = rng.uniform(size=(3, B))
    quat = np.stack([
        np.sqrt(1 - u1) * np.sin(2 * np.pi * u2),
        np.sqrt(1 - u1) * np.cos(2 * np.pi * u2),
        np.sqrt(u1) * np.sin(2 * np.pi * u3),
        np.sqrt(u1) * np.cos(2 * np.pi * u3),
    ], axis=1)
    z = rng.uniform(1.2, 2.0, B)
    cx, cy = width / 2, height / 2
    u = rng.uniform(0.25 * width, 0.75 * width, B)
    v = rng.uniform(0.25 * height, 0.75 * height, B)
    pos = np.stack([(u - cx) * z / f, -(v - cy) * z / f, -z], axis=1)
    scale = rng.uniform(0.4, 0.6, B)
    return pos.astype(np.float32), quat.astype(np.float32), (1.0 / scale).astype(np.float32)


MUG_INIT_BACKBONE = {"in_size": 3, "mlp_out_sizes": [128, 128, 128, 128, 1024], "batchnorm": True, "dense": True,
                     "residual": True}                         # estimation/configs/models/mug.yaml:98-103
MUG_INIT_HEAD = {"in_size": 1024, "mlp_out_sizes": [512, 256, 128], "batchnorm": True,
                 "orientation_repr": "discretized", "orientation_grid_resolution": 1}   # mug.yaml:106-111


def init_network_state(seed, backbone=MUG_INIT_BACKBONE, head=MUG_INIT_HEAD, shape_dimension=8, num_cells=576):
    """Seeded random weights with the reference's SDFPoseNet state-dict keys (the trained weights are not in
    the reference repository): He-scaled Linear weights, small biases, BatchNorm parameters and running
    statistics away from their defaults.  float32 numpy arrays; the same numbers in tools/make_goldens.py
    and in the tests, so the fixtures hold outputs only."""
    rng = np.random.default_rng(seed)
    state = {}

    def linear(prefix, cin, cout):
        state[prefix + ".weight"] = (rng.normal(size=(cout, cin)) * np.sqrt(2.0 / cin)).astype(np.float32)
        state[prefix + ".bias"] = rng.uniform(-0.1, 0.1, cout).astype(np.float32)

    def bn(prefix, c):
        state[prefix + ".weight"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        state[prefix + ".bias"] = rng.uniform(-0.2, 0.2, c).astype(np.float32)
        state[prefix + ".running_mean"] = rng.normal(0, 0.1, c).astype(np.float32)
        state[prefix + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    sizes = backbone["mlp_out_sizes"]
    for i, c in enumerate(sizes):
        cin = backbone["in_size"] if i == 0 else (2 if backbone.get("dense") else 1) * sizes[i - 1]
        linear(f"_backbone._linear_layers.{i}", cin, c)
        if backbone["batchnorm"]:
            bn(f"_backbone._bn_layers.{i}", c)
    hs = head["mlp_out_sizes"]
    for i, c in enumerate(hs):
        linear(f"_head._linear_layers.{i}", head["in_size"] if i == 0 else hs[i - 1], c)
        if head["batchnorm"]:
            bn(f"_head._bn_layers.{i}", c)
    n_out = shape_dimension + (8 if head.get("orientation_repr", "quaternion") == "quaternion" else 4 + num_cells)
    linear("_head._final_layer", hs[-1], n_out)
    return state


def plausible_init_network_state(seed=7, scale=0.06, position_offset=(0.004, -0.003, 0.005)):
    """``init_network_state`` with a final layer that answers like a trained network would -- latent ~ 0, position ~
    the centroid of the observed points (+ a small offset), half-width ~ `scale`, and whatever orientation cell its
    small logits favour: a usable starting point for the loop when the front door is driven end to end (tests,
    bench.py's time to result) without the trained weights, which are not in the reference repository."""
    st = {k: v.copy() for k, v in init_network_state(seed).items()}
    st["_head._final_layer.weight"] *= 0.01
    b = st["_head._final_layer.bias"]
    b *= 0.01
    b[8:11] += np.asarray(position_offset, dtype=np.float32)
    b[11] = scale
    return st
