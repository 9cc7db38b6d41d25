"""Seeded synthetic inputs: nuScenes-shaped multi-sweep clouds and random-init weights.

There is no dataset or checkpoint in the image, so bench / tests use
  * ``synthetic_cloud(seed, target_points)``: rows ``[x, y, z, intensity, dt]`` float32 in the layout
    LoadPointCloudFromFile produces (det3d/datasets/pipelines/loading.py:128-140; points closer
    than 1 m removed like ``remove_close`` :36-51): 10 sweeps x 32 beams (-30..+10 deg) x A azimuths,
    ground plane at z = -1.84 m, per-azimuth-sector wall distance U(4, 70) m, 5 % drop-out,
    per-sweep ego shift;
  * ``seeded_state_dict(module, seed)``: fills every parameter / BN statistic from a numpy RNG keyed
    by the tensor's state_dict key, so the same weights can be regenerated on any box for any
    implementation that keeps the reference's key names.
"""
import zlib

import numpy as np
import torch


def synthetic_cloud(seed=0, target_points=300000, n_sweeps=10, n_beams=32, profile="dense"):
    """profile "dense" (default; the bench workload): every sweep sees its own random walls and azimuth phase and the ego shifts by
    ~0.6 m per sweep, so almost every point opens its own 7.5 cm voxel -- a 300k-point cloud fills the 160k-voxel cap (a stress case).
    profile "street": ONE static scene (half of the azimuth sectors open to beyond the range), motion-compensated sweeps (8 mm
    residual shift per sweep, nearly the same azimuth phase): returns of the ten sweeps fall into the same voxels, ~55k voxels with
    ~4 points each for 300k points -- the sparsity of a real 10-sweep nuScenes frame."""
    if profile == "street":
        return _street_cloud(seed, target_points, n_sweeps, n_beams)
    assert profile == "dense", profile
    rng = np.random.default_rng(seed)
    n_az = max(8, int(round(target_points / 0.95 / 0.94 / (n_sweeps * n_beams))))
    elev = np.deg2rad(np.linspace(-30.0, 10.0, n_beams)).astype(np.float64)
    n_sectors = 360
    sweeps = []
    for s in range(n_sweeps):
        wall = rng.uniform(4.0, 70.0, n_sectors)
        az = (np.arange(n_az) + rng.uniform(0, 1)) * (2 * np.pi / n_az)
        A, E = np.meshgrid(az, elev, indexing="ij")
        sector = (A / (2 * np.pi) * n_sectors).astype(np.int64) % n_sectors
        r_wall = wall[sector] / np.maximum(np.cos(E), 1e-3)
        sensor_h = 1.84
        with np.errstate(divide="ignore"):
            r_ground = np.where(E < -1e-3, sensor_h / np.maximum(-np.sin(E), 1e-6), np.inf)
        r = np.minimum(r_wall, r_ground) + rng.normal(0, 0.02, A.shape)
        x = r * np.cos(E) * np.cos(A)
        y = r * np.cos(E) * np.sin(A)
        z = r * np.sin(E)
        shift = rng.normal(0, 0.6, 2) * s  # ego motion between sweeps
        x = x + shift[0]
        y = y + shift[1]
        keep = rng.random(A.shape) > 0.05
        keep &= (np.abs(x) >= 1.0) | (np.abs(y) >= 1.0)
        inten = rng.uniform(0, 255, A.shape)
        dt = np.full(A.shape, 0.05 * s)
        pts = np.stack([x, y, z, inten, dt], axis=-1)[keep]
        sweeps.append(pts.astype(np.float32))
    return np.ascontiguousarray(np.concatenate(sweeps, axis=0))


def synthetic_sweeps(seed=0, target_points=300000, n_sweeps=10, profile="dense"):
    """The same scene as ``synthetic_cloud`` one step earlier in the reference's pipeline: what LoadPointCloudFromFile reads
    (det3d/datasets/pipelines/loading.py:100-141) -- the key frame and ``n_sweeps - 1`` sweeps as RAW sensor-frame rows [x, y, z,
    intensity, ring] with, per sweep, the 4x4 ``transform_matrix`` into the key frame (a small yaw + the ego shift, float64) and its
    ``time_lag``.  Returns (raw [R,5] float32, rows int64 [n_sweeps+1], transforms (None for the key frame), time_lags, remove_close flags):
    the arguments of hip_ops.sweep_descriptors.  Sweep s of the cloud is recognised by its time column (0.05 s)."""
    cloud = synthetic_cloud(seed, target_points, n_sweeps=n_sweeps, profile=profile)
    rng = np.random.default_rng([seed, 4242])
    sweep_of = np.rint(cloud[:, 4] / 0.05).astype(np.int64)
    chunks, transforms, lags, close = [], [], [], []
    for s in range(n_sweeps):
        pts = cloud[sweep_of == s].astype(np.float64)
        if s == 0:
            transforms.append(None)
        else:
            yaw, t = rng.normal(0, 0.01) * s, np.array([rng.normal(0, 0.3) * s, rng.normal(0, 0.3) * s, rng.normal(0, 0.01)])
            M = np.eye(4)
            M[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
            M[:3, 3] = t
            pts[:, :3] = (pts[:, :3] - t) @ M[:3, :3]  # sensor-frame rows: M^-1 applied (row vectors: (R^T (p - t))^T = (p - t)^T R)
            transforms.append(M)
        raw = np.concatenate([pts[:, :4], rng.integers(0, 32, (len(pts), 1)).astype(np.float64)], 1).astype(np.float32)
        chunks.append(raw)
        lags.append(0.05 * s)
        close.append(s > 0)
    rows = np.cumsum([0] + [len(c) for c in chunks]).astype(np.int64)
    return np.ascontiguousarray(np.concatenate(chunks, 0)), rows, transforms, lags, close


def _street_cloud(seed, target_points, n_sweeps, n_beams):
    rng = np.random.default_rng([seed, 77])
    n_az = max(8, int(round(target_points / 0.95 / 0.94 / (n_sweeps * n_beams))))
    elev = np.deg2rad(np.linspace(-30.0, 10.0, n_beams)).astype(np.float64)
    n_sectors = 360
    wall = rng.uniform(5.0, 60.0, n_sectors)
    wall = np.where(rng.random(n_sectors) < 0.5, 400.0, wall)  # open sectors: returns beyond the detection range
    phase = rng.uniform(0, 1)
    sweeps = []
    for s in range(n_sweeps):
        az = (np.arange(n_az) + phase + rng.uniform(-0.1, 0.1)) * (2 * np.pi / n_az)
        A, E = np.meshgrid(az, elev, indexing="ij")
        sector = (A / (2 * np.pi) * n_sectors).astype(np.int64) % n_sectors
        r_wall = wall[sector] / np.maximum(np.cos(E), 1e-3)
        with np.errstate(divide="ignore"):
            r_ground = np.where(E < -1e-3, 1.84 / np.maximum(-np.sin(E), 1e-6), np.inf)
        r = np.minimum(r_wall, r_ground) + rng.normal(0, 0.01, A.shape)
        shift = rng.normal(0, 0.008, 2) * s
        x = r * np.cos(E) * np.cos(A) + shift[0]
        y = r * np.cos(E) * np.sin(A) + shift[1]
        z = r * np.sin(E)
        keep = rng.random(A.shape) > 0.05
        keep &= (np.abs(x) >= 1.0) | (np.abs(y) >= 1.0)
        pts = np.stack([x, y, z, rng.uniform(0, 255, A.shape), np.full(A.shape, 0.05 * s)], axis=-1)[keep]
        sweeps.append(pts.astype(np.float32))
    return np.ascontiguousarray(np.concatenate(sweeps, axis=0))


def tame_scores(sd, scale=0.25, bias=-6.0):
    """Scales the last convolution of every heat-map head and lowers its bias: with plain random weights half of the 32400 cells
    pass the score threshold (logit mean -2.2 = the threshold, sigma 1.5 .. 5) and every one of the 7 x 83 output slots is taken
    in every comparison.  Scaled, a few hundred cells pass and a few dozen boxes survive NMS (a scene with few objects): the
    top-1000 cut and the post-max cut stay inactive.  Returns ``sd``."""
    for k in list(sd.keys()):
        if k.endswith(".hm.3.weight"):
            sd[k] = sd[k] * scale
        elif k.endswith(".hm.3.bias"):
            sd[k] = torch.full_like(sd[k], bias)
    return sd


def _rng_for(key, seed):
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def seeded_state_dict(module, seed=0):
    """Returns {key: tensor} for every entry of ``module.state_dict()`` (except num_batches_tracked)."""
    out = {}
    sd = module.state_dict()
    for key, t in sd.items():
        if key.endswith("num_batches_tracked"):
            continue
        rng = _rng_for(key, seed)
        shape = tuple(t.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            v = rng.normal(0, 0.1, shape)
        elif leaf == "weight" and len(shape) == 1:  # norm scale
            v = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("hm.3.bias"):  # heat-map prior, init_bias=-2.19 (center_head.py:145-146)
            v = np.full(shape, -2.19)
        elif leaf == "bias":
            v = rng.normal(0, 0.1, shape)
        elif leaf == "weight" and len(shape) == 5:  # spconv layout (kD,kH,kW,Cin,Cout)
            fan_in = shape[0] * shape[1] * shape[2] * shape[3]
            v = rng.normal(0, np.sqrt(2.0 / fan_in), shape)
        elif leaf == "weight" and len(shape) == 4:  # Conv2d (Cout,Cin,kh,kw) / ConvT (Cin,Cout,kh,kw)
            fan_in = shape[1] * shape[2] * shape[3]
            v = rng.normal(0, np.sqrt(2.0 / fan_in), shape)
        else:
            v = rng.normal(0, 0.1, shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).to(t.dtype)
    return out


def load_seeded(module, seed=0):
    sd = seeded_state_dict(module, seed)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    missing = [m for m in missing if not m.endswith("num_batches_tracked")]
    assert not missing and not unexpected, (missing, unexpected)
    return module


def tame_box_dims(sd, factor=0.01):
    """Scales the last convolution of every ``dim`` head in a seeded state dict.  With plain random weights the size logits
    reach +-150, i.e. boxes of exp(150) metres: the reference's float32 rotated-IoU arithmetic has no correct digits left for
    such boxes (its result changes by O(1) with the last ulp of sin / cos), so NMS decisions between them are noise in ANY
    implementation.  Scaled, the synthetic network emits boxes of 0.2 .. 5 m like a trained one.  Returns ``sd``."""
    for k in list(sd.keys()):
        if k.endswith(".dim.3.weight") or k.endswith(".dim.3.bias"):
            sd[k] = sd[k] * factor
    return sd
