"""Deterministic synthetic 640x480 RGB-D stream (depth + exact normals) for tests and bench.

Stands in for the reference's dataset iterator (`pytorch/dataset/production/icl_nuim.py:56-77`),
which needs cv2 and PNG files; the pinhole intrinsics are the ICL-NUIM ones the reference uses
(`icl_nuim.py:16`: fx=481.2 fy=480 cx=319.5 cy=239.5) and the depth cut follows
`pytorch/main.py:67-68` / `configs/fusion-lr-kt.yaml:20-21` (outside [0.5, 5.0] m -> NaN).

Scenes are analytic ray casts (sphere seen from inside, or an axis-aligned room with inner boxes) so
that surface normals are exact and no kd-tree normal estimation (reference `ext/pcproc`, out of scope)
is needed.  All geometry is evaluated in float64 with correctly-rounded ops only (+ - * / sqrt), then
cast to float32, so a frame generated on the CPU and on the GPU is bit-identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import torch


@dataclass
class Intrinsic:
    fx: float = 481.2
    fy: float = 480.0
    cx: float = 319.5
    cy: float = 239.5
    width: int = 640
    height: int = 480

    def scaled(self, s: float) -> "Intrinsic":
        return Intrinsic(self.fx * s, self.fy * s, self.cx * s, self.cy * s,
                         int(round(self.width * s)), int(round(self.height * s)))


@dataclass
class Box:
    lo: Tuple[float, float, float]
    hi: Tuple[float, float, float]


@dataclass
class Scene:
    kind: str = "sphere"                  # "sphere" | "room"
    radius: float = 1.5                   # sphere
    room: Box = field(default_factory=lambda: Box((-3.0, -1.5, -3.0), (3.0, 1.5, 3.0)))
    boxes: List[Box] = field(default_factory=list)


def default_room() -> Scene:
    """A 6 m x 3 m x 6 m room with a few boxes (the 'ScanNet-shape' workload of BASELINE.json configs[2])."""
    return Scene(kind="room",
                 room=Box((-3.0, -1.5, -3.0), (3.0, 1.5, 3.0)),
                 boxes=[Box((-2.2, 0.3, 1.2), (-0.9, 1.5, 2.4)),
                        Box((0.8, 0.7, 1.6), (2.1, 1.5, 2.6)),
                        Box((-0.6, 0.9, 2.0), (0.5, 1.5, 2.9)),
                        Box((1.6, -0.2, -2.6), (2.7, 1.5, -1.4)),
                        Box((-2.8, 0.1, -2.4), (-1.8, 1.5, -0.8))])


def orbit_pose(i: int, radius: float = 0.3, deg_per_frame: float = 0.5,
               phase_deg: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """Camera-to-world pose of frame i: yaw about +y by i*deg_per_frame on a circle of `radius`."""
    th = math.radians(phase_deg + i * deg_per_frame)
    c, s = math.cos(th), math.sin(th)
    R = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=np.float64)
    t = R @ np.array([0.0, 0.0, -radius], dtype=np.float64)
    return R, t


def _ray_box(o, d, lo, hi, inside: bool):
    """Slab test. Returns (s, normal) with s=+inf where missed. o:(3,), d:(...,3) float64."""
    inv = 1.0 / d
    t0 = (lo - o) * inv
    t1 = (hi - o) * inv
    tmin = torch.minimum(t0, t1)
    tmax = torch.maximum(t0, t1)
    if inside:
        s, axis = tmax.min(dim=-1)          # exit point
        hit = torch.ones_like(s, dtype=torch.bool)
        sign = torch.gather(d, -1, axis.unsqueeze(-1)).squeeze(-1).sign()   # wall we hit is in +d dir
        nrm = torch.zeros_like(d)
        nrm.scatter_(-1, axis.unsqueeze(-1), (-sign).unsqueeze(-1))         # pointing inward
    else:
        s_in, axis = tmin.max(dim=-1)
        s_out = tmax.min(dim=-1).values
        hit = (s_in < s_out) & (s_in > 0)
        s = torch.where(hit, s_in, torch.full_like(s_in, float("inf")))
        sign = torch.gather(d, -1, axis.unsqueeze(-1)).squeeze(-1).sign()
        nrm = torch.zeros_like(d)
        nrm.scatter_(-1, axis.unsqueeze(-1), (-sign).unsqueeze(-1))         # pointing outward (towards camera)
    return s, nrm, hit


def render_frame(scene: Scene, R: np.ndarray, t: np.ndarray, intr: Intrinsic,
                 device: torch.device = torch.device("cpu"),
                 depth_cut: Tuple[float, float] = (0.5, 5.0),
                 noise_seed: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ray-cast one frame.

    :return: depth (H, W) float32 metres along the camera z axis (NaN where cut / missed),
             normal (H, W, 3) float32 unit normals in the CAMERA frame, oriented towards the camera.
    """
    H, W = intr.height, intr.width
    f64 = torch.float64
    u = torch.arange(W, device=device, dtype=f64)
    v = torch.arange(H, device=device, dtype=f64)
    dx = ((u - intr.cx) / intr.fx).unsqueeze(0).expand(H, W)
    dy = ((v - intr.cy) / intr.fy).unsqueeze(1).expand(H, W)
    d_cam = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)             # (H,W,3), z == 1
    Rt = torch.tensor(R, device=device, dtype=f64)
    o = torch.tensor(t, device=device, dtype=f64)
    d = d_cam @ Rt.T                                                         # world ray, param = cam depth

    if scene.kind == "sphere":
        # |o + s d|^2 = r^2, camera inside -> positive root
        a = (d * d).sum(-1)
        b = 2.0 * (d * o).sum(-1)
        c = (o * o).sum() - scene.radius ** 2
        disc = b * b - 4.0 * a * c
        s = (-b + torch.sqrt(disc)) / (2.0 * a)
        p = o + s.unsqueeze(-1) * d
        n_world = -p / scene.radius
    elif scene.kind == "room":
        lo = torch.tensor(scene.room.lo, device=device, dtype=f64)
        hi = torch.tensor(scene.room.hi, device=device, dtype=f64)
        s, n_world, _ = _ray_box(o, d, lo, hi, inside=True)
        for bx in scene.boxes:
            lo = torch.tensor(bx.lo, device=device, dtype=f64)
            hi = torch.tensor(bx.hi, device=device, dtype=f64)
            s2, n2, hit = _ray_box(o, d, lo, hi, inside=False)
            closer = hit & (s2 < s)
            s = torch.where(closer, s2, s)
            n_world = torch.where(closer.unsqueeze(-1), n2, n_world)
    else:
        raise ValueError(scene.kind)

    if noise_seed is not None:
        g = torch.Generator(device="cpu").manual_seed(noise_seed)
        noise = torch.randn((H, W), generator=g, dtype=f64).to(device)
        s = s + 0.002 * s * s * noise
    depth = s.to(torch.float32)
    bad = ~((depth >= depth_cut[0]) & (depth <= depth_cut[1]))
    depth = torch.where(bad, torch.full_like(depth, float("nan")), depth)
    n_cam = (n_world @ Rt).to(torch.float32)                                 # R^T n
    return depth.contiguous(), n_cam.contiguous()


def unproject_reference_order(depth: torch.Tensor, intr: Intrinsic) -> torch.Tensor:
    """float32 back-projection in the op order of the reference kernel
    (`ext/imgproc/imgproc.cu:18-20`: (u - cx) / fx * d).  Plain torch; test/bench input helper only."""
    H, W = depth.shape
    u = torch.arange(W, device=depth.device, dtype=torch.float32).unsqueeze(0)
    v = torch.arange(H, device=depth.device, dtype=torch.float32).unsqueeze(1)
    x = (u - np.float32(intr.cx)) / np.float32(intr.fx) * depth
    y = (v - np.float32(intr.cy)) / np.float32(intr.fy) * depth
    return torch.stack([x, y, depth], dim=-1)


def frame_points(scene: Scene, i: int, intr: Intrinsic, device=torch.device("cpu"),
                 orbit_radius: float = 0.3, deg_per_frame: float = 0.5, noise: bool = False,
                 phase_deg: float = 0.0):
    """World-frame (xyz, normal) of frame i, the pair the reference hands to `integrate_keyframe`
    (`pytorch/main.py:83-85`).  float32, NaN pixels removed, row-major pixel order."""
    R, t = orbit_pose(i, orbit_radius, deg_per_frame, phase_deg)
    depth, n_cam = render_frame(scene, R, t, intr, device, noise_seed=(1234 + i) if noise else None)
    pc = unproject_reference_order(depth, intr).reshape(-1, 3)
    nc = n_cam.reshape(-1, 3)
    ok = ~torch.isnan(pc[:, 0])
    pc, nc = pc[ok], nc[ok]
    Rf = torch.tensor(R, dtype=torch.float64, device=device).float()
    tf = torch.tensor(t, dtype=torch.float64, device=device).float()
    xyz = transform_points(pc, Rf, tf)
    nrm = transform_points(nc, Rf, None)
    return xyz.contiguous(), nrm.contiguous()


def frame_cloud_camera(scene: Scene, i: int, intr: Intrinsic, device=torch.device("cpu"), orbit_radius: float = 0.3,
                       deg_per_frame: float = 0.5, phase_deg: float = 0.0):
    """Camera-space points of frame i (NaN pixels removed, row-major pixel order) and the frame's camera-to-world pose (R, t) float64: what
    the tracker is handed as `obs_xyz` (reference `system/tracker.py:220`) and the pose it is asked to find."""
    R, t = orbit_pose(i, orbit_radius, deg_per_frame, phase_deg)
    depth, _ = render_frame(scene, R, t, intr, device)
    pc = unproject_reference_order(depth, intr).reshape(-1, 3)
    return pc[~torch.isnan(pc[:, 0])].contiguous(), R, t


def transform_points(p: torch.Tensor, R: torch.Tensor, t: Optional[torch.Tensor]) -> torch.Tensor:
    """p @ R^T (+ t) with an explicit, unfused float32 op order ((r0*x + r1*y) + r2*z) + t — the order the
    HIP kernel `dif_unproject_transform` uses — so CPU and GPU agree bit-for-bit.  (The reference does this
    with a 3x3 GEMM, `utils/motion_util.py:322-327`, whose summation order is backend-defined.)"""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    cols = []
    for r in range(3):
        acc = (R[r, 0] * x + R[r, 1] * y) + R[r, 2] * z
        if t is not None:
            acc = acc + t[r]
        cols.append(acc)
    return torch.stack(cols, dim=-1)


@dataclass
class MapConfig:
    """`args.mapping` keys of `pytorch/configs/fusion-lr-kt.yaml:27-35`."""
    bound_min: Tuple[float, float, float]
    bound_max: Tuple[float, float, float]
    voxel_size: float
    prune_min_vox_obs: int = 16
    ignore_count_th: float = 16.0
    encoder_count_th: float = 600.0

    def namespace(self):
        import argparse
        return argparse.Namespace(bound_min=list(self.bound_min), bound_max=list(self.bound_max),
                                  voxel_size=self.voxel_size, prune_min_vox_obs=self.prune_min_vox_obs,
                                  ignore_count_th=self.ignore_count_th, encoder_count_th=self.encoder_count_th,
                                  optim_n_iters=0, code_regularization=False, code_reg_lambda=0.0)


# BASELINE.json configs
def config_c1() -> Tuple[Scene, MapConfig]:
    return Scene(kind="sphere", radius=1.5), MapConfig((-1.6, -1.6, -1.6), (1.6, 1.6, 1.6), 0.1)


def config_c2() -> Tuple[Scene, MapConfig]:
    return default_room(), MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.1)


def config_c3() -> Tuple[Scene, MapConfig]:
    return default_room(), MapConfig((-3.2, -3.2, -3.2), (3.2, 3.2, 3.2), 0.05)
