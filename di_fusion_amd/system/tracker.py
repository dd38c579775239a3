"""The tracker's side of the map: the SDF term of `SDFTracker.gauss_newton` on libdifusion.so.

Reference: `pytorch/system/tracker.py`.  Its `compute_sdf_Hg` (tracker.py:174-218) is what calls `map.get_sdf` once per Gauss-Newton
iteration — tens of times per frame, where the map is integrated once in twenty frames — and spreads the rest of the iteration over ~25
torch launches, the autograd engine and three device -> host round trips.  Here the whole term is ONE C call of two launches (`dif_sdf_hg`:
the decoder kernel over all points of the posed cloud — pose, validity test, latent look-up, values and analytic input gradient —, then
Jacobian, robust weights and the 6x6 / 6 / 1 sums in double in a fixed order) and the 44 numbers come back through pinned host memory,
without a copy and without a stream synchronisation.

What is here: `Pose` (the part of `utils.motion_util.Isometry` the loop needs, on plain rotation matrices: pyquaternion is not a
dependency), `SDFTracker` with `compute_sdf_Hg`, `gauss_newton`, `track_camera` (the point-cloud preparation of tracker.py:87-118 on the HIP
operators of `system.ext`).  What is not: the photometric term (`compute_rgb_Hg` needs the reference's `rgb_odometry` / `gradient_xy` image
kernels, outside SURVEY.md section 8): `iter_config` entries naming 'rgb' raise NotImplementedError unless a subclass supplies
`compute_rgb_Hg`.  A reference `SDFTracker` can use the fused term as it is: `compute_sdf_Hg` takes the reference's `Isometry` objects too
(anything with `.q.rotation_matrix` and `.t`), see INTEGRATION.md.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace

import numpy as np
import torch

from .. import _lib
from . import ext

ROBUST_KERNELS = {None: 0, "none": 0, "huber": 1, "tukey": 2}


def _wedge(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


class Pose:
    """Rigid transform x -> R x + t (float64).  `q.rotation_matrix`, `t`, `dot`, `inv`, `@` and `from_twist` behave like the reference's
    `Isometry` (utils/motion_util.py:162-333) for the calls the tracker makes."""

    def __init__(self, R=None, t=None):
        self.R = np.eye(3) if R is None else np.asarray(R, dtype=np.float64).reshape(3, 3)
        self.t = np.zeros(3) if t is None else np.asarray(t, dtype=np.float64).reshape(3)

    # -- what the reference reads off an Isometry ------------------------------------------------------------
    @property
    def q(self):
        return SimpleNamespace(rotation_matrix=self.R)

    @property
    def matrix(self):
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = self.R, self.t
        return m

    @staticmethod
    def of(pose) -> "Pose":
        """A Pose from a Pose, a 4x4 matrix, or any object with `.q.rotation_matrix` and `.t` (the reference's Isometry)."""
        if isinstance(pose, Pose):
            return pose
        if isinstance(pose, np.ndarray) and pose.shape == (4, 4):
            return Pose(pose[:3, :3], pose[:3, 3])
        return Pose(np.asarray(pose.q.rotation_matrix), np.asarray(pose.t))

    def dot(self, right) -> "Pose":
        right = Pose.of(right)
        return Pose(self.R @ right.R, self.R @ right.t + self.t)

    def inv(self) -> "Pose":
        return Pose(self.R.T, -(self.R.T @ self.t))

    def __matmul__(self, other):
        if isinstance(other, torch.Tensor):                        # (N,3) float32, like motion_util.py:324-327
            R = torch.from_numpy(self.R).to(other.device).float()
            t = torch.from_numpy(self.t).to(other.device).float()
            return other @ R.t() + t.unsqueeze(0)
        if isinstance(other, np.ndarray):
            return other @ self.R.T + self.t
        return self.dot(other)

    @staticmethod
    def from_twist(xi) -> "Pose":
        """exp of a twist (rho, phi): R = exp(phi^), t = V(phi) rho with V the left Jacobian of SO(3) (motion_util.py:204-228, 45-57;
        first-order forms below |phi| ~ 1e-8 like the reference)."""
        xi = np.asarray(xi, dtype=np.float64)
        rho, phi = xi[:3], xi[3:6]
        th = float(np.linalg.norm(phi))
        if np.isclose(th, 0.0):
            R = np.eye(3) + _wedge(phi)
            # (the reference turns this into a unit quaternion: the nearest rotation)
            u, _, vt = np.linalg.svd(R)
            return Pose(u @ vt, (np.eye(3) + 0.5 * _wedge(phi)) @ rho)
        a = phi / th
        s, c = np.sin(th), np.cos(th)
        aa, ax = np.outer(a, a), _wedge(a)
        R = c * np.eye(3) + (1.0 - c) * aa + s * ax
        V = (s / th) * np.eye(3) + (1.0 - s / th) * aa + ((1.0 - c) / th) * ax
        return Pose(R, V @ rho)

    def __repr__(self):
        return f"Pose(t={self.t}, R={self.R.tolist()})"


def _rows34(p: Pose):
    return np.concatenate([p.R.astype(np.float32), p.t.astype(np.float32)[:, None]], axis=1).reshape(12)


class _HgState:
    """Per-map scratch of `sdf_hg`: workspace, the device result, four pinned result slots (a slot is 44 doubles + the sequence word)."""

    def __init__(self, dev):
        self.dev = dev
        self.ws = None
        self.n = -1
        self.out = torch.zeros((44,), dtype=torch.float64, device=dev)
        self.slots = [torch.zeros((45,), dtype=torch.float64).pin_memory() for _ in range(4)]
        self.slots_np = [s.numpy() for s in self.slots]
        self.words = [s.numpy().view(np.int64) for s in self.slots]
        self.seq = 0


def sdf_hg(map_, obs_xyz: torch.Tensor, last_pose, cur_delta_pose, robust_kernel=None, robust_k: float = 0.0, no_grad: bool = False):
    """The SDF term for one pose (reference tracker.py:174-218): (H (6,6) float64, g (6,) float64, sum_error float, M int), or
    (None, None, sum_error, M) with `no_grad`.  `obs_xyz` (N,3) float32 on the map's device, camera space."""
    if robust_kernel not in ROBUST_KERNELS:
        raise NotImplementedError(robust_kernel)
    obs = obs_xyz.detach()
    if obs.dtype != torch.float32 or not obs.is_contiguous():
        obs = obs.float().contiguous()
    _lib.require_cuda(obs)
    if obs.dim() != 2 or obs.size(1) != 3:
        raise RuntimeError("obs_xyz must be (N,3)")
    N = int(obs.size(0))
    last, delta = Pose.of(last_pose), Pose.of(cur_delta_pose)
    a = _lib.DifSdfHg()
    a.T_cur[:] = _rows34(last.dot(delta)).tolist()
    a.T_delta[:] = _rows34(delta).tolist()
    a.last_Rt[:] = last.R.astype(np.float32).T.reshape(9).tolist()
    a.robust_kernel = ROBUST_KERNELS[robust_kernel]
    a.robust_k = float(robust_k)
    a.no_grad = 1 if no_grad else 0
    lib = _lib.load()
    st = getattr(map_, "_hg_state", None)
    if st is None or st.dev != map_.device:
        st = map_._hg_state = _HgState(map_.device)
    from .map import _on_device
    with _on_device(map_.device):
        if st.n < N:
            need = int(lib.dif_sdf_hg_workspace_bytes(N))
            st.ws = torch.zeros((need + 256,), dtype=torch.uint8, device=map_.device)
            st.n = N
        off = (-st.ws.data_ptr()) % 256
        st.seq += 1
        k = st.seq & 3
        w = map_.model.packed.weights_struct(map_.device)
        _lib.check(lib.dif_sdf_hg(ctypes.byref(map_._cmap), ctypes.byref(w), _lib.ptr(obs), N, ctypes.byref(a),
                                  ctypes.c_void_p(st.ws.data_ptr() + off), st.ws.numel() - off, _lib.ptr(st.out), _lib.ptr(st.slots[k]), st.seq,
                                  _lib.stream_ptr()), "dif_sdf_hg")
        _lib.spin_until(st.words[k], 44, st.seq, "the tracker's SDF term")
    r = st.slots_np[k]
    M = int(r[43])
    if no_grad:
        return None, None, float(r[42]), M
    return r[:36].reshape(6, 6).copy(), r[36:42].copy(), float(r[42]), M


class SDFTracker:
    """reference `system/tracker.py:26-283`, the SDF term on the fused path.  `args`: the reference's `tracking` block (a namespace or dict
    with `sdf`, `rgb` (optional) and `iter_config`, configs/fusion-lr-kt.yaml:38-56)."""

    def __init__(self, map, args):
        self.map = map
        self.args = args if not isinstance(args, dict) else SimpleNamespace(**args)
        as_ns = lambda d: d if not isinstance(d, dict) else SimpleNamespace(**d)   # noqa: E731
        self.sdf_args = as_ns(self.args.sdf)
        self.rgb_args = as_ns(getattr(self.args, "rgb", None) or dict(weight=0.0, robust_kernel=None, robust_k=0.0))
        self.last_intensity = None
        self.last_depth = None
        self.all_pd_pose = []
        self.last_processed_pc = None        # [xyz, normal] of the last frame: what integrate_keyframe takes (tracker.py:117)
        self.last_colored_pcd = None
        self.cur_gt_pose = None
        self.n_unstable = 0

    # -- terms -----------------------------------------------------------------------------------------------
    def compute_sdf_Hg(self, n_iter: int, last_pose, cur_delta_pose, obs_xyz: torch.Tensor, no_grad: bool = False):
        """tracker.py:174-218: (H, g, energy), H / g None with `no_grad`.  An empty valid set divides by zero there; here too."""
        H, g, e, M = sdf_hg(self.map, obs_xyz, last_pose, cur_delta_pose, self.sdf_args.robust_kernel, self.sdf_args.robust_k, no_grad)
        if M == 0:
            raise ZeroDivisionError("no observation falls into a tracked voxel (tracker.py:209)")
        return H, g, e

    def compute_rgb_Hg(self, pyramid_level, cur_delta_pose, cur_intensity_pyramid, cur_depth_pyramid, cur_dIdxy_pyramid, calib, no_grad=False):
        raise NotImplementedError("the photometric term needs the reference's rgb_odometry / gradient_xy kernels (tracker.py:131-172), which are "
                                  "outside the fusion path; use an iter_config of 'sdf' terms or override compute_rgb_Hg")

    def _uses_rgb(self):
        return any(t[0] == "rgb" for grp in self.args.iter_config for t in grp["type"])

    # -- the loop (tracker.py:220-283) -----------------------------------------------------------------------
    def gauss_newton(self, init_pose, cur_intensity_pyramid, cur_depth_pyramid, cur_dIdxy_pyramid, obs_xyz: torch.Tensor, calib):
        last_pose = Pose.of(self.all_pd_pose[-1])
        delta = last_pose.inv().dot(Pose.of(init_pose))
        accepted = delta
        it = 0
        for group in self.args.iter_config:
            best = np.inf
            for it in list(range(group["n"])) + [-1]:        # -1: one last evaluation without derivatives
                final = it == -1
                H, g, energy = np.zeros((6, 6)), np.zeros(6), 0.0
                for term in group["type"]:
                    if term[0] == "sdf":
                        tH, tg, te = self.compute_sdf_Hg(it, last_pose, delta, obs_xyz, final)
                    elif term[0] == "rgb":
                        tH, tg, te = self.compute_rgb_Hg(term[1], delta, cur_intensity_pyramid, cur_depth_pyramid, cur_dIdxy_pyramid, calib, final)
                    else:
                        raise NotImplementedError(term[0])
                    energy += te
                    if not final:
                        H += tH
                        g += tg
                if energy > best:                            # the step made it worse: back to the last accepted pose, next group
                    delta = accepted
                    break
                accepted, best = delta, energy
                if not final:
                    delta = Pose.from_twist(np.linalg.solve(H, -g)).dot(delta)
        if it >= 10:
            self.n_unstable += 1
            if self.n_unstable >= 3:
                self.rgb_args.weight = max(self.rgb_args.weight, 500.0)
        return last_pose.dot(accepted)

    # -- per frame (tracker.py:74-129) -----------------------------------------------------------------------
    def track_camera(self, rgb_data, depth_data: torch.Tensor, calib, set_pose=None):
        """Point-cloud preparation of the frame (half-resolution unprojection, radius-outlier removal, PCA normals, 2 cm box filter: the
        cloud `integrate_keyframe` takes, left in `last_processed_pc`), then the pose: `set_pose`, or Gauss-Newton from the previous one."""
        if self._uses_rgb():
            self.compute_rgb_Hg(None, None, None, None, None, calib)          # raises unless a subclass supplies the term and its pyramids
        sc = float(self.sdf_args.subsample)
        d = torch.nn.functional.interpolate(depth_data[None, None], scale_factor=sc, mode="nearest", recompute_scale_factor=False)[0, 0].contiguous()
        pc = ext.unproject_depth(d, calib.fx * sc, calib.fy * sc, calib.cx * sc, calib.cy * sc).reshape(-1, 3)
        pc = pc[~torch.isnan(pc[:, 0])].contiguous()
        with torch.cuda.device(self.map.device):
            pc = pc[ext.remove_radius_outlier(pc, 16, 0.05)].contiguous()
            nrm = ext.estimate_normals(pc, 16, 0.1, [0.0, 0.0, 0.0])
            ok = ~torch.isnan(nrm[:, 0])
            pc, nrm = pc[ok].contiguous(), nrm[ok].contiguous()
            pc, nrm = ext.point_box_filter(pc, nrm, 0.02)
        self.last_processed_pc = [pc, nrm]
        if set_pose is not None:
            pose = Pose.of(set_pose)
        else:
            assert len(self.all_pd_pose) > 0
            pose = self.gauss_newton(self.all_pd_pose[-1], None, None, None, pc, calib)
        self.all_pd_pose.append(pose)
        return pose
