"""ORS projection: the occupancy-ray-sampled condition of the ORS-3D ControlNet branch.

Mirrors `OccupancyRay` of magicdrive/networks/occ3d_proj.py:10-113 (constructor arguments, `compute_rays`,
`project(sample_token)` -> (6, h, w, sample_point) int64) and the collate step of
magicdrive/dataset/utils.py:409-420 (`condition`).  The reference one-hot-encodes the 200 x 200 x 16
volume into 18 fp32 channels and runs a 5-D `grid_sample` + argmax per camera on the CPU data-loader
(≈ 2.7 M samples x 18 channels per scene); here the per-sample work is one gather kernel
(`dd_ors_project`, include/dualdiff_hip.h) that reads the uint8 volume directly and writes either the
integer labels or the normalised channels-first condition the branch consumes.

What stays on the host: the 6 x h x w ray table (K^-1, R, normalisation — 8400 rays), computed with the
same torch fp32 calls as the reference so that the sampled coordinates are the reference's bit for bit,
and cached per camera rig (nuScenes calibrations repeat across the samples of a log).

Data access is injected rather than hard-wired: `camera_data[token][cam]` holds 'intrinsic' plus either
'rotation_matrix' or 'rotation' (w, x, y, z quaternion) and 'translation'; `occ_loader(token)` returns
the 200 x 200 x 16 class volume.  `from_reference_files` reads the reference's own pickles / labels.npz
layout (occ3d_proj.py:12-16,50-52).
"""
import os
import pickle

import numpy as np
import torch

from .. import ops as O

CAMERAS = ("CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_RIGHT", "CAM_BACK", "CAM_BACK_LEFT")


def quaternion_rotation_matrix(q):
    """Unit-quaternion (w, x, y, z) -> 3 x 3 rotation, float64 (what pyquaternion's `rotation_matrix`
    returns after normalising; pyquaternion is not in this image, so this one function has no pinned
    reference — the parity tests feed rotation matrices)."""
    q = np.asarray(q, dtype=np.float64)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class OccupancyRay:
    def __init__(self, image_shape=(900, 1600), sample_point=200, sample_step=0.2, compress_ratio=8,
                 dataroot="./data/nuscenes/", device="cuda", camera_data=None, occ_loader=None):
        self.device = torch.device(device)
        self.dataroot = dataroot
        self.image_shape = image_shape
        self.sample_point = sample_point
        self.sample_step = sample_step
        self.compress_ratio = compress_ratio
        self.image_shape_compress = [int(image_shape[0] * compress_ratio), int(image_shape[1] * compress_ratio)]
        self.camera_data = camera_data if camera_data is not None else {}
        self.occ_loader = occ_loader
        self._ray_cache = {}

    @classmethod
    def from_reference_files(cls, pkl_root="magicdrive/networks", **kw):
        with open(os.path.join(pkl_root, "camera.pkl"), "rb") as f:
            cams = pickle.load(f)
        with open(os.path.join(pkl_root, "occ3d_idx.pkl"), "rb") as f:
            idx = pickle.load(f)
        self = cls(camera_data=cams, **kw)
        self.occ_loader = lambda tok: np.load(os.path.join(self.dataroot, idx[tok], "labels.npz"))["semantics"]
        return self

    # ---- rays (host, fp32, the reference's own call sequence: occ3d_proj.py:26-42) ----
    @staticmethod
    def compute_rays(K, Rt, u_array, v_array):
        u, v = u_array.to(torch.float32), v_array.to(torch.float32)
        K, Rt = K.to(torch.float32), Rt.to(torch.float32)
        if len(u) != len(v):
            raise AssertionError("u_array and v_array must have the same length")
        pix = torch.stack([u, v, torch.ones_like(u)], dim=1)
        d = torch.matmul(Rt[:3, :3], torch.matmul(torch.inverse(K), pix.T)).T
        d = d / torch.norm(d, dim=1, keepdim=True)
        return Rt[:3, 3].expand_as(d), d

    def camera_matrices(self, cam):
        """One camera record -> (K (3, 3) fp32, Rt (4, 4) fp32), occ3d_proj.py:69-78."""
        rot = cam["rotation_matrix"] if "rotation_matrix" in cam else quaternion_rotation_matrix(cam["rotation"])
        m = np.eye(4)
        m[:3, :3] = np.asarray(rot, dtype=np.float64)
        m[:3, 3] = np.asarray(cam["translation"], dtype=np.float64)
        return torch.tensor(np.asarray(cam["intrinsic"]), dtype=torch.float32), torch.from_numpy(m).to(torch.float32)

    def rays(self, intrinsics, extrinsics):
        """(origin (n, 3), direction (n, h*w, 3)) fp32 on self.device for a rig, cached by value."""
        key = (tuple(np.asarray(k, dtype=np.float32).tobytes() for k in intrinsics),
               tuple(np.asarray(e, dtype=np.float32).tobytes() for e in extrinsics))
        hit = self._ray_cache.get(key)
        if hit is not None:
            return hit
        h, w = self.image_shape_compress
        xx, yy = torch.meshgrid(torch.arange(0, w), torch.arange(0, h), indexing="ij")
        gx, gy = xx.flatten() // self.compress_ratio, yy.flatten() // self.compress_ratio
        origins, dirs = [], []
        for K, Rt in zip(intrinsics, extrinsics):
            K, Rt = torch.as_tensor(K), torch.as_tensor(Rt)
            o, d = self.compute_rays(K, Rt, gx, gy)
            origins.append(o[0])
            dirs.append(d.view(w, h, 3).permute(1, 0, 2).reshape(h * w, 3))      # pixel-major (y, x)
        out = (torch.stack(origins).contiguous().to(self.device), torch.stack(dirs).contiguous().to(self.device))
        if len(self._ray_cache) > 64:
            self._ray_cache.clear()
        self._ray_cache[key] = out
        return out

    # ---- sampling (GPU) ----
    def _volume(self, occ):
        occ = torch.as_tensor(np.asarray(occ) if not torch.is_tensor(occ) else occ)
        if tuple(occ.shape) != (200, 200, 16):
            raise ValueError("occupancy volume must be 200 x 200 x 16, got %s" % (tuple(occ.shape),))
        return occ.to(torch.uint8).contiguous().to(self.device, non_blocking=True)

    def project_volume(self, occ, intrinsics, extrinsics):
        """Labels (n_cam, h, w, sample_point) int64 for one volume and rig (`project` without file IO)."""
        h, w = self.image_shape_compress
        origin, direction = self.rays(intrinsics, extrinsics)
        labels, _ = O.ors_project(self._volume(occ), origin, direction, self.sample_point, self.sample_step)
        return labels.view(len(intrinsics), h, w, self.sample_point).to(torch.int64)

    def condition_volume(self, occ, intrinsics, extrinsics, dtype=torch.bfloat16, use_fg=True, use_bg=True):
        """(n_cam, sample_point, h, w) `dtype`: labels -> optional fg / bg filtering -> / 17, fused into the
        sampling kernel (dataset/utils.py:412-420; no label tensor is materialised)."""
        h, w = self.image_shape_compress
        origin, direction = self.rays(intrinsics, extrinsics)
        _, cond = O.ors_project(self._volume(occ), origin, direction, self.sample_point, self.sample_step,
                                want_labels=False, cond_dtype=dtype, keep_fg=use_fg, keep_bg=use_bg)
        return cond.view(len(intrinsics), self.sample_point, h, w)

    def _rig(self, sample_token):
        mats = [self.camera_matrices(self.camera_data[sample_token][c]) for c in CAMERAS]
        return [m[0] for m in mats], [m[1] for m in mats]

    def project(self, sample_token):
        if self.occ_loader is None:
            raise RuntimeError("OccupancyRay.project needs an occ_loader (token -> 200 x 200 x 16 class volume)")
        return self.project_volume(self.occ_loader(sample_token), *self._rig(sample_token))

    def condition(self, sample_tokens, dtype=torch.bfloat16, use_occ_3d_fg=True, use_occ_3d_bg=True):
        """The collate step for a batch of tokens: (bs * n_cam, sample_point, h, w)."""
        if self.occ_loader is None:
            raise RuntimeError("OccupancyRay.condition needs an occ_loader")
        per = [self.condition_volume(self.occ_loader(t), *self._rig(t), dtype=dtype, use_fg=use_occ_3d_fg,
                                     use_bg=use_occ_3d_bg) for t in sample_tokens]
        return torch.cat(per, dim=0)
