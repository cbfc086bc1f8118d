"""CPU restatement of the ORS (occupancy ray sampling) projection — TEST INFRASTRUCTURE ONLY.

SURVEY.md §8f N3: the step BEFORE the denoising loop for the ORS-3D ControlNet branch
(`use_occ_3d`): every latent pixel of every camera casts a ray through a 200 x 200 x 16 semantic
occupancy volume and records the class at `sample_point` equidistant samples; the result, scaled to
[0, 1], is the (n_cam, sample_point, h, w) condition the branch adds to its first feature map.

Restates /root/reference/MD_txt_con_fusion/magicdrive/networks/occ3d_proj.py:26-113
(`OccupancyRay.compute_rays`, `OccupancyRay.project`) and the post-processing of
magicdrive/dataset/utils.py:409-420.  PINNED: tests/golden/mint.py runs the reference's own
`project()` (its pickled camera tables and the label file replaced by seeded stand-ins, `cv2` /
`pyquaternion` as name-only stubs: the stub Quaternion hands back a rotation matrix we supply, no
quaternion arithmetic of ours is involved) and tests/test_oracle_golden.py checks this file against
it bit-for-bit (integer labels).

The arithmetic is fp32 with the reference's operation order (separate multiply and add, the same
normalisation constants, `grid_sample(mode="nearest", padding_mode="zeros", align_corners=False)`
index rule) because the output is an INDEX: a last-bit difference in a coordinate can flip a label.
"""
import torch

N_CLASSES = 18          # 17 semantic classes + 17 = "free / outside" (occ3d_proj.py:67,103-106)
GRID = (200, 200, 16)   # occupancy volume, 0.4 m voxels: x, y in [-40, 40) m, z in [-1, 5.4) m


def compute_rays(K, Rt, u, v):
    """occ3d_proj.py:26-42: origin = camera centre, direction = R K^-1 [u, v, 1]^T normalised."""
    u, v = u.to(torch.float32), v.to(torch.float32)
    K, Rt = K.to(torch.float32), Rt.to(torch.float32)
    k_inv = torch.inverse(K)
    r, t = Rt[:3, :3], Rt[:3, 3]
    pix = torch.stack([u, v, torch.ones_like(u)], dim=1)
    p_c = torch.matmul(k_inv, pix.T).T
    d = torch.matmul(r, p_c.T).T
    d = d / torch.norm(d, dim=1, keepdim=True)
    return t.expand_as(d), d


def pixel_rays(K, Rt, h, w, compress_ratio):
    """Ray origin / direction per latent pixel, (h, w, 3) each (occ3d_proj.py:79-88): latent pixel
    (y, x) looks through image pixel (x // compress_ratio, y // compress_ratio)."""
    xs, ys = torch.arange(0, w), torch.arange(0, h)
    xx, yy = torch.meshgrid(xs, ys, indexing="ij")
    gx = xx.flatten() // compress_ratio
    gy = yy.flatten() // compress_ratio
    o, d = compute_rays(K, Rt, gx, gy)
    return (o.view(w, h, 3).permute(1, 0, 2).contiguous(), d.view(w, h, 3).permute(1, 0, 2).contiguous())


def sample_labels(occ, origin, direction, sample_point, sample_step):
    """Labels along the rays, (h, w, sample_point) int64 (occ3d_proj.py:91-108).

    occ: (200, 200, 16) integer class volume.  A sample at metric point p = o + s d is looked up at
    the NEAREST voxel of normalised coordinates (p_x / 40, p_y / 40, p_z / 3.2 - 2.2 / 3.2) in
    [-1, 1]^3 (grid_sample convention, align_corners=False); outside the volume -> class 17."""
    occ = occ.to(torch.int64)
    steps = torch.arange(sample_point).float() * sample_step
    pts = origin.unsqueeze(2) + steps.view(1, 1, -1, 1) * direction.unsqueeze(2)         # (h, w, S, 3) metres
    g = pts / 40
    gz = g[..., 2] * 40 / 3.2 - 2.2 / 3.2
    gx, gy = g[..., 0], g[..., 1]

    def index(coord, size):          # grid_sample: ((c + 1) * size - 1) / 2, round half to even
        return torch.round(((coord + 1) * size - 1) / 2).to(torch.int64)

    ix, iy, iz = index(gx, GRID[0]), index(gy, GRID[1]), index(gz, GRID[2])
    inside = (ix >= 0) & (ix < GRID[0]) & (iy >= 0) & (iy < GRID[1]) & (iz >= 0) & (iz < GRID[2])
    lab = occ[ix.clamp(0, GRID[0] - 1), iy.clamp(0, GRID[1] - 1), iz.clamp(0, GRID[2] - 1)]
    return torch.where(inside, lab, torch.full_like(lab, N_CLASSES - 1))


def ors_project(occ, intrinsics, extrinsics, h, w, compress_ratio, sample_point=320, sample_step=0.2):
    """`OccupancyRay.project` for given volume and cameras: (n_cam, h, w, sample_point) int64."""
    out = []
    for K, Rt in zip(intrinsics, extrinsics):
        o, d = pixel_rays(K, Rt, h, w, compress_ratio)
        out.append(sample_labels(occ, o, d, sample_point, sample_step))
    return torch.stack(out, dim=0)


def ors_condition(labels, use_fg=True, use_bg=True):
    """dataset/utils.py:412-420: optional foreground (<= 10) / background (>= 11) filtering to class
    17, channels-first, scaled to [0, 1]: (n_cam, h, w, S) int -> (n_cam, S, h, w) float32."""
    t = labels
    if not use_fg:
        t = torch.where(t <= 10, torch.full_like(t, 17), t)
    if not use_bg:
        t = torch.where(11 <= t, torch.full_like(t, 17), t)
    t = t.permute(0, 3, 1, 2).float()
    return t / 17
