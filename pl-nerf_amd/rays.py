"""Ray helpers with the reference's signatures (run_nerf_helpers.py:162-201).

Row a12 of SURVEY.md section 8: negligible work (one pass over H*W pixels), kept as torch
ops on whatever device `c2w` lives on -- except the NDC warp of rays already on the GPU, which is one launch
(plnerf_ndc_rays) instead of the expression's twenty.
"""
import math

import numpy as np
import torch


def get_rays(H, W, K, c2w):
    """Pinhole rays for every pixel: directions ((i-cx)/fx, -(j-cy)/fy, -1) rotated by c2w,
    origin = camera centre; no half-pixel offset (run_nerf_helpers.py:162-171)."""
    dev = c2w.device if isinstance(c2w, torch.Tensor) else None
    c2w = torch.as_tensor(c2w, dtype=torch.float32, device=dev)
    col = torch.arange(W, dtype=torch.float32, device=dev)[None, :].expand(H, W)
    row = torch.arange(H, dtype=torch.float32, device=dev)[:, None].expand(H, W)
    cam = torch.stack([(col - K[0][2]) / K[0][0], -(row - K[1][2]) / K[1][1], -torch.ones_like(col)], -1)
    rays_d = torch.sum(cam[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def get_rays_np(H, W, K, c2w):
    """NumPy twin of get_rays (run_nerf_helpers.py:174-181)."""
    col, row = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    cam = np.stack([(col - K[0][2]) / K[0][0], -(row - K[1][2]) / K[1][1], -np.ones_like(col)], -1)
    rays_d = np.sum(cam[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Warp forward-facing rays to normalised device coordinates
    (run_nerf_helpers.py:184-201): shift origins to the near plane, then project.
    fp32 rays on the GPU that carry no gradient go through plnerf_ndc_rays -- one launch instead of ~20, the same bits;
    anything else (host tensors: the loaders' rays; rays that need a gradient) is the expression itself."""
    if (isinstance(rays_o, torch.Tensor) and rays_o.is_cuda and rays_o.dtype == torch.float32 and
            isinstance(rays_d, torch.Tensor) and rays_d.is_cuda and rays_d.dtype == torch.float32 and
            rays_o.shape == rays_d.shape and rays_o.shape[-1] == 3 and
            not (torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad))):
        from . import _lib as L
        o_c, d_c = rays_o.detach().contiguous(), rays_d.detach().contiguous()
        o_ndc, d_ndc = torch.empty_like(o_c), torch.empty_like(d_c)
        L.check(L.lib().plnerf_ndc_rays(int(H), int(W), float(focal), float(near), L.dptr(o_c), L.dptr(d_c),
                                        o_c.numel() // 3, L.dptr(o_ndc), L.dptr(d_ndc), L.stream()), "plnerf_ndc_rays")
        return o_ndc, d_ndc
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx = -1. / (W / (2. * focal))
    sy = -1. / (H / (2. * focal))
    ox_oz, oy_oz = o[..., 0] / o[..., 2], o[..., 1] / o[..., 2]
    # (the reference writes "s * x / z": the product first, then the division -- one rounding apart from s * (x / z))
    o_ndc = torch.stack([sx * o[..., 0] / o[..., 2], sy * o[..., 1] / o[..., 2], 1. + 2. * near / o[..., 2]], -1)
    d_ndc = torch.stack([sx * (rays_d[..., 0] / rays_d[..., 2] - ox_oz),
                         sy * (rays_d[..., 1] / rays_d[..., 2] - oy_oz),
                         -2. * near / o[..., 2]], -1)
    return o_ndc, d_ndc


def pose_spherical(theta, phi, radius):
    """Blender orbit camera (load_blender.py:29-34), used to synthesise benchmark rays."""
    th, ph = theta / 180. * math.pi, phi / 180. * math.pi
    trans = np.eye(4); trans[2, 3] = radius
    rphi = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0],
                     [0, 0, 0, 1.]])
    rth = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0],
                    [0, 0, 0, 1.]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.]])
    return torch.from_numpy((flip @ rth @ rphi @ trans).astype(np.float32))


def synthetic_blender_rays(n_rays, seed=0, near=2.0, far=6.0, H=800, W=800, theta=30.0, device="cpu"):
    """Synthetic 800x800 Blender-style training batch: focal from
    camera_angle_x = 0.6911112, pose_spherical(theta, -30, 4), n_rays distinct pixels.
    Returns (batch_rays [2,n,3], target [n,3], K)."""
    focal = .5 * W / math.tan(.5 * 0.6911112070083618)
    K = [[focal, 0, .5 * W], [0, focal, .5 * H], [0, 0, 1]]
    c2w = pose_spherical(theta, -30.0, 4.0)[:3, :4]
    o, d = get_rays(H, W, K, c2w)
    rng = np.random.default_rng(seed)
    pix = torch.from_numpy(rng.choice(H * W, n_rays, replace=False))
    batch = torch.stack([o.reshape(-1, 3)[pix], d.reshape(-1, 3)[pix]], 0).to(device)
    target = torch.from_numpy(rng.random((n_rays, 3), dtype=np.float32)).to(device)
    return batch, target, K
