"""TEST INFRASTRUCTURE ONLY -- seeded synthetic batch in the reference schema."""
import torch


def make_batch(B, size, seed, center_idx=0):
    """Seeded synthetic batch in the reference's schema."""
    g = torch.Generator().manual_seed(seed)
    image = torch.rand((B, 3, size, size), generator=g) - 0.5
    root = torch.stack([0.05 * torch.randn(B, generator=g), 0.05 * torch.randn(B, generator=g),
                        0.5 + 0.05 * torch.randn(B, generator=g)], dim=1)
    f = 435.0 * size / 256.0 * (1.0 + 0.2 * torch.rand(B, generator=g))
    intr = torch.zeros(B, 3, 3)
    intr[:, 0, 0] = f
    intr[:, 1, 1] = f
    intr[:, 0, 2] = size / 2 + 5 * torch.randn(B, generator=g)
    intr[:, 1, 2] = size / 2 + 5 * torch.randn(B, generator=g)
    intr[:, 2, 2] = 1
    ext = 0.03 + 0.07 * torch.rand(B, 1, 3, generator=g)
    signs = torch.tensor([[sx, sy, sz] for sx in (-1., 1.) for sy in (-1., 1.) for sz in (-1., 1.)])
    corners_can = ext * signs[None]
    joints = 0.06 * torch.randn(B, 21, 3, generator=g)
    joints[:, center_idx] = 0
    corners = 0.08 * torch.randn(B, 8, 3, generator=g)
    jv = (torch.rand(B, 21, generator=g) > 0.15).float()
    cv = (torch.rand(B, 8, generator=g) > 0.15).float()
    return {"image": image, "root_joint": root, "cam_intr": intr, "corners_can": corners_can,
            "joints_3d": joints, "corners_3d": corners, "joints_vis": jv, "corners_vis": cv}


