"""Synthetic Lego-style scene generator (CPU, tiny sizes)."""
import numpy as np
import torch

from raw_ngp_amd.nerf.options import Options
from raw_ngp_amd.nerf.scene import SyntheticDataset, hemisphere_poses, make_bricks


def test_bricks_fit_the_unit_cube():
    boxes, albedo = make_bricks()
    assert boxes.shape[1] == 6 and albedo.shape == (boxes.shape[0], 3)
    assert np.all(boxes[:, :3] < boxes[:, 3:]) and np.all(np.abs(boxes) <= 0.8)


def test_poses_look_at_origin():
    poses = hemisphere_poses(8, 3.2, seed=0)
    pos = poses[:, :3, 3]
    np.testing.assert_allclose(np.linalg.norm(pos, axis=1), 3.2, rtol=1e-5)
    fwd = -poses[:, :3, 2]
    np.testing.assert_allclose((fwd * pos).sum(1) / 3.2, -1.0, atol=1e-5)      # -z axis points at the origin
    R = poses[:, :3, :3]
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (8, 1, 1)), atol=1e-5)
    assert np.all(pos[:, 2] > 0)


def test_dataset_renders_object_in_frame():
    opt = Options(bound=1.0)
    ds = SyntheticDataset(opt, torch.device("cpu"), "train", n_views=2, H=32, W=32)
    assert ds.images.shape == (2, 32, 32, 4) and ds.images.dtype == torch.uint8
    alpha = ds.images[..., 3].float() / 255
    assert 0.1 < alpha.mean() < 0.9                              # object visible, background visible
    assert alpha[:, 12:22, 10:22].mean() > 0.6                   # roughly centred
    batch = ds.sample_rays(64, torch.Generator().manual_seed(0))
    assert batch["rays_o"].shape == (64, 3) and batch["images"].shape == (64, 4)
    occ = ds.occupancy_grid(32, 1.0)
    assert 0.01 < occ.float().mean() < 0.3
