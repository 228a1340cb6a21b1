"""Blender-format scene writer/reader (CPU): round trip through the on-disk layout the reference's provider reads."""
import json
import os

import numpy as np
import torch

from raw_ngp_amd.nerf.options import Options
from raw_ngp_amd.nerf.provider import BlenderDataset, nerf_matrix_to_ngp, write_blender_scene
from raw_ngp_amd.nerf.scene import SyntheticDataset


def test_blender_round_trip(tmp_path):
    opt = Options(bound=1.0)
    dev = torch.device("cpu")
    train = SyntheticDataset(opt, dev, "train", n_views=3, H=24, W=32)
    val = SyntheticDataset(opt, dev, "val", n_views=2, H=24, W=32)
    root = str(tmp_path / "scene")
    write_blender_scene(root, {"train": train, "val": val})
    # layout the reference expects (nerf/provider.py:113-147, :194-198)
    meta = json.load(open(os.path.join(root, "transforms_train.json")))
    assert set(meta) >= {"camera_angle_x", "frames"} and len(meta["frames"]) == 3
    assert os.path.exists(os.path.join(root, meta["frames"][0]["file_path"] + ".png"))

    back = BlenderDataset(opt, root, "train", scale=1.0, offset=(0, 0, 0))
    assert (back.H, back.W) == (24, 32) and len(back) == 3
    assert torch.equal(back.images, train.images)                      # PNG is lossless, RGBA kept straight
    np.testing.assert_allclose(back.poses.numpy(), train.poses.numpy(), atol=0)
    np.testing.assert_allclose(back.intrinsics, train.intrinsics, rtol=1e-12)
    both = BlenderDataset(opt, root, "trainval", scale=1.0, offset=(0, 0, 0))
    assert len(both) == 5
    # scale / offset as in nerf_matrix_to_ngp (provider.py:16-19); downscale halves the image and the focal length
    small = BlenderDataset(opt, root, "val", downscale=2, scale=0.5, offset=(0.1, 0, 0))
    assert (small.H, small.W) == (12, 16)
    np.testing.assert_allclose(small.intrinsics, val.intrinsics / 2)
    want = nerf_matrix_to_ngp(val.poses[0].numpy(), 0.5, (0.1, 0, 0))
    np.testing.assert_allclose(small.poses[0].numpy(), want)
    np.testing.assert_allclose(small.poses[0, :3, :3].numpy(), val.poses[0, :3, :3].numpy())
    # camera_angle_x only (the original NeRF-synthetic files carry nothing else)
    for k in ("fl_x", "fl_y", "cx", "cy", "w", "h"):
        meta.pop(k)
    json.dump(meta, open(os.path.join(root, "transforms_train.json"), "w"))
    angle = BlenderDataset(opt, root, "train", scale=1.0, offset=(0, 0, 0))
    np.testing.assert_allclose(angle.intrinsics, train.intrinsics, rtol=1e-6)
    batch = angle.sample_rays(64, torch.Generator().manual_seed(0))
    assert batch["rays_d"].shape == (64, 3) and batch["images"].shape == (64, 4)
