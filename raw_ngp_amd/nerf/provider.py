"""Blender-format ("NeRF-synthetic") scenes on disk: `transforms_{train,val,test}.json` + PNG frames.

Reader = the `mode == 'blender'` branch of the reference's NeRFDataset (nerf/provider.py:113-254): which JSON a split
loads (:125-147), `file_path` + '.png' when the name has no extension (:197-198), RGBA kept as straight alpha (:211-214),
pose translation scaled/offset by `nerf_matrix_to_ngp` (:16-19, :204), focal length from `fl_x/fl_y` or
`camera_angle_x/y` (:238-248), principal point defaulting to the image centre (:250-251), mean camera radius (:230).
Images are decoded with PIL instead of cv2 (not installed here); `downscale` uses PIL's box filter where the
reference uses cv2.INTER_AREA (same area averaging for integer factors).

Writer: dumps any dataset exposing `poses [V,4,4]`, `images [V,H,W,4] uint8`, `intrinsics`, `H`, `W` (e.g. the
procedural SyntheticDataset) in that layout, so the same bytes can be read back by this reader -- or by the reference.

The result exposes what the fused engine's device-side sampler wants: `images` (uint8, on the device), `poses`,
`intrinsics`, `H`, `W`, plus `sample_rays` / `view` like SyntheticDataset.
"""
import glob
import json
import math
import os

import numpy as np
import torch

from .utils import get_rays


def nerf_matrix_to_ngp(pose, scale=0.33, offset=(0, 0, 0)):
    pose = np.array(pose, dtype=np.float32)
    pose[:3, 3] = pose[:3, 3] * scale + np.asarray(offset, dtype=np.float32)
    return pose


def write_blender_scene(root, splits, camera_angle_x=None):
    """splits: {"train": dataset, "val": dataset, ...}.  Poses are written as they are (use scale=1, offset=0 to read
    them back unchanged)."""
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    for split, ds in splits.items():
        os.makedirs(os.path.join(root, split), exist_ok=True)
        fl_x, fl_y, cx, cy = [float(v) for v in ds.intrinsics]
        meta = {"camera_angle_x": camera_angle_x if camera_angle_x is not None else 2 * math.atan(0.5 * ds.W / fl_x),
                "fl_x": fl_x, "fl_y": fl_y, "cx": cx, "cy": cy, "w": int(ds.W), "h": int(ds.H), "frames": []}
        images = ds.images.cpu().numpy()
        poses = ds.poses.cpu().numpy()
        for v in range(poses.shape[0]):
            name = f"r_{v}"
            Image.fromarray(images[v]).save(os.path.join(root, split, name + ".png"))
            meta["frames"].append({"file_path": f"./{split}/{name}", "transform_matrix": poses[v].tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump(meta, f, indent=1)


class BlenderDataset:
    def __init__(self, opt, root, ttype="train", device="cpu", downscale=1, scale=None, offset=None):
        from PIL import Image
        self.opt, self.device, self.training = opt, torch.device(device), ttype in ("train", "all", "trainval")
        self.root_path, self.downscale = root, downscale
        self.scale = getattr(opt, "scale", 1.0) if scale is None else scale
        if self.scale == -1:                    # "--data_format nerf cannot auto-choose --scale" (provider.py:106-108)
            self.scale = 1.0
        self.offset = getattr(opt, "offset", (0, 0, 0)) if offset is None else offset
        if not os.path.exists(os.path.join(root, "transforms_train.json")):
            raise NotImplementedError(f"[BlenderDataset] cannot find transforms_train.json under {root}")
        if ttype == "all":
            transform = None
            for path in sorted(glob.glob(os.path.join(root, "*.json"))):
                with open(path) as f:
                    t = json.load(f)
                if transform is None:
                    transform = t
                else:
                    transform["frames"].extend(t["frames"])
        elif ttype == "trainval":
            with open(os.path.join(root, "transforms_train.json")) as f:
                transform = json.load(f)
            with open(os.path.join(root, "transforms_val.json")) as f:
                transform["frames"].extend(json.load(f)["frames"])
        else:
            with open(os.path.join(root, f"transforms_{ttype}.json")) as f:
                transform = json.load(f)

        if "h" in transform and "w" in transform:
            self.H, self.W = int(transform["h"]) // downscale, int(transform["w"]) // downscale
        else:
            self.H = self.W = None
        poses, images = [], []
        for fr in transform["frames"]:
            path = os.path.join(root, fr["file_path"])
            if "." not in os.path.basename(path):
                path += ".png"
            if not os.path.exists(path):
                print(f"[WARN] {path} not exists!")
                continue
            img = Image.open(path)
            img = img.convert("RGBA" if img.mode in ("RGBA", "LA", "PA") else "RGB")
            if self.H is None:
                self.H, self.W = img.size[1] // downscale, img.size[0] // downscale
            if img.size != (self.W, self.H):
                img = img.resize((self.W, self.H), Image.BOX)
            images.append(np.asarray(img, dtype=np.uint8))
            poses.append(nerf_matrix_to_ngp(fr["transform_matrix"], scale=self.scale, offset=self.offset))
        self.poses = torch.from_numpy(np.stack(poses, 0)).to(self.device)
        self.images = torch.from_numpy(np.stack(images, 0)).to(self.device).contiguous()     # [V, H, W, C] uint8
        self.radius = float(self.poses[:, :3, 3].norm(dim=-1).mean())

        if "fl_x" in transform or "fl_y" in transform:
            fl_x = transform.get("fl_x", transform.get("fl_y")) / downscale
            fl_y = transform.get("fl_y", transform.get("fl_x")) / downscale
        elif "camera_angle_x" in transform or "camera_angle_y" in transform:
            fl_x = self.W / (2 * np.tan(transform["camera_angle_x"] / 2)) if "camera_angle_x" in transform else None
            fl_y = self.H / (2 * np.tan(transform["camera_angle_y"] / 2)) if "camera_angle_y" in transform else None
            fl_x = fl_y if fl_x is None else fl_x
            fl_y = fl_x if fl_y is None else fl_y
        else:
            raise RuntimeError("Failed to load focal length, please check the transforms.json!")
        cx = transform["cx"] / downscale if "cx" in transform else self.W / 2.0
        cy = transform["cy"] / downscale if "cy" in transform else self.H / 2.0
        self.intrinsics = np.array([fl_x, fl_y, cx, cy])

    def __len__(self):
        return self.poses.shape[0]

    def sample_rays(self, num_rays, generator=None, pose_fn=None):
        """random_image_batch collate: every ray picks its own (view, pixel).  pose_fn(poses, index): the pose
        optimiser's hook (provider.py:298-300); `self.ldirs` [V,3], when set, yields per-ray light directions
        (colmap_provider.py:619-620)."""
        V = self.poses.shape[0]
        index = torch.randint(0, V, size=(num_rays,), device=self.device, generator=generator)
        poses = self.poses[index]
        if pose_fn is not None:
            poses = pose_fn(poses, index)
        ldirs = getattr(self, "ldirs", None)
        rays = get_rays(poses, self.intrinsics, self.H, self.W, num_rays, generator=generator,
                        ldirs=ldirs[index] if ldirs is not None else None)
        images = self.images[index, rays["j"], rays["i"]].float() / 255
        out = {"rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "images": images, "index": index,
               "H": self.H, "W": self.W}
        if ldirs is not None:
            out["rays_ldir"] = rays["rays_ldir"]
        return out

    def view(self, v):
        ldirs = getattr(self, "ldirs", None)
        rays = get_rays(self.poses[v:v + 1], self.intrinsics, self.H, self.W, -1)
        out = {"rays_o": rays["rays_o"], "rays_d": rays["rays_d"], "images": self.images[v].float() / 255,
               "H": self.H, "W": self.W}
        if ldirs is not None:           # one light per image: the inference march repeats it per sample
            out["rays_ldir"] = ldirs[v:v + 1]
        return out
