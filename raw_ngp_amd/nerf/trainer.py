"""Minimal training / evaluation harness for the hot path.

What the reference's Trainer does around the hot path (nerf/train_utils.py:481-568 train_step,
:570-581 post_train_step, :863-975 train_one_epoch, main.py:245,261 optimiser + schedule), without
its logging / checkpoint / tensorboard machinery (out of scope):
  density-grid refresh every `update_extra_interval` steps -> random (view, pixel) rays -> render ->
  MSE against rgb*a + bg*(1-a) -> backward -> optional TV / weight-decay gradients -> Adam(eps 1e-15)
  -> lr = lr0 * 0.1^(step/iters).
With `opt.pose_opt != "none"` (train_utils.py:397, :488-489, :891-909): one se(3) correction per training camera
(nerf/pose.py) refines the poses the rays are cast from, the level-window annealing value follows step / iters, and
the pose optimiser steps beside the field's while annealing < end_annealing.
Data parallelism (one process per GPU, ray-batch sharding, gradient all-reduce over RCCL) lives in
raw_ngp_amd/parallel.py and is applied here when a process group is initialised.
"""
import time

import numpy as np
import torch

from . import utils
from .. import parallel


class Trainer:
    def __init__(self, opt, model, dataset, device="cuda", seed=0):
        self.opt, self.model, self.data, self.device = opt, model.to(device), dataset, device
        self.rank, self.world_size = parallel.rank(), parallel.world_size()
        self.global_step = 0
        self.optimizer = torch.optim.Adam(self.model.get_params(opt.lr), eps=1e-15)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer,
                                                           lambda it: 0.1 ** min(it / opt.iters, 1))
        self.criterion = torch.nn.MSELoss(reduction="none")
        self.scaler = torch.amp.GradScaler("cuda", enabled=bool(opt.fp16))
        # every rank draws different rays but the same density-grid jitter
        self.ray_gen = torch.Generator(device=device).manual_seed(seed * 1000 + self.rank)
        self.reducer = parallel.GradReducer(self.model) if self.world_size > 1 else None
        if self.world_size > 1:
            parallel.broadcast_module(self.model)
        self.pose_optimizer = None
        if opt.pose_opt != "none":
            from .pose import CameraOptimizer
            self.pose_optimizer = CameraOptimizer(len(dataset), device, opt, seed=seed)
        self.annealing = 0.0
        self.last_loss = None
        self.last_num_points = 0

    # ------------------------------------------------------------------ one optimiser step
    def bg_color(self, n):
        mode = self.opt.background
        if mode == "random":
            return torch.rand(n, 3, device=self.device, generator=self.ray_gen)
        return 1 if mode in ("white", "last_sample") else 0

    def train_step(self):
        opt, model = self.opt, self.model
        model.train()
        if model.cuda_ray and self.global_step % opt.update_extra_interval == 0:
            if self.world_size > 1:
                torch.manual_seed(1234567 + self.global_step)      # identical jitter on every rank
            model.update_extra_state()
        refine = self.pose_optimizer is not None
        if refine:
            # a numpy float16, like the reference (train_utils.py:488): the level windows see its rounding
            # (global_step is incremented in front of train_step there, train_utils.py:887-888: step s sees (s + 1) / iters;
            # the density-grid refresh above still saw the previous step's value, 0.0 before the first one, :411)
            self.annealing = np.clip((self.global_step + 1) / opt.iters, 0, 1).astype(np.float16)
            model.update_annealing(self.annealing)
            self.pose_optimizer.update_annealing(self.annealing)
            data = self.data.sample_rays(opt.num_rays, self.ray_gen, pose_fn=self.pose_optimizer)
        else:
            data = self.data.sample_rays(opt.num_rays, self.ray_gen)
        pose_step = refine and self.annealing < opt.end_annealing
        images = data["images"]
        bg = self.bg_color(images.shape[0])
        gt = images[..., :3] * images[..., 3:] + bg * (1 - images[..., 3:]) if images.shape[-1] == 4 else images

        self.optimizer.zero_grad(set_to_none=True)
        if pose_step:
            self.pose_optimizer.optimizer.zero_grad(set_to_none=True)
        out = model.render(data["rays_o"], data["rays_d"], rays_ldir=data.get("rays_ldir"), bg_color=bg, perturb=True)
        if getattr(opt, "image_mode", "LDR") == "HDR":      # train_utils.py:512-536
            exposures = getattr(self.data, "exposures", None)
            exposure = (torch.as_tensor(exposures, dtype=torch.float32, device=gt.device)[data["index"]]
                        if exposures is not None else torch.ones(gt.shape[0], device=gt.device))
            loss = utils.hdr_loss(out["image"], gt, exposure, getattr(opt, "loss_weight", "none"))
        else:
            loss = self.criterion(out["image"], gt).mean(-1).mean()
        if "proposal_loss" in out and opt.lambda_proposal > 0:
            loss = loss + opt.lambda_proposal * out["proposal_loss"]
        if "orientation_loss" in out and opt.lambda_orientation > 0:
            loss = loss + opt.lambda_orientation * out["orientation_loss"]
        if opt.lambda_entropy > 0:
            w = out["weights_sum"].clamp(1e-5, 1 - 1e-5)
            loss = loss + opt.lambda_entropy * (-w * torch.log2(w) - (1 - w) * torch.log2(1 - w)).mean()
        self.scaler.scale(loss).backward()
        if self.reducer is not None:
            self.reducer.all_reduce()
        self.scaler.unscale_(self.optimizer)
        if opt.lambda_tv > 0:
            model.apply_total_variation(opt.lambda_tv)
        if opt.lambda_wd > 0:
            model.apply_weight_decay(opt.lambda_wd)
        self.scaler.step(self.optimizer)
        if pose_step:
            if self.world_size > 1:
                for p in self.pose_optimizer.parameters():
                    if p.grad is not None:
                        torch.distributed.all_reduce(p.grad)
                        p.grad.div_(self.world_size)
            self.scaler.step(self.pose_optimizer.optimizer)
        self.scaler.update()
        self.scheduler.step()
        if pose_step:
            self.pose_optimizer.lr_scheduler.step()
        self.global_step += 1
        self.last_loss = loss.detach()
        self.last_num_points = out.get("num_points", 0)
        if opt.adaptive_num_rays and self.last_num_points:
            opt.num_rays = int(round((opt.num_points / self.last_num_points) * opt.num_rays))
        return self.last_loss

    def train(self, steps, log_every=0):
        t0 = time.time()
        for _ in range(steps):
            self.train_step()
            if log_every and self.rank == 0 and self.global_step % log_every == 0:
                print(f"[step {self.global_step}] loss {float(self.last_loss):.5f} samples {self.last_num_points} "
                      f"({time.time() - t0:.1f}s)", flush=True)

    # ------------------------------------------------------------------ evaluation
    @torch.no_grad()
    def evaluate(self, dataset, max_views=None, chunk=1 << 16):
        """PSNR over held-out views, averaged per image (train_utils.py:221-233)."""
        self.model.eval()
        meter = utils.PSNRMeter()
        n = len(dataset) if max_views is None else min(max_views, len(dataset))
        for v in range(n):
            data = dataset.view(v)
            preds = []
            for s in range(0, data["rays_o"].shape[0], chunk):
                ld = data.get("rays_ldir")
                out = self.model.render(data["rays_o"][s:s + chunk], data["rays_d"][s:s + chunk],
                                        rays_ldir=ld, bg_color=0,
                                        perturb=False)
                preds.append(out["image"])
            pred = torch.cat(preds, 0).view(data["H"], data["W"], 3)
            img = data["images"]
            gt = img[..., :3] * img[..., 3:] if img.shape[-1] == 4 else img
            meter.update(pred.clamp(0, 1), gt)
        return meter.measure()
