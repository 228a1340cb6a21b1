#!/usr/bin/env python3
"""Benchmark of the hot path: training rays/s on a synthetic 800x800 Lego-style scene.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.  Started WITHOUT that environment and
--gpus N > 1, the script launches the N ranks itself (child processes, before this process touches the GPU) or exits
non-zero: it never reports a one-GPU number for an N-GPU request.

  step      = one optimiser step of the density-grid path: 4096 random rays -> near/far -> occupancy march
              -> hash-grid + SH encode -> tiny MLPs -> composite -> MSE -> backward (composite, MLPs, SH,
              hash-grid scatter) -> Adam, plus the density-grid refresh every 16 steps (amortised).
              Rays, images, tables and the occupancy grid are resident in HBM before the timed region.
  value     = world_size * rays_per_step * K / max-over-ranks wall time of the K timed steps.
  burn-in   = untimed training steps before the W warm-up steps, so that the 16 full density-grid sweeps
              are over and the occupancy grid has converged (metric definition):
              setup, not part of W or K.
  roofline  = the dominant kernel of the step, timed live with HIP events around its launches in a PROBE TAIL: the
              timed region itself carries no probes (a probed step splits its graph around the probed launches); right
              after it -- same graphs' kernels, same scene state, the schedule simply continues -- `--probe-launches`
              steps are probed, one in `--probe-every`, half way between two density-grid refreshes.  Algorithmic bytes
              per sample from SURVEY.md section 8d.  On one GPU that
              launch also applies Adam to the table (fused into its reduce kernel): the optimiser's 24 B per
              table entry are then part of its algorithmic bytes (`optimizer_bytes_per_launch`; the figure without
              them is `frac_grid_only`).
  cpu_baseline = the same step restated on the CPU oracle (oracle/ngp_oracle.c kernels + torch CPU MLPs and
              Adam) on a bounded ray sample, all host cores (kind "port": the reference has no CPU path for
              the encoders and its CUDA kernels cannot be built here).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raw_ngp_amd import _lib, parallel  # noqa: E402
from raw_ngp_amd.nerf.network import NeRFNetwork  # noqa: E402
from raw_ngp_amd.nerf.options import Options  # noqa: E402
from raw_ngp_amd.nerf.scene import SyntheticDataset  # noqa: E402
from raw_ngp_amd.nerf.engine import FusedTrainer  # noqa: E402
from raw_ngp_amd.nerf.trainer import Trainer  # noqa: E402

METRIC = "training rays/sec + PSNR@5k-iters, NeRF-synthetic Lego 800², 1/2/4/8 MI355X"
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

# HBM-side traffic of the hash-grid kernels, measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE ON THIS SCRIPT's own
# steady state (tools/pmc_bench.sh, profiles/r04_pmc_bench_traffic.csv: last 30 steps before iteration 5000, 140.4 k samples
# per step, --no-graph; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950 -- full-line reads are tallied at
# 64 B).  Counters cannot be read while this script times itself, so the per-sample figures are carried as constants and
# scaled by the samples of the run.  Round 4 (tile-local records, pair records on the hashed levels, backward over the list
# of samples in front of the compositor's early stop):
#   binned backward, Adam fused (one GPU):  fill 2 x 24.1 + 51.0 MB, reduce 2 x 129.7 + 149.6 MB = 508.2 MB per launch, of
#     which 292.7 MB are the optimiser state (24 B x 12.2 M table entries, independent of the samples) -> 1 535 B/sample +
#     292.7 MB (round 3: 1 169 B/sample at 143.5 k samples; round 2, global bins: 723 B/sample)
#   binned backward, gradient written (data parallel / --no-fuse-adam; round 3's pass): fill 76.6 MB + reduce 2 x 40.8 + 47.6
#     MB = 205.9 MB -> 1 435 B/sample (the separate Adam launch, 2 x 95.4 + 143.1 MB, is not part of the probed entry point)
#   slab forward: 59.6 + 25.4 MB = 85.0 MB per launch -> 605 B/sample (gathers of 8 / 16 bytes, counted as reported)
PMC_TRAFFIC_BYTES_PER_SAMPLE = {"ngp_x_grid_backward_binned": 1435.0}
PMC_FUSED_TRAFFIC_BYTES_PER_SAMPLE = 1535.0
PMC_FWD_TRAFFIC_BYTES_PER_SAMPLE = 605.0
FWD_BYTES_PER_SAMPLE = 12 + 16 * (64 + 8)      # 1164 B/sample, SURVEY.md section 8d
# What the forward's ADDRESS STREAM can reach with the arithmetic taken away: tools/ubench/gather_lines.hip issues exactly the
# slab forward's loads (same table, same ray-ordered samples, same level -> XCD placement) and nothing else.  Best variant
# (8-byte gathers, 8 in flight per lane, snake placement of the levels over the XCDs) on 139 264 samples: 36.3 us = 3.84 G
# samples/s, i.e. 4 465 GB/s in the forward's algorithmic bytes (profiles/r03_ubench_gather_lines.txt; 4 670 GB/s at 204 800
# samples).  The table is cache-resident: this, not the 8 TB/s of HBM, is the kernel's ceiling -- `frac_of_line_rate` is
# measured against it.
FWD_LINE_RATE_CEILING_GBPS = 4465.0

# entry point(s) timed with HIP events -> (index of the samples-per-launch argument, algorithmic bytes per sample)
ROOFLINE_KERNELS = {
    # the table gradient of the fused step is one operation launched in two halves (bins sized with the march, on the
    # side stream; fill + reduce after the MLP backward): both are timed and their durations added
    "ngp_x_grid_backward_binned": (5, 12 + 16 * (8 + 64)),           # 1164 B/sample, SURVEY.md section 8d
    "ngp_x_grid_encode_backward_binned": (5, 12 + 16 * (8 + 64)),   # the same in one call (per-op autograd path)
    "ngp_grid_encode_backward": (5, 12 + 16 * (8 + 64)),             # the reference-shaped float-atomic scatter
    "ngp_grid_encode_forward": (4, 12 + 16 * (64 + 8)),              # 1164 B/sample
}


def cpu_baseline(opt, n_rays, steps, seed=0):
    """One training step restated on the CPU oracle; returns rays/s and the thread count."""
    from oracle import oracle as orc
    import torch.nn.functional as F
    # the GPU box gives one GPU's share of host cores (16); more OpenMP threads than that only thrash
    threads = min(len(os.sched_getaffinity(0)), 16)
    orc.set_threads(threads)
    torch.set_num_threads(threads)
    rng = np.random.default_rng(seed)
    dev = torch.device("cpu")
    data = SyntheticDataset(opt, dev, "train", n_views=2, H=100, W=100)
    H = opt.grid_size
    occ = data.occupancy_grid(H, opt.bound).numpy()
    xs, ys, zs = np.nonzero(occ)
    grid = np.zeros((1, H ** 3), dtype=np.float32)
    grid[0, orc.morton3D(np.stack([xs, ys, zs], 1).astype(np.int32))] = 1.0
    bits = orc.packbits(grid, 0.5)
    D, C, L, Hb = 3, 2, 16, 16
    offsets, scale = orc.grid_offsets(desired_resolution=opt.hashgrid_resolution * opt.bound,
                                      log2_hashmap_size=opt.hashmap_size)
    S = float(np.log2(scale))
    table = torch.nn.Parameter(torch.from_numpy(rng.uniform(-1e-4, 1e-4, (int(offsets[-1]), C)).astype(np.float32)))
    w_grid = [torch.nn.Parameter(torch.randn(o, i) * (1.0 / i) ** 0.5) for i, o in ((32, 64), (64, 64), (64, 16))]
    w_view = [torch.nn.Parameter(torch.randn(o, i) * (1.0 / i) ** 0.5) for i, o in ((31, 64), (64, 64), (64, 3))]
    optim = torch.optim.Adam([table] + w_grid + w_view, lr=opt.lr, eps=1e-15)
    aabb = np.array([-opt.bound] * 3 + [opt.bound] * 3, dtype=np.float32)

    def mlp(x, ws):
        for k, w in enumerate(ws):
            x = F.linear(x, w)
            if k < len(ws) - 1:
                x = F.relu(x)
        return x

    def step():
        r = data.sample_rays(n_rays)
        o, d = r["rays_o"].numpy().copy(), r["rays_d"].numpy().copy()
        img = r["images"]
        gt = (img[:, :3] * img[:, 3:]).numpy()
        nears, fars = orc.near_far_from_aabb(o, d, aabb, n_rays, opt.min_near)
        noises = rng.uniform(0, 1, n_rays).astype(np.float32)
        xyzs, dirs, ts, rays, _, M = orc.march_rays_train(o, d, None, bits, opt.bound, False, 0.0, opt.max_steps, 1, H,
                                                          nears, fars, noises)
        if M == 0:
            return 0
        x01 = ((xyzs + opt.bound) / (2 * opt.bound)).astype(np.float32)
        enc, _ = orc.grid_encode_forward(x01, table.detach().numpy(), offsets, M, D, C, L, L, S, Hb)
        feat = torch.from_numpy(np.ascontiguousarray(enc.transpose(1, 0, 2).reshape(M, L * C))).requires_grad_(True)
        dn = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
        sh, _ = orc.sh_encode_forward(dn.astype(np.float32), M, 3, 4)
        h = mlp(feat, w_grid)
        sigma = torch.exp(h[:, 0])
        color = torch.clamp(torch.exp(mlp(torch.cat([h[:, 1:], torch.from_numpy(sh)], 1), w_view) - 5.0), max=5.0)
        w, ws, dep, image = orc.composite_rays_train_forward(sigma.detach().numpy(), color.detach().numpy(), ts, rays, M,
                                                            n_rays, opt.T_thresh)
        g_img = (2.0 / (3 * n_rays)) * (image - gt)
        gs, gc = orc.composite_rays_train_backward(np.zeros(M, np.float32), np.zeros(n_rays, np.float32),
                                                   np.zeros(n_rays, np.float32), g_img.astype(np.float32),
                                                   sigma.detach().numpy(), color.detach().numpy(), ts, rays, ws, dep, image,
                                                   M, n_rays, opt.T_thresh)
        optim.zero_grad()
        ((sigma * torch.from_numpy(gs)).sum() + (color * torch.from_numpy(gc)).sum()).backward()
        g_enc = np.ascontiguousarray(feat.grad.numpy().reshape(M, L, C).transpose(1, 0, 2))
        g_tab, _ = orc.grid_encode_backward(g_enc, x01, table.detach().numpy(), offsets, M, D, C, L, L, S, Hb)
        table.grad = torch.from_numpy(g_tab)
        optim.step()
        return M

    step()
    t0 = time.perf_counter()
    samples = sum(step() for _ in range(steps))
    dt = time.perf_counter() - t0
    return {"value": n_rays * steps / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{steps} training steps of {n_rays} rays ({samples // max(steps, 1)} samples/step) on the "
                      f"oracle kernels + torch-CPU MLP/Adam, {dt:.1f} s"}


def run_config(args, bound, background, rank, world, dev, probe_on=True):
    """Train one configuration up to the timed region, time K steps, evaluate PSNR.  Returns a dict of raw results."""
    torch.manual_seed(0)                    # every configuration starts from the same initial weights
    opt = Options(bound=bound, background=background, num_rays=args.rays, iters=max(args.psnr_iters, 5000), arena_capacity=args.arena,
                  fused_mlp=not args.torch_mlp, prefetch_march=not args.no_prefetch,
                  capture_graph=not args.no_graph, device_sampler=not args.torch_sampler,
                  aux_stream=args.aux, grad_wire=args.grad_wire, fuse_adam=not args.no_fuse_adam,
                  dp_rehearsal=args.dp_rehearsal, dp_exchange=args.dp_exchange, group_steps=args.group_steps,
                  group_any=not args.group_pow2, group_ramp=not args.no_group_ramp,
                  dp_mode=args.dp_mode, dynamic_loss_scale=not args.static_loss_scale, dp_split_level=args.dp_split_level,
                  **({"loss_scale": args.loss_scale} if args.loss_scale else {}))
    data = SyntheticDataset(opt, dev, "train", n_views=args.views, H=args.res, W=args.res)
    model = NeRFNetwork(opt)
    fused = not (args.autograd or args.torch_mlp)
    if fused:
        trainer = FusedTrainer(opt, model, data, device=dev, seed=args.seed, capacity=args.arena or None)    # default: rays * 160 * ceil(bound)
    else:
        trainer = Trainer(opt, model, data, device=dev)
    # the reference's -O preset marks cells no training camera sees as never-to-sample (train_utils.py: mark_untrained)
    trainer.model.mark_untrained_grid(data)
    untrained_cells = int((trainer.model.density_grid < 0).sum())

    # set before the step is captured into graphs: the engine keeps the probed entry point out of them
    arg_idx, bytes_per_sample = ROOFLINE_KERNELS[args.roofline_kernel]
    symbols = (args.roofline_kernel,)
    if args.roofline_kernel == "ngp_x_grid_backward_binned":
        # (one GPU, plain field: the apply call that also carries the MLP's weight-gradient reduction along)
        ride = fused and trainer.rides_mlp_tail()
        symbols = ("ngp_x_grid_backward_binned_apply" + ("_mlp" if ride else ""), "ngp_x_grid_backward_binned_prepare")
    # the north star also names the encoder's forward: timed the same way, reported as roofline_forward
    fwd_symbol = "ngp_x_grid_encode_forward_slab"
    probed = symbols + ((fwd_symbol,) if fused and fwd_symbol not in symbols else ())
    # (every probe_every-th step, in the middle of the period: with the default 16 these are the steps half way between two
    # density-grid refreshes -- a timed step splits its graphs around the probed launches, ~ 50 us the other steps do not pay)
    probing = probe_on and not args.no_probe
    _lib.set_probe(None)                    # (the timed region carries no probes: they follow it, see `probe tail` below)
    # The fused engine replays runs of consecutive steps from hipGraphs whose shape depends on where a run starts inside the
    # 16-step density-grid cycle, on its length and on which steps are probed; a variant that is first needed inside the
    # timed region would be CAPTURED there (milliseconds of host time: the driver's 20-step region is 7 ms long).  So the
    # burn-in contains a dress rehearsal: the same number of steps, launched by the same call, a whole number of
    # refresh / probe periods before the timed region -- every graph the timed region replays has then been captured and
    # launched once, inside ordinary (untimed) burn-in training.  The schedule of steps is unchanged.
    period = int(opt.update_extra_interval)
    back = -(-(args.steps + args.warmup) // period) * period
    start = args.burnin + args.warmup                    # first step of the timed region
    if fused and start - back >= 2 * period:
        trainer.train(start - back)
        trainer.train(args.steps)                        # the rehearsal (same grouping decisions as the timed call)
        trainer.train(args.burnin - (start - back) - args.steps)
    else:
        trainer.train(args.burnin)
    trainer.train(args.warmup)
    if fused and args.precapture:
        # every run of up to --group-steps regular steps between two refreshes / timed steps becomes ONE graph launch; the
        # captures (ms each) happen here, outside the timed region, and execute nothing.  Pays on long regions (200 steps:
        # -1 %); on the driver's 20-step region the first launch of a never-launched graph costs more than it saves
        trainer.precapture_groups()

    # ---- timed region -------------------------------------------------------------------------
    parallel.barrier()
    torch.cuda.synchronize()
    seen0 = int(trainer.samples_seen) if fused else 0
    t0 = time.perf_counter()
    samples = 0
    host = 0.0
    if fused:       # the engine's own loop: consecutive steps are replayed from one graph where nothing forbids it
        h0 = time.perf_counter()
        trainer.train(args.steps)
        host = time.perf_counter() - h0
    else:
        for _ in range(args.steps):
            h0 = time.perf_counter()
            trainer.train_step()
            host += time.perf_counter() - h0
            samples += trainer.last_num_points      # (host value of the per-op path; the fused step never syncs)
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    overflow = False
    if fused:
        samples = int(trainer.samples_seen) - seen0
        overflow = int(trainer.arena.counter[1]) > trainer.cap
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    in_sync = None
    if world > 1 and fused:
        # replicas must hold identical parameters and the same occupancy bitfield after the timed steps
        sums = torch.stack([trainer.table.double().sum(), trainer.w_flat.double().sum(),
                            model.density_bitfield.double().sum()])
        lo, hi = sums.clone(), sums.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi))

    psnr, eval_check = None, None
    if args.psnr_iters > trainer.global_step:
        trainer.train(args.psnr_iters - trainer.global_step)
    psnr_step = trainer.global_step
    if args.psnr_iters > 0:
        val = SyntheticDataset(opt, dev, "val", n_views=4, H=args.res, W=args.res)
        psnr = trainer.evaluate(val)        # (more than one rank: views dealt to the ranks, images all-gathered)
        if world > 1 and fused:
            # ... which must be the value ONE rank computes over all views from its replica of the parameters
            eval_check = bool(psnr == trainer.evaluate(val, distributed=False))
    scaler_state = trainer.scaler.state() if fused and getattr(trainer, "scaler", None) is not None else None

    # ---- probe tail: the roofline kernels under HIP events, untimed --------------------------------------------
    # The schedule simply continues (same weights, same occupancy grid, the graphs' own kernels); one step in
    # --probe-every is probed, half way between two density-grid refreshes, until --probe-launches launches are in.  A probed
    # step splits its graph around the probed launches (~ 50 us it would cost the timed region).  Under data parallelism the
    # collectives of the probed steps are timed the same way (they stay outside those steps' graphs).
    probe, probe_fwd, collective_ms = (0, 0, 0.0), (0, 0, 0.0), None
    if probing:
        _lib.set_probe(probed, arg_idx, every=args.probe_every, phase=args.probe_every // 2)
        tail = args.probe_every * (args.probe_launches + 2)
        if fused and trainer.xchg is not None and trainer.xchg.carrier != "none":
            trainer.collective_events, trainer.collective_steps = [], 0
        trainer.train(2 * args.probe_every)               # (the probed variants of the step are captured here)
        torch.cuda.synchronize()
        _lib.probe_reset()
        if fused and trainer.collective_events is not None:
            trainer.collective_events, trainer.collective_steps = [], 0
        seen1 = int(trainer.samples_seen) if fused else 0
        if fused:
            trainer.train(tail - 2 * args.probe_every)
        else:
            for _ in range(tail - 2 * args.probe_every):
                trainer.train_step()
        torch.cuda.synchronize()
        if fused and getattr(trainer, "collective_events", None):
            collective_ms = sum(a.elapsed_time(b) for a, b in trainer.collective_events) / max(trainer.collective_steps, 1)
        if fused:
            trainer.collective_events = None
        probe = _lib.probe_results(symbols)
        probe_fwd = _lib.probe_results((fwd_symbol,)) if fwd_symbol in probed else (0, 0, 0.0)
        _lib.set_probe(None)
        if fused:
            # live samples per launch (the launch argument is the arena capacity): the tail's own average
            per_step = (int(trainer.samples_seen) - seen1) / max(tail - 2 * args.probe_every, 1)
            probe = (probe[0], per_step * probe[0], probe[2])
            probe_fwd = (probe_fwd[0], per_step * probe_fwd[0], probe_fwd[2])

    return dict(dt=dt, host=host, samples=samples, probe=probe, probe_fwd=probe_fwd, probed=probed, fwd_symbol=fwd_symbol,
                bytes_per_sample=bytes_per_sample, in_sync=in_sync, psnr=psnr, trainer=trainer, fused=fused,
                collective_ms=collective_ms, psnr_step=psnr_step, scaler_state=scaler_state, eval_check=eval_check,
                untrained_cells=untrained_cells, overflow=bool(fused and overflow), model=model)


def run_config4(args, dev):
    """BASELINE configs[3] as far as it can be run without its (private) data: the light-conditioned field (rfield:
    47 -> 80 -> 80 -> 3 view MLP over [features, SH(view), SH(light)]), BARF pose refinement from perturbed cameras, the HDR
    loss with per-view exposures, bound 2 as `--lightstage` sets it (main.py:129-143), on the procedural scene with synthetic
    light directions / exposures (SURVEY 8d "Config 4").  One fused step per iteration (raw_ngp_amd.nerf.engine with
    rfield + pose_opt + image_mode HDR); timed like the headline: K steps between synchronisations after a burn-in."""
    from raw_ngp_amd.nerf import pose as P
    torch.manual_seed(0)
    iters = 3000
    # (adaptive_num_rays: `--lightstage` switches it on, main.py:142 -- the batch follows num_points = 2^18 samples per step)
    opt = Options(bound=2.0, num_rays=args.rays, iters=iters, rfield=True, pose_opt="barf", noise=0.03, image_mode="HDR",
                  background="black", adaptive_num_rays=True)
    views, res = 40, 200
    data = SyntheticDataset(opt, dev, "train", n_views=views, H=res, W=res)
    data.ldirs = torch.from_numpy(P.synthetic_light_dirs(views)).to(dev)
    data.exposures = torch.from_numpy(np.random.default_rng(5).choice([0.5, 1.0, 2.0], views).astype(np.float32)).to(dev)
    # what a camera with that exposure records of the same radiance: colour x exposure, clipped at white
    data.images[..., :3] = (data.images[..., :3].float() * data.exposures.view(-1, 1, 1, 1)).clamp(max=255).to(torch.uint8)
    trainer = FusedTrainer(opt, NeRFNetwork(opt), data, device=dev, seed=args.seed)
    co = trainer.pose_optimizer
    err0 = P.pose_error(co.get_refined_poses(data.poses), data.poses)
    moving = int(opt.end_annealing * iters)             # the cameras move while annealing < end_annealing
    out = {}
    for tag, upto in (("cameras_moving", moving // 2), ("cameras_frozen", iters - args.steps)):
        trainer.train(upto - trainer.global_step - args.warmup)
        trainer.train(args.warmup)
        torch.cuda.synchronize()
        seen0, rays0 = int(trainer.samples_seen), int(trainer.rays_seen)
        t0 = time.perf_counter()
        trainer.train(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays = int(trainer.rays_seen) - rays0           # adaptive batches: what the steps really carried
        out[tag] = {"at_step": upto, "ms_per_step": round(dt / args.steps * 1e3, 4),
                    "value": round(rays / dt, 1), "unit": "rays/s", "rays_per_step": round(rays / args.steps),
                    "samples_per_step": round((int(trainer.samples_seen) - seen0) / args.steps)}
    err1 = P.pose_error(co.get_refined_poses(data.poses), data.poses)
    out.update({"workload": f"configs[3] stand-in: rfield + barf + HDR loss, bound 2, adaptive ray batches towards {opt.num_points} "
                            f"samples/step (first batch {args.rays} rays), {views} views of "
                            f"{res}x{res} (synthetic light directions, exposures, 0.03 se(3) noise), {iters} iterations",
                "pose_error_deg_dist_start": [round(float(v), 4) for v in err0],
                "pose_error_deg_dist_end": [round(float(v), 4) for v in err1],
                "loss": round(float(trainer.loss), 6), "arena_overflow": int(trainer.arena.counter[1]) > trainer.cap})
    return out


def launch_ranks(n):
    """`--gpus n` without torchrun's environment: start the n ranks as CHILD processes of this one (torch.distributed.run,
    one per GPU, rendezvous on 127.0.0.1) and return their exit code.  Called before anything in this process has
    initialised the GPU -- counting devices does not -- and never as an exec; the children print the JSON line."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and "NGP_LOCAL_DEVICE" not in os.environ:
        print(f"bench.py: --gpus {n} but this node shows {have} GPU(s); refusing to report a smaller job under that name "
              "(rehearsal on one device: NGP_DIST_BACKEND=gloo NGP_LOCAL_DEVICE=0)", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sock:           # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--burnin", type=int, default=-1,
                    help="untimed training steps before warm-up.  Default: as many as it takes for the timed region to "
                         "END at iteration --psnr-iters (5000): the metric pairs rays/s with PSNR@5k, so the throughput is "
                         "quoted at that point of the 5000-iteration schedule (learned occupancy grid), and the PSNR is "
                         "evaluated right after the timed steps")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--background", default="random", choices=["black", "white", "random"],
                    help="training background (main.py:46).  random (per-ray colours, what torch-ngp trains RGBA "
                         "NeRF-synthetic data with) is the default here: against the reference's default, black, dark fog in "
                         "empty space is invisible to the loss, and how much of it survives decides samples per ray")
    ap.add_argument("--bound", type=float, default=1.0,
                    help="scene bound (1 = the benchmark framing; 2 = the reference's default, two cascades)")
    ap.add_argument("--views", type=int, default=100)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--roofline-kernel", default="ngp_x_grid_backward_binned", choices=sorted(ROOFLINE_KERNELS))
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--cpu-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--psnr-iters", type=int, default=5000,
                    help="report PSNR on held-out views at this iteration (training continues to it after the timed "
                         "region if needed; 0 = skip)")
    ap.add_argument("--arena", type=int, default=0, help="sample arena capacity (0 = reference two-pass march)")
    ap.add_argument("--torch-mlp", action="store_true", help="fp32 nn.Linear MLPs instead of the fused f16 MFMA field")
    ap.add_argument("--group-steps", type=int, default=15,
                    help="fused step: consecutive steps per captured graph (1 = one graph launch per step)")
    ap.add_argument("--group-pow2", action="store_true", help="step groups of 2, 4, 8 steps only (default: any length up to --group-steps)")
    ap.add_argument("--no-group-ramp", action="store_true",
                    help="let the first group of a train() call be long too (default: at most 2 steps -- the stream may be idle)")
    ap.add_argument("--precapture", action="store_true", help="capture step groups of every length up front instead of 2/4/8 on demand")
    ap.add_argument("--probe-every", type=int, default=16,
                    help="probe tail: time every N-th launch of the roofline entry points with HIP events (16 = the step half way "
                         "between two density-grid refreshes)")
    ap.add_argument("--probe-launches", type=int, default=12,
                    help="probe tail: how many launches of each roofline entry point to time after the timed region")
    ap.add_argument("--no-probe", action="store_true", help="skip the HIP-event roofline probe (roofline: null)")
    ap.add_argument("--no-graph", action="store_true", help="fused step: launch kernels one by one (no hipGraph replay)")
    ap.add_argument("--torch-sampler", action="store_true", help="fused step: draw rays with torch ops (implies no graph)")
    ap.add_argument("--aux", action="store_true", help="fused step: MLP-weight tail on a third stream (slower)")
    ap.add_argument("--no-fuse-adam", action="store_true",
                    help="separate Adam pass over the table (what data-parallel ranks run), on one GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--loss-scale", type=float, default=0.0,
                    help="(initial) loss scale of the fused MLP backward's f16 deltas (0 = the Options default, 2^16)")
    ap.add_argument("--static-loss-scale", action="store_true",
                    help="keep the loss scale fixed (deltas saturate) instead of GradScaler's rule on the device")
    ap.add_argument("--grad-wire", default="f32", choices=["f32", "bf16"],
                    help="data parallel: wire format of the table-gradient all-reduce")
    ap.add_argument("--dp-mode", default="shard", choices=["shard", "allreduce"],
                    help="data parallel: reduce_scatter -> Adam on 1/R of the table -> all_gather (shard), or gradient "
                         "all-reduce + full Adam on every rank (allreduce)")
    ap.add_argument("--dp-exchange", default=None, choices=["rccl", "torch"],
                    help="data parallel, carrier of the collectives: rccl = bare RCCL calls captured inside the step graphs "
                         "(default on an nccl process group), torch = torch.distributed, eager between graph segments")
    ap.add_argument("--dp-split-level", type=int, default=None,
                    help="data parallel (shard, f32 wire): exchange in two level groups split at this level (default: 8 with more "
                         "than one rank, off on one; 0 = one group)")
    ap.add_argument("--dp-rehearsal", action="store_true",
                    help="one GPU: run the data-parallel step (separate Adam, RCCL collectives on a one-rank group)")
    ap.add_argument("--no-prefetch", action="store_true", help="fused step: march on the main stream (no overlap)")
    ap.add_argument("--autograd", action="store_true", help="per-op autograd path (Trainer) instead of the fused step")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary rows (reference defaults: black background; bound 2 + black; the configs[3] stand-in)")
    args = ap.parse_args()
    if args.burnin < 0:
        args.burnin = max((args.psnr_iters or 5000) - args.warmup - args.steps, 300)

    if args.dp_rehearsal:
        os.environ["NGP_DP_REHEARSAL"] = "1"
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        # not under a launcher: become one (children do the work), or fail -- never a 1-GPU line for an N-GPU request
        sys.exit(launch_ranks(args.gpus))
    rank, world, local = parallel.init_from_env("cuda")
    if world != max(args.gpus, 1):
        print(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr, flush=True)
        sys.exit(2)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.load()
    torch.manual_seed(0)
    # how many ranks the collectives really span (an all-reduce of ones), for the record
    ranks_seen = 1
    if parallel.is_dist():
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        ranks_seen = int(ones.item())

    res = run_config(args, args.bound, args.background, rank, world, dev)
    dt, host, samples, probe, probe_fwd = res["dt"], res["host"], res["samples"], res["probe"], res["probe_fwd"]
    probed, fwd_symbol, bytes_per_sample, in_sync, psnr = (res["probed"], res["fwd_symbol"], res["bytes_per_sample"],
                                                           res["in_sync"], res["psnr"])
    trainer, fused, untrained_cells, overflow = res["trainer"], res["fused"], res["untrained_cells"], res["overflow"]
    collective_ms, psnr_step, scaler_state = res["collective_ms"], res["psnr_step"], res["scaler_state"]
    eval_check = res["eval_check"]
    # what the JSON line needs from the primary run's trainer (it is freed before the secondary runs)
    tinfo = {"fuse_adam": bool(fused and getattr(trainer, "fuse_adam", False)),
             "table_numel": int(trainer.table.numel()) if fused else 0,
             "graph": bool(fused and trainer.use_graph), "prefetch": bool(fused and trainer.prefetch),
             "device_sampler": bool(fused and trainer.device_sampler), "cap": trainer.cap if fused else 0,
             "step": psnr_step,
             "dp_exchange": trainer.xchg.carrier if fused and getattr(trainer, "xchg", None) is not None else None,
             "dp_split_level": trainer.split["level"] if fused and getattr(trainer, "split", None) is not None else None}
    if fused:
        trainer.close()
    del res, trainer
    # the reference's own defaults as secondary rows (main.py:31,46: --bound 2, black background): same schedule and
    # timing protocol, no probes.  The headline stays the benchmark framing (bound 1, random background).
    secondary = {}
    if not args.no_secondary:
        for name, (b2, bg2) in {"black_background": (args.bound, "black"), "bound2_black": (2.0, "black")}.items():
            if (b2, bg2) == (args.bound, args.background):
                continue
            torch.cuda.empty_cache()
            r2 = run_config(args, b2, bg2, rank, world, dev, probe_on=False)
            secondary[name] = {"value": round(world * args.rays * args.steps / r2["dt"], 1), "unit": "rays/s",
                               "ms_per_step": round(r2["dt"] / args.steps * 1e3, 4), "bound": b2, "background": bg2,
                               "samples_per_step": round(r2["samples"] / max(args.steps, 1)),
                               "host_enqueue_ms_per_step": round(r2["host"] / max(args.steps, 1) * 1e3, 4),
                               "psnr": None if r2["psnr"] is None else round(float(r2["psnr"]), 3),
                               "arena_overflow": r2["overflow"], "untrained_cells": r2["untrained_cells"]}
            if r2["fused"]:
                r2["trainer"].close()
            del r2
        if world == 1 and not (args.autograd or args.torch_mlp):
            torch.cuda.empty_cache()
            secondary["config4"] = run_config4(args, dev)
    if rank == 0:
        launches, units, ksec = probe
        roof = None
        # (exchange in two level groups: the probed entry point is then the fill alone -- the reduce runs as one launch per
        # group, each feeding its own collective -- so no table-backward roofline is formed from it)
        if launches and tinfo["dp_split_level"] is None:
            # one GPU: the same launch also runs Adam on the table (inside the reduce kernel): its algorithmic bytes are
            # the read + write of parameter, exp_avg and exp_avg_sq -- 24 B per table entry, the gradient never
            # reaches HBM (SURVEY 8d prices a separate optimiser pass at 28 B)
            fused_adam = tinfo["fuse_adam"] and args.roofline_kernel == "ngp_x_grid_backward_binned"
            opt_bytes = 24 * tinfo["table_numel"] if fused_adam else 0
            grid_only = units * bytes_per_sample / ksec / 1e9
            ach = (units * bytes_per_sample + launches * opt_bytes) / ksec / 1e9
            per_sample = PMC_TRAFFIC_BYTES_PER_SAMPLE.get(args.roofline_kernel)
            if fused_adam:
                per_sample = PMC_FUSED_TRAFFIC_BYTES_PER_SAMPLE
            traffic = round(per_sample * units / launches + opt_bytes) if per_sample else None
            roof = {"bound": "hbm", "kernel": args.roofline_kernel + (" + Adam on the table (fused)" if fused_adam else ""),
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_source": ("FETCH_SIZE + WRITE_SIZE measured on this script's steady state (profiles/r04_pmc_bench_"
                                       "traffic.csv), per-sample part scaled to this run's samples"
                                       + (", plus the optimiser's 24 B per table entry" if fused_adam else "")),
                    "launches": launches, "timed": f"probe tail after the timed region and the PSNR evaluation, one step in "
                                                   f"{args.probe_every}", "avg_us": round(ksec / launches * 1e6, 2),
                    "bytes_per_sample": bytes_per_sample, "samples_per_launch": round(units / launches),
                    "optimizer_bytes_per_launch": opt_bytes, "achieved_grid_only": round(grid_only, 1),
                    "frac_grid_only": round(grid_only / HBM_PEAK_GBPS, 4)}
        roof_fwd = None
        if fused and probe_fwd[0]:
            per_launch = probe_fwd[1] / probe_fwd[0]
            ach = per_launch * FWD_BYTES_PER_SAMPLE * probe_fwd[0] / probe_fwd[2] / 1e9
            roof_fwd = {"bound": "hbm", "kernel": fwd_symbol, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                        "line_rate_ceiling": FWD_LINE_RATE_CEILING_GBPS,
                        "frac_of_line_rate": round(ach / FWD_LINE_RATE_CEILING_GBPS, 4),
                        "traffic": round(PMC_FWD_TRAFFIC_BYTES_PER_SAMPLE * per_launch),
                        "launches": probe_fwd[0], "avg_us": round(probe_fwd[2] / probe_fwd[0] * 1e6, 2),
                        "bytes_per_sample": FWD_BYTES_PER_SAMPLE, "samples_per_launch": round(per_launch)}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(Options(bound=1.0), args.cpu_rays, args.cpu_steps)
            cpu["value"] = round(cpu["value"], 1)
        line = {
            "metric": METRIC, "value": round(world * args.rays * args.steps / dt, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.torch_mlp else "f32 (grid/SH/march/composite) + f16-MFMA/f32-acc MLP", "data": "synthetic",
            "config": {"bound": args.bound, "background": args.background, "workload": "configs[1]: Lego-style 800x800 procedural scene, hashgrid L=16 F=2 T=2^19, "
                                   "density-grid march (cuda_ray path), 4096 rays/batch/GPU, "
                                   + ("fp32 nn.Linear MLPs" if args.torch_mlp else "fused tiny-MLP (configs[2])"),
                       "rays_per_step_per_gpu": args.rays, "samples_per_step": round(samples / max(args.steps, 1)),
                       "views": args.views, "resolution": args.res, "burnin_steps": args.burnin,
                       "parallelism": f"dp{world}", "grad_wire": args.grad_wire if (world > 1 or args.dp_rehearsal) else None,
                       "dp_mode": args.dp_mode if (world > 1 or args.dp_rehearsal) else None,
                       "dp_exchange": tinfo["dp_exchange"], "dp_split_level": tinfo["dp_split_level"], "ranks_seen": ranks_seen,
                       "collective_ms_per_step": None if collective_ms is None else round(collective_ms, 4),
                       "multi_gpu_measured": "RCCL over >1 rank has not been measured by the builder (no multi-GPU box in reach): "
                                             "this line is the first measurement" if world > 1 else None,
                       "replicas_in_sync": in_sync, "step": "fused" if fused else "autograd",
                       "graph": tinfo["graph"], "group_steps": args.group_steps, "prefetch": tinfo["prefetch"],
                       "device_sampler": tinfo["device_sampler"],
                       "host_enqueue_ms_per_step": round(host / max(args.steps, 1) * 1e3, 4),
                       "untrained_cells": untrained_cells, "arena_capacity": tinfo["cap"], "arena_overflow": bool(fused and overflow),
                       # torch.cuda.amp.GradScaler's rule on the device (train_utils.py:404,897-904): state at the PSNR iteration
                       "loss_scaler": scaler_state},
            "roofline": roof, "roofline_forward": roof_fwd, "cpu_baseline": cpu,
            "secondary": secondary or None,
        }
        if psnr is not None:
            line["psnr"] = {"iters": tinfo["step"], "value": round(float(psnr), 3)}
            if eval_check is not None:      # distributed evaluation == one rank's evaluation of all views
                line["psnr"]["distributed_equals_local"] = eval_check
        print(json.dumps(line), flush=True)
    if parallel.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
