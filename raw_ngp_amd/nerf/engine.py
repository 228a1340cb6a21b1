"""Fused training step of the density-grid path: the reference's train_step + backward + optimiser
(nerf/train_utils.py:481-568, :890-907; nerf/renderer.py:515-556; main.py:245,261) and its density-grid refresh
(renderer.py:811-897) as a fixed sequence of HIP kernel launches on pre-allocated buffers, replayed from hipGraphs.

Compared with the per-op autograd path (raw_ngp_amd.nerf.trainer.Trainer, which mirrors the reference op by op)
the arithmetic is the same and the plumbing is not:
  * ray batches are drawn on the device (Philox) and marched ONCE into a sample arena by the chain-parallel march; the
    batch of step i+1 is prepared on a second stream while step i trains; nothing waits for a sample count on the host
    -- every kernel reads it from the arena's device counter
  * the hash-grid encoder writes the level-major slab the fused MFMA MLP reads (no permutes, no concatenations); with the
    global-bins record layout it also counts the records of the table backward (the default tile-local layout needs no counts)
  * compositing runs one wave per ray; forward, MSE loss, background mix and backward are ONE kernel
  * the table gradient is binned and reduced in LDS (64-bit fixed point); on one GPU Adam is applied inside that
    reduction (the gradient never reaches HBM), under data parallelism the gradient is all-reduced first
  * learning-rate schedule, loss, sample counters live on the device; autograd, GradScaler, zeros_like and the foreach
    optimiser are gone (their work is in the kernels)
  * the density-grid refresh draws its cells, evaluates, EMA-maxes and re-packs the bitfield without a host round trip
Main stream per step (one graph): encoder forward -> MLP forward (+ step_begin as a passenger) -> composite forward +
loss + backward -> MLP backward (2) -> dW reduction (+Adam on the MLP weights, + their entries in the f16 operand image)
-> fill -> reduce (+Adam on the table).  Same seed, same bits: no float atomic feeds back into the state.
"""
import math
import os

import numpy as np
import torch

from .. import parallel, raymarching
from . import utils
from .._lib import engine_backend as eb
from .._lib import gridencoder_backend as gb
from .._lib import mlp_backend as _mlp_plain
from .._lib import mlp_rf_backend as _mlp_rf
from .._lib import raymarching_backend as rb


def _level_cost(env, L):
    """Per-level tile costs for the slab encoder's level -> XCD placement: `env` = "c0,c1,..." (L positive numbers).  Unset
    (the default) or "0": the fixed pairing of levels, which measures best for both callers."""
    text = os.environ.get(env, "").strip()
    if text in ("", "0"):
        return None
    cost = [float(t) for t in text.split(",")]
    if len(cost) != L or min(cost) <= 0:
        raise ValueError(f"{env}: expected {L} positive numbers")
    return cost


class _Slot:
    """One ray batch and everything derived from it before the field is evaluated."""

    def __init__(self, N, max_steps, cap, dev, chain_cap=0, lit=False, indexed=False):
        f32 = dict(dtype=torch.float32, device=dev)
        self.arena = raymarching.MarchArena(N, max_steps, cap, dev, chain_cap=chain_cap, with_ldirs=lit)
        self.rays_ldir = torch.zeros(N, 3, **f32) if lit else None            # rfield: one light direction per ray
        self.index = torch.zeros(N, 2, dtype=torch.int32, device=dev) if indexed else None    # (view, pixel) of each ray
        self.exposure = torch.ones(N, **f32) if indexed else None
        self.live = torch.full((1,), N, dtype=torch.int32, device=dev)        # rays the batch really carries (adaptive)
        self.rays_o, self.rays_d = torch.empty(N, 3, **f32), torch.empty(N, 3, **f32)
        self.gt, self.bg = torch.empty(N, 4, **f32), torch.empty(N, 3, **f32)
        self.noises = torch.empty(N, **f32)
        self.nears, self.fars = torch.empty(N, **f32), torch.empty(N, **f32)
        self.step = -1                     # training step whose rays the slot holds
        self.head_step = -1                # ... or whose rays are drawn and half marched (in front of a grid refresh)


class FusedTrainer:
    """Drop-in for Trainer.train_step / train on models that satisfy NeRFNetwork._fused()."""

    def __init__(self, opt, model, dataset, device="cuda", seed=0, capacity=None, betas=(0.9, 0.999), eps=1e-15):
        assert opt.cuda_ray, "fused step: density-grid path only"
        # light-conditioned field (rfield: 47 -> 80 -> 80 -> 3 view MLP over [features, SH(view), SH(light)]), BARF pose
        # refinement (level window + se(3) corrections) and the HDR loss: BASELINE configs[3]
        self.rfield = bool(opt.rfield)
        self.pose = opt.pose_opt != "none"
        self.hdr = getattr(opt, "image_mode", "LDR") == "HDR"
        assert opt.pose_opt in ("none", "barf", "baangp"), f"fused step: unknown pose_opt {opt.pose_opt!r}"
        # BAA-NGP (network.py:77-97): masked levels are replaced by the finest active one -- a blend on the encoder slab in
        # front of the field kernels (and its adjoint behind them) instead of BARF's per-level factors inside them
        self.baa = opt.pose_opt == "baangp"
        assert not (getattr(opt, "adaptive_num_rays", False) and getattr(opt, "loss_weight", "none") != "none"), \
            "fused step: adaptive ray batches and a loss weight are not combined"
        assert getattr(opt, "loss_weight", "none") in ("none", "planck", "gaussian", "hanning"), opt.loss_weight
        # the orientation term (renderer.py:558-571, train_utils.py:546-548): per sample, from d sigma / d xyz -- one more
        # pass through the density network and the encoder's Jacobian slab in front of the compositor.  (lambda_distort
        # needs no code: only the renderer WITHOUT the density grid returns a distort_loss, renderer.py:504-505 -- on this
        # path the reference's train_step never sees one, train_utils.py:550.)
        self.orient = float(getattr(opt, "lambda_orientation", 0.0)) > 0
        # (with pose refinement the term also reaches the cameras through the view directions -- ray_gradients takes that
        # part -- and the level window's adjoint enters d sigma / d xyz)
        assert opt.loss_scale > 0, "fused step: the f16 deltas of the MLP backward need a positive loss scale"
        self.opt, self.model, self.data, self.device = opt, model.to(device), dataset, torch.device(device)
        opt.fused_mlp = True
        assert model._fused(), "fused step needs the default field configuration"
        self.mb = mb = _mlp_rf if self.rfield else _mlp_plain
        # the field's output activations (network.py:115,131-135); None = the defaults (trunc_exp density, clamped_exp colour)
        from .._lib import field_activations, _default_act
        act = field_activations(opt)
        self.act = None if _default_act(act) else act
        assert self.act is None or act[3] == 0 or not float(getattr(opt, "lambda_orientation", 0.0)) > 0, \
            "fused step: softplus hidden layers are built without the orientation term (its density-gradient pass is ReLU)"
        self.rank, self.world_size = parallel.rank(), parallel.world_size()
        # data-parallel step (separate Adam pass, gradient collectives); `dp_rehearsal` runs it on one rank as well
        self.dp = self.world_size > 1 or (bool(getattr(opt, "dp_rehearsal", False)) and parallel.is_dist())
        # adaptive ray batches (train_utils.py:563-564, part of the reference's -O preset): the ray SLOTS are fixed
        # (max_ray_batch of them), how many carry rays is decided on the device from the previous batch's sample count
        self.adaptive = bool(getattr(opt, "adaptive_num_rays", False))
        self.N = N = max(opt.max_ray_batch, opt.num_rays) if self.adaptive else opt.num_rays
        # sample arena: ~145 samples/ray are needed while the occupancy grid is still full at bound 1; rays longer with the bound
        self.cap = cap = int(capacity or max(opt.arena_capacity, opt.num_rays * 160 * int(math.ceil(model.real_bound)),
                                             2 * opt.num_points if self.adaptive else 0))
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        enc = model.grid_encoder
        self.L = enc.num_levels
        self.S = float(np.log2(enc.per_level_scale))
        self.H = enc.base_resolution
        self.table = enc.embeddings.data
        self.rows = self.table.shape[0]
        layers = list(model.grid_mlp.net) + list(model.view_mlp.net)
        sizes = [l.weight.numel() for l in layers]
        n_t, n_w = self.table.numel(), sum(sizes)
        plain = opt.lambda_tv == 0 and opt.lambda_wd == 0
        # Adam on the hash table fused into the table-gradient reduction: one rank, nothing else touching the gradient
        self.fuse_adam = bool(getattr(opt, "fuse_adam", True)) and not self.dp and plain
        # Everything else -- data parallel, or a separate optimiser pass on one GPU -- is the EXCHANGE step: the reduce kernel
        # overwrites a flat gradient buffer, collectives average it (none on one rank), ONE Adam launch updates what this
        # rank owns, collectives publish it (parallel.Exchange; `dp_exchange` picks the carrier).  "shard": reduce_scatter ->
        # Adam on this rank's 1/R of the flat parameter, moments for that shard only -> all_gather (variant 2);
        # "allreduce": gradient all-reduce + Adam over everything on every rank.  TV / weight decay need the whole gradient.
        self.dp_mode = (getattr(opt, "dp_mode", "shard") if plain else "allreduce") if not self.fuse_adam else None
        # 16-bit wire format of the table gradient: the reduce kernel stores bfloat16, RCCL averages it in place, Adam
        # reads it -- no conversion passes, half the bytes on xGMI (the MLP gradients then travel on their own, in f32)
        self.wire16 = self.dp_mode is not None and getattr(opt, "grad_wire", "f32") == "bf16" and plain
        self.xchg, self.flat, self.gflat, self.split, self.comm = None, None, None, None, None
        self.collective_events = None           # bench.py: [(start, stop)] HIP events around the collectives of timed steps
        self.collective_steps = 0               # ... and how many steps they cover
        if self.dp_mode is not None:
            want = getattr(opt, "dp_exchange", None) if self.dp else "none"
            direct = self.dp and self.world_size > 1 and dev.type == "cuda" and want in (None, "rccl") and \
                parallel.is_dist() and torch.distributed.get_backend() == "nccl"
            if direct:
                # bare RCCL calls, captured inside the step graphs -- after known answers (eagerly and replayed from a graph,
                # agreed on by all ranks, under a deadline: parallel.guarded_rccl_exchange); else torch.distributed's calls
                self.xchg = parallel.guarded_rccl_exchange(dev)
                if self.xchg is None:
                    if self.rank == 0:
                        print("[raw_ngp_amd] direct RCCL exchange failed its self-test: falling back to torch.distributed", flush=True)
                    self.xchg = parallel.Exchange(dev, carrier="torch")
            else:
                self.xchg = parallel.Exchange(dev, carrier=want)
            # f32 wire: table and MLP weights share ONE flat parameter (and one flat gradient): one collective each way
            n_flat = parallel.padded_numel(n_t + (0 if self.wire16 else n_w))
            # The exchange in TWO LEVEL GROUPS ("shard", f32 wire, tile-local records): levels [0, a) and [a, L) + MLP weights.
            # The table's levels are independent, so the reduce-scatter of group A can run on a second stream while the table
            # backward still reduces group B, and the all-gather of group B while the next step's encoder already works on
            # group A's levels -- communication the one-group exchange leaves exposed between two steps.  Each
            # group is sharded over the ranks on its own; group A's tail that does not fill a multiple of 4 R floats (level
            # boundaries are multiples of 16 floats: 16 floats at R = 8, nothing at R <= 4) is the SEAM: all-reduced, and
            # stepped by every rank with replicated moments.
            self.split = None
            # (default: with more than one rank; on one rank -- rehearsal, --no-fuse-adam -- there is no communication to hide and
            # the two-launch forms of encoder, reduce and Adam cost ~ 25 us: 0.352 against 0.325 ms/step)
            a = getattr(opt, "dp_split_level", None)
            a = int(a) if a is not None else (8 if self.world_size > 1 else 0)
            if self.dp_mode == "shard" and not self.wire16 and 0 < a < self.L and dev.type == "cuda" \
                    and os.environ.get("NGP_DP_SPLIT", "1") != "0" \
                    and not gb.backward_needs_counts(cap, self.L, enc.offsets):
                offs = enc.offsets.detach().cpu().tolist()
                q = 4 * self.xchg.R
                bA = 2 * int(offs[a])                       # floats of the levels in front of the split
                sA = bA // q * q                            # ... of them sharded; [sA, bA) is the seam
                n_flat = bA + parallel.padded_numel(n_t + n_w - bA)
                self.split = dict(level=a, bA=bA, sA=sA, chunks_a=gb.level_chunks(enc.offsets, 0, a),
                                  chunks_b=gb.level_chunks(enc.offsets, a, self.L))
            self.flat = torch.zeros(n_flat, **f32)
            self.flat[:n_t].copy_(self.table.reshape(-1))
            self.table = self.flat[:n_t].view(self.rows, 2)
            enc.embeddings.data = self.table                # the module's parameter lives in the padded buffer
            self.gflat = torch.zeros(n_flat, dtype=torch.bfloat16 if self.wire16 else torch.float32, device=dev)
        # the six MLP matrices become views of one flat buffer (one Adam launch, one all-reduce)
        if self.flat is not None and not self.wire16:
            self.w_flat, self.w_grad = self.flat[n_t:n_t + n_w], self.gflat[n_t:n_t + n_w]
        else:
            self.w_flat = torch.empty(n_w, **f32)
            self.w_grad = torch.zeros(n_w, **f32)
        self.weights, self.dws, off = [], [], 0
        for l, n in zip(layers, sizes):
            view = self.w_flat[off:off + n].view_as(l.weight)
            view.copy_(l.weight.data)
            l.weight.data = view
            self.weights.append(view)
            self.dws.append(self.w_grad[off:off + n].view_as(l.weight))
            off += n
        # optimiser state
        self.betas, self.eps, self.lr0 = betas, eps, opt.lr
        self._wire = None
        if self.xchg is None:
            self.t_m, self.t_v = torch.zeros_like(self.table), torch.zeros_like(self.table)
            self.w_m, self.w_v = torch.zeros_like(self.w_flat), torch.zeros_like(self.w_flat)
            self.table_grad = torch.zeros_like(self.table)
        elif self.split is not None:
            # moments of what this rank updates: [its shard of group A | the seam (replicated) | its shard of group B]
            sp, R = self.split, self.xchg.R
            nA, nS, nB = sp["sA"] // R, sp["bA"] - sp["sA"], (self.flat.numel() - sp["bA"]) // R
            self.t_m, self.t_v = torch.zeros(nA + nS + nB, **f32), torch.zeros(nA + nS + nB, **f32)
            sp.update(nA=nA, nS=nS, nB=nB)
            self.w_m = self.w_v = None
            self.table_grad = self.gflat[:n_t].view(self.rows, 2)
            # the collectives of the two groups live on a stream of their own when they can be captured with the step
            self.comm = torch.cuda.Stream(device=dev) if self.xchg.capturable else None
            self._ev = {k: torch.cuda.Event() for k in ("rsA", "rsB", "agA", "agB")}
        else:
            own = self.xchg.shard_of(self.flat) if self.dp_mode == "shard" else self.flat
            self.t_m, self.t_v = torch.zeros_like(own), torch.zeros_like(own)      # moments of what this rank updates
            if self.wire16:
                self.w_m, self.w_v = torch.zeros_like(self.w_flat), torch.zeros_like(self.w_flat)
                self._wire = self.gflat[:n_t].view(self.rows, 2)
                self.table_grad = None
            else:
                self.w_m = self.w_v = None                                           # (inside t_m / t_v)
                self.table_grad = self.gflat[:n_t].view(self.rows, 2)
        # per-ray and per-sample buffers
        # two ray-batch slots: while step i trains out of one, step i+1's rays are drawn and marched into the
        # other on a second stream (the march is a long, narrow kernel -- 64 waves -- that hides under backward)
        # (pose refinement: the rays of step i + 1 are cast from the poses step i has just updated -- nothing to draw ahead)
        self.prefetch = bool(getattr(opt, "prefetch_march", True)) and dev.type == "cuda" and not self.pose
        # march pass 1: "chain" (all candidate parameters classified in parallel), "index" (serial loop, occupancy
        # index in LDS) or "serial" (serial loop on the bitfield)
        self.march_mode = getattr(opt, "march_mode", "chain")
        chain_cap = opt.max_steps * int(math.ceil(model.real_bound)) + 2 if self.march_mode == "chain" else 0
        if chain_cap >= 65536:
            self.march_mode, chain_cap = "index", 0
        self._slot_kw = dict(lit=self.rfield, indexed=self.pose or self.hdr)
        self.slots = [_Slot(N, opt.max_steps, cap, dev, chain_cap, **self._slot_kw) for _ in range(2 if self.prefetch else 1)]
        self.arena = self.slots[0].arena
        if self.adaptive:
            assert bool(getattr(opt, "device_sampler", True)), "adaptive ray batches are drawn by the device sampler"
            for sl in self.slots:       # "the previous batch": num_rays rays that produced exactly num_points samples
                sl.live.fill_(opt.num_rays)
                sl.arena.counter[0] = opt.num_points
            self.rays_seen = torch.zeros(1, dtype=torch.int64, device=dev)
        self.side = torch.cuda.Stream(device=dev) if self.prefetch else None
        self.aux = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        # compressed occupancy bitfield the march keeps in LDS (rebuilt after every density-grid refresh)
        self.occ_index = None
        # (chain mode with dt_gamma == 0: the constant-step march kernel stages the same index in LDS when it fits)
        indexed = self.march_mode == "index" or (self.march_mode == "chain" and opt.dt_gamma == 0 and dev.type == "cuda"
                                                 and os.environ.get("NGP_MARCH_INDEX", "1") != "0")
        if indexed and (model.cascade * model.grid_size ** 3) % 2048 == 0:
            self.occ_index = torch.zeros(rb.occupancy_index_bytes(model.cascade, model.grid_size) // 4,
                                         dtype=torch.int32, device=dev)
        self._occ_version = None
        # ray batches drawn on the device (one kernel, counter-based RNG) when the dataset keeps uint8 images there
        imgs = getattr(dataset, "images", None)
        self.device_sampler = bool(getattr(opt, "device_sampler", True)) and torch.is_tensor(imgs) \
            and imgs.dtype == torch.uint8 and imgs.is_cuda and imgs.dim() == 4
        assert self.device_sampler or not (self.pose or self.hdr), "fused pose / HDR step: needs the device-side ray sampler"
        self.view_ldirs = None
        if self.rfield:
            ld = getattr(dataset, "ldirs", None)
            assert ld is not None, "rfield: the dataset must carry one light direction per view (dataset.ldirs [V,3])"
            self.view_ldirs = torch.as_tensor(ld, dtype=torch.float32, device=dev).contiguous()
        self.view_exposure = None
        if self.hdr:
            ex = getattr(dataset, "exposures", None)
            assert ex is not None, "HDR loss: the dataset must carry one exposure value per view (dataset.exposures [V])"
            self.view_exposure = torch.as_tensor(ex, dtype=torch.float32, device=dev).contiguous()
        self.seed64 = (seed * 1000 + self.rank) & (2 ** 64 - 1)
        i32 = dict(dtype=torch.int32, device=dev)
        self.draw_ctr, self.step_ctr = torch.zeros(1, **i32), torch.zeros(1, **i32)
        # torch.cuda.amp.GradScaler (train_utils.py:404,897-904) as eight device words: the MLP backward reads the scale,
        # an overflowing f16 delta makes the step's weight gradients non-finite, every optimiser kernel then leaves its
        # parameters alone and the next step's step_begin halves the scale -- inside the step graphs, no host read
        from .._lib import LossScaler
        self.scaler = LossScaler(dev, init_scale=opt.loss_scale, growth_interval=getattr(opt, "scale_growth_interval", 2000)) \
            if getattr(opt, "dynamic_loss_scale", True) and dev.type == "cuda" else None
        self.hyper = torch.zeros(4, **f32)                  # {lr, 1 - b1^t, 1/sqrt(1 - b2^t)} of the current step
        # the main stream's part of a step replayed from captured hipGraphs (one per ray slot)
        self.use_graph = bool(getattr(opt, "capture_graph", True)) and opt.lambda_tv == 0 and dev.type == "cuda"
        # run merging in the binned backward pays while consecutive samples share cells: res * (step in [0,1]) < ~0.7
        step01 = (2 * math.sqrt(3) / opt.max_steps) / (2 * model.bound)
        self.merge_max_res = int(min(1024, max(16, 0.7 / step01)))
        self.graphs, self.graph_pool, self.last_graph_key, self._graphs_alive = {}, None, None, []
        self._refresh_graph = {}                       # steady-state refresh as graphs: 0 whole, 1 cell draw, 2 the rest
        self._refresh_head_step = -1                   # step whose refresh already has its cells drawn
        self._eval_slot = None
        self._main_symbols = {"ngp_x_mlp_rf_forward", "ngp_x_mlp_rf_backward", "ngp_x_mlp_rf_prepare",
                              "ngp_x_grid_encode_forward_slab_jac", "ngp_x_composite_hdr_train",
                              "ngp_x_grid_backward_binned_apply", "ngp_x_grid_backward_binned_apply_mlp",
                              "ngp_x_grid_encode_forward_slab", "ngp_x_mlp_forward", "ngp_x_mlp_forward_step_begin",
                              "ngp_x_mlp_backward", "ngp_x_composite_rays_train_forward",
                              "ngp_x_composite_mse_backward", "ngp_x_composite_mse_train", "ngp_x_adam_step_dev2", "ngp_x_adam_step_dev",
                              "ngp_x_step_begin", "ngp_x_mlp_prepare", "ngp_x_mlp_reduce_dw",
                              "ngp_x_grid_backward_binned_prepare"}
        # density-grid refresh on the device (no host round trips)
        self.native_refresh = bool(getattr(opt, "native_grid_refresh", True)) and model.grid_size ** 3 % 64 == 0
        if self.native_refresh:
            cells = model.grid_size ** 3
            self.dg_indices = torch.empty(cells, **i32)
            self.dg_xyzs, self.dg_sigma = torch.empty(cells, 3, **f32), torch.empty(cells, **f32)
            self.dg_tmp = torch.full_like(model.density_grid, -1.0)
            self.dg_stats = torch.zeros(4 + 1024, **f32)
            self.dg_draw = torch.zeros(1, **i32)
            self.dg_ws = torch.empty(eb.density_grid_workspace_bytes(model.grid_size), dtype=torch.uint8, device=dev)
            self.dg_seed = ((seed * 1000) ^ 0x9E3779B97F4A7C15) & (2 ** 64 - 1)     # same on every rank
        # BARF: per-level weights of the current step, {pose step?, step} flags; the se(3) corrections and their Adam state
        self.level_w = torch.ones(self.L, **f32) if self.pose else None
        self.flags = torch.zeros(2, **i32) if self.pose else None
        self.pose_optimizer = None
        if self.pose:
            from .pose import CameraOptimizer, compose
            V = len(dataset)
            self.pose_optimizer = co = CameraOptimizer(V, dev, opt, seed=seed)
            self.xi = co.se3_refine.weight.data                          # [V,6], updated in place by the device kernel
            with torch.no_grad():                                         # what the corrections refine (pose.py: forward)
                base = dataset.poses[:, :3, :].float()
                if co.pose_noise is not None:
                    base = compose([co.pose_noise, base])
                if getattr(opt, "identity", False):
                    base = torch.eye(4, device=dev)[None, :3, :4].expand(V, 3, 4)
            self.pose_base = base.reshape(V, 12).contiguous()
            self.poses_refined = torch.zeros(V, 4, 4, **f32)
            self.pose_m, self.pose_v = torch.zeros(V, 6, **f32), torch.zeros(V, 6, **f32)
            self.grad_pose = torch.zeros(V, 12, **f32)
            self.g_rays_o, self.g_rays_d = torch.zeros(N, 3, **f32), torch.zeros(N, 3, **f32)
            self.pose_lr0 = float(getattr(opt, "c_lr", 1e-3))
            self.pose_gamma = 1e-2 ** (1.0 / opt.iters)                   # ExponentialLR of camera_optimizers.py:44-50
            eb.pose_update(self.xi, self.pose_base, None, None, None, None, self.pose_lr0, self.pose_gamma, 0.9, 0.999, 1e-8,
                           self.poses_refined)
            self.ddirs = torch.empty(cap, 3, **f32)
            # before the first step the reference's annealing value is 0.0 (train_utils.py:411): the first density-grid
            # refresh sees that window (level 0 only)
            eb.step_window(self.step_ctr, 0, float(opt.iters), opt.start_annealing, opt.end_annealing, self.L, self.level_w,
                           self.flags, baa=self.baa)
        if self.pose or self.orient:                   # d enc / d x01, written by the encoder's forward
            self.dydx = torch.empty(self.L, cap, 3, 2, **f32)
        if self.orient:
            self.orient_term = torch.zeros(cap, **f32)
            if self.pose:       # d term / d dirs per sample, and lambda * weights from the compositor step
                self.orient_ddirs, self.orient_weight = torch.zeros(cap, 3, **f32), torch.zeros(cap, **f32)
        self.enc = torch.empty(self.L, cap, 2, **f32)
        self.denc = torch.empty(self.L, cap, 2, **f32)
        self.x01 = torch.empty(cap, 3, **f32)
        self.sigma, self.rgb = torch.empty(cap, **f32), torch.empty(cap, 3, **f32)
        self.dsigma, self.drgb = torch.empty(cap, **f32), torch.empty(cap, 3, **f32)
        # the samples in front of the compositor's early stop, as a list the backward kernels run over (the others -- a third
        # of the batch late in training -- have exactly zero gradients): plain field, MSE loss, tile-local table backward
        self.live_n = torch.zeros(N, dtype=torch.int32, device=dev)
        self.live_idx = torch.zeros(cap, dtype=torch.int32, device=dev)
        self.live_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.live_off = torch.zeros(N, dtype=torch.int32, device=dev)    # where each ray's entries start (ray gradients)
        self.live_list = dev.type == "cuda" and os.environ.get("NGP_LIVE_LIST", "1") != "0"   # (and tile-local records: below)
        # level -> XCD placement of the slab encoder by per-level costs, per caller (ray-ordered samples in the step,
        # scattered cell draws in the refresh): an experiment's knob, None = the fixed pairing
        self.level_cost_step = _level_cost("NGP_LEVEL_COST_STEP", self.L)
        self.level_cost_refresh = _level_cost("NGP_LEVEL_COST_REFRESH", self.L)
        self.weights_buf = torch.empty(cap, **f32)
        self.ws, self.depth, self.image = torch.empty(N, **f32), torch.empty(N, **f32), torch.empty(N, 3, **f32)
        self.loss = torch.zeros(1, **f32)
        self.mlp_image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device=dev)
        self.ws_mlp = torch.empty(mb.backward_workspace_bytes(cap), dtype=torch.uint8, device=dev)
        for slot in self.slots:         # the binned backward's bookkeeping is per ray batch (prepared with the march)
            slot.ws_grid = torch.empty(gb.backward_workspace_bytes(cap, self.L, self.rows), dtype=torch.uint8, device=dev)
        # record layout of the binned table backward at this capacity: does it want the encoder's forward to count records?
        self.binned_counts = gb.backward_needs_counts(cap, self.L, model.grid_encoder.offsets)
        self._image_ready = False                      # the step path expects the f16 weight image of the current weights
        self.global_step = 0
        self._groups_precaptured = False
        # the march's first kernel does not read the occupancy bitfield: in front of a density-grid refresh the next batch
        # can still be drawn and that kernel run (chain-parallel march only)
        self._split_march = self.prefetch and self.slots[0].arena.chain is not None and \
            os.environ.get("NGP_SPLIT_MARCH", "1") != "0"
        self.samples_seen = torch.zeros(1, dtype=torch.int64, device=dev)     # running total, never read per step
        self.ray_gen = torch.Generator(device=dev).manual_seed(seed * 1000 + self.rank)   # torch sampling path
        self.last_loss = None
        if self.world_size > 1:
            parallel.broadcast_module(self.model)

    def close(self):
        """Give back what outlives the object otherwise: the RCCL communicator of the exchange step."""
        if self.xchg is not None:
            self.xchg.close()

    # ------------------------------------------------------------------ pieces
    def lr(self):
        return self.lr0 * 0.1 ** min(self.global_step / self.opt.iters, 1)

    def _adaptive_args(self, slot):
        """(previous batch's sample count, its ray count, this batch's ray count, target) for the device sampler."""
        if not self.adaptive:
            return None
        prev = self.slots[(self.slots.index(slot) + 1) % len(self.slots)] if slot in self.slots else slot
        return prev.arena.counter, prev.live, slot.live, self.opt.num_points

    def _mlp_prepare(self):
        self.mb.prepare(self.weights, self.mlp_image)

    def _mlp_forward(self, stride, dirs, ldirs, cnt, M, sigma, rgb, step_begin=None):
        """Field evaluation on self.enc (rgb None: density only).  rfield: light directions + the level window.
        step_begin: the step's scalar bookkeeping rides along with this launch (plain field only)."""
        if self.rfield:
            if self.baa:            # f'_l = w_l f_l + (1 - w_l) f_c on the slab, in place
                eb.slab_window(self.enc, stride, self.L, self.level_w, cnt, M)
            self.mb.forward(self.enc, stride, dirs, ldirs, None if self.baa else self.level_w, cnt, M, self.mlp_image, sigma, rgb,
                            act=self.act)
        else:
            if self.pose:           # the plain field kernels carry no level window: BARF scale / BAA blend on the slab
                eb.slab_window(self.enc, stride, self.L, self.level_w, cnt, M, scale_only=not self.baa)
            self.mb.forward(self.enc, stride, dirs, cnt, M, self.mlp_image, sigma, rgb, step_begin=step_begin, act=self.act)

    def march(self, slot, rays_o, rays_d, noises, aabb=None, plan=True, stage=0):
        """rays -> sample arena of `slot` (near/far, count, scan, expand); runs on the current stream.
        stage 1: only what does not read the occupancy bitfield (near/far, the chain kernel); 2: the rest."""
        opt, m, ar, N = self.opt, self.model, slot.arena, self.N
        if stage != 1:
            self._sync_occ_index()
        if stage != 2:
            eb.near_far_from_aabb_v2(rays_o, rays_d, m.aabb_train if aabb is None else aabb, N, m.min_near, slot.nears,
                                     slot.fars)
        rb.march_rays_train_arena(rays_o, rays_d, slot.rays_ldir, m.density_bitfield, m.real_bound, opt.contract,
                                  opt.dt_gamma, opt.max_steps, N, m.cascade, m.grid_size, slot.nears, slot.fars, noises,
                                  ar.t_scratch, self.cap, ar.xyzs, ar.dirs, ar.ts, ar.ldirs, ar.rays, ar.counter, None,
                                  self.occ_index, ar.chain, stage=stage)
        if not plan or stage == 1:
            return
        # reset the bookkeeping of the binned table backward for this batch (stage 1: plan); the encoder's forward
        # pass counts the records per chunk while it has the rows in registers, a scan (stage 2) follows it
        gb.grid_backward_binned_prepare(None, 0.0, m.grid_encoder.offsets, self.rows, ar.counter, self.cap, self.L,
                                        self.L, self.S, self.H, slot.ws_grid, merge_max_res=self.merge_max_res, stage=1)

    def forward_backward(self, rays_o, rays_d, gt_rgba, noises, bg_rgb=None, bg_const=0.0):
        """march -> encode -> MLP -> composite -> loss -> backward into self.table_grad / self.w_grad."""
        slot = self.slots[0]
        self.march(slot, rays_o, rays_d, noises)
        self.field_forward_backward(slot, gt_rgba, bg_rgb, bg_const)

    def field_forward_backward(self, slot, gt_rgba, bg_rgb=None, bg_const=0.0):
        # (the orientation term and the HDR loss live in the one-launch compositor step)
        for _, op in self._field_ops(slot, gt_rgba, bg_rgb, bg_const, fuse_composite=self.orient or self.hdr):
            op()

    def _field_ops(self, slot, gt_rgba, bg_rgb, bg_const, zero_loss=True, fused_adam=False, split_weights=False,
                   overwrite=False, fuse_composite=False, mlp_tail=None, split=None):
        """The field part of the step as (C entry point, thunk) pairs, in launch order.  split_weights: the step path --
        the f16 weight image was prepared at the end of the previous step and the weight-gradient reduction is left to
        the caller (it goes to the aux stream together with the MLP's Adam step)."""
        opt, m, ar, N, cap = self.opt, self.model, slot.arena, self.N, self.cap
        cnt, offsets = ar.counter, m.grid_encoder.offsets
        # single GPU: the table's Adam step happens inside the reduce kernel (the gradient never reaches HBM)
        adam = (self.table, self.t_m, self.t_v, self.hyper, *self.betas, self.eps) if fused_adam else None

        def loss_and_composite_backward():
            if zero_loss:
                self.loss.zero_()
            eb.composite_mse_backward(gt_rgba, bg_rgb, bg_const, self.sigma, self.rgb, ar.ts, ar.rays, self.ws,
                                      self.depth, self.image, cap, N, opt.T_thresh, self.dsigma, self.drgb, self.loss)

        def composite_train():      # forward + loss + backward of the compositor in one launch (the step path)
            if zero_loss:
                self.loss.zero_()
            lam = float(getattr(opt, "lambda_entropy", 0.0))
            exposure = slot.exposure if self.hdr else None      # exposure-scaled, clipped loss of train_utils.py:512-536
            weight = utils.hdr_loss_weight(opt.loss_weight, self._target(gt_rgba, bg_rgb, bg_const)) if self.hdr else None
            if weight is not None:
                weight = weight.contiguous()
            if self.orient or self.adaptive or lam > 0 or (live is not None and (self.hdr or live[3] is not None)):
                # the general entry: the loss over the rays the batch really carries (adaptive), the entropy of the accumulated
                # opacity, the term over the samples' weights, the HDR loss -- and the list with its per-ray offsets
                eb.composite_train_live(gt_rgba, bg_rgb, bg_const, exposure, weight, 1.0 / (3 * N),
                                        slot.live if self.adaptive else None, self.sigma, self.rgb, ar.ts, ar.rays, cap, N,
                                        opt.T_thresh, self.ws, self.depth, self.image, self.dsigma, self.drgb, self.loss,
                                        lambda_entropy=lam, live=live, sample_term=self.orient_term if self.orient else None,
                                        lambda_sample=float(opt.lambda_orientation) if self.orient else 0.0,
                                        term_weight=self.orient_weight if self.orient and self.pose else None)
                if self.adaptive:
                    self.rays_seen.add_(slot.live)
            elif self.hdr:
                eb.composite_hdr_train(gt_rgba, bg_rgb, bg_const, exposure, weight, 1.0 / (3 * N), self.sigma, self.rgb,
                                       ar.ts, ar.rays, cap, N, opt.T_thresh, self.ws, self.depth, self.image, self.dsigma,
                                       self.drgb, self.loss)
            else:
                eb.composite_mse_train(gt_rgba, bg_rgb, bg_const, self.sigma, self.rgb, ar.ts, ar.rays, cap, N, opt.T_thresh,
                                       self.ws, self.depth, self.image, self.dsigma, self.drgb, self.loss,
                                       live=live[:3] if live is not None else None)

        # the backward over the list of samples that can have a gradient (see __init__): the step path (one-launch compositor
        # step) with tile-local records
        live_list = self._lists_live_samples(fuse_composite)
        live = (self.live_n, self.live_idx, self.live_count, self.live_off if self.pose else None) if live_list else None
        back_n, back_idx = (self.live_count, self.live_idx) if live_list else (cnt, None)

        def orientation_term():     # min(0, n . -v)^2 per sample; d enc is free until the backward writes it
            if self.rfield:         # (its kernels apply BARF's window themselves; BAA-NGP's blend is on the slab already)
                self.mb.density_gradient(self.enc, cap, cnt, cap, self.mlp_image, self.denc,
                                         level_w=self.level_w if self.pose and not self.baa else None)
            else:
                self.mb.density_gradient(self.enc, cap, cnt, cap, self.mlp_image, self.denc)
            if self.pose and (self.baa or not self.rfield):     # the window's adjoint: d enc' -> d enc, as in the backward
                eb.slab_window(self.denc, cap, self.L, self.level_w, cnt, cap, backward=True,
                               scale_only=not self.baa and not self.rfield)
            eb.orientation_term(self.denc, self.dydx, cap, self.L, m.bound, self.sigma, ar.dirs, cnt, cap, self.orient_term,
                                dterm_ddirs=self.orient_ddirs if self.pose else None, act=self.act)

        def mlp_backward():
            if self.rfield:         # one call: both view kernels, the density kernel, the weight-gradient reduction
                self.mb.backward(self.enc, cap, ar.dirs, ar.ldirs, None if self.baa else self.level_w, self.dsigma, self.drgb,
                                 back_n, cap, self.mlp_image, opt.loss_scale, self.denc, self.ddirs if self.pose else None,
                                 self.dws, self.ws_mlp, sample_index=back_idx, scaler=self.scaler, act=self.act)
                if self.baa:        # the blend's adjoint: d enc' -> d enc (what the table backward and the ray gradients read)
                    eb.slab_window(self.denc, cap, self.L, self.level_w, back_n, cap, backward=True)
            else:
                self.mb.backward(self.enc, cap, ar.dirs, self.dsigma, self.drgb, back_n, cap, self.mlp_image, opt.loss_scale,
                                 self.denc, None if split_weights else self.dws, self.ws_mlp,
                                 ddirs=self.ddirs if self.pose else None, sample_index=back_idx, scaler=self.scaler,
                                 act=self.act)
                if self.pose:       # the window's adjoint: d enc' -> d enc
                    eb.slab_window(self.denc, cap, self.L, self.level_w, back_n, cap, backward=True, scale_only=not self.baa)

        ops = [
            ("ngp_x_grid_encode_forward_slab", lambda: eb.grid_encode_forward_slab(
                ar.xyzs, m.bound, self.table, offsets, self.enc, self.x01, cnt, cap, cap, self.L, self.L, self.S, self.H,
                binned_workspace=slot.ws_grid if self.binned_counts else None,
                dydx=self.dydx if self.pose or self.orient else None, level_cost=self.level_cost_step)),
            ("ngp_x_grid_backward_binned_prepare", lambda: gb.grid_backward_binned_prepare(
                None, 0.0, offsets, self.rows, cnt, cap, self.L, self.L, self.S, self.H, slot.ws_grid,
                single_segment=fused_adam or overwrite, stage=2)),
            ("ngp_x_mlp_prepare", self._mlp_prepare),
            ("ngp_x_mlp_forward", lambda: self._mlp_forward(cap, ar.dirs, ar.ldirs, cnt, cap, self.sigma, self.rgb)),
            *([("ngp_x_orientation_term", orientation_term)] if self.orient else []),
            ("ngp_x_composite_rays_train_forward", lambda: eb.composite_rays_train_forward(
                self.sigma, self.rgb, ar.ts, ar.rays, cap, N, opt.T_thresh, self.weights_buf, self.ws, self.depth,
                self.image)),
            ("ngp_x_composite_mse_backward", loss_and_composite_backward),
            ("ngp_x_mlp_backward", mlp_backward),
            ("ngp_x_grid_backward_binned_apply" + ("_mlp" if mlp_tail is not None else ""), lambda: gb.grid_backward_binned_apply(
                self.denc, self.x01, offsets, self._wire if overwrite and self.wire16 else self.table_grad, back_n, cap, cap,
                self.L, self.L, self.S, self.H, slot.ws_grid, adam=adam, overwrite=overwrite, mlp_tail=mlp_tail,
                sample_index=back_idx, scaler=self.scaler)),
        ]
        if split is not None:
            # the exchange in two level groups: the encoder runs group by group (each group's parameters arrive with their own
            # all-gather) and the table backward is a fill followed by one reduce per group (each feeds its own reduce-scatter)
            a, ca, cb = split["level"], split["chunks_a"], split["chunks_b"]
            jac = self.dydx if self.pose or self.orient else None

            def fwd(lo, hi):
                return lambda: eb.grid_encode_forward_slab_levels(ar.xyzs, m.bound, self.table, offsets, self.enc, self.x01, cnt,
                                                                  cap, cap, self.L, lo, hi, self.S, self.H, dydx=jac)

            def red(c):
                return lambda: gb.grid_backward_binned_reduce_range(offsets, self.table_grad, back_n, cap, self.L, self.S, self.H,
                                                                    slot.ws_grid, c[0], c[1], scaler=self.scaler)
            out = []
            for o in ops:
                if o[0] == "ngp_x_grid_encode_forward_slab":     # (group b first, on both ends of the step: see _split_step_ops)
                    out += [("fwd_b", fwd(a, self.L)), ("fwd_a", fwd(0, a))]
                elif o[0].startswith("ngp_x_grid_backward_binned_apply"):
                    out += [(o[0], lambda: gb.grid_backward_binned_apply(
                                self.denc, self.x01, offsets, None, back_n, cap, cap, self.L, self.L, self.S, self.H, slot.ws_grid,
                                mlp_tail=mlp_tail, sample_index=back_idx, scaler=self.scaler, n_rows=self.rows)),
                            ("reduce_b", red(cb)), ("reduce_a", red(ca))]
                else:
                    out.append(o)
            ops = out
        if split_weights:
            ops = [o for o in ops if o[0] != "ngp_x_mlp_prepare"]
        assert fuse_composite or not (self.hdr or self.orient), "the HDR loss / the orientation term live in the fused compositor step"
        if fuse_composite:
            ops = [o for o in ops if o[0] != "ngp_x_composite_rays_train_forward"]
            ops = [("ngp_x_composite_mse_train", composite_train) if o[0] == "ngp_x_composite_mse_backward" else o
                   for o in ops]
        return ops

    def _lists_live_samples(self, fuse_composite=True):
        """Does the step's backward run over the list of samples in front of the compositor's early stop?"""
        return bool(self.live_list and fuse_composite and not self.binned_counts)

    @staticmethod
    def _target(gt_rgba, bg_rgb, bg_const):
        """The target colour the loss compares with: the pixel over the step's background (train_utils.py:500-505)."""
        a = gt_rgba[:, 3:]
        return gt_rgba[:, :3] * a + (bg_rgb if bg_rgb is not None else bg_const) * (1 - a)

    @torch.no_grad()
    def refined_poses(self):
        """[V,3,4] camera-to-world matrices the rays are currently cast from (pose refinement)."""
        return self.poses_refined[:, :3, :].clone() if self.pose else self.data.poses[:, :3, :].clone()

    @staticmethod
    def _without(ops, name):
        return [o for o in ops if o[0] != name]

    @torch.no_grad()
    def refresh_density_grid(self, decay=0.95):
        """NeRFRenderer.update_extra_state (nerf/renderer.py:811-897) as ~10 launches per cascade with nothing read
        back: draw the cells (every cell for the first 16 calls, then H^3/4 uniform + H^3/4 occupied ones), evaluate
        the density there with the slab encoder + the density half of the fused MLP, EMA-max into the grid, re-pack
        the bitfield with thresh = min(mean, density_thresh)."""
        m = self.model
        full = m.iter_density < 16
        # (the cells may have been drawn already, on the side stream of the step in front of this one: refresh_head)
        part = 2 if self._refresh_head_step == self.global_step and not full else 0
        if self.use_graph and not full:             # the steady-state variant has fixed launch arguments: replay it
            if self._refresh_graph.get(part) is None:
                self._refresh_graph[part] = self._capture_ops([lambda: self._refresh_launches(decay, False, part)])
            for piece in self._refresh_graph[part]:
                piece()
        else:
            self._refresh_launches(decay, full)
        m.iter_density += 1
        m.bitfield_version = getattr(m, "bitfield_version", 0) + 1
        self._occ_version = m.bitfield_version      # (_refresh_launches rebuilt the index)

    def _refresh_head_ok(self):
        """May the cell draw of the next refresh (a third kernel chain that reads the density grid but no weights) run
        ahead, beside the step in front of it?  Steady state, one cascade (the draws of several share their buffers)."""
        m = self.model
        return bool(self._split_march and self.native_refresh and self.use_graph and m.cascade == 1 and m.iter_density >= 16)

    def refresh_head(self, for_step):
        """The weight-independent head of the refresh that step `for_step` will do: draw the cells (current stream)."""
        if self._refresh_graph.get(1) is None:
            self._refresh_graph[1] = self._capture_ops([lambda: self._refresh_launches(0.95, False, 1)])
        for piece in self._refresh_graph[1]:
            piece()
        self._refresh_head_step = for_step

    def _refresh_launches(self, decay, full, part=0):
        """part 1: only the cell draw (one cascade); part 2: everything after it; 0: all."""
        m, cap = self.model, self.cap
        H, cells = m.grid_size, m.grid_size ** 3
        n_uni, n_occ = (cells, 0) if full else (cells // 4, cells // 4)
        total = n_uni + n_occ
        offsets = m.grid_encoder.offsets
        if part != 1 and not self._image_ready:     # (the step keeps the f16 operand image of the current weights up to date)
            self._mlp_prepare()
        for cas in range(m.cascade):
            bound = min(2 ** cas, m.bound)
            half = bound / H
            if part != 2:
                eb.density_grid_sample(m.density_grid[cas], H, bound - half, half, n_uni, n_occ, full, self.dg_seed,
                                       self.dg_draw, self.dg_ws, self.dg_indices[:total], self.dg_xyzs[:total])
                eb.counter_add(self.dg_draw, 1)
            if part == 1:
                return
            for s in range(0, total, cap):
                k = min(cap, total - s)
                eb.grid_encode_forward_slab(self.dg_xyzs[s:s + k], m.bound, self.table, offsets, self.enc, None, None, k,
                                            cap, self.L, self.L, self.S, self.H, level_cost=self.level_cost_refresh)
                if self.rfield:
                    self._mlp_forward(cap, None, None, None, k, self.dg_sigma[s:s + k], None)
                else:               # the scatter is the field kernel's epilogue: no sigma array, no second launch
                    if self.pose:
                        eb.slab_window(self.enc, cap, self.L, self.level_w, None, k, scale_only=not self.baa)
                    self.mb.density_scatter(self.enc, cap, k, self.mlp_image, self.dg_indices[s:s + k], self.dg_tmp[cas],
                                            act=self.act)
            if self.rfield:
                eb.density_grid_scatter(self.dg_indices[:total], self.dg_sigma[:total], total, self.dg_tmp[cas])
        eb.density_grid_update(m.density_grid, self.dg_tmp, decay, self.dg_stats)
        eb.packbits_mean(m.density_grid, self.dg_stats, m.density_thresh, m.density_bitfield)
        if self.occ_index is not None:              # the march's LDS copy of the bitfield: rebuilt with it
            rb.build_occupancy_index(m.density_bitfield, m.cascade, m.grid_size, self.occ_index)

    def _sync_occ_index(self):
        """Rebuild the occupancy index if the bitfield was re-packed behind this object's back (model.update_extra_state)."""
        m = self.model
        version = getattr(m, "bitfield_version", 0)
        if self.occ_index is not None and version != self._occ_version:
            rb.build_occupancy_index(m.density_bitfield, m.cascade, m.grid_size, self.occ_index)
            self._occ_version = version

    @property
    def mean_density(self):
        return float(self.dg_stats[1]) if self.native_refresh else float(self.model.mean_density)   # host read

    def _timed(self, fn):
        """Run fn() between two HIP events on the current stream when bench.py asked for collective timings."""
        if self.collective_events is None or torch.cuda.is_current_stream_capturing():
            return fn()         # (events recorded during a capture are never executed: nothing to time there)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.collective_events.append((a, b))

    # ---- the exchange step's optimiser half (data parallel, or a separate optimiser pass on one GPU) ----------------
    def _xchg_pre(self):
        """Average the gradients over the ranks: afterwards `_xchg_grad()` is the mean gradient of what this rank updates."""
        x = self.xchg
        if self.collective_events is not None and not torch.cuda.is_current_stream_capturing():
            self.collective_steps += 1

        def run():
            if self.scaler is not None:
                # GradScaler under DDP sees the all-reduced gradients, inf on one rank is inf on all: here every rank owns a
                # shard of them, so the overflow word itself travels (4 bytes, MAX) and all ranks skip or step together
                x.all_reduce_max(self.scaler.found)
            if self.dp_mode == "shard":
                x.reduce_scatter_avg(self.gflat)
                if self.wire16:
                    x.all_reduce_avg(self.w_grad)
            elif self.wire16:
                x.all_reduce_avg(self.gflat, self.w_grad)
            else:
                x.all_reduce_avg(self.gflat)
        if x.carrier != "none":
            self._timed(run)

    def _skip(self):
        """The overflow word of the dynamic loss scale (None without one): optimiser kernels do nothing while it is set."""
        return self.scaler.found if self.scaler is not None else None

    def _xchg_adam(self):
        """ONE launch: Adam on this rank's part of the flat parameter (f32 wire: the MLP weights are part of it)."""
        x = self.xchg
        if self.dp_mode == "shard":
            lo, hi = x.shard_bounds(self.flat)
            own_p, own_g = self.flat[lo:hi], x.shard_of(self.gflat)
        else:
            own_p, own_g = self.flat, self.gflat
        if self.wire16:
            eb.adam_step_dev2((own_p, own_g, self.t_m, self.t_v, False),
                              (self.w_flat, self.w_grad, self.w_m, self.w_v, False), self.hyper, *self.betas, self.eps,
                              skip=self._skip())
        else:
            eb.adam_step_dev(own_p, own_g, self.t_m, self.t_v, self.hyper, *self.betas, self.eps, skip=self._skip())

    # ---- ... in two level groups (self.split): collectives on the comm stream, events between the two streams ------------
    def _group(self, which):
        """(parameter span, gradient span, seam parameter, seam gradient, moment offsets) of level group "a" / "b"."""
        sp = self.split
        if which == "a":
            return self.flat[:sp["sA"]], self.gflat[:sp["sA"]], self.flat[sp["sA"]:sp["bA"]], self.gflat[sp["sA"]:sp["bA"]], 0
        return self.flat[sp["bA"]:], self.gflat[sp["bA"]:], None, None, sp["nA"] + sp["nS"]

    def _on_comm(self, fn, done, inline=False):
        """fn() behind everything the main stream has queued so far -- on the comm stream when there is one (capturable
        carrier: the main stream carries on), in line otherwise -- then `done` marks its end."""
        main = torch.cuda.current_stream(self.device)
        if self.comm is None or inline:         # (same stream: the order is the dependency)
            self._timed(fn)
            return
        self.comm.wait_stream(main)
        with torch.cuda.stream(self.comm):
            self._timed(fn)
            self._ev[done].record(self.comm)

    def _split_reduce_scatter(self, which, inline=False, first=None):
        """Average group `which`'s gradient over the ranks (shards by reduce-scatter, the seam by all-reduce); group b's goes
        first and carries the loss scaler's overflow word (MAX) along."""
        x = self.xchg
        _, g, _, gs, _ = self._group(which)
        if which == "b" and self.collective_events is not None and not torch.cuda.is_current_stream_capturing():
            self.collective_steps += 1

        def run():
            if first is not None:       # group b's table reduction: on the comm stream too, BESIDE group a's on the main one
                first()                 # (one after the other the two launches take 73 us, side by side what one launch takes)
            if x.carrier == "none":
                return
            if which == "b" and self.scaler is not None:     # (the step's first collective: the word is final after the fill)
                x.all_reduce_max(self.scaler.found)
            x.reduce_scatter_avg(g)
            if gs is not None and gs.numel():
                x.all_reduce_avg(gs)
        self._on_comm(run, "rs" + which.upper(), inline)

    def _split_adam(self, which, inline=False):
        """Adam on this rank's shard of group `which` (+ the seam, on every rank alike), once its gradient has arrived."""
        x, sp = self.xchg, self.split
        p, g, ps, gs, at = self._group(which)
        if self.comm is not None and not inline:
            torch.cuda.current_stream(self.device).wait_event(self._ev["rs" + which.upper()])
        lo, hi = x.shard_bounds(p)
        n = hi - lo
        own = (p[lo:hi], x.shard_of(g), self.t_m[at:at + n], self.t_v[at:at + n], False)
        if ps is not None and ps.numel():
            k = ps.numel()
            eb.adam_step_dev2(own, (ps, gs, self.t_m[at + n:at + n + k], self.t_v[at + n:at + n + k], False), self.hyper,
                              *self.betas, self.eps, skip=self._skip())
        else:
            eb.adam_step_dev(*own[:4], self.hyper, *self.betas, self.eps, skip=self._skip())

    def _split_all_gather(self, which, inline=False):
        """Publish group `which`'s updated shards (behind its Adam step)."""
        p = self._group(which)[0]
        self._on_comm(lambda: self.xchg.all_gather(p) if self.xchg.carrier != "none" else None, "ag" + which.upper(), inline)

    def _wait_gather(self, which, inline=False):
        if self.comm is not None and not inline:
            torch.cuda.current_stream(self.device).wait_event(self._ev["ag" + which.upper()])

    def _xchg_post(self):
        """Publish the updated shards (shard mode)."""
        if self.dp_mode == "shard" and self.xchg.carrier != "none":
            self._timed(lambda: self.xchg.all_gather(self.flat))

    def optimizer_step(self, device_hyper=False):
        """Adam on the table and the MLP weights.  device_hyper: learning rate and bias corrections come from
        self.hyper (written by schedule_step earlier in the step) instead of host scalars."""
        if self.opt.lambda_tv > 0:
            self.model.grid_encoder.embeddings.grad = self.table_grad
            self.model.apply_total_variation(self.opt.lambda_tv)
        if self.opt.lambda_wd > 0:
            self.model.grid_encoder.embeddings.grad = self.table_grad
            self.model.apply_weight_decay(self.opt.lambda_wd)
        if device_hyper:                                # (step path: the gradient is overwritten next step, no zeroing)
            self._xchg_adam()
            return
        assert self.xchg is None or (self.xchg.carrier == "none" and not self.wire16), \
            "optimizer_step with host scalars: one rank, whole-table moments (data parallel: train_step owns the optimiser)"
        self._image_ready = False                       # the f16 operand image is of the weights before this update
        step, lr = self.global_step + 1, self.lr()
        if self.xchg is not None:                       # one flat parameter: table, MLP weights, padding
            eb.adam_step(self.flat, self.gflat, self.t_m, self.t_v, lr, *self.betas, self.eps, step, zero_grad=True)
            return
        eb.adam_step(self.table, self.table_grad, self.t_m, self.t_v, lr, *self.betas, self.eps, step, zero_grad=True)
        eb.adam_step(self.w_flat, self.w_grad, self.w_m, self.w_v, lr, *self.betas, self.eps, step, zero_grad=False)

    def invalidate_weights(self):
        """Call after changing the MLP weights from outside the step (load_state_dict into the w_flat views, an external
        optimiser): the next step / density-grid refresh rebuilds the f16 operand image instead of trusting the one the
        fused Adam keeps in step."""
        self._image_ready = False

    # ------------------------------------------------------------------ one optimiser step
    def _load_slot(self, slot, batch=None, noises=None, stage=0):
        """Draw (or take) a ray batch into the slot's fixed buffers and march it, on the current stream.
        stage 1: the part that does not read the occupancy bitfield (draw the rays, near/far, the march's chain kernel) --
        what a step in front of a density-grid refresh can still do for the next batch; 2: the rest of the march."""
        opt = self.opt
        if stage == 2:
            self.march(slot, slot.rays_o, slot.rays_d, slot.noises, stage=2)
            return
        if batch is None and self.device_sampler:
            d = self.data
            # pose refinement: rays are cast from the refined cameras (the dataset's pose_fn hook, provider.py:298-300)
            eb.sample_rays(d.images, self.poses_refined if self.pose else d.poses, d.intrinsics, self.N, self.seed64,
                           self.draw_ctr, slot.rays_o, slot.rays_d, slot.gt, slot.noises,
                           slot.bg if opt.background == "random" else None, slot.index, self.view_ldirs, slot.rays_ldir,
                           adaptive=self._adaptive_args(slot),
                           # the exposure of each ray's image (colmap_provider.py:605-606); parked ray slots get 1
                           exposure=(self.view_exposure, slot.exposure) if self.hdr else None)
            eb.counter_add(self.draw_ctr, 1)
        else:
            if batch is None:
                batch = self.data.sample_rays(self.N, self.ray_gen)
            if self.rfield:
                slot.rays_ldir.copy_(batch["rays_ldir"])
            gt = batch["images"]
            slot.gt[:, :gt.shape[-1]].copy_(gt)
            if gt.shape[-1] == 3:
                slot.gt[:, 3] = 1.0
            slot.rays_o.copy_(batch["rays_o"])
            slot.rays_d.copy_(batch["rays_d"])
            if opt.background == "random":
                torch.rand(slot.bg.shape, out=slot.bg, generator=self.ray_gen)
            if noises is None:
                torch.rand(slot.noises.shape, out=slot.noises, generator=self.ray_gen)
            else:
                slot.noises.copy_(noises)
        self.march(slot, slot.rays_o, slot.rays_d, slot.noises, stage=stage)

    def _step_ops(self, slot, gathered=True, gather=True, inline=False):
        """Everything one step does after the rays are marched, as (name, thunk, lane) triples.
        Exchange in two level groups only -- gathered: the parameters are complete when the step starts (False: the
        previous step of the same graph left its all-gathers in flight; the encoder waits group by group); gather: wait for
        this step's all-gathers at its end and rebuild the MLP's operand image (False: the next step of the graph does);
        inline: the collectives stay on the main stream (they are launched outside the step's graphs: eager carrier, or a
        step whose collectives bench.py times).  Lane "aux" marks the
        MLP-weight tail (gradient reduction, Adam, next step's f16 weight image): it depends only on the MLP backward,
        so on one GPU it runs on a third stream beside the table's fill + reduce instead of after them."""
        opt = self.opt
        bg_const = 1.0 if opt.background in ("white", "last_sample") else 0.0
        split = self.fuse_adam
        # lr / Adam bias corrections of this step, loss = 0, samples_seen += this batch's sample count -- and the record
        # offsets of the binned backward (the scan after the encoder's counting pass): one launch, right after the forward
        begin = ("ngp_x_step_begin", lambda: eb.step_begin(
            self.step_ctr, self.hyper, self.lr0, float(opt.iters), *self.betas, self.loss, self.samples_seen,
            slot.arena.counter, binned_workspace=slot.ws_grid if self.binned_counts else None, L=self.L,
            n_rows_total=self.rows, single_segment=True, scaling=self.scaler))
        ops = []
        # separate Adam (data parallel, or fuse_adam off): the reduction writes every row of the gradient, so nothing
        # has to zero it and the accumulate's read disappears (TV / weight decay are added afterwards, in optimizer_step)
        # one GPU, plain field: the MLP's weight-gradient reduction (+ Adam on the MLP weights + their entries in the f16
        # operand image) rides along with the table backward's fill launch instead of being a kernel of its own
        # (exchange step: the same passenger without Adam -- it leaves the weight gradients in the flat gradient buffer)
        ride = self.rides_mlp_tail()
        mlp_adam = (self.w_flat, self.w_grad, self.w_m, self.w_v, self.hyper, *self.betas, self.eps) if split else None
        mlp_tail = (self.cap, opt.loss_scale, self.dws, self.ws_mlp, mlp_adam, self.mlp_image if split else None) if ride else None
        self.table_backward_symbol = "ngp_x_grid_backward_binned_apply" + ("_mlp" if ride else "")   # (what bench.py times)
        field = self._field_ops(slot, slot.gt, slot.bg if opt.background == "random" else None, bg_const, zero_loss=False,
                                fused_adam=self.fuse_adam, split_weights=True, overwrite=not self.fuse_adam,
                                fuse_composite=True, mlp_tail=mlp_tail, split=self.split)
        field = self._without(field, "ngp_x_grid_backward_binned_prepare")          # folded into step_begin
        if not self.rfield and not self.pose and self.act is None and os.environ.get("NGP_STEP_BEGIN_RIDES", "1") != "0":
            # ... which in turn is one more workgroup of the MLP forward's launch (nothing reads its results before the
            # compositor): one kernel and one dependent-launch gap fewer on the critical path
            ar, cap = slot.arena, self.cap
            sb = (self.step_ctr, self.hyper, self.lr0, float(opt.iters), *self.betas, self.loss, self.samples_seen,
                  ar.counter, slot.ws_grid if self.binned_counts else None, self.L, self.rows, True, self.scaler)
            field = [("ngp_x_mlp_forward_step_begin", lambda: self._mlp_forward(cap, ar.dirs, ar.ldirs, ar.counter, cap, self.sigma,
                                                                               self.rgb, step_begin=sb))
                     if o[0] == "ngp_x_mlp_forward" else o for o in field]
        else:
            field.insert(1, begin)                                                  # right after the encoder's forward
        if self.pose:
            # the level window of THIS step and whether the cameras still move (annealing < end_annealing), from the step
            # counter before step_begin advances it; + 1: the reference counts the step before it trains it
            # (train_utils.py:887-888 in front of :488)
            field.insert(1, ("ngp_x_step_window", lambda: eb.step_window(
                self.step_ctr, 1, float(opt.iters), opt.start_annealing, opt.end_annealing, self.L, self.level_w, self.flags,
                baa=self.baa)))
        # pose refinement, after the field's own backward: ray gradients (encoder input backward + segment sums) ->
        # per-camera pose gradients -> se(3) Adam step and the refined poses the next batch is cast from
        pose_tail = []
        if self.pose:
            ar, d = slot.arena, self.data
            pose_tail = [
                ("ngp_x_ray_gradients", lambda: eb.ray_gradients(
                    self.denc, self.dydx, self.cap, self.L, self.model.bound, self.ddirs, ar.ts, ar.rays, self.N, self.cap,
                    self.g_rays_o, self.g_rays_d,
                    live=(self.live_n, self.live_off) if self._lists_live_samples() else None,
                    terms=(self.orient_weight, self.orient_ddirs) if self.orient else None)),
                ("ngp_x_pose_gradient", lambda: eb.pose_gradient(slot.index, self.g_rays_o, self.g_rays_d, self.N, len(d),
                                                                 d.W, d.intrinsics, self.grad_pose)),
                ("ngp_x_pose_update", lambda: eb.pose_update(self.xi, self.pose_base, self.grad_pose, self.flags, self.pose_m,
                                                             self.pose_v, self.pose_lr0, self.pose_gamma, 0.9, 0.999, 1e-8,
                                                             self.poses_refined, scaler=self.scaler)),
            ]
        if split:
            if self.rfield:
                # (the light-conditioned kernels reduce their weight gradients themselves: Adam and next step's operand
                # image are two more small launches)
                tail = [("ngp_x_adam_step_dev", lambda: eb.adam_step_dev(self.w_flat, self.w_grad, self.w_m, self.w_v,
                                                                         self.hyper, *self.betas, self.eps, skip=self._skip())),
                        ("ngp_x_mlp_prepare", self._mlp_prepare)]
            else:
                # weight gradients out of the partial sums and, element by element, Adam on the flat MLP weights and the
                # new value's two entries in the f16 operand image (so the next step needs no prepare pass)
                tail = [("ngp_x_mlp_reduce_dw", lambda: self.mb.reduce_dw(
                            self.cap, opt.loss_scale, self.dws, self.ws_mlp,
                            adam=(self.w_flat, self.w_grad, self.w_m, self.w_v, self.hyper, *self.betas, self.eps),
                            image=self.mlp_image, scaler=self.scaler))]
            # (dynamic loss scale: the weights' optimiser step waits for the table backward's verdict on the batch)
            late = self.scaler is not None
            for name, op in field:
                if name == "ngp_x_grid_backward_binned_apply" and not late:   # right after the MLP backward, beside the apply
                    ops += [(n, o, "aux") for n, o in tail]
                ops.append((name, op, "main"))
                if name == "ngp_x_grid_backward_binned_apply" and late:
                    ops += [(n, o, "main") for n, o in tail]
            ops += [(n, o, "main") for n, o in pose_tail]
            return ops
        if self.split is not None:
            return self._split_step_ops(field, pose_tail, ride, gathered, gather, inline)
        # ---- the exchange step: gradients -> collectives -> ONE Adam launch -> collectives -> next step's operand image
        ops += [(n, o, "main") for n, o in field]
        if not ride and not self.rfield:            # (the light-conditioned backward reduces its weight gradients itself)
            ops.append(("ngp_x_mlp_reduce_dw", lambda: self.mb.reduce_dw(self.cap, opt.loss_scale, self.dws, self.ws_mlp,
                                                                         scaler=self.scaler), "main"))
        if self.pose:
            # the cameras are replicated: every rank folds its rays into its own pose gradient, the ranks average them and
            # all take the same se(3) step
            pose_tail.insert(2, ("xchg_pose", lambda: self.xchg.all_reduce_avg(self.grad_pose.view(-1))))
            ops += [(n, o, "main") for n, o in pose_tail if n != "xchg_pose" or self.xchg.carrier != "none"]
        ops.append(("xchg_pre", self._xchg_pre, "main"))
        ops.append(("ngp_x_adam_step_dev", lambda: self.optimizer_step(device_hyper=True), "main"))
        ops.append(("xchg_post", self._xchg_post, "main"))
        ops.append(("ngp_x_mlp_prepare", self._mlp_prepare, "main"))
        return ops

    def _split_step_ops(self, field, pose_tail, ride, gathered, gather, inline):
        """The exchange step in two level groups (see __init__): a = levels [0, split), b = the finer levels + the MLP weights
        (69 % of the bytes at the default split of 8).  Main stream: encoder b, encoder a, field, compositor, field backward,
        fill, reduce b, reduce a, Adam b, Adam a; the collectives go to the comm stream as soon as their inputs exist --
        reduce-scatter b beside reduce a (the dense levels' heavy chunks: the longer of the two), all-gather b beside Adam a --
        and the NEXT step of the same graph starts its encoder on group b's levels while the (small) all-gather of group a is
        still running.  The order of the collectives is the same on every rank: rs b, rs a, ag b, ag a."""
        opt = self.opt
        ops = []
        for name, op in field:
            if name in ("fwd_a", "fwd_b") and not gathered:
                g = name[-1]
                ops.append(("wait_ag_" + g, lambda g=g: self._wait_gather(g, inline), "main"))
            ops.append((name, op, "main"))
            if name == "fwd_b" and not gathered:        # the MLP weights travel with group b: their operand image follows
                ops.append(("ngp_x_mlp_prepare", self._mlp_prepare, "main"))
            if name == "reduce_b":                      # (runs inside xchg_rs_b, on the comm stream: see _split_reduce_scatter)
                ops.pop()
                if not ride and not self.rfield:        # (weight gradients not reduced by the fill launch's passengers)
                    ops.append(("ngp_x_mlp_reduce_dw", lambda: self.mb.reduce_dw(self.cap, opt.loss_scale, self.dws, self.ws_mlp,
                                                                                 scaler=self.scaler), "main"))
                ops.append(("xchg_rs_b", lambda op=op: self._split_reduce_scatter("b", inline, first=op), "main"))
            if name == "reduce_a":
                ops.append(("xchg_rs_a", lambda: self._split_reduce_scatter("a", inline), "main"))
        if self.pose:
            pose_tail.insert(2, ("xchg_pose", lambda: self.xchg.all_reduce_avg(self.grad_pose.view(-1))))
            ops += [(n, o, "main") for n, o in pose_tail if n != "xchg_pose" or self.xchg.carrier != "none"]
        ops += [("adam_b", lambda: self._split_adam("b", inline), "main"),
                ("xchg_ag_b", lambda: self._split_all_gather("b", inline), "main"),
                ("adam_a", lambda: self._split_adam("a", inline), "main"),
                ("xchg_ag_a", lambda: self._split_all_gather("a", inline), "main")]
        if gather:
            ops += [("wait_ag", lambda: (self._wait_gather("a", inline), self._wait_gather("b", inline)), "main"),
                    ("ngp_x_mlp_prepare", self._mlp_prepare, "main")]
        return ops

    def _run_ops(self, ops, fork=True):
        """Launch a run of ops on the current stream; "aux" ops go to the aux stream (fork at the first one, join at the
        end of the run) when `fork`, else in line."""
        main = torch.cuda.current_stream(self.device)
        forked = False
        for _, op, lane in ops:
            if lane == "aux" and fork:
                if not forked:
                    self.aux.wait_stream(main)
                    forked = True
                with torch.cuda.stream(self.aux):
                    op()
            else:
                op()
        if forked:
            main.wait_stream(self.aux)

    def _capture_ops(self, ops):
        """A list of thunks as one hipGraph (list with its replay callable)."""
        if self.graph_pool is None:
            self.graph_pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.graph_pool, capture_error_mode="thread_local"):
            for op in ops:
                op()
        self._graphs_alive.append(g)
        return [g.replay]

    def _load_slot_fast(self, slot, stage=0):
        """_load_slot (device sampler + march + binning reset) replayed from a graph: 9 launches -> 1."""
        if not (self.use_graph and self.device_sampler and self.global_step >= 2 and self.march_mode != "index"):
            return self._load_slot(slot, stage=stage)
        key = ("load", id(slot), stage)
        if key not in self.graphs:
            self.graphs[key] = self._capture_ops([lambda: self._load_slot(slot, stage=stage)])
        for part in self.graphs[key]:
            part()

    def rides_mlp_tail(self):
        """Does the MLP's weight-gradient reduction ride on the table backward's fill launch (ngp_x_..._apply_mlp)?"""
        return not self.rfield and not bool(getattr(self.opt, "aux_stream", False)) and \
            os.environ.get("NGP_MLP_TAIL_RIDES", "1") != "0"

    def _loose_collectives(self, timed):
        """Do this step's collectives stay outside its graph?  Yes when their carrier cannot be captured (gloo,
        torch.distributed) and on the steps bench.py times (HIP events around them)."""
        return self.xchg is not None and self.xchg.carrier != "none" and \
            (not self.xchg.capturable or (timed and self.collective_events is not None))

    def _capture(self, slot, timed):
        """The step as hipGraphs.  Two kinds of op stay outside: the gradient all-reduce (RCCL, under DP) and, on the
        steps where bench.py times it with HIP events, the probed entry point (events inside a graph cannot be
        timed); the runs of ops between them become one graph each."""
        from .. import _lib
        if self.graph_pool is None:
            self.graph_pool = torch.cuda.graph_pool_handle()
        # the collectives stay outside when their carrier cannot be captured (gloo, torch.distributed) and on the steps
        # bench.py times (HIP events around them)
        loose = self._loose_collectives(timed)
        eager = {*(("xchg_pre", "xchg_post", "xchg_pose", "xchg_rs_a", "xchg_rs_b", "xchg_ag_a", "xchg_ag_b") if loose else ()),
                 *(_lib.probed_symbols() if timed else ())}
        ops = self._step_ops(slot, inline=loose)
        # one graph: the aux lane may fork inside it
        whole = not any(name in eager for name, _, _ in ops) and bool(getattr(self.opt, "aux_stream", False))
        parts, run = [], []

        def flush():
            if run and len(run) <= 2 and all(lane == "main" for _, _, lane in run):
                # (one or two launches between two eager ops: launched directly -- a graph launch costs more than it saves)
                parts.extend(op for _, op, _ in run)
                run.clear()
            if run:
                g, seg = torch.cuda.CUDAGraph(), list(run)
                with torch.cuda.graph(g, pool=self.graph_pool, capture_error_mode="thread_local"):
                    self._run_ops(seg, fork=whole)
                parts.append(g.replay)
                self._graphs_alive.append(g)
                run.clear()

        for name, op, lane in ops:
            if name in eager:
                flush()
                parts.append(op)
            else:
                run.append((name, op, lane))
        flush()
        return parts

    def train_step(self, batch=None, noises=None):
        opt, model = self.opt, self.model
        model.train()
        step = self.global_step
        if self._refresh_step_folded(step, batch):
            return self.loss
        if step % opt.update_extra_interval == 0:
            if self.native_refresh:
                from .. import _lib
                with _lib.probe_paused():               # its encoder calls are not the training step's
                    self.refresh_density_grid()
            else:
                if self.world_size > 1:
                    torch.manual_seed(1234567 + step)
                model.update_extra_state()
                self._sync_occ_index()
        if not self._image_ready:                       # later steps prepare it right after their Adam step
            self._mlp_prepare()
            self._image_ready = True
        slot = self.slots[step % len(self.slots)]
        if batch is not None:
            self._load_slot(slot, batch, noises)
            slot.step = step
        elif slot.step != step:                         # first step, or just after a grid refresh
            # (after a refresh: the previous step has already drawn this batch and run the part of its march that does
            # not read the bitfield)
            self._load_slot_fast(slot, stage=2 if slot.head_step == step else 0)
            slot.step = step
        # the occupancy bitfield the next step marches through is final unless that step refreshes it first
        ahead = self.prefetch and batch is None and (step + 1) % opt.update_extra_interval != 0
        head = self.prefetch and batch is None and not ahead and self._split_march
        nxt = self.slots[(step + 1) % 2] if (ahead or head) else None
        main = torch.cuda.current_stream(self.device) if self.prefetch else None
        if nxt is not None:
            # fork: the next step's rays are drawn and marched on the side stream, concurrently with everything
            # this step does on the main stream (nxt's previous user, step - 1, is already behind this point).  In front
            # of a refresh: only what does not depend on the bitfield (a third of the march)
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self._load_slot_fast(nxt, stage=0 if ahead else 1)
                if not ahead and self._refresh_head_ok():
                    self.refresh_head(step + 1)
            if ahead:
                nxt.step = step + 1
            else:
                nxt.head_step = step + 1
        if self.use_graph and batch is None and step >= 2:      # the first steps run eagerly (lazy init, caches)
            from .. import _lib
            probed = [n for n in _lib.probed_symbols() if n in self._main_symbols]
            timed = bool(probed) and _lib.probe_next_timed()
            key = (step % 2, timed, self._loose_collectives(timed))
            if key not in self.graphs:
                self.graphs[key] = self._capture(slot, timed)
            for part in self.graphs[key]:
                part()
            if probed and not timed:
                _lib.probe_skip(probed)                         # the call happened inside the graph
            self.last_graph_key = key
        else:
            self._run_ops(self._step_ops(slot), fork=bool(getattr(self.opt, "aux_stream", False)))
            self.last_graph_key = None
        if nxt is not None:
            main.wait_stream(self.side)                 # join
        self.global_step += 1
        self.last_loss = self.loss
        return self.loss

    def _refresh_step_folded(self, step, batch):
        """Steady-state refresh step as ONE graph: the refresh (its cells were drawn ahead, beside the previous step), the
        rest of this batch's march (it reads the new bitfield), the fork of the side stream for the next batch -- BEHIND the
        refresh: that march reads the new bitfield too -- and the step.  Launched as three graphs in a row the same work
        left ~ 40 us of gaps on the critical path (14 us in front of the refresh, 9 in front of the march, 15 in front of
        the step).  Returns False when the step is not of that kind (train_step then takes it piece by piece)."""
        from .. import _lib
        opt, model = self.opt, self.model
        every = opt.update_extra_interval
        slot = self.slots[step % len(self.slots)]
        if not (step % every == 0 and self.native_refresh and self.use_graph and self.prefetch and batch is None and step >= 2
                and model.iter_density >= 16 and self._refresh_head_step == step and slot.head_step == step
                and slot.step != step and self._image_ready and every > 1
                and (self.xchg is None or self.xchg.capturable)
                and os.environ.get("NGP_FOLD_REFRESH", "1") != "0"):
            return False
        probed = [n for n in _lib.probed_symbols() if n in self._main_symbols]
        if probed and _lib.probe_next_timed():
            return False
        nxt = self.slots[(step + 1) % 2]
        key = ("rstep", step % 2)
        if key not in self.graphs:
            if self.graph_pool is None:
                self.graph_pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.graph_pool, capture_error_mode="thread_local"):
                main = torch.cuda.current_stream(self.device)
                self._refresh_launches(0.95, False, 2)
                self._load_slot(slot, stage=2)
                self.side.wait_stream(main)             # fork: the next step's rays, marched through the NEW bitfield
                with torch.cuda.stream(self.side):
                    self._load_slot(nxt, stage=0)
                self._run_ops(self._step_ops(slot), fork=bool(getattr(opt, "aux_stream", False)))
                main.wait_stream(self.side)             # join
            self._graphs_alive.append(g)
            self.graphs[key] = [g.replay]
        for part in self.graphs[key]:
            part()
        # (what refresh_density_grid, _load_slot_fast and train_step note down)
        model.iter_density += 1
        model.bitfield_version = getattr(model, "bitfield_version", 0) + 1
        self._occ_version = model.bitfield_version
        slot.step, nxt.step = step, step + 1
        if probed:
            _lib.probe_skip(probed)
        self.global_step += 1
        self.last_loss = self.loss
        self.last_graph_key = key
        return True

    @property
    def last_num_points(self):
        return int(self.arena.counter[0])          # host read: only for logging

    # ------------------------------------------------------------------ several steps per graph
    def _multi_step(self, limit, first=False):
        """Run up to `limit` (and at most self.group_steps) consecutive regular steps from ONE captured graph; returns how
        many it ran (0: the caller takes a single train_step).  Between two graph launches the GPU idles for ~ 20 us (the
        replay floor of a dependent launch); a step is ~ 0.4 ms, so one launch per step costs 5 %.  A group never contains
        a density-grid refresh (those steps go through train_step), nor a step whose kernels bench.py times with events."""
        from .. import _lib
        opt = self.opt
        every = opt.update_extra_interval
        s = self.global_step
        if not (self.use_graph and self.prefetch and self.device_sampler and self.march_mode != "index"
                and (self.xchg is None or self.xchg.capturable)
                and s >= 2 and s % every != 0 and self._image_ready):
            return 0
        G = min(limit, every - s % every, _lib.probe_untimed_run())
        slot = self.slots[s % 2]
        if G < 2 or slot.step != s:
            return 0
        # (longer groups save launch boundaries, but a graph launch costs host time in proportion to its nodes, which is
        # exposed whenever the GPU has nothing queued -- e.g. at the start of a short timed region)
        G = min(G, int(getattr(opt, "group_steps", 8)))
        # (`group_ramp`: the FIRST group of a train() call is short -- the stream may be idle, e.g. right after a synchronise,
        # and the host time of a long graph's launch would be idle GPU time; behind it the launches run ahead of the GPU)
        if first and getattr(opt, "group_ramp", False):
            G = min(G, 2)
        if not self._groups_precaptured and not getattr(opt, "group_any", False):
            G = 1 << (G.bit_length() - 1)          # 2, 4, 8: a handful of graph variants, all captured early in a run
                                                   # (a capture takes milliseconds: none may fall into a timed region)
        last_ahead = (s + G) % every != 0          # does the group's last step draw the rays of the step after it?
        head = not last_ahead and self._refresh_head_ok()      # ... or the cells of the refresh that follows it?
        key = ("multi", s % 2, G, last_ahead, head)
        if key not in self.graphs:
            self._capture_group(s % 2, G, last_ahead, head)
        for part in self.graphs[key]:           # (capturing does not execute anything)
            part()
        probed = [n for n in _lib.probed_symbols() if n in self._main_symbols]
        for k in range(G):
            nxt_step = s + k + 1
            if k + 1 < G or last_ahead:
                self.slots[nxt_step % 2].step = nxt_step
            elif self._split_march:
                self.slots[nxt_step % 2].head_step = nxt_step
                if head:
                    self._refresh_head_step = nxt_step
            if probed:
                _lib.probe_skip(probed)
        self.global_step += G
        self.last_loss = self.loss
        self.last_graph_key = key
        return G

    def _capture_group(self, parity, G, last_ahead, head=False):
        """G consecutive regular steps (the first one on slot `parity`) as one hipGraph; nothing is executed."""
        if self.graph_pool is None:
            self.graph_pool = torch.cuda.graph_pool_handle()
        opt = self.opt
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.graph_pool, capture_error_mode="thread_local"):
            main = torch.cuda.current_stream(self.device)
            for k in range(G):
                cur = self.slots[(parity + k) % 2]
                whole = k + 1 < G or last_ahead         # (else a refresh follows: only the bitfield-independent part)
                nxt = self.slots[(parity + k + 1) % 2] if (whole or self._split_march) else None
                # (exchange in two level groups: inside the graph a step's all-gathers run into the next step's encoder)
                ops = self._step_ops(cur, gathered=k == 0, gather=k + 1 == G)
                at = min(max(int(os.environ.get("NGP_SIDE_FORK_AT", "0")), 0), len(ops) - 1) if nxt is not None else 0
                self._run_ops(ops[:at], fork=False)
                if nxt is not None:         # fork: the next step's rays, on the side stream
                    self.side.wait_stream(main)
                    with torch.cuda.stream(self.side):
                        self._load_slot(nxt, stage=0 if whole else 1)
                        if not whole and head:
                            self._refresh_launches(0.95, False, 1)
                self._run_ops(ops[at:], fork=bool(getattr(opt, "aux_stream", False)) and at == 0)
                if nxt is not None:
                    main.wait_stream(self.side)     # join
        self._graphs_alive.append(g)
        self.graphs[("multi", parity, G, last_ahead, head)] = [g.replay]

    def precapture_groups(self):
        """Capture the step groups of every length (2 .. update_extra_interval - 1, both slot parities, with and without
        the trailing prefetch) now, so that any run of regular steps between two density-grid refreshes -- or between two
        steps bench.py times -- is ONE graph launch and no capture (milliseconds) falls into a timed region.  Call after
        the first few steps (lazy initialisation done); nothing is executed.  Returns the number of graphs captured."""
        if not (self.use_graph and self.prefetch and self.device_sampler and self.march_mode != "index"
                and (self.xchg is None or self.xchg.capturable)
                and self.global_step >= 2 and self._image_ready):
            return 0
        n = 0
        for parity in (0, 1):
            for G in range(2, min(self.opt.update_extra_interval - 1, int(getattr(self.opt, "group_steps", 8))) + 1):
                for last_ahead in (True, False):
                    head = not last_ahead and self._refresh_head_ok()
                    if ("multi", parity, G, last_ahead, head) not in self.graphs:
                        self._capture_group(parity, G, last_ahead, head)
                        n += 1
        self._groups_precaptured = True
        return n

    def train(self, steps, log_every=0):
        done = 0
        while done < steps:
            n = 0 if log_every else self._multi_step(steps - done, first=done == 0)
            if n == 0:
                self.train_step()
                n = 1
            done += n
            if log_every and self.rank == 0 and self.global_step % log_every == 0:
                needed = int(self.arena.counter[1])
                print(f"[step {self.global_step}] loss {float(self.loss):.5f} samples {self.last_num_points}"
                      + (f" (ARENA OVERFLOW: {needed} > {self.cap})" if needed > self.cap else ""), flush=True)

    @torch.no_grad()
    def render_rays(self, rays_o, rays_d, bg_const=0.0, ldir=None):
        """Images for evaluation out of the TRAINING kernels, forward only: per block of N rays one chain-parallel march
        (no jitter), slab encoder, fused MLP, wave compositor -- instead of the reference's alive-ray loop
        (renderer.py:573-616: up to max_steps rounds of march_rays / field / composite_rays with a host sync each).
        Same sample positions, same T_thresh rule; ~ 10 x faster.  Returns (image [T,3], overflowed: bool)."""
        opt, m, N, cap = self.opt, self.model, self.N, self.cap
        if self._eval_slot is None:
            chain_cap = self.slots[0].arena.chain[0].shape[0] if self.slots[0].arena.chain is not None else 0
            self._eval_slot = _Slot(N, opt.max_steps, cap, self.device, chain_cap, lit=self.rfield)
            self._eval_slot.noises.zero_()
        slot, ar = self._eval_slot, self._eval_slot.arena
        if self.rfield:             # one light per rendered view (colmap_provider.py:619: rays_ldir of the image)
            assert ldir is not None, "render_rays: the light-conditioned field needs the view's light direction"
            slot.rays_ldir.copy_(torch.as_tensor(ldir, dtype=torch.float32, device=self.device).view(-1, 3).expand(N, 3))
        self._mlp_prepare()
        total = rays_o.shape[0]
        out = torch.empty(total, 3, device=self.device)
        # image rays are coherent: a block of N neighbouring pixels can need more samples than the arena holds (training
        # batches are random pixels, most of which miss the object).  So a block carries `real` rays and N - real rays
        # that miss the volume; `real` halves when a block overflows and recovers when blocks are light (one 16-byte
        # read of the sample counter per block).
        miss_o = torch.tensor([0.0, 0.0, 1e3 * m.bound], device=self.device)
        miss_d = torch.tensor([0.0, 0.0, 1.0], device=self.device)
        real, s = N, 0
        while s < total:
            n = min(real, total - s)
            slot.rays_o[:n].copy_(rays_o[s:s + n])
            slot.rays_d[:n].copy_(rays_d[s:s + n])
            if n < N:
                slot.rays_o[n:] = miss_o
                slot.rays_d[n:] = miss_d
            self.march(slot, slot.rays_o, slot.rays_d, slot.noises, aabb=m.aabb_infer, plan=False)
            needed = int(ar.counter[1])                                     # host read
            if needed > cap:
                if real == 1:
                    return out, True
                real = max(1, real // 2)
                continue
            eb.grid_encode_forward_slab(ar.xyzs, m.bound, self.table, m.grid_encoder.offsets, self.enc, None, ar.counter,
                                        cap, cap, self.L, self.L, self.S, self.H)
            self._mlp_forward(cap, ar.dirs, ar.ldirs, ar.counter, cap, self.sigma, self.rgb)
            eb.composite_rays_train_forward(self.sigma, self.rgb, ar.ts, ar.rays, cap, N, opt.T_thresh, self.weights_buf,
                                            self.ws, self.depth, self.image)
            out[s:s + n] = self.image[:n] + (1.0 - self.ws[:n, None]) * bg_const
            s += n
            if needed < cap // 4 and real < N:
                real = min(N, real * 2)
        return out, False

    @torch.no_grad()
    def evaluate(self, dataset, max_views=None, chunk=1 << 16, fast=True, distributed=True):
        """PSNR over held-out views (train_utils.py:221-233).  fast: render with render_rays; a view whose samples do
        not fit the arena falls back to the reference-shaped inference loop.
        distributed (more than one rank): the views are dealt to the ranks round-robin, every round ends with an all-gather
        of the ranks' predictions and targets, and every rank feeds the meter with all of them in view order -- the
        reference's evaluate_one_epoch (train_utils.py:1033-1048: all_gather of preds / truths, then the metrics) with the
        result on every rank instead of on rank 0 only.  Same images, same order: the value a single rank computes."""
        from . import utils
        from .trainer import Trainer
        if not fast:        # (the training background does not matter here: evaluation composites over a constant 0)
            return Trainer.evaluate(self, dataset, max_views, chunk)
        self.model.eval()
        meter = utils.PSNRMeter()
        n = len(dataset) if max_views is None else min(max_views, len(dataset))
        R, r = (self.world_size, self.rank) if (distributed and self.world_size > 1) else (1, 0)

        def render(v):
            data = dataset.view(v)
            pred, overflow = self.render_rays(data["rays_o"].contiguous(), data["rays_d"].contiguous(), 0.0,
                                              ldir=data.get("rays_ldir"))
            if overflow:
                preds = [self.model.render(data["rays_o"][s:s + chunk], data["rays_d"][s:s + chunk],
                                           rays_ldir=data.get("rays_ldir"), bg_color=0,
                                           perturb=False)["image"] for s in range(0, pred.shape[0], chunk)]
                pred = torch.cat(preds, 0)
            img = data["images"]
            gt = img[..., :3] * img[..., 3:] if img.shape[-1] == 4 else img
            return pred.view(data["H"], data["W"], 3).clamp(0, 1), gt.reshape(data["H"], data["W"], 3).float()

        for base in range(0, n, R):
            mine = min(base + r, n - 1)         # (a rank without a view in the last round renders the last one again: the
            pred, gt = render(mine)             # collective wants equal shapes; its copy is not counted)
            if R == 1:
                meter.update(pred, gt)
                continue
            pair = torch.stack([pred, gt]).contiguous()
            parts = [torch.empty_like(pair) for _ in range(R)]
            torch.distributed.all_gather(parts, pair)
            for k in range(min(R, n - base)):
                meter.update(parts[k][0], parts[k][1])
        return meter.measure()
