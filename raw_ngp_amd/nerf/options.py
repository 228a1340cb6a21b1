"""Hot-path options with the reference's flag names and defaults (main.py:10-125).

The reference passes one argparse Namespace (`opt`) everywhere; NeRFNetwork / NeRFRenderer here read
the same attribute names, so either this dataclass or the reference's own Namespace can be handed in.
Only flags that reach the hot path are listed; dataset / logging / mesh flags are out of scope.
"""
from dataclasses import dataclass, field
from typing import List


@dataclass
class Options:
    # scene / marching (main.py:31-53)
    bound: float = 2.0
    contract: bool = False
    min_near: float = 0.05
    T_thresh: float = 1e-8
    cuda_ray: bool = True
    max_steps: int = 1024
    num_steps: List[int] = field(default_factory=lambda: [256, 96, 48])
    update_extra_interval: int = 16
    max_ray_batch: int = 4096 * 4
    grid_size: int = 128
    dt_gamma: float = 0.0
    density_thresh: float = 10.0
    background: str = "black"
    # encoder (main.py:55-56)
    hashgrid_resolution: int = 2048
    hashmap_size: int = 19
    # batch (main.py:59-61)
    num_rays: int = 4096
    adaptive_num_rays: bool = False
    num_points: int = 2 ** 18
    # regularisers (main.py:64-69)
    lambda_entropy: float = 0.0
    lambda_tv: float = 0.0
    lambda_wd: float = 0.0
    lambda_orientation: float = 0.0
    lambda_proposal: float = 1.0
    lambda_distort: float = 0.0
    # field (main.py:90-92,101,106-109,121)
    internal_activation: str = "relu"
    color_activation: str = "clamped_exp"
    density_activation: str = "clamped_exp"
    rfield: bool = False
    pose_opt: str = "none"
    c_lr: float = 1e-3           # pose refinement (main.py:110-113)
    noise: float = 0.0
    identity: bool = False
    scale: float = 1.0
    start_annealing: float = 0.0
    end_annealing: float = 0.33
    beta: float = 2.0
    compute_normals: bool = False
    image_mode: str = "LDR"      # HDR: exposure-scaled, clipped RawNeRF loss (main.py:86, train_utils.py:512-536)
    loss_weight: str = "none"    # HDR loss weighting (main.py:118): none | planck | gaussian | hanning
    # training (main.py:16,40-41)
    fp16: bool = False
    iters: int = 20000
    lr: float = 1e-2
    device: str = "cuda"
    # --- extensions of this implementation (no reference counterpart) -----------------------------
    fused_mlp: bool = False       # hand-written MFMA tiny-MLP instead of nn.Linear stacks
    loss_scale: float = 65536.0   # loss scale of the fused MLP backward's f16 deltas: GradScaler's initial value
                                  # (train_utils.py:404); the per-op path keeps it static (deltas saturate), the fused engine
                                  # adapts it on the device:
    dynamic_loss_scale: bool = True   # GradScaler's rule inside the step graphs: x0.5 and the step skipped on overflow, x2 after
    scale_growth_interval: int = 2000  # this many clean steps (torch.cuda.amp.GradScaler defaults, train_utils.py:897-904)
    arena_capacity: int = 0       # > 0: sample arena (no host sync per step); 0 = reference 2-pass protocol
    native_grid_refresh: bool = True  # fused engine: density-grid refresh as device kernels (no host syncs)
    dp_exchange: str = None       # data parallel, carrier of the collectives: "rccl" = bare RCCL calls on the step's stream,
                                  # captured inside the step graphs (default on an "nccl" process group), "torch" =
                                  # torch.distributed, eager between graph segments (gloo always takes this route)
    dp_rehearsal: bool = False    # run the data-parallel step on ONE rank (needs an initialised process group)
    grad_wire: str = "f32"        # data parallel: wire format of the gradient exchange (f32 | bf16; bf16 is an opt-in:
                                  # the cross-rank sum is then formed in bfloat16, narrower than anything the reference does)
    dp_mode: str = "shard"        # data parallel: "shard" = reduce_scatter -> Adam on 1/R of the table -> all_gather
                                  # (SURVEY 8e), "allreduce" = gradient all-reduce + full Adam on every rank
    dp_split_level: int = None    # data parallel, "shard", f32 wire: exchange the table in two level groups -- levels below this
                                  # one, and the finer ones + the MLP weights -- so that reduce-scatter / all-gather of one group
                                  # overlap the table reduction / the next step's encoder of the other (engine.py).  None: 8 with
                                  # more than one rank, off on one; 0: off
    aux_stream: bool = False      # fused engine: MLP-weight tail (dW reduction, Adam, f16 image) on a third stream
                                  # (measured slower: the fork/join costs more than the ~20 us it takes off the main stream)
    fuse_adam: bool = True        # fused engine, one rank: Adam on the table inside the gradient reduction kernel
    march_mode: str = "chain"     # fused engine, march pass 1: chain | index | serial (see engine.py)
    device_sampler: bool = True   # fused engine: draw ray batches with one kernel (Philox) instead of torch ops
    capture_graph: bool = True    # fused engine: replay whole steps from captured hipGraphs
    group_steps: int = 15         # fused engine: consecutive regular steps replayed from ONE graph (1 = a graph per step): a
                                  # graph boundary costs ~ 26 us of idle GPU; 15 = everything between two grid refreshes
    group_any: bool = True        # ... of any length up to group_steps (False: 2, 4, 8 only -- fewer variants to capture)
    group_ramp: bool = True       # ... but the first group of a train() call at most 2 steps long (an idle stream waits for
                                  # the host while it launches a long graph)
    prefetch_march: bool = True   # fused engine: march step i+1's rays on a second stream during step i's backward
