"""Tiny-MLP radiance field on top of the hash-grid and SH encoders.

Host-side mirror of the reference's nerf/network.py: `MLP` (:12-35, bias-free Linear stack with
in-place ReLU or softplus(beta, threshold 20)), `NeRFNetwork` (:37-184: sigma = trunc_exp(f0) |
softplus, feat = f[1:16]; colour = view_mlp(cat[feat, SH(d)[, SH(l)]]) with exp / sigmoid /
clamped_exp activation; BARF and BAA-NGP level windows :77-109).  Parameter names are the
reference's (`grid_encoder.embeddings`, `grid_encoder.offsets`, `grid_mlp.net.{0,1,2}.weight`,
`view_mlp.net.*`) so its checkpoints load with `load_state_dict(strict=False)`.
The BARF pose optimiser itself (barf/camera_optimizers.py) is out of scope; `pose_opt` here only
selects the level-window weighting, which is the part that touches the hot path.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..activation import trunc_exp
from ..encoding import get_encoder
from .renderer import NeRFRenderer


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T for a tall activation matrix (hundreds of thousands of samples, <= 96 features).  Forward and d/dx are
    ordinary GEMMs; the weight gradient dY^T X contracts over ALL samples into a tiny matrix, a shape hipBLASLt runs at
    0.3-1.3 ms per layer here -- as 64 batched GEMMs over sample slices plus a sum it takes 30-60 us (same products, the
    sum merely regrouped)."""
    SLICES = 64

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            b, M = _SplitKLinear.SLICES, x.shape[0]
            main = (M // b) * b
            dyc, xc = dy.contiguous(), x.contiguous()
            dw = torch.bmm(dyc[:main].view(b, main // b, -1).transpose(1, 2), xc[:main].view(b, main // b, -1)).sum(0)
            if main < M:
                dw = dw + dyc[main:].t() @ xc[main:]
        return dx, dw


def _linear(x, layer):
    if (layer.bias is None and x.is_cuda and x.dim() == 2 and x.shape[0] >= 16384 and x.dtype == torch.float32
            and not torch.is_autocast_enabled() and torch.is_grad_enabled()):
        return _SplitKLinear.apply(x, layer.weight)
    return layer(x)


class MLP(nn.Module):
    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, opt, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.opt = opt
        widths = [dim_in] + [dim_hidden] * (num_layers - 1) + [dim_out]
        self.net = nn.ModuleList(nn.Linear(a, b, bias=bias) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, x):
        last = self.num_layers - 1
        for i, layer in enumerate(self.net):
            x = _linear(x, layer)
            if i == last:
                break
            if self.opt.internal_activation == "relu":
                x = F.relu(x, inplace=True)
            elif self.opt.internal_activation == "softplus":
                x = F.softplus(x, beta=self.opt.beta, threshold=20)
        return x


def level_window(annealing, start, end, n_levels, device):
    """Cosine window over levels, w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2 (network.py:81-86,101-106)."""
    if end == 0:
        end = 1e-12
    alpha = (annealing - start) / (end - start) * n_levels
    k = torch.arange(n_levels, dtype=torch.float32, device=device)
    return (1 - (alpha - k).clamp_(min=0, max=1).mul_(np.pi).cos_()) / 2


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt):
        super().__init__(opt)
        self.annealing = 0.0
        self.level_dim = 2
        self.grid_encoder, self.grid_in_dim = get_encoder(
            "hashgrid", input_dim=3, level_dim=self.level_dim, num_levels=16,
            log2_hashmap_size=self.opt.hashmap_size, desired_resolution=self.opt.hashgrid_resolution * self.bound)
        self.grid_mlp = MLP(self.grid_in_dim, 16, 64, 3, opt, bias=False)
        self.view_encoder, self.view_in_dim = get_encoder("sh", input_dim=3, degree=4)
        ldir_dim = self.view_in_dim if self.opt.rfield else 0
        self.view_mlp = MLP(15 + self.view_in_dim + ldir_dim, 3, 64 + ldir_dim, 3, opt, bias=False)

        if not self.opt.cuda_ray:      # proposal networks of the run() sampler (network.py:59-72)
            self.prop_encoders = nn.ModuleList()
            self.prop_mlp = nn.ModuleList()
            for desired in (128, 256):
                enc, dim = get_encoder("hashgrid", input_dim=3, level_dim=2, num_levels=5, log2_hashmap_size=17,
                                       desired_resolution=desired)
                self.prop_encoders.append(enc)
                self.prop_mlp.append(MLP(dim, 1, 16, 2, opt, bias=False))

    # -- level windows -----------------------------------------------------------------------
    def _apply_level_window(self, f):
        mode = self.opt.pose_opt
        dev = f.device
        if mode == "baangp":
            L = self.grid_mlp.dim_out - 1
            w = level_window(self.annealing, self.opt.start_annealing, self.opt.end_annealing, L, dev)
            weights = torch.cat([torch.ones(self.level_dim, device=dev), w.repeat_interleave(self.level_dim)])
            assert f.shape[-1] == len(weights)
            active = f[..., weights > 0]
            coarse = active[..., -self.level_dim:]
            reps = [1] * (f.dim() - 1) + [L + 1]
            weights[0:2] = 1
            return f * weights + coarse.repeat(*reps) * (1 - weights)
        if mode == "barf":
            L = self.grid_mlp.dim_out
            w = level_window(self.annealing, self.opt.start_annealing, self.opt.end_annealing, L, dev)
            weights = w.repeat_interleave(self.level_dim)
            weights[0:2] = 1
            return f * weights
        return f

    # -- fused MFMA path (extension; Options.fused_mlp) ------------------------------------------
    def _fused(self, plain_only=False):
        """The field configuration the MFMA kernels implement: ReLU hidden layers, trunc_exp density and clamped_exp colour, and
        the reference's other activations (network.py:31-34,115,131-135): softplus density and exp / sigmoid colour for both
        fields, softplus hidden layers for the plain one.
        plain_only: additionally the 31-input view MLP without level windows -- what the autograd op `fused_field`
        covers; the light-conditioned / BARF variants exist for the fused training step only (nerf/engine.py)."""
        from .._lib import field_activations, _default_act
        o = self.opt
        act = field_activations(o)
        # (the light-conditioned field: other OUTPUT activations yes, softplus hidden layers no)
        ok = (getattr(o, "fused_mlp", False) and act is not None and (_default_act(act) or not o.rfield or act[3] == 0)
              and self.grid_encoder.embeddings.is_cuda)
        if plain_only:
            ok = ok and not o.rfield and o.pose_opt == "none"
        return ok

    def _mlp_weights(self):
        return [l.weight for l in self.grid_mlp.net] + [l.weight for l in self.view_mlp.net]

    # -- field ------------------------------------------------------------------------------
    def common_forward(self, x):
        f = self._apply_level_window(self.grid_encoder(x, bound=self.bound))
        f = self.grid_mlp(f)
        if self.opt.density_activation == "clamped_exp":
            sigma = trunc_exp(f[..., 0])
        else:
            sigma = F.softplus(f[..., 0], beta=self.opt.beta, threshold=20)
        return sigma, f[..., 1:]

    def forward(self, x, d, ld=None, **kwargs):
        """x [N,3] in [-bound, bound], d [N,3] unit view dirs, ld [N,3] light dirs (rfield)."""
        if self._fused(plain_only=True) and not d.requires_grad:
            from .fused_field import fused_field
            enc = self.grid_encoder(x.reshape(-1, 3), bound=self.bound, slab=True)
            from .._lib import field_activations
            sigma, color = fused_field(enc, d.reshape(-1, 3), self._mlp_weights(),
                                       getattr(self.opt, "loss_scale", 65536.0), field_activations(self.opt))
            return {"sigma": sigma.view(x.shape[:-1]), "color": color.view(*x.shape[:-1], 3)}
        sigma, feat = self.common_forward(x)
        parts = [feat, self.view_encoder(d)]
        if self.opt.rfield:
            parts.append(self.view_encoder(ld))
        color = self.view_mlp(torch.cat(parts, dim=-1))
        act = self.opt.color_activation
        if act == "exp":
            color = torch.exp(color - 5.0)
        elif act == "sigmoid":
            color = torch.sigmoid(color)
        elif act == "clamped_exp":
            color = torch.clamp(torch.exp(color - 5.0), max=5.0)
        return {"sigma": sigma, "color": color}

    def density(self, x, proposal=-1):
        if self._fused(plain_only=True) and not torch.is_grad_enabled() and not (0 <= proposal < len(getattr(self, "prop_encoders", ()))):
            from .fused_field import fused_density
            enc = self.grid_encoder(x.reshape(-1, 3), bound=self.bound, slab=True)
            from .._lib import field_activations
            return {"sigma": fused_density(enc, self._mlp_weights(), field_activations(self.opt)).view(x.shape[:-1])}
        if 0 <= proposal < len(getattr(self, "prop_encoders", ())):
            h = self.prop_encoders[proposal](x, bound=self.bound)
            sigma = trunc_exp(self.prop_mlp[proposal](h).squeeze(-1))
        else:
            sigma, _ = self.common_forward(x)
        return {"sigma": sigma}

    def apply_total_variation(self, w):
        self.grid_encoder.grad_total_variation(w)

    def apply_weight_decay(self, w):
        self.grid_encoder.grad_weight_decay(w)

    def update_annealing(self, new_value):
        self.annealing = new_value

    def get_params(self, lr):
        groups = [self.grid_encoder, self.grid_mlp, self.view_mlp]
        if not self.opt.cuda_ray:
            groups += [self.prop_encoders, self.prop_mlp]
        return [{"params": g.parameters(), "lr": lr} for g in groups]
