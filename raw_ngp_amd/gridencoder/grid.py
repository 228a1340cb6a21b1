"""Multiresolution hash-grid encoder: autograd op + module.

Host-side mirror of the reference's gridencoder/grid.py (same public names, argument orders,
state-dict keys and error behaviour): `_grid_encode` <-> grid.py:24-99, `GridEncoder` <->
grid.py:102-211.  The device work goes through `_backend` = raw_ngp_amd._lib.gridencoder_backend,
i.e. the C ABI ngp_grid_* functions of libngp_hip.so.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from .._lib import gridencoder_backend as _backend

_gridtype_to_id = {"hash": 0, "tiled": 1}
_interp_to_id = {"linear": 0, "smoothstep": 1}


class _grid_encode(Function):
    """inputs [B, D] in [0, 1], embeddings [rows, C], offsets [L+1] int32  ->  [B, L*C]."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                gridtype=0, align_corners=False, interpolation=0, max_level=None, slab=False):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)      # float64 here, narrowed to float32 at the ABI (grid.py:38)
        H = base_resolution
        max_level = L if max_level is None else min(max_level, L)

        # level-major slab, as the kernel writes it; levels >= max_level must read as zero
        alloc = torch.zeros if max_level < L else torch.empty
        outputs = alloc(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = alloc(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        # the Jacobian is private to this op: level-major [L, B, D*C] where the backward route allows it (coalesced)
        lm = calc_grad_inputs and _backend.level_major_jacobian(B, D, C, L)
        if lm:
            _backend.grid_encode_forward_jac(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx,
                                             gridtype, align_corners, interpolation, True)
        else:
            _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, max_level, S, H, dy_dx,
                                         gridtype, align_corners, interpolation)
        ctx.lm = lm

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, C, L, S, H, gridtype, interpolation, max_level)
        ctx.align_corners = align_corners
        ctx.slab = slab
        if slab:        # level-major [L, B, C] exactly as the kernel wrote it (consumed by the fused MLP)
            return outputs
        # feature f[:, l*C:(l+1)*C] belongs to level l (the BARF / BAA windows rely on it)
        return outputs.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation, max_level = ctx.dims

        if ctx.slab:
            grad = grad.contiguous()                                        # already [L, B, C]
        else:
            grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()        # [B, L*C] -> [L, B, C]
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None

        _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, max_level, S,
                                      H, dy_dx, grad_inputs, gridtype, ctx.align_corners, interpolation,
                                      dy_dx_level_major=ctx.lm)

        if grad_inputs is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


def level_table(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size):
    """Row offsets of every level (grid.py:124-134): min(2^log2T, res^D) rounded up to 8 rows."""
    cap = 2 ** log2_hashmap_size
    offsets, total = [], 0
    for level in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** level))
        rows = int(np.ceil(min(cap, res ** input_dim) / 8) * 8)
        offsets.append(total)
        total += rows
    offsets.append(total)
    return np.asarray(offsets, dtype=np.int32)


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:      # overrides per_level_scale (grid.py:107-108)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))

        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        offsets = level_table(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.n_params = self.offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        top = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {top} per_level_scale={self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners} interpolation={self.interpolation}")

    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(self, inputs, bound=1, max_level=None, slab=False):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim]
        (slab=True: the kernel's own level-major [num_levels, B, level_dim] layout, no permute copy)."""
        inputs = (inputs + bound) / (2 * bound)
        lead = list(inputs.shape[:-1])
        flat = inputs.view(-1, self.input_dim)
        out = grid_encode(flat, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                          flat.requires_grad, self.gridtype_id, self.align_corners, self.interp_id, max_level, slab)
        return out if slab else out.view(lead + [self.output_dim])

    def _grad_or_raise(self):
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        return self.embeddings.grad

    @torch.autocast("cuda", enabled=False)
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """Adds the TV sub-gradient at random (or given) positions to embeddings.grad in place."""
        grad = self._grad_or_raise()
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        _backend.grad_total_variation(inputs.contiguous(), self.embeddings, grad, self.offsets, weight, B,
                                      self.input_dim, self.embeddings.shape[1], self.offsets.shape[0] - 1,
                                      np.log2(self.per_level_scale), self.base_resolution, self.gridtype_id,
                                      self.align_corners)

    @torch.autocast("cuda", enabled=False)
    def grad_weight_decay(self, weight=0.1):
        """Level-wise mean-normalised weight decay added to embeddings.grad in place."""
        grad = self._grad_or_raise()
        _backend.grad_weight_decay(self.embeddings, grad, self.offsets, weight, self.embeddings.shape[0],
                                   self.embeddings.shape[1], self.offsets.shape[0] - 1)
