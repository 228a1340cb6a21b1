"""Spherical-harmonics direction encoder: autograd op + module.

Host-side mirror of the reference's shencoder/sphere_harmonics.py (`_sh_encoder` <-> :14-54,
`SHEncoder` <-> :60-90); device work = ngp_sh_encode_* of libngp_hip.so.
"""
import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from .._lib import shencoder_backend as _backend


class _sh_encoder(Function):
    """inputs [B, 3] (unit vectors) -> [B, degree^2]; Jacobian kept only when inputs need grad."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        n_out = degree ** 2
        outputs = torch.empty(B, n_out, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, D * n_out, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        _backend.sh_encode_forward(inputs, outputs, B, D, degree, dy_dx)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = (B, D, degree)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, D, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _backend.sh_encode_backward(grad.contiguous(), inputs, B, D, degree, dy_dx, grad_inputs)
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        """inputs [..., 3] in [-size, size]; normalised to unit length before encoding."""
        inputs = inputs / size
        # (floor: unused rows of a fixed-capacity sample arena hold zero vectors; valid directions are unaffected)
        inputs = inputs / torch.norm(inputs, dim=-1, keepdim=True).clamp_min(1e-30)
        lead = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        out = sh_encode(flat, self.degree, flat.requires_grad)
        return out.reshape(lead + [self.output_dim])
