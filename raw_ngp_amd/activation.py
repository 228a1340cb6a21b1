"""trunc_exp: exp() whose backward clamps the argument (mirror of the reference's activation.py:9-21)."""
import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function


class _trunc_exp(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-80, 80))


trunc_exp = _trunc_exp.apply
