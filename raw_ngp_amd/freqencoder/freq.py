"""NeRF frequency (sin/cos) encoder: autograd op + module.

Host-side mirror of the reference's freqencoder/freq.py; device work = ngp_freq_encode_*.
Not used by NeRFNetwork (encoding.py only offers it); kept so the native inventory is complete.
"""
import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from .._lib import freqencoder_backend as _backend


class _freq_encoder(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        if not inputs.is_cuda:
            inputs = inputs.cuda()
        inputs = inputs.contiguous()
        B, D = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        _backend.freq_encode_forward(inputs, B, D, degree, output_dim, outputs)
        ctx.save_for_backward(outputs)
        ctx.dims = (B, D, degree, output_dim)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        (outputs,) = ctx.saved_tensors
        B, D, degree, output_dim = ctx.dims
        grad_inputs = torch.zeros(B, D, dtype=outputs.dtype, device=outputs.device)
        _backend.freq_encode_backward(grad.contiguous(), outputs, B, D, degree, output_dim, grad_inputs)
        return grad_inputs, None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        lead = list(inputs.shape[:-1])
        out = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return out.reshape(lead + [self.output_dim])
