"""Autograd wrapper of the fused MFMA tiny-MLP field (csrc/fused_mlp*.hip).

Evaluates what NeRFNetwork.forward computes after the hash-grid encoder for the default configuration
(network.py:111-138 of the reference: grid_mlp -> trunc_exp density + 15 features; SH(d); view_mlp ->
clamped_exp colour) in two kernels (forward) / three kernels (backward) instead of six GEMMs and a dozen
elementwise kernels, in the precision of the reference's --fp16 path (f16 operands, f32 accumulation)."""
import torch
from torch.autograd import Function

from .._lib import mlp_backend


class _FusedField(Function):
    @staticmethod
    def forward(ctx, enc, dirs, loss_scale, act, w1, w2, w3, w4, w5, w6):
        """enc [16, B, 2] level-major hash features, dirs [B, 3] -> sigma [B], rgb [B, 3]."""
        enc = enc.contiguous()
        dirs = dirs.contiguous().float()
        B = enc.shape[1]
        weights = [w.detach().float().contiguous() for w in (w1, w2, w3, w4, w5, w6)]
        image = torch.empty(mlp_backend.image_bytes(), dtype=torch.uint8, device=enc.device)
        mlp_backend.prepare(weights, image)
        sigma = torch.empty(B, dtype=torch.float32, device=enc.device)
        rgb = torch.empty(B, 3, dtype=torch.float32, device=enc.device)
        mlp_backend.forward(enc, B, dirs, None, B, image, sigma, rgb, act=act)
        ctx.save_for_backward(enc, dirs, image, *weights)
        ctx.loss_scale, ctx.act = loss_scale, act
        return sigma, rgb

    @staticmethod
    def backward(ctx, dsigma, drgb):
        enc, dirs, image, *weights = ctx.saved_tensors
        B = enc.shape[1]
        denc = torch.empty_like(enc)
        dws = [torch.empty_like(w) for w in weights]
        dsigma = dsigma.contiguous().float() if dsigma is not None else torch.zeros(B, device=enc.device)
        drgb = drgb.contiguous().float() if drgb is not None else torch.zeros(B, 3, device=enc.device)
        mlp_backend.backward(enc, B, dirs, dsigma, drgb, None, B, image, ctx.loss_scale, denc, dws, act=ctx.act)
        return (denc, None, None, None) + tuple(dws)


def fused_field(enc, dirs, weights, loss_scale=65536.0, act=None):
    """act = (color_act, density_act, beta) from _lib.field_activations(opt); None: the default activations."""
    return _FusedField.apply(enc, dirs, loss_scale, act, *weights)


@torch.no_grad()
def fused_density(enc, weights, act=None):
    """sigma only (density-grid refresh): the colour MLP is skipped."""
    enc = enc.contiguous()
    B = enc.shape[1]
    image = torch.empty(mlp_backend.image_bytes(), dtype=torch.uint8, device=enc.device)
    mlp_backend.prepare([w.detach().float().contiguous() for w in weights], image)
    sigma = torch.empty(B, dtype=torch.float32, device=enc.device)
    mlp_backend.forward(enc, B, None, None, B, image, sigma, None, act=act)
    return sigma
