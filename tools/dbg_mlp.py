import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_fused_mlp import make_weights, torch_field
from raw_ngp_amd import _lib
mb = _lib.mlp_backend
for M in (33, 1000):
    W = [w.requires_grad_(True) for w in make_weights(2)]
    g = torch.Generator(device="cuda").manual_seed(100 + M)
    stride = M + 5
    enc = torch.randn(16, stride, 2, device="cuda", generator=g) * 0.5
    dirs = torch.randn(M, 3, device="cuda", generator=g)
    dsigma = torch.randn(M, device="cuda", generator=g) * 1e-3
    drgb = torch.randn(M, 3, device="cuda", generator=g) * 1e-3
    enc_bf = enc[:, :M].permute(1, 0, 2).reshape(M, 32).clone().requires_grad_(True)
    rs, rc, h, c = torch_field(enc_bf, dirs, W)
    ((rs * dsigma).sum() + (rc * drgb).sum()).backward()
    ref = enc_bf.grad.view(M, 16, 2).permute(1, 0, 2)
    image = torch.empty(mb.image_bytes(), dtype=torch.uint8, device="cuda")
    mb.prepare([w.detach() for w in W], image)
    denc = torch.zeros(16, stride, 2, device="cuda")
    dws = [torch.zeros_like(w) for w in W]
    mb.backward(enc, stride, dirs, dsigma, drgb, None, M, image, 1024.0, denc, dws)
    err = (denc[:, :M] - ref).permute(1, 0, 2).reshape(M, 32)
    per = err.norm(dim=1) / (ref.permute(1, 0, 2).reshape(M, 32).norm(dim=1) + 1e-30)
    print(M, "per-sample rel err: median %.2e max %.2e argmax %d" % (per.median(), per.max(), per.argmax()))
    print("  sigma of worst:", float(rs[per.argmax()]), "color", rc[per.argmax()].tolist(), "raw c", c[per.argmax()].tolist())
    print("  worst 5:", per.topk(5).values.tolist(), per.topk(5).indices.tolist())
    for k in range(6):
        print("  dW%d rel l2 %.2e" % (k + 1, float((dws[k] - W[k].grad).norm() / W[k].grad.norm())))
    for k in (0, 3):
        e = (dws[k] - W[k].grad).abs()
        rel = e / (W[k].grad.abs().max())
        print("  dW%d: err by row (max over cols):" % (k + 1), [round(float(v), 4) for v in rel.max(dim=1).values[:64]])
        print("  dW%d: err by col (max over rows):" % (k + 1), [round(float(v), 4) for v in rel.max(dim=0).values])
