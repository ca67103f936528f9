"""One small invocation of the hot path on the GPU, checked against the CPU oracle (used by __graft_entry__.smoke)."""
import torch


def run(device):
    from neuralsvb_amd import kernels as K
    from oracle import ops as oops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 48, 281, generator=g)
    v = torch.randn(96, 48, 5, generator=g) * 0.2
    gn = torch.rand(96, 1, 1, generator=g) + 0.5
    ref = oops.conv1d(x, oops.weight_norm(v, gn), None, 1, 2)
    pa, _ = K.weight_pack(v.to(device), gn.to(device))
    y = K.conv1d_forward(x.to(device), pa, 96, 5, 1, 2)
    torch.cuda.synchronize()
    err = (y.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-5, f"smoke: conv mismatch {err}"
    try:
        from neuralsvb_amd import smoke_model
    except ImportError:
        smoke_model = None
    if smoke_model is not None:
        smoke_model.run(device)
    print(f"smoke OK (conv rel err {err:.2e})")
