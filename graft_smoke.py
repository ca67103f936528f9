"""TEST INFRASTRUCTURE (used by __graft_entry__.smoke only; lives beside it, outside the product package, because it imports the
CPU oracle): one small invocation of the hot path on the GPU, checked against the oracle.  The tile-walking conv kernel is exercised
too (forced configuration), so a driver smoke run loads and runs both conv families."""
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
    qa, _ = K.weight_pack_q(oops.weight_norm(v, gn).to(device), None, 1)
    for cfg in (2, 13):                 # a conv1d_bf16.hip tile and the tile-walking kernel of conv1d_tw.hip, bf16x3 arithmetic
        yq = K.conv1d_forward(x.to(device), qa, 96, 5, 1, 2, force_cfg=cfg)
        torch.cuda.synchronize()
        eq = (yq.cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert eq < 6e-5, f"smoke: bf16x3 conv (configuration {cfg}) mismatch {eq}"
    import graft_smoke_model
    graft_smoke_model.run(device)
    print(f"smoke OK (conv rel err {err:.2e})")
