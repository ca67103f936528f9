"""The pre-split activation layout ("Q image", csrc/svb_q.h) of the bf16x3 convs: layout, producers, and bit-identity of
the convs fed from it with the convs that split fp32 input themselves.  emu (CPU) + gpu."""
import numpy as np
import pytest
import torch

from neuralsvb_amd import kernels as K


def _bf16_rne(v):
    """numpy float32 -> bf16 (round to nearest even) as uint16 bit patterns + the float32 value."""
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return r.astype(np.uint16), (r << 16).astype(np.uint32).view(np.float32)


def q_ref(x):
    """numpy restatement of svb_split_q: [B,C,T] fp32 -> [B, ceil(C/16), T, 32] uint16."""
    B, C, T = x.shape
    kc = -(-C // 16)
    xp = np.zeros((B, kc * 16, T), np.float32)
    xp[:, :C] = x
    hi, hif = _bf16_rne(xp)
    lo, _ = _bf16_rne(xp - hif)
    hi = hi.reshape(B, kc, 16, T).transpose(0, 1, 3, 2)
    lo = lo.reshape(B, kc, 16, T).transpose(0, 1, 3, 2)
    return np.concatenate([hi, lo], -1)


def as_u16(q):
    return q.cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("B,C,T", [(2, 16, 8), (1, 40, 37), (3, 192, 52)])
def test_split_q_layout(dev, B, C, T):
    g = torch.Generator().manual_seed(B * 100 + C)
    x = torch.randn(B, C, T, generator=g) * 3
    x[0, 0, :3] = torch.tensor([0.0, 1e-30, -65504.0])[:min(3, T)]
    assert np.array_equal(as_u16(K.split_q(x.to(dev))), q_ref(x.numpy()))
    m = (torch.rand(B, T, generator=g) > 0.3).float()
    assert np.array_equal(as_u16(K.split_q(x.to(dev), m.to(dev))), q_ref((x * m[:, None]).numpy()))


@pytest.mark.parametrize("T", [36, 41])            # 16-byte vector path and the scalar path
def test_gated_stack_producers_emit_q_of_their_result(dev, T):
    g = torch.Generator().manual_seed(5)
    B, C, gch = 2, 32, 96
    xin = torch.randn(B, 2 * C, T, generator=g).to(dev)
    G = torch.randn(B, gch, T, generator=g).to(dev)
    acts, acts_q = K.wn_gate_fwd(xin, G, 16, want_q=True)
    assert torch.equal(acts, K.wn_gate_fwd(xin, G, 16))
    assert np.array_equal(as_u16(acts_q), q_ref(acts.cpu().numpy()))
    dacts = torch.randn(B, C, T, generator=g).to(dev)
    dG = torch.zeros_like(G)
    dxin, dxin_q = K.wn_gate_bwd(xin, G, dacts, 16, dg=dG, want_q=True)
    assert torch.equal(dxin, K.wn_gate_bwd(xin, G, dacts, 16))
    assert torch.equal(dG[:, 16:16 + 2 * C], dxin)
    assert np.array_equal(as_u16(dxin_q), q_ref(dxin.cpu().numpy()))
    x = torch.randn(B, C, T, generator=g).to(dev)
    rs = torch.randn(B, 2 * C, T, generator=g).to(dev)
    out = torch.randn(B, C, T, generator=g).to(dev)
    mask = (torch.rand(B, T, generator=g) > 0.3).float().to(dev)
    xn, on, xnq = K.wn_res_skip(x, rs, mask, out, False, want_q=True)
    xn0, on0 = K.wn_res_skip(x, rs, mask, out, False)
    assert torch.equal(xn, xn0) and torch.equal(on, on0)
    assert torch.allclose(xn.cpu(), ((x + rs[:, :C]) * mask[:, None]).cpu()) and torch.allclose(on.cpu(), (out + rs[:, C:]).cpu())
    assert np.array_equal(as_u16(xnq), q_ref(xn.cpu().numpy()))
    _, ol = K.wn_res_skip(None, rs[:, :C].contiguous(), mask, out, True)
    assert torch.allclose(ol.cpu(), (out + rs[:, :C]).cpu())
    dxn, dout = torch.randn(B, C, T, generator=g).to(dev), torch.randn(B, C, T, generator=g).to(dev)
    drs, dxm, drs_q = K.wn_res_skip_bwd(dxn, dout, mask, want_q=True)
    assert torch.equal(drs[:, :C], dxn * mask[:, None]) and torch.equal(drs[:, C:], dout) and torch.equal(dxm, drs[:, :C])
    assert np.array_equal(as_u16(drs_q), q_ref(drs.cpu().numpy()))


CASES = [  # B, Cin, Cout, T, k, stride, pad, dil, groups
    (2, 32, 48, 70, 5, 1, 2, 1, 1),
    (1, 48, 40, 133, 1, 1, 0, 1, 1),
    (2, 40, 64, 61, 3, 1, 2, 2, 1),          # Cin not a multiple of 16: zero-padded last chunk
    (2, 32, 32, 64, 8, 4, 2, 1, 1),          # strided
    (2, 64, 64, 50, 3, 1, 1, 1, 2),          # grouped, 32 channels per group
]


@pytest.mark.parametrize("case", CASES)
def test_conv_from_q_image_is_bit_identical(dev, case):
    B, cin, cout, T, k, s, pad, dil, groups = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, cin, T, generator=g).to(dev)
    w = (torch.randn(cout, cin // groups, k, generator=g) * 0.2).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    qa, qb = K.weight_pack_q(w, None, groups)
    xq = K.split_q(x)
    for cfg in range(1, 13):
        y0 = K.conv1d_forward(x, qa, cout, k, s, pad, dil, groups, bias=bias, force_cfg=cfg)
        y1 = K.conv1d_forward(x, qa, cout, k, s, pad, dil, groups, bias=bias, force_cfg=cfg, x_q=xq)
        assert torch.equal(y0, y1), (case, cfg, (y0 - y1).abs().max().item())
    # data gradient (transposed form) from the Q image of dy
    dy = torch.randn_like(y0)
    dyq = K.split_q(dy)
    res = torch.randn(B, cin, T, generator=g).to(dev)
    for cfg in (1, 2, 4):
        d0 = K.conv1d_transposed(dy, qb, cin, T, k, s, pad, dil, groups, residual=res, force_cfg=cfg)
        d1 = K.conv1d_transposed(dy, qb, cin, T, k, s, pad, dil, groups, residual=res, force_cfg=cfg, x_q=dyq)
        assert torch.equal(d0, d1), (case, cfg, (d0 - d1).abs().max().item())


def test_gated_stack_with_q_images_is_bit_identical(dev):
    """functional.USE_Q: the whole WN stack (forward, data gradients, weight gradients) gives identical bits either way."""
    from neuralsvb_amd import functional as SF
    g_ = torch.Generator().manual_seed(9)
    B, C, T, gin, n, ks = 2, 16, 44, 20, 2, 5
    x = torch.randn(B, C, T, generator=g_)
    gcond = torch.randn(B, gin, T, generator=g_)
    mask = (torch.rand(B, T, generator=g_) > 0.2).float()
    cond = [torch.randn(2 * C * n, gin, 1, generator=g_) * 0.3, torch.rand(2 * C * n, 1, 1, generator=g_) + 0.5,
            torch.randn(2 * C * n, generator=g_) * 0.1]
    layers = []
    for i in range(n):
        rc = 2 * C if i < n - 1 else C
        layers.append([torch.randn(2 * C, C, ks, generator=g_) * 0.3, torch.rand(2 * C, 1, 1, generator=g_) + 0.5,
                       torch.randn(2 * C, generator=g_) * 0.1, torch.randn(rc, C, 1, generator=g_) * 0.3,
                       torch.rand(rc, 1, 1, generator=g_) + 0.5, torch.randn(rc, generator=g_) * 0.1])
    dy = torch.randn(B, C, T, generator=g_)
    res = []
    SF.set_precision("bf16x3")
    try:
        for useq in (False, True):
            SF.USE_Q = useq
            leaves = [t.clone().to(dev).requires_grad_(True) for t in [x, gcond] + cond + [t for lp in layers for t in lp]]
            ld = [leaves[5 + 6 * i: 11 + 6 * i] for i in range(n)]
            y = SF.wn_stack(leaves[0], mask.to(dev), leaves[1], leaves[2:5], ld, ks)
            y.backward(dy.to(dev))
            res.append([y.detach()] + [t.grad for t in leaves])
    finally:
        SF.USE_Q = False
        SF.set_precision("fp32")
    for a, b in zip(*res):
        assert torch.equal(a, b)
