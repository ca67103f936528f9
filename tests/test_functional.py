"""Autograd-level parity: neuralsvb_amd.functional (HIP forward + HIP backward) vs torch autograd over the
op-level oracle.  Runs on the emulator (CPU) and on the MI355X (gpu mark)."""
import pytest
import torch
import torch.nn.functional as F

from neuralsvb_amd import functional as SF
from oracle import ops as oops
from tests.test_kernels import rel_err


def _leaf(t, dev):
    return t.detach().clone().to(dev).requires_grad_(True)


@pytest.mark.parametrize("wn", [False, True])
@pytest.mark.parametrize("variant", ["plain", "relu", "lrelu_in_res", "mask", "strided_grouped"])
def test_conv1d_autograd(dev, wn, variant):
    g_ = torch.Generator().manual_seed(17)
    B, Cin, Cout, T, k = 2, 12, 16, 41, 5
    stride, pad, dil, groups = 1, 2, 1, 1
    kw = {}
    if variant == "strided_grouped":
        stride, pad, groups = 2, 1, 2
        k = 3
    x = torch.randn(B, Cin, T, generator=g_)
    v = torch.randn(Cout, Cin // groups, k, generator=g_) * 0.3
    gn = torch.rand(Cout, 1, 1, generator=g_) + 0.5
    b = torch.randn(Cout, generator=g_)
    Tout = (T + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = torch.randn(B, Cout, Tout, generator=g_)
    mask = (torch.rand(B, Tout, generator=g_) > 0.3).float()

    def ref_fn(x, v, gn, b, res):
        w = oops.weight_norm(v, gn) if wn else v
        if variant == "relu":
            return torch.relu(oops.conv1d(x, w, b, stride, pad, dil, groups))
        if variant == "lrelu_in_res":
            return oops.conv1d(F.leaky_relu(x, 0.1), w, b, stride, pad, dil, groups) + res
        if variant == "mask":
            return oops.conv1d(x, w, b, stride, pad, dil, groups) * mask[:, None]
        return oops.conv1d(x, w, b, stride, pad, dil, groups)

    rl = [t.clone().requires_grad_(True) for t in (x, v, gn, b, res)]
    yr = ref_fn(*rl)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)

    xd, vd, gd, bd, rd = (_leaf(t, dev) for t in (x, v, gn, b, res))
    if variant == "relu":
        kw = dict(out_act=SF.ACT_RELU)
    elif variant == "lrelu_in_res":
        kw = dict(in_slope=0.1, residual=rd)
    elif variant == "mask":
        kw = dict(mask=mask.to(dev))
    y = SF.conv1d(xd, vd, bd, stride, pad, dil, groups, weight_g=gd if wn else None, **kw)
    assert rel_err(y, yr) < 2e-5
    y.backward(dy.to(dev))
    assert rel_err(xd.grad, rl[0].grad) < 3e-5
    assert rel_err(vd.grad, rl[1].grad) < 5e-5
    assert rel_err(bd.grad, rl[3].grad) < 3e-5
    if wn:
        assert rel_err(gd.grad, rl[2].grad) < 5e-5
    if variant == "lrelu_in_res":
        assert rel_err(rd.grad, rl[4].grad) < 1e-6


@pytest.mark.parametrize("wn", [False, True])
def test_conv_transpose1d_autograd(dev, wn):
    g_ = torch.Generator().manual_seed(23)
    B, Cin, Cout, T, k, s, pad = 2, 10, 6, 19, 8, 4, 2
    x = torch.randn(B, Cin, T, generator=g_)
    v = torch.randn(Cin, Cout, k, generator=g_) * 0.3
    gn = torch.rand(Cin, 1, 1, generator=g_) + 0.5
    b = torch.randn(Cout, generator=g_)
    rl = [t.clone().requires_grad_(True) for t in (x, v, gn, b)]
    w = oops.weight_norm(rl[1], rl[2]) if wn else rl[1]
    yr = oops.conv_transpose1d(F.leaky_relu(rl[0], 0.1), w, rl[3], s, pad)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    xd, vd, gd, bd = (_leaf(t, dev) for t in (x, v, gn, b))
    y = SF.conv_transpose1d(xd, vd, bd, s, pad, weight_g=gd if wn else None, in_slope=0.1)
    assert rel_err(y, yr) < 2e-5
    y.backward(dy.to(dev))
    assert rel_err(xd.grad, rl[0].grad) < 3e-5
    assert rel_err(vd.grad, rl[1].grad) < 5e-5
    assert rel_err(bd.grad, rl[3].grad) < 3e-5
    if wn:
        assert rel_err(gd.grad, rl[2].grad) < 5e-5


def _wn_ref(x, mask, gcond, cond, layers, ks):
    """WN.forward restated with stock torch ops (reference modules/fastspeech/fs2_vae.py:61-91)."""
    C = x.shape[1]
    m = mask[:, None, :] if mask is not None else 1
    out = torch.zeros_like(x)
    G = oops.conv1d(gcond, oops.weight_norm(cond[0], cond[1]), cond[2]) if gcond is not None else None
    n = len(layers)
    for i, (iv, ig, ib, rv, rg, rb) in enumerate(layers):
        xin = oops.conv1d(x, oops.weight_norm(iv, ig), ib, 1, (ks - 1) // 2)
        gl = G[:, i * 2 * C:(i + 1) * 2 * C] if G is not None else torch.zeros_like(xin)
        acts = oops.wn_gate(xin, gl)
        rs = oops.conv1d(acts, oops.weight_norm(rv, rg), rb)
        if i < n - 1:
            x = (x + rs[:, :C]) * m
            out = out + rs[:, C:]
        else:
            out = out + rs
    return out * m


@pytest.mark.parametrize("with_cond,with_mask", [(True, True), (False, False)])
def test_wn_stack_autograd(dev, with_cond, with_mask):
    g_ = torch.Generator().manual_seed(31)
    B, C, T, gin, n, ks = 2, 8, 45, 10, 3, 5
    x = torch.randn(B, C, T, generator=g_)
    gcond = torch.randn(B, gin, T, generator=g_) if with_cond else None
    mask = (torch.rand(B, T, generator=g_) > 0.25).float() if with_mask else None
    cond = [torch.randn(2 * C * n, gin, 1, generator=g_) * 0.3, torch.rand(2 * C * n, 1, 1, generator=g_) + 0.5,
            torch.randn(2 * C * n, generator=g_) * 0.1]
    layers = []
    for i in range(n):
        rc = 2 * C if i < n - 1 else C
        layers.append([torch.randn(2 * C, C, ks, generator=g_) * 0.3, torch.rand(2 * C, 1, 1, generator=g_) + 0.5,
                       torch.randn(2 * C, generator=g_) * 0.1,
                       torch.randn(rc, C, 1, generator=g_) * 0.3, torch.rand(rc, 1, 1, generator=g_) + 0.5,
                       torch.randn(rc, generator=g_) * 0.1])
    # reference
    xr = x.clone().requires_grad_(True)
    gr = gcond.clone().requires_grad_(True) if with_cond else None
    cr = [t.clone().requires_grad_(True) for t in cond]
    lr = [[t.clone().requires_grad_(True) for t in lp] for lp in layers]
    yr = _wn_ref(xr, mask, gr, cr, lr, ks)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    # HIP
    xd = _leaf(x, dev)
    gd = _leaf(gcond, dev) if with_cond else None
    cd = [_leaf(t, dev) for t in cond]
    ld = [[_leaf(t, dev) for t in lp] for lp in layers]
    y = SF.wn_stack(xd, mask.to(dev) if with_mask else None, gd, cd if with_cond else None, ld, ks)
    assert rel_err(y, yr) < 3e-5
    y.backward(dy.to(dev))
    assert rel_err(xd.grad, xr.grad) < 5e-5
    if with_cond:
        assert rel_err(gd.grad, gr.grad) < 5e-5
        for a, b in zip(cd, cr):
            assert rel_err(a.grad, b.grad) < 1e-4
    for la, lb in zip(ld, lr):
        for a, b in zip(la, lb):
            assert rel_err(a.grad, b.grad) < 1e-4


def test_layer_norm_autograd(dev):
    g_ = torch.Generator().manual_seed(2)
    x = torch.randn(2, 9, 64, generator=g_)
    gm, bt = torch.randn(64, generator=g_), torch.randn(64, generator=g_)
    rl = [t.clone().requires_grad_(True) for t in (x, gm, bt)]
    yr = oops.layernorm(*rl)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    dl = [_leaf(t, dev) for t in (x, gm, bt)]
    y = SF.layer_norm(*dl)
    y.backward(dy.to(dev))
    assert (y.detach().cpu() - yr.detach()).abs().max() < 2e-5
    for a, b in zip(dl, rl):
        assert rel_err(a.grad, b.grad) < 3e-5


def test_conv2d_lrelu_autograd(dev):
    """mel-critic block: Conv2d(c,128,3x3,s2,p1)+LeakyReLU(0.2) (multi_window_disc.py:14-22) via im2col + GEMM kernel."""
    g_ = torch.Generator().manual_seed(41)
    for cin, H, W in ((1, 32, 80), (6, 16, 40), (6, 5, 10)):
        x = torch.randn(2, cin, H, W, generator=g_)
        w = torch.randn(10, cin, 3, 3, generator=g_) * 0.3
        b = torch.randn(10, generator=g_)
        rl = [t.clone().requires_grad_(True) for t in (x, w, b)]
        yr = F.leaky_relu(F.conv2d(rl[0], rl[1], rl[2], 2, 1), 0.2)
        dy = torch.randn(yr.shape, generator=g_)
        yr.backward(dy)
        dl = [_leaf(t, dev) for t in (x, w, b)]
        y = SF.conv2d_lrelu(dl[0], dl[1], dl[2], 2, 1, 0.2)
        assert y.shape == yr.shape and rel_err(y, yr) < 2e-5
        y.backward(dy.to(dev))
        for a, r in zip(dl, rl):
            assert rel_err(a.grad, r.grad) < 3e-5


def test_critic_block_autograd(dev):
    """Whole critic block -- Conv2d(3x3,s2,p1) -> LeakyReLU(0.2) -> Dropout2d -> InstanceNorm2d(affine)
    (multi_window_disc.py:14-31) -- as space-to-depth conv + one crop/drop/norm pass, against stock torch with the SAME
    keep mask; with and without the norm, with and without dropout; gradients of input, kernel, bias, gamma, beta."""
    g_ = torch.Generator().manual_seed(43)
    for cin, H, W, norm, p in ((1, 32, 80, False, 0.25), (6, 16, 40, True, 0.25), (6, 8, 20, True, 0.0), (5, 4, 10, True, 0.5)):
        N, cout = 3, 10
        x = torch.randn(N, cin, H, W, generator=g_)
        w = torch.randn(cout, cin, 3, 3, generator=g_) * 0.3
        b = torch.randn(cout, generator=g_)
        gm, bt = torch.rand(cout, generator=g_) + 0.5, torch.randn(cout, generator=g_)
        keep = (torch.rand(N, cout, generator=g_) >= p).float() / (1.0 - p)
        rl = [t.clone().requires_grad_(True) for t in (x, w, b, gm, bt)]
        yr = F.leaky_relu(F.conv2d(rl[0], rl[1], rl[2], 2, 1), 0.2) * keep[:, :, None, None]
        if norm:
            yr = F.instance_norm(yr, weight=rl[3], bias=rl[4], eps=1e-5)
        dy = torch.randn(yr.shape, generator=g_)
        yr.backward(dy)
        dl = [_leaf(t, dev) for t in (x, w, b, gm, bt)]
        old = SF.dropout2d_keep
        SF.dropout2d_keep = lambda n, c, pp, device: keep.to(device)
        try:
            y = SF.critic_block(dl[0], dl[1], dl[2], 0.2, p, dl[3] if norm else None, dl[4] if norm else None)
        finally:
            SF.dropout2d_keep = old
        assert y.shape == yr.shape and rel_err(y, yr) < 3e-5
        y.backward(dy.to(dev))
        for a, r in list(zip(dl, rl))[:5 if norm else 3]:
            assert rel_err(a.grad, r.grad) < 1e-4, (cin, H, W, norm, p)


def test_critic_blocks_chained_through_the_conv_layout(dev):
    """Two blocks where the first writes the second's space-to-depth conv input directly (s2d_out / planes): outputs and all
    gradients equal the plain composition's (stock torch)."""
    g_ = torch.Generator().manual_seed(47)
    N, c0, c1, c2, H, W, p = 3, 1, 6, 5, 16, 40, 0.25
    x = torch.randn(N, c0, H, W, generator=g_)
    w1, b1 = torch.randn(c1, c0, 3, 3, generator=g_) * 0.4, torch.randn(c1, generator=g_)
    w2, b2 = torch.randn(c2, c1, 3, 3, generator=g_) * 0.3, torch.randn(c2, generator=g_)
    gm, bt = torch.rand(c2, generator=g_) + 0.5, torch.randn(c2, generator=g_)
    keeps = [(torch.rand(N, c, generator=g_) >= p).float() / (1.0 - p) for c in (c1, c2)]
    rl = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2, gm, bt)]
    h1 = F.leaky_relu(F.conv2d(rl[0], rl[1], rl[2], 2, 1), 0.2) * keeps[0][:, :, None, None]
    yr = F.instance_norm(F.leaky_relu(F.conv2d(h1, rl[3], rl[4], 2, 1), 0.2) * keeps[1][:, :, None, None], weight=rl[5],
                         bias=rl[6], eps=1e-5)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    dl = [_leaf(t, dev) for t in (x, w1, b1, w2, b2, gm, bt)]
    it = iter(keeps)
    old = SF.dropout2d_keep
    SF.dropout2d_keep = lambda n, c, pp, device: next(it).to(device)
    try:
        x4, planes = SF.critic_block(dl[0], dl[1], dl[2], 0.2, p, None, None, s2d_out=True)
        assert planes == (N, c1, H // 2, W // 2)
        y = SF.critic_block(x4, dl[3], dl[4], 0.2, p, dl[5], dl[6], planes=planes)
    finally:
        SF.dropout2d_keep = old
    assert y.shape == yr.shape and rel_err(y, yr) < 3e-5
    y.backward(dy.to(dev))
    for a, r in zip(dl, rl):
        assert rel_err(a.grad, r.grad) < 1e-4


def test_critic_block_accumulates_into_grad_buffers(dev):
    """A kernel that already owns a .grad buffer gets its gradient added there by the gather kernel (no autograd add)."""
    g_ = torch.Generator().manual_seed(44)
    x = torch.randn(2, 4, 8, 10, generator=g_)
    w = torch.randn(6, 4, 3, 3, generator=g_) * 0.3
    rl = [t.clone().requires_grad_(True) for t in (x, w)]
    yr = F.leaky_relu(F.conv2d(rl[0], rl[1], None, 2, 1), 0.2)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    xd, wd = _leaf(x, dev), _leaf(w, dev)
    wd.grad = torch.full_like(wd, 0.5)
    buf = wd.grad
    SF.critic_block(xd, wd, None, 0.2, 0.0, None, None).backward(dy.to(dev))
    assert wd.grad is buf and rel_err(wd.grad - 0.5, rl[1].grad) < 1e-4


def test_critic_block_sees_fused_optimizer_updates(dev):
    """A fused AdamW step changes the kernel without moving its version counter: inside a Trainer-managed weight epoch the
    re-laid-out kernel image must follow note_weights_updated(); outside one it is rebuilt on every call."""
    g_ = torch.Generator().manual_seed(46)
    x = torch.randn(2, 3, 8, 10, generator=g_)
    w = torch.randn(5, 3, 3, 3, generator=g_) * 0.3
    xd, wd = x.to(dev), _leaf(w, dev)
    opt = torch.optim.AdamW([wd], lr=0.05, fused=True)

    def ref():
        return F.leaky_relu(F.conv2d(x, wd.detach().cpu(), None, 2, 1), 0.2)

    def run():
        return SF.critic_block(xd, wd, None, 0.2, 0.0, None, None)
    for managed in (True, False):
        if managed:
            SF.begin_weight_epoch()
        try:
            y = run()
            assert rel_err(y, ref()) < 3e-5
            y.sum().backward()
            ver = wd._version
            opt.step()
            assert wd._version == ver                       # the premise of this test
            if managed:
                SF.note_weights_updated([wd])
            assert rel_err(run(), ref()) < 3e-5
        finally:
            SF.end_weight_epoch()


def test_embedding_nct_autograd(dev):
    """pitch_embed(pitch).transpose(1, 2) (svb_vae.py:66; nn.Embedding(300, H, padding_idx=0)) as one gather, with the
    deterministic weight gradient: padding row zero, repeated bins summed, two runs bit-identical."""
    g_ = torch.Generator().manual_seed(48)
    B, T, V, H = 3, 150, 300, 20
    idx = torch.randint(0, 12, (B, T), generator=g_)
    idx[0, 100:] = 0                                   # padding frames
    idx[1, :5] = 299
    w = torch.randn(V, H, generator=g_)
    wr = w.clone().requires_grad_(True)
    yr = F.embedding(idx, wr, padding_idx=0).transpose(1, 2)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    grads = []
    for _ in range(2):
        wd = _leaf(w, dev)
        y = SF.embedding_nct(idx.to(dev), wd, 0)
        assert torch.equal(y.cpu(), yr.detach())
        y.backward(dy.to(dev))
        grads.append(wd.grad.cpu())
    assert torch.equal(grads[0], grads[1])
    assert rel_err(grads[0], wr.grad) < 1e-5 and float(grads[0][0].abs().max()) == 0.0


def test_dropout2d_keep_is_a_bernoulli_field(dev):
    k = SF.dropout2d_keep(64, 128, 0.25, dev)
    vals = sorted(torch.unique(k).tolist())
    assert k.shape == (64, 128) and len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / 0.75) < 1e-6
    assert abs(float((k > 0).float().mean()) - 0.75) < 0.03


def test_dropout2d_pool_serves_the_fields_of_one_draw(dev):
    """SF.dropout2d_pool: the Dropout2d fields of a stacked critic pass come out of ONE Bernoulli draw, in order, disjoint;
    a field the pool cannot serve (other p, pool exhausted) is drawn on its own; a replaced dropout2d_keep (mask replay in the
    golden-step tests) opens no pool and consumes no draw."""
    shapes = [(6, 32), (6, 31), (6, 64)]
    with SF.dropout2d_pool(shapes, 0.25, dev) as pool:
        assert pool.opened
        flat = SF._KEEP_POOL[0]
        ks = [SF.dropout2d_keep(n, c, 0.25, dev) for n, c in shapes]
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        spans = [(k.data_ptr(), k.data_ptr() + k.numel() * 4) for k in ks]
        assert all(lo <= a and b <= hi and a % 16 == lo % 16 for a, b in spans)
        assert all(spans[i][1] <= spans[i + 1][0] for i in range(2))
        other = SF.dropout2d_keep(6, 8, 0.5, dev)                      # another p: its own draw
        extra = SF.dropout2d_keep(6, 64, 0.25, dev)                    # beyond the reserved fields: its own draw
        assert not (lo <= other.data_ptr() < hi) and not (lo <= extra.data_ptr() < hi)
    assert SF._KEEP_POOL is None
    for k, (n, c) in zip(ks, shapes):
        vals = sorted(torch.unique(k).tolist())
        assert k.shape == (n, c) and k.is_contiguous() and vals[0] == 0.0 and abs(vals[-1] - 1.0 / 0.75) < 1e-6
    allk = torch.cat([k.flatten() for k in ks])
    assert abs(float((allk > 0).float().mean()) - 0.75) < 0.05
    old = SF.dropout2d_keep
    SF.dropout2d_keep = lambda n, c, pp, device: torch.ones(n, c, device=device)
    try:
        with SF.dropout2d_pool(shapes, 0.25, dev) as pool:
            assert not pool.opened and SF._KEEP_POOL is None
    finally:
        SF.dropout2d_keep = old
    with SF.dropout2d_pool([], 0.0, dev) as pool:
        assert not pool.opened


def test_plane_score_autograd(dev):
    """adv_layer (nn.Linear(C*H*W, 1), multi_window_disc.py:62-64) on channel-major feature maps."""
    g_ = torch.Generator().manual_seed(45)
    N, C, H, W = 5, 12, 4, 10
    base = torch.randn(C, N, H, W, generator=g_)
    wt, bs = torch.randn(1, C * H * W, generator=g_) * 0.1, torch.randn(1, generator=g_)
    hr = base.permute(1, 0, 2, 3).clone().requires_grad_(True)
    rl = [hr, wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)]
    yr = F.linear(hr.flatten(1), rl[1], rl[2])
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    hb = base.to(dev).requires_grad_(True)
    wd, bd = _leaf(wt, dev), _leaf(bs, dev)
    y = SF.plane_score(hb.permute(1, 0, 2, 3), wd, bd)
    assert y.shape == yr.shape and rel_err(y, yr) < 2e-5
    y.backward(dy.to(dev))
    assert rel_err(hb.grad.permute(1, 0, 2, 3), hr.grad) < 2e-5
    assert rel_err(wd.grad, rl[1].grad) < 2e-5 and rel_err(bd.grad, rl[2].grad) < 2e-5
    # the score's cotangent as a strided column of the stacked [N,1,windows] scores (no re-layout copy in backward)
    hb2, wd2, bd2 = base.detach().clone().to(dev).requires_grad_(True), _leaf(wt, dev), _leaf(bs, dev)
    y3 = torch.stack([SF.plane_score(hb2.permute(1, 0, 2, 3), wd2, bd2), torch.zeros(N, 1, device=dev)], -1)
    dy3 = torch.zeros(N, 1, 2)
    dy3[:, :, 0] = dy
    y3.backward(dy3.to(dev))
    assert rel_err(hb2.grad.permute(1, 0, 2, 3), hr.grad) < 2e-5 and rel_err(wd2.grad, rl[1].grad) < 2e-5


def test_wn_stack_bf16x3_mode(dev):
    """WN stack with forward + data-gradient convs in bf16x3 mode (weight gradients stay fp32): 2e-4 relative."""
    g_ = torch.Generator().manual_seed(33)
    B, C, T, gin, n, ks = 2, 16, 50, 20, 2, 5
    x = torch.randn(B, C, T, generator=g_)
    gcond = torch.randn(B, gin, T, generator=g_)
    cond = [torch.randn(2 * C * n, gin, 1, generator=g_) * 0.3, torch.rand(2 * C * n, 1, 1, generator=g_) + 0.5,
            torch.randn(2 * C * n, generator=g_) * 0.1]
    layers = []
    for i in range(n):
        rc = 2 * C if i < n - 1 else C
        layers.append([torch.randn(2 * C, C, ks, generator=g_) * 0.3, torch.rand(2 * C, 1, 1, generator=g_) + 0.5,
                       torch.randn(2 * C, generator=g_) * 0.1, torch.randn(rc, C, 1, generator=g_) * 0.3,
                       torch.rand(rc, 1, 1, generator=g_) + 0.5, torch.randn(rc, generator=g_) * 0.1])
    xr = x.clone().requires_grad_(True)
    lr = [[t.clone().requires_grad_(True) for t in lp] for lp in layers]
    cr = [t.clone().requires_grad_(True) for t in cond]
    yr = _wn_ref(xr, None, gcond, cr, lr, ks)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    SF.set_precision("bf16x3")
    try:
        xd = _leaf(x, dev)
        cd = [_leaf(t, dev) for t in cond]
        ld = [[_leaf(t, dev) for t in lp] for lp in layers]
        y = SF.wn_stack(xd, None, gcond.to(dev), cd, ld, ks)
        y.backward(dy.to(dev))
    finally:
        SF.set_precision("fp32")
    assert rel_err(y, yr) < 2e-4
    assert rel_err(xd.grad, xr.grad) < 2e-4
    for la, lb in zip(ld, lr):
        for a, b in zip(la, lb):
            assert rel_err(a.grad, b.grad) < 3e-4


def test_weight_pack_cache_semantics(dev):
    """Packed weight images: frozen weights are packed once; trainable weights are repacked on every call unless a
    Trainer-managed weight epoch is open, and an epoch bump (optimizer step) invalidates them."""
    from neuralsvb_amd import kernels as K
    calls = {"n": 0}
    orig = K.weight_pack

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    K.weight_pack = counting
    try:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(1, 4, 20, generator=g).to(dev)
        frozen = torch.nn.Parameter((torch.randn(6, 4, 3, generator=g) * 0.3).to(dev), requires_grad=False)
        train = torch.nn.Parameter((torch.randn(6, 4, 3, generator=g) * 0.3).to(dev))
        SF.end_weight_epoch()
        y0 = SF.conv1d(x, frozen, None, 1, 1)
        SF.conv1d(x, frozen, None, 1, 1)
        assert calls["n"] == 1                                   # frozen: second call served from the cache
        with torch.no_grad():
            frozen.mul_(2.0)                                     # in-place update bumps the version counter
        y1 = SF.conv1d(x, frozen, None, 1, 1)
        assert calls["n"] == 2 and torch.allclose(y1, 2 * y0, rtol=1e-5, atol=1e-6)
        n = calls["n"]
        SF.conv1d(x, train, None, 1, 1); SF.conv1d(x, train, None, 1, 1)
        assert calls["n"] == n + 2                               # trainable, no epoch: always repacked
        SF.begin_weight_epoch()
        SF.conv1d(x, train, None, 1, 1); SF.conv1d(x, train, None, 1, 1)
        assert calls["n"] == n + 3                               # inside an epoch: packed once
        SF.note_weights_updated()
        SF.conv1d(x, train, None, 1, 1)
        assert calls["n"] == n + 4                               # optimizer step announced: repacked
        SF.end_weight_epoch()
        SF.begin_weight_epoch()                                  # next training step: epoch numbers are never reused
        SF.conv1d(x, train, None, 1, 1)
        assert calls["n"] == n + 5
    finally:
        K.weight_pack = orig
        SF.end_weight_epoch()


def test_wn_stack_frozen_weights_receive_no_gradient(dev):
    """Latent-map pass (svb_vae_task.py:634-661): the generator is frozen (requires_grad False) but keeps its flat-buffer
    `.grad` views; the decoder WN backward must only produce dx -- nothing may be accumulated into the frozen grads."""
    g_ = torch.Generator().manual_seed(77)
    B, C, T, gin, n, ks = 2, 8, 37, 6, 2, 5
    x = torch.randn(B, C, T, generator=g_)
    gcond = torch.randn(B, gin, T, generator=g_)
    cond = [torch.randn(2 * C * n, gin, 1, generator=g_) * 0.3, torch.rand(2 * C * n, 1, 1, generator=g_) + 0.5,
            torch.randn(2 * C * n, generator=g_) * 0.1]
    layers = []
    for i in range(n):
        rc = 2 * C if i < n - 1 else C
        layers.append([torch.randn(2 * C, C, ks, generator=g_) * 0.3, torch.rand(2 * C, 1, 1, generator=g_) + 0.5,
                       torch.randn(2 * C, generator=g_) * 0.1,
                       torch.randn(rc, C, 1, generator=g_) * 0.3, torch.rand(rc, 1, 1, generator=g_) + 0.5,
                       torch.randn(rc, generator=g_) * 0.1])
    xr = x.clone().requires_grad_(True)
    yr = _wn_ref(xr, None, gcond, cond, layers, ks)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    xd = _leaf(x, dev)
    frozen = [t.to(dev).requires_grad_(False) for t in cond] + [t.to(dev).requires_grad_(False) for lp in layers for t in lp]
    for t in frozen:
        t.grad = torch.zeros_like(t)          # FlatGradSync leaves such views on frozen parameters
    cd, ld = frozen[:3], [frozen[3 + 6 * i: 9 + 6 * i] for i in range(n)]
    y = SF.wn_stack(xd, None, gcond.to(dev), cd, ld, ks)
    y.backward(dy.to(dev))
    assert rel_err(xd.grad, xr.grad) < 5e-5
    for t in frozen:
        assert float(t.grad.abs().max()) == 0.0


def test_persistent_weight_images_and_multi_tensor_repack(dev):
    """bf16x3 + Trainer-managed step: trainable conv weights keep persistent packed images.  A silent in-place update (fused
    AdamW does not bump version counters) is announced with note_weights_updated(params); repack_registered(params) refills
    all of them with ONE launch; unrelated weights stay valid; a versioned update is caught without any announcement."""
    from neuralsvb_amd import kernels as K
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 8, 30, generator=g).to(dev)
    mk = lambda *s: torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(dev))
    v1, g1, b1 = mk(12, 8, 3), torch.nn.Parameter((torch.rand(12, 1, 1, generator=g) + 0.5).to(dev)), mk(12)
    v2 = mk(6, 8, 5)
    counts = {"single": 0, "multi": 0}
    o_into, o_multi = K.weight_pack_q_into, K.weight_pack_q_multi

    def c_into(*a, **k):
        counts["single"] += 1
        return o_into(*a, **k)

    def c_multi(*a, **k):
        counts["multi"] += 1
        return o_multi(*a, **k)
    K.weight_pack_q_into, K.weight_pack_q_multi = c_into, c_multi

    def fresh(v, gg, b, pad):
        SF.end_weight_epoch()                          # no epoch: packed from scratch on every call
        try:
            return SF.conv1d(x, v, b, 1, pad, weight_g=gg).detach().clone()
        finally:
            SF.begin_weight_epoch()
    SF.set_precision("bf16x3")
    try:
        SF.begin_weight_epoch()
        y1 = SF.conv1d(x, v1, b1, 1, 1, weight_g=g1)
        y2 = SF.conv1d(x, v2, None, 1, 2)
        n0 = counts["single"]
        assert torch.equal(SF.conv1d(x, v1, b1, 1, 1, weight_g=g1), y1) and counts["single"] == n0      # image reused
        SF.end_weight_epoch()
        SF.begin_weight_epoch()                                                # next step, nothing changed: still valid
        assert torch.equal(SF.conv1d(x, v1, b1, 1, 1, weight_g=g1), y1) and counts["single"] == n0
        v1.data.mul_(1.5)                                                      # silent update (no version bump)
        g1.data.add_(0.25)
        SF.note_weights_updated([v1, g1, b1])
        assert SF.repack_registered([v1, g1, b1]) == 1 and counts["multi"] == 1
        ya = SF.conv1d(x, v1, b1, 1, 1, weight_g=g1)
        assert counts["single"] == n0 and torch.equal(ya, fresh(v1, g1, b1, 1)) and not torch.equal(ya, y1)
        assert torch.equal(SF.conv1d(x, v2, None, 1, 2), y2) and counts["single"] == n0     # the other weight was not touched
        with torch.no_grad():
            v2.mul_(2.0)                                                       # versioned update, no announcement
        yb = SF.conv1d(x, v2, None, 1, 2)
        assert counts["single"] == n0 + 1 and torch.equal(yb, fresh(v2, None, None, 2))
        # backward through the image (data gradient uses the second layout, allocated on demand)
        xg = x.clone().requires_grad_(True)
        SF.conv1d(xg, v1, b1, 1, 1, weight_g=g1).sum().backward()
        assert xg.grad is not None and torch.isfinite(xg.grad).all()
        SF.note_weights_updated([v1, g1, b1, v2])
        assert SF.repack_registered([v1, g1, b1, v2]) == 2 and counts["multi"] == 2
    finally:
        K.weight_pack_q_into, K.weight_pack_q_multi = o_into, o_multi
        SF.end_weight_epoch()
        SF.set_precision("fp32")


@pytest.mark.parametrize("shape", [(3, 48, 37, 3),        # T % 4 != 0: the streaming kernels
                                   (2, 32, 40, 2),        # row-resident kernels (one wave per channel), 2 quads per lane
                                   (2, 64, 1124, 4),      # the step's row length: 5 quads per lane
                                   (1, 16, 2000, 1),      # 8 quads per lane
                                   (2, 24, 64, 3),        # 8 channels per group: 512-thread workgroups
                                   (1, 128, 48, 2),       # 64 channels per group: streaming kernels
                                   (2, 16, 4, 1)])        # one quad per row
def test_group_norm_relu_residual_matches_torch(dev, shape):
    """ConvBlock's GroupNorm(C/16) + ReLU and ConvStacks' residual (common_layers.py:688-707,739-773): values and all
    gradients against stock torch fp32."""
    import torch.nn.functional as F
    g_ = torch.Generator().manual_seed(21)
    B, Cc, T, G = shape
    h = (torch.randn(B, Cc, T, generator=g_) * 1.7 + 0.3).requires_grad_(True)
    res = torch.randn(B, Cc, T, generator=g_).requires_grad_(True)
    gm = (torch.rand(Cc, generator=g_) + 0.5).requires_grad_(True)
    bt = (torch.randn(Cc, generator=g_) * 0.3).requires_grad_(True)
    ref = res + F.relu(F.group_norm(h, G, gm, bt, 1e-5))
    gy = torch.randn(ref.shape, generator=g_)
    ref.backward(gy)
    hd, rd, gd, bd = (t.detach().to(dev).requires_grad_(True) for t in (h, res, gm, bt))
    y = SF.group_norm_relu(hd, gd, bd, G, 1e-5, residual=rd)
    assert (y.cpu() - ref.detach()).abs().max() < 2e-5
    y.backward(gy.to(dev))
    for mine, r in ((hd, h), (rd, res), (gd, gm), (bd, bt)):
        assert (mine.grad.cpu() - r.grad).abs().max() < 3e-5 * max(1.0, r.grad.abs().max().item())
    y2 = SF.group_norm_relu(hd.detach(), gd.detach(), bd.detach(), G, 1e-5)
    assert (y2.cpu() - F.relu(F.group_norm(h, G, gm, bt, 1e-5)).detach()).abs().max() < 2e-5


@pytest.mark.parametrize("groups", [1, 2])
def test_vae_latent_head_matches_torch(dev, groups):
    """Latent head of the global VAE encoder + KL (vae_models.py:24-41,100-105) against the stock-torch expression, incl.
    the positivity guard (a logs_q whose exp underflows) and every gradient."""
    g_ = torch.Generator().manual_seed(22)
    N, Lc, Tp, Tq = 4, 24, 9, 13
    xp = torch.randn(N, 2 * Lc, Tp, generator=g_)
    xp[1, Lc + 3] = -200.0                                    # exp underflows: guarded
    xp = xp.requires_grad_(True)
    eps = torch.randn(N, Lc, 1, generator=g_)
    mask = (torch.rand(N, Tq, generator=g_) > 0.3).float()

    def ref_fn(x):
        xm = x.mean(-1, keepdim=True)
        m_q, logs_q = torch.split(xm, Lc, dim=1)
        z = m_q + eps * torch.exp(logs_q)
        with torch.no_grad():
            bad = ~(logs_q.exp() > 0)
        lq = torch.where(bad, torch.zeros_like(logs_q), logs_q)
        kl = 0.5 * (torch.exp(2 * lq) + m_q ** 2 - 1.0) - lq
        ms = mask[:, None, :]
        klm = ((kl * ms).reshape(groups, -1).sum(1) / ms.reshape(groups, -1).sum(1)) / Lc
        return z, m_q, lq, klm
    z, mq, lq, kl = ref_fn(xp)
    cz, cm, cl = (torch.randn(N, Lc, 1, generator=g_) for _ in range(3))
    ck = torch.randn(groups, generator=g_)
    ((z * cz).sum() + (mq * cm).sum() + (lq * cl).sum() + (kl * ck).sum()).backward()
    xd = xp.detach().to(dev).requires_grad_(True)
    z2, mq2, lq2, kl2 = SF.vae_head(xd, eps.to(dev), mask.to(dev), groups)
    for a, b in ((z2, z), (mq2, mq), (lq2, lq), (kl2, kl)):
        assert (a.cpu() - b.detach()).abs().max() < 1e-5 * max(1.0, b.detach().abs().max().item())
    ((z2 * cz.to(dev)).sum() + (mq2 * cm.to(dev)).sum() + (lq2 * cl.to(dev)).sum() + (kl2 * ck.to(dev)).sum()).backward()
    assert (xd.grad.cpu() - xp.grad).abs().max() < 1e-5 * max(1.0, xp.grad.abs().max().item())
    # only z and kl used (the phase-2 step): absent cotangents are zeros
    xd2 = xp.detach().to(dev).requires_grad_(True)
    z3, _, _, kl3 = SF.vae_head(xd2, eps.to(dev), mask.to(dev), groups)
    ((z3 * cz.to(dev)).sum() + (kl3 * ck.to(dev)).sum()).backward()
    xr = xp.detach().clone().requires_grad_(True)
    zr, _, _, klr = ref_fn(xr)
    ((zr * cz).sum() + (klr * ck).sum()).backward()
    assert (xd2.grad.cpu() - xr.grad).abs().max() < 1e-5 * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("masked", [True, False])
@pytest.mark.parametrize("grad_buffers", ["all", "none", "frozen_in"])
def test_wn_stack_c_executor_equals_per_launch_path(dev, masked, grad_buffers):
    """SF.STACK_EXECUTOR (bf16x3): the gated stack issued by one C-ABI call per direction (csrc/wn_stack.hip) must give
    bit-identical outputs and gradients to the per-launch Python sequence -- same kernels, same arguments, same order.
    grad_buffers: 'all' = every parameter owns a `.grad` buffer (the Trainer's state: gradients are accumulated in place by the
    reduce kernel, the backward runs in C too); 'none' = no buffers (the C backward declines, the per-launch backward runs on
    the executor's stacked saved tensors); 'frozen_in' = buffers, but the in-layers do not train (their gradients are skipped)."""
    g_ = torch.Generator().manual_seed(35)
    B, C, T, gin, n, ks = 2, 16, 70, 12, 3, 3
    x = torch.randn(B, C, T, generator=g_)
    mask = torch.ones(B, T)
    mask[1, 55:] = 0.0
    gcond = torch.randn(B, gin, T, generator=g_)
    cond = [torch.randn(2 * C * n, gin, 1, generator=g_) * 0.3, torch.rand(2 * C * n, 1, 1, generator=g_) + 0.5,
            torch.randn(2 * C * n, generator=g_) * 0.1]
    layers = []
    for i in range(n):
        rc = 2 * C if i < n - 1 else C
        layers.append([torch.randn(2 * C, C, ks, generator=g_) * 0.3, torch.rand(2 * C, 1, 1, generator=g_) + 0.5,
                       torch.randn(2 * C, generator=g_) * 0.1, torch.randn(rc, C, 1, generator=g_) * 0.3,
                       torch.rand(rc, 1, 1, generator=g_) + 0.5, torch.randn(rc, generator=g_) * 0.1])
    dy = torch.randn(B, C, T, generator=g_)
    res = {}
    SF.set_precision("bf16x3")
    old = SF.STACK_EXECUTOR
    try:
        for cexec in (False, True):
            SF.STACK_EXECUTOR = cexec
            xd = _leaf(x, dev)
            cd = [_leaf(t, dev) for t in cond]
            ld = [[_leaf(t, dev) for t in lp] for lp in layers]
            leaves = cd + [t for lp in ld for t in lp]
            if grad_buffers != "none":
                for t in leaves:
                    t.grad = torch.full_like(t, 0.125)              # (accumulated into, not overwritten)
            if grad_buffers == "frozen_in":
                for lp in ld:
                    for t in lp[:3]:
                        t.requires_grad_(False)
            y = SF.wn_stack(xd, mask.to(dev) if masked else None, gcond.to(dev), cd, ld, ks, dilation_rate=2)
            y.backward(dy.to(dev))
            if dev.type == "cuda":
                torch.cuda.synchronize()
            res[cexec] = [y.detach(), xd.grad] + [t.grad for t in leaves]
    finally:
        SF.STACK_EXECUTOR = old
        SF.set_precision("fp32")
    for a, b in zip(res[False], res[True]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", [
    # period, H, Cin, Cout, k, stride, pad
    (2, 17, 3, 8, 5, 3, 2),       # MPD strided layer (hifigan.py:178), H % 3 == 2
    (3, 12, 4, 6, 5, 3, 2),       # H % 3 == 0
    (11, 10, 16, 24, 5, 3, 2),    # the deep layers' shape family: 10 rows of 11
    (5, 7, 1, 32, 5, 3, 2),       # first layer (one input channel), H % 3 == 1
    (7, 9, 20, 20, 5, 1, 2),      # stride-1 layer (hifigan.py:179) = dilation-p conv
    (11, 6, 24, 1, 3, 1, 1),      # conv_post (hifigan.py:180)
    (3, 20, 4, 8, 7, 2, 3),       # another (k, stride, pad) family: taps over 4 row offsets, 2 phases
    (2, 3, 2, 4, 5, 3, 2),        # fewer rows than taps (one output row)
    (13, 1, 2, 4, 5, 1, 2),       # a single row
])
def test_period_strided_conv_matches_conv2d(dev, case, precision):
    """The period discriminators' weight_norm(Conv2d((k,1), (stride,1), padding=(pad,0))) + LeakyReLU on [B,C,H,p] planes
    (reference modules/hifigan/hifigan.py:171-223), computed as a dilation-p 1-D conv -- strided layers over the row
    space-to-depth image (csrc/period_ops.hip) -- against torch's conv2d: output and all four gradients."""
    p, H, cin, cout, k, stride, pad = case
    g_ = torch.Generator().manual_seed(p * 100 + H)
    B = 2
    x = torch.randn(B, cin, H, p, generator=g_)
    v = torch.randn(cout, cin, k, 1, generator=g_) * 0.3
    gn = torch.rand(cout, 1, 1, 1, generator=g_) + 0.5
    b = torch.randn(cout, generator=g_)
    rl = [t.clone().requires_grad_(True) for t in (x, v, gn, b)]
    w = rl[2] * rl[1] / rl[1].flatten(1).norm(dim=1).view(-1, 1, 1, 1)
    yr = F.leaky_relu(F.conv2d(rl[0], w, rl[3], (stride, 1), (pad, 0)), 0.1)
    dy = torch.randn(yr.shape, generator=g_)
    yr.backward(dy)
    xd, vd, gd, bd = (_leaf(t, dev) for t in (x, v, gn, b))
    with SF.precision_scope(precision):
        y, h_out = SF.period_strided_conv(xd.view(B, cin, H * p), H, p, vd, gd, bd, stride, pad, out_act=SF.ACT_LRELU, out_slope=0.1)
        assert h_out == yr.shape[2] and y.shape == (B, cout, h_out * p)
        y.backward(dy.reshape(B, cout, -1).to(dev))
    tol = 2e-5 if precision == "fp32" else 6e-5
    assert rel_err(y.view(yr.shape), yr) < tol
    for got, ref in ((xd.grad, rl[0].grad), (vd.grad, rl[1].grad), (gd.grad, rl[2].grad), (bd.grad, rl[3].grad)):
        assert rel_err(got, ref) < 5 * tol


@pytest.mark.parametrize("case", [(5, 12, 8, 12, 5, 3, 2), (3, 10, 1, 8, 5, 3, 2), (4, 9, 16, 12, 5, 1, 2), (2, 14, 4, 8, 7, 2, 3)])
def test_period_conv_node_accumulates_into_gradient_buffers(dev, case):
    """The period discriminators' convs as ONE autograd node on the 4-D parameters (functional._PeriodConvFn, round 6): inside a
    Trainer-managed weight epoch, with `.grad` buffers in place, the weight / WeightNorm / bias gradients are ADDED into the buffers
    (strided layers through the slot gather of svb_period_weight; a single input channel -- rows of 6 floats -- through the tensor
    fallback), the kernel image and its packed forms persist across calls and follow an in-place weight update announced by
    note_weights_updated(); results equal the round-5 composition (torch pad / view / permute + the generic conv node) and torch's
    conv2d."""
    p, H, cin, cout, k, stride, pad = case
    g_ = torch.Generator().manual_seed(p * 10 + H)
    B = 2
    x = torch.randn(B, cin, H, p, generator=g_)
    v = torch.randn(cout, cin, k, 1, generator=g_) * 0.3
    gn = torch.rand(cout, 1, 1, 1, generator=g_) + 0.5
    b = torch.randn(cout, generator=g_)

    def reference(v_):
        rl = [t.clone().requires_grad_(True) for t in (x, v_, gn, b)]
        w = rl[2] * rl[1] / rl[1].flatten(1).norm(dim=1).view(-1, 1, 1, 1)
        yr = F.leaky_relu(F.conv2d(rl[0], w, rl[3], (stride, 1), (pad, 0)), 0.1)
        dy = torch.randn(yr.shape, generator=torch.Generator().manual_seed(7))
        yr.backward(dy)
        return yr.detach(), dy, [t.grad for t in rl]
    yr, dy, gr = reference(v)
    xd, vd, gd, bd = (_leaf(t, dev) for t in (x, v, gn, b))
    for t in (vd, gd, bd):
        t.grad = torch.full_like(t, 0.25)

    def run():
        y, h_out = SF.period_strided_conv(xd.view(B, cin, H * p), H, p, vd, gd, bd, stride, pad, out_act=SF.ACT_LRELU, out_slope=0.1)
        y.backward(dy.reshape(B, cout, -1).to(dev))
        return y.detach()
    with SF.precision_scope("bf16x3"):
        SF.begin_weight_epoch()
        try:
            y = run()
            assert rel_err(y.view(yr.shape), yr) < 6e-5
            for got, ref in ((vd.grad - 0.25, gr[1]), (gd.grad - 0.25, gr[2]), (bd.grad - 0.25, gr[3])):
                assert rel_err(got, ref) < 3e-4
            assert rel_err(xd.grad, gr[0]) < 3e-4
            n_reg = len(SF._REG)
            run()                                                    # second call of the epoch: same persistent images, gradients add up
            assert len(SF._REG) == n_reg and rel_err(vd.grad - 0.25, 2 * gr[1]) < 3e-4
            # the Trainer's deferred stage-2 reduces (a pass only RECORDS them; one multi-tensor launch finishes them): the strided
            # form's slot gather must see the finished gradient of the kernel image, not the zeros it was given
            from neuralsvb_amd import kernels as K
            K.begin_deferred_reduces()
            try:
                run()
            finally:
                K.flush_deferred_reduces()
            assert rel_err(vd.grad - 0.25, 3 * gr[1]) < 3e-4 and rel_err(gd.grad - 0.25, 3 * gr[2]) < 3e-4
            # an optimizer step the Trainer announces: the images follow the new weights
            with torch.no_grad():
                vd.mul_(1.5).add_(0.01)
            SF.note_weights_updated([vd])
            SF.repack_registered([vd, gd])
            yr2, _, _ = reference(v * 1.5 + 0.01)
            for t in (vd, gd, bd):
                t.grad.fill_(0.0)
            xd.grad = None
            y2 = run()
            assert rel_err(y2.view(yr2.shape), yr2) < 6e-5
            # the round-5 composition on the same tensors
            SF.PERIOD_CONV_NODE = False
            v_ref = vd.grad.clone()
            for t in (vd, gd, bd):
                t.grad.fill_(0.0)
            y3 = run()
            assert rel_err(y3, y2) < 1e-6 and rel_err(vd.grad, v_ref) < 1e-5
        finally:
            SF.PERIOD_CONV_NODE = True
            SF.end_weight_epoch()


def test_period_s2d_index_walk(dev):
    """svb_period_s2d walks the element index as a mixed-radix counter advancing by the grid's stride (no division per element).
    Forward and inverse against index arithmetic in torch, bit for bit."""
    _period_s2d_index_walk(dev, False)


@pytest.mark.gpu
def test_period_s2d_index_walk_big(gpu_only):
    """The same on a tensor of 5.9 M elements, more than the 16384 x 256 threads of the capped grid, so every thread wraps its
    digits several times (MI355X only: more elements than the lane emulator walks in reasonable time)."""
    _period_s2d_index_walk(gpu_only, True)


def _period_s2d_index_walk(dev, big):
    from neuralsvb_amd import kernels as K
    B, C, H, p, s, lead = (6, 96, 173, 7, 3, 2) if big else (2, 5, 23, 3, 3, 1)
    h_out = (H + 2 * 2 - 5) // s + 1
    R = lead + h_out + 1
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H * p, generator=g)
    x3 = x.view(B, C, H, p)
    img = torch.zeros(B, C, s, R, p)
    for r in range(s):
        for row in range(lead, R):
            hh = s * (row - lead) + r
            if hh < H:
                img[:, :, r, row] = x3[:, :, hh]
    got = K.period_s2d(x.to(dev), H, p, s, lead, R)
    assert torch.equal(got.cpu(), img.view(B, C * s, R * p))
    gi = torch.randn(B, C * s, R * p, generator=g)
    gi5 = gi.view(B, C, s, R, p)
    back = torch.zeros(B, C, H, p)
    for hh in range(H):
        r, row = hh % s, lead + hh // s
        if row < R:
            back[:, :, hh] = gi5[:, :, r, row]
    got_b = K.period_s2d(gi.to(dev), H, p, s, lead, R, inverse=True)
    assert torch.equal(got_b.cpu(), back.view(B, C, H * p))


@pytest.mark.parametrize("shape", [(2, 5, 37, 4), (1, 3, 8, 3), (3, 2, 1, 4), (2, 4, 33, 1)])
def test_upsample_nearest_nct_matches_interpolate(dev, shape):
    """SF.upsample_nearest_nct (reference svb_vae.py:39-45: nn.Upsample(scale_factor=s, mode='nearest')) against F.interpolate,
    forward bit for bit and the adjoint against torch autograd (window sums: same values, fp32 summation order may differ)."""
    from neuralsvb_amd import functional as SF
    B, C_, T, s = shape
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, C_, T, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.interpolate(xr, scale_factor=s, mode="nearest")
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    assert torch.equal(SF.upsample_nearest_nct(x.to(dev), s).cpu(), yr.detach())
    xd = x.to(dev).requires_grad_(True)
    y = SF.upsample_nearest_nct(xd, s)
    assert torch.equal(y.detach().cpu(), yr.detach())
    y.backward(dy.to(dev))
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= 1e-6 * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("n,shape", [(3, (2, 8, 333)), (2, (1, 4, 4096)), (3, (3, 5, 7))])
def test_mean_of_parallel_resblocks_one_pass(dev, n, shape):
    """SF.mean_of (the HifiGAN generator's `xs / num_kernels`, reference hifigan.py:157-163) in one pass against the stock
    sequence of adds and a division: value to one rounding, every input's gradient dy / n."""
    g_ = torch.Generator().manual_seed(n + shape[-1])
    xs = [torch.randn(shape, generator=g_).to(dev).requires_grad_(True) for _ in range(n)]
    y = SF.mean_of(xs)
    ref = sum(x.detach().cpu().double() for x in xs) / n
    assert rel_err(y.detach(), ref.float()) < 2e-7
    dy = torch.randn(shape, generator=g_).to(dev)
    y.backward(dy)
    for x in xs:
        assert torch.allclose(x.grad.cpu(), dy.cpu() / n, rtol=3e-7, atol=0)


def test_lsgan_terms_multi_tensor_launches_match_the_reference_formula(dev):
    """`discriminator_loss` / `generator_loss` (reference modules/hifigan/hifigan.py:338-365: means over the discriminators of
    mean((1 - D(x))^2), mean(D(G)^2)) through the mode-1 terms of the multi-tensor launches: 8 discriminator outputs of different
    sizes (one longer than a workgroup's 4096 elements, slices of stacked real / generated outputs as the task hands them over),
    value and gradients against the stock-torch formula, and the switch back gives the same numbers."""
    from neuralsvb_amd.modules import hifigan as H
    g_ = torch.Generator().manual_seed(8)
    shapes = [(2, 1, 37), (2, 1, 19), (2, 1, 5000), (2, 1, 64), (2, 1, 1), (2, 1, 130), (2, 1, 77), (2, 1, 512)]
    stacked = [torch.randn((2 * s[0],) + s[1:], generator=g_) for s in shapes]

    def run(fused):
        SF.FUSED_GAN_LOSS = fused
        outs = [t.clone().to(dev).requires_grad_(True) for t in stacked]
        real, gen = [o[:2] for o in outs], [o[2:] for o in outs]
        r, g = H.discriminator_loss(real, gen)
        a = H.generator_loss(gen)
        (r * 0.5 + g * 1.5 + a * 0.25).backward()
        return [float(r), float(g), float(a)], [o.grad.cpu() for o in outs]
    try:
        v1, g1 = run(True)
        v0, g0 = run(False)
        for a, b in zip(v1, v0):
            assert abs(a - b) <= 3e-6 * max(1.0, abs(b))
        for a, b in zip(g1, g0):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-9)
    finally:
        SF.FUSED_GAN_LOSS = True


def test_feature_loss_multi_tensor_launches_match_the_reference_formula(dev):
    """`feature_loss` (reference modules/hifigan/hifigan.py:328-335: 2 * sum over the feature-map pairs of mean(|r - g|)) through the
    multi-tensor kernels (csrc/loss_ops.hip): 37 pairs (two launches of <= 32), sizes that are and are not multiples of 4 / of a
    workgroup's 4096 elements, exact zeros in r - g (torch's sign(0) = 0), gradients into the generated maps only (the generator
    pass) and into both -- against the stock-torch formula; and the switch back to that formula gives the same numbers."""
    from neuralsvb_amd.modules import hifigan as H
    g_ = torch.Generator().manual_seed(5)
    shapes = [(2, 3, 7), (1, 4, 1024), (2, 8, 300), (3, 5, 11, 2), (1, 1, 4097), (2, 16, 513)] * 6 + [(4, 4, 4)]
    fr = [torch.randn(s, generator=g_) for s in shapes]
    fg = [torch.randn(s, generator=g_) for s in shapes]
    fg[1][0, :, :50] = fr[1][0, :, :50]                              # exact ties
    groups = [slice(0, 12), slice(12, 30), slice(30, 37)]             # "discriminators" with 12 / 18 / 7 maps

    def run(fused, r_grad):
        SF.FUSED_FEATURE_LOSS = fused
        r = [t.clone().to(dev).requires_grad_(r_grad) for t in fr]
        g = [t.clone().to(dev).requires_grad_(True) for t in fg]
        loss = H.feature_loss([r[s] for s in groups], [g[s] for s in groups])
        (loss * 0.7).backward()
        return loss.detach().cpu(), [t.grad.cpu() for t in g], [t.grad.cpu() if r_grad else None for t in r]
    try:
        for r_grad in (False, True):
            l1, dg1, dr1 = run(True, r_grad)
            l0, dg0, dr0 = run(False, r_grad)
            assert abs(float(l1) - float(l0)) <= 2e-6 * abs(float(l0))
            for a, b in zip(dg1, dg0):
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)
            if r_grad:
                for a, b in zip(dr1, dr0):
                    assert torch.allclose(a, b, rtol=1e-6, atol=1e-9)
    finally:
        SF.FUSED_FEATURE_LOSS = True
