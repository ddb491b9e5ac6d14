"""Operator parity on the GPU: every HIP convolution form against torch's CPU convolution on the
same seeded inputs.  bf16 runs feed the oracle bf16-rounded operands, so products are exact and
only fp32 accumulation order and the final bf16 store rounding differ."""
import pytest
import torch
import torch.nn.functional as F

from helpers import bf16_round, from_cl, rel_err, to_cl
from unet_amd import ops

pytestmark = pytest.mark.gpu

DT = {"fp32": torch.float32, "bf16": torch.bfloat16}
TOL_OUT = {"fp32": 2e-5, "bf16": 6e-3}      # relative to max |reference|
TOL_W = {"fp32": 5e-5, "bf16": 2e-4}        # weight gradients are produced in fp32 in both modes


def _conv_ref(x, w, b, stride, transposed, nd):
    if transposed:
        f = F.conv_transpose2d if nd == 2 else F.conv_transpose3d
        return f(x, w, b, stride=2, padding=1, output_padding=1)
    f = F.conv2d if nd == 2 else F.conv3d
    return f(x, w, b, stride=stride, padding=(w.shape[-1] - 1) // 2)


CASES = [
    # nd, spatial, cin, cout, k, stride, transposed
    (2, (32, 32), 16, 16, 3, 1, False),
    (2, (24, 40), 32, 32, 3, 1, False),
    (2, (16, 16), 64, 128, 3, 1, False),
    (2, (8, 8), 128, 256, 3, 1, False),
    (2, (32, 32), 16, 64, 3, 2, False),
    (2, (16, 48), 32, 128, 3, 2, False),
    (2, (8, 8), 128, 256, 1, 1, False),
    (2, (8, 8), 384, 64, 3, 2, True),
    (2, (16, 16), 64, 16, 3, 2, True),
    (2, (12, 20), 128, 32, 3, 2, True),
    (2, (18, 30), 32, 64, 3, 2, False),      # odd output extents: ragged tiles of the stride-2 gather / scatter kernels
    (2, (36, 20), 64, 128, 3, 2, False),     # two contraction chunks with resident weights
    (2, (10, 14), 64, 32, 3, 2, True),
    (2, (20, 36), 128, 64, 3, 2, False),     # wide contraction: the stride-2 gather streams its tap tiles with the patches
    (2, (10, 14), 192, 128, 3, 2, True),     # the same in the data gradient of a transposed convolution (two sources' worth of channels)
    (2, (40, 24), 64, 64, 3, 1, False),      # several tiles per workgroup, last tile row ragged
    (2, (24, 40), 16, 32, 3, 1, False),      # 16 contraction channels: two taps per MFMA step (bf16)
    (2, (20, 20), 16, 64, 3, 1, False),
    (2, (36, 28), 16, 32, 3, 2, False),      # the same with stride 2
    (2, (14, 10), 64, 16, 3, 2, True),       # its data gradient gathers 16 channels with stride 2
    (3, (8, 8, 8), 16, 16, 3, 1, False),
    (3, (6, 10, 20), 16, 16, 3, 1, False),    # 3-D, 16 contraction channels (bf16: tap pairs, fragments shared over kernel rows): ragged in z, y and x
    (3, (9, 5, 33), 16, 32, 3, 1, False),     # the same with two channel blocks and odd extents
    (3, (4, 8, 12), 32, 64, 3, 1, False),
    (3, (9, 6, 20), 32, 32, 3, 1, False),     # 3-D, 32 contraction channels: 16 output channels per workgroup, ragged tiles
    (3, (8, 8, 8), 16, 64, 3, 2, False),
    (3, (4, 4, 4), 64, 16, 3, 2, True),
    (3, (4, 4, 4), 32, 64, 1, 1, False),
    (3, (10, 12, 20), 32, 64, 3, 2, False),   # 3-D stride-2 gather staged by polyphase components: ragged tiles
    (3, (12, 8, 16), 128, 32, 3, 2, False),   # the same with several contraction chunks
    (3, (6, 10, 6), 128, 32, 3, 2, True),     # data gradient of a 3-D transposed convolution (the other stride-2 gather)
]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "nd%d_%s_%d-%d_k%d_s%d_%s" % (c[0], "x".join(map(str, c[1])), c[2], c[3], c[4], c[5], "T" if c[6] else "C"))
def test_conv_forward_dgrad_wgrad(dev, mode, case):
    nd, sp, cin, cout, k, stride, transposed = case
    dt = DT[mode]
    g = torch.Generator().manual_seed(1234 + cin * 7 + cout)
    nb = 2
    x = torch.randn(nb, cin, *sp, generator=g)
    wshape = ((cin, cout) if transposed else (cout, cin)) + (k,) * nd
    w = torch.randn(*wshape, generator=g) / (cin * k ** nd) ** 0.5
    b = torch.randn(cout, generator=g)
    if mode == "bf16":
        x, w = bf16_round(x), bf16_round(w)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = _conv_ref(xr, wr, br, stride, transposed, nd)
    dy = torch.randn(y_ref.shape, generator=g)
    if mode == "bf16":
        dy = bf16_round(dy)
    y_ref.backward(dy)

    xc = to_cl(x, dt, dev)
    y, stats = ops.conv_forward(xc, w.to(dev), b.to(dev), ksize=k, stride=stride, transposed=transposed, want_stats=True)
    torch.cuda.synchronize()
    assert rel_err(from_cl(y), y_ref.detach()) < TOL_OUT[mode]
    # epilogue statistics: per-(image, channel) sum and sum of squares of the bias-free result
    nob = (y_ref.detach() - b.view(1, -1, *([1] * nd)))
    s1 = nob.flatten(2).sum(-1)
    s2 = (nob * nob).flatten(2).sum(-1)
    st = stats.cpu().sum(1).float()
    assert rel_err(st[..., 0], s1) < 1e-4 and rel_err(st[..., 1], s2) < 1e-4

    dyc = to_cl(dy, dt, dev)
    dx = ops.conv_dgrad(dyc, w.to(dev), xc.shape, ksize=k, stride=stride, transposed=transposed)
    assert rel_err(from_cl(dx), xr.grad) < TOL_OUT[mode]
    # addend in the epilogue (gradient accumulation at a fan-out)
    dx2 = ops.conv_dgrad(dyc, w.to(dev), xc.shape, ksize=k, stride=stride, transposed=transposed, addend=xc)
    ref2 = xr.grad + x
    assert rel_err(from_cl(dx2), ref2) < TOL_OUT[mode]

    dw, db = ops.conv_wgrad(xc, dyc, wshape, ksize=k, stride=stride, transposed=transposed)
    assert rel_err(dw.cpu(), wr.grad) < TOL_W[mode]
    if db is not None:
        assert rel_err(db.cpu(), br.grad) < TOL_W[mode]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_conv_exact_integer_data(dev, mode):
    """Small-integer operands make every product and sum exact in both dtypes: any mismatch is an
    indexing error, not rounding.  Asymmetric weights catch transposed fragments."""
    dt = DT[mode]
    g = torch.Generator().manual_seed(7)
    for (cin, cout, stride, transposed) in [(16, 32, 1, False), (32, 16, 2, False), (48, 16, 2, True), (64, 64, 1, False)]:
        x = torch.randint(-2, 3, (1, cin, 20, 28), generator=g).float()
        wshape = ((cin, cout) if transposed else (cout, cin)) + (3, 3)
        w = torch.randint(-2, 3, wshape, generator=g).float()
        y_ref = _conv_ref(x, w, None, stride, transposed, 2)
        assert y_ref.abs().max() < 256  # exactly representable in bf16
        y = ops.conv_forward(to_cl(x, dt, dev), w.to(dev), None, ksize=3, stride=stride, transposed=transposed)
        assert torch.equal(from_cl(y), y_ref)
    # 3-D with 16 contraction channels (bf16: the kernel that pairs taps inside a contraction step)
    x = torch.randint(-2, 3, (2, 16, 7, 9, 21), generator=g).float()
    w = torch.randint(-1, 2, (16, 16, 3, 3, 3), generator=g).float()
    y_ref = _conv_ref(x, w, None, 1, False, 3)
    assert y_ref.abs().max() < 256
    y = ops.conv_forward(to_cl(x, dt, dev), w.to(dev), None, ksize=3, stride=1, transposed=False)
    assert torch.equal(from_cl(y), y_ref)


C1_CASES = [
    # nd, fine spatial, channels of the first source, channels of the second source
    (2, (32, 48), 16, 0),
    (2, (36, 20), 32, 0),
    (2, (24, 136), 16, 16),
    (2, (40, 72), 32, 32),        # 64 channels from two sources (the first layer of the 32-512 network)
    (3, (8, 8, 8), 16, 0),
    (3, (12, 20, 72), 16, 0),     # 10 coarse rows (ragged four-row tiles), 36 coarse pixels per row (ragged 64-pixel tile)
    (3, (6, 36, 136), 16, 16),    # two sources; 68 coarse pixels: two tiles per row, the second nearly empty
    (3, (10, 12, 40), 32, 0),     # 20 coarse pixels: the 32-pixel tile
    (3, (4, 8, 24), 16, 32),      # 48 channels: the vector-ALU kernel in both modes
]


def _conv1_ref(x, dy, nd):
    n = dy.shape[1]
    w = torch.zeros([n, 1] + [3] * nd, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(n, dtype=torch.float64, requires_grad=True)
    f = F.conv2d if nd == 2 else F.conv3d
    f(x[:, None].double(), w, b, stride=2, padding=1).backward(dy.double())
    return w.grad, b.grad


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("case", C1_CASES, ids=lambda c: "nd%d_%s_%d+%d" % (c[0], "x".join(map(str, c[1])), c[2], c[3]))
def test_one_channel_stride2_wgrad(dev, mode, case):
    """The weight gradient of the two layers with one channel on the fine side (first encoder convolution, last
    transposed convolution) against torch's CPU convolution in fp64: 5e-5 of the largest element in fp32; in bf16 the
    channel tensor is bf16 on both sides and the fp32 image is taken as a bf16 head plus remainder (2e-4).  Then with
    small integers, where every product and sum is exact: bit for bit."""
    nd, sp, c1, c2 = case
    dt = DT[mode]
    g = torch.Generator().manual_seed(99 + c1 + 3 * c2 + sp[-1])
    nb = 2
    csp = [v // 2 for v in sp]
    for exact in (False, True):
        if exact:
            x = torch.randint(-4, 5, (nb, *sp), generator=g).float()
            dy = torch.randint(-2, 3, (nb, c1 + c2, *csp), generator=g).float()
        else:
            x = torch.randn(nb, *sp, generator=g)
            dy = torch.randn(nb, c1 + c2, *csp, generator=g)
            if mode == "bf16":
                dy = bf16_round(dy)
        dw_ref, db_ref = _conv1_ref(x, dy, nd)
        ya = to_cl(dy[:, :c1].contiguous(), dt, dev)
        yb = to_cl(dy[:, c1:].contiguous(), dt, dev) if c2 else None
        dw, db = ops.conv1_wgrad(x.to(dev), ya, yb)
        torch.cuda.synchronize()
        if exact:
            assert torch.equal(dw.cpu().double(), dw_ref) and torch.equal(db.cpu().double(), db_ref)
        else:
            assert rel_err(dw.cpu().double(), dw_ref) < TOL_W[mode]
            assert rel_err(db.cpu().double(), db_ref) < TOL_W[mode]


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("case", C1_CASES + [(2, (256, 256), 32, 32), (2, (64, 512), 16, 16)],
                         ids=lambda c: "nd%d_%s_%d+%d" % (c[0], "x".join(map(str, c[1])), c[2], c[3]))
def test_one_channel_stride2_forward_and_transposed(dev, mode, case):
    """The two layers with one channel on the fine side, forward, against torch's CPU convolution in fp64.  1 -> C
    convolution: fp32 image and weights in both modes (2e-5 of the largest output in fp32; in bf16 the bound is the
    rounding of the stored output), epilogue statistics to 1e-4, two destinations.  C -> 1 transposed convolution: the
    channel tensor and the weights are bf16 in bf16 mode (the oracle gets the same rounded values), the output fp32.  Small integers
    make both exact."""
    nd, sp, c1, c2 = case
    dt = DT[mode]
    n = c1 + c2
    g = torch.Generator().manual_seed(5 + c1 + 3 * c2 + sp[-1])
    nb = 2
    csp = [v // 2 for v in sp]
    conv = F.conv2d if nd == 2 else F.conv3d
    convT = F.conv_transpose2d if nd == 2 else F.conv_transpose3d
    for exact in (False, True):
        if exact:
            x = torch.randint(-4, 5, (nb, *sp), generator=g).float()
            w = torch.randint(-2, 3, [n, 1] + [3] * nd, generator=g).float()
            b = torch.randint(-3, 4, (n,), generator=g).float()
            yin = torch.randint(-2, 3, (nb, n, *csp), generator=g).float()
        else:
            x = torch.randn(nb, *sp, generator=g)
            w = torch.randn([n, 1] + [3] * nd, generator=g) / 3 ** (nd / 2)
            b = torch.randn(n, generator=g)
            yin = torch.randn(nb, n, *csp, generator=g)
            if mode == "bf16":
                yin = bf16_round(yin)
        y_ref = conv(x[:, None].double(), w.double(), b.double(), stride=2, padding=1)
        out, stats = ops.conv1_forward(x.to(dev), w.to(dev), b.to(dev), dtype=dt, split=c1 if c2 else None, want_stats=True)
        torch.cuda.synchronize()
        y = torch.cat([from_cl(t) for t in out], 1) if c2 else from_cl(out)
        if exact:
            assert y_ref.abs().max() < 256 and torch.equal(y.double(), y_ref)
        else:
            assert rel_err(y.double(), y_ref) < TOL_OUT[mode]
            nob = y_ref - b.double().view(1, -1, *([1] * nd))
            st = stats.cpu().sum(1)
            assert rel_err(st[..., 0], nob.flatten(2).sum(-1)) < 1e-4 and rel_err(st[..., 1], (nob * nob).flatten(2).sum(-1)) < 1e-4

        wt = bf16_round(w) if mode == "bf16" else w      # the bf16 kernel multiplies on the matrix cores: weights rounded like every packed weight
        t_ref = convT(yin.double(), wt.double(), b[:1].double(), stride=2, padding=1, output_padding=1)[:, 0]
        ya = to_cl(yin[:, :c1].contiguous(), dt, dev)
        yb = to_cl(yin[:, c1:].contiguous(), dt, dev) if c2 else None
        xo, tstats = ops.conv1_transposed(ya, w.to(dev), b[:1].to(dev), y2=yb, want_stats=True)
        torch.cuda.synchronize()
        if exact:
            assert torch.equal(xo.cpu().double(), t_ref)
        else:
            assert rel_err(xo.cpu().double(), t_ref) < (2e-5 if mode == "fp32" else 2e-4)
            nob = t_ref - float(b[0])
            st = tstats.cpu().sum(1)[:, 0]
            assert rel_err(st[:, 0], nob.flatten(1).sum(-1)) < 1e-4 and rel_err(st[:, 1], (nob * nob).flatten(1).sum(-1)) < 1e-4


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("nd,sp,c", [(2, (32, 32), 16), (2, (20, 12), 64), (3, (4, 6, 8), 32), (2, (16, 16), 256)])
def test_instnorm_prelu_fwd_bwd(dev, mode, nd, sp, c):
    dt = DT[mode]
    g = torch.Generator().manual_seed(99 + c)
    nb = 3
    x = torch.randn(nb, c, *sp, generator=g) * 2 + 0.5
    res = torch.randn(nb, c, *sp, generator=g)
    dy = torch.randn(nb, c, *sp, generator=g)
    if mode == "bf16":
        x, res, dy = bf16_round(x), bf16_round(res), bf16_round(dy)
    alpha = torch.tensor([0.25])
    xr = x.clone().requires_grad_(True)
    ar = alpha.clone().requires_grad_(True)
    y_ref = F.prelu(F.instance_norm(xr, eps=1e-5), ar) + res
    y_ref.backward(dy)
    # statistics as the producing convolution would deliver them (sums split over the 8 slots)
    s1 = x.flatten(2).sum(-1)
    s2 = (x * x).flatten(2).sum(-1)
    stats = torch.zeros(nb, 8, c, 2, dtype=torch.float64)
    stats[:, 0, :, 0], stats[:, 3, :, 1] = s1, s2
    stats = stats.to(dev)
    xc, rc, dyc = to_cl(x, dt, dev), to_cl(res, dt, dev), to_cl(dy, dt, dev)
    y = ops.instnorm_prelu_fwd(xc, stats, alpha.to(dev), res=rc)
    tol = 2e-5 if mode == "fp32" else 8e-3
    assert rel_err(from_cl(y), y_ref.detach()) < tol
    dx, dalpha = ops.instnorm_prelu_bwd(dyc, xc, stats, alpha.to(dev))
    assert rel_err(from_cl(dx), xr.grad) < (1e-4 if mode == "fp32" else 1e-2)
    assert abs(float(dalpha.cpu()) - float(ar.grad)) < 1e-4 * max(1.0, abs(float(ar.grad))) * (1 if mode == "fp32" else 50)


def test_dropout_statistics_and_mask_replay(dev):
    """The stateless dropout keeps ~(1-p) of the elements, scales by 1/(1-p), and backward replays
    the very same mask (gradient is zero exactly where the forward output is zero)."""
    nb, c, sp = 2, 32, (64, 64)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(nb, c, *sp, generator=g)
    s1 = x.flatten(2).sum(-1)
    s2 = (x * x).flatten(2).sum(-1)
    stats = torch.zeros(nb, 8, c, 2, dtype=torch.float64)
    stats[:, 0, :, 0], stats[:, 0, :, 1] = s1, s2
    stats = stats.to(dev)
    alpha = torch.tensor([0.25], device=dev)
    xc = to_cl(x, torch.float32, dev)
    p = 0.1
    y = ops.instnorm_prelu_fwd(xc, stats, alpha, dropout=p, seed=42, layer=3)
    y0 = ops.instnorm_prelu_fwd(xc, stats, alpha, dropout=0.0)
    dropped = (y == 0)
    frac = float(dropped.float().mean())
    assert abs(frac - p) < 0.01
    kept = ~dropped
    assert torch.allclose(y[kept], y0[kept] / (1 - p), rtol=1e-5, atol=1e-6)
    y_again = ops.instnorm_prelu_fwd(xc, stats, alpha, dropout=p, seed=42, layer=3)
    assert torch.equal(y, y_again)
    y_other = ops.instnorm_prelu_fwd(xc, stats, alpha, dropout=p, seed=43, layer=3)
    assert not torch.equal(y, y_other)
    y_layer = ops.instnorm_prelu_fwd(xc, stats, alpha, dropout=p, seed=42, layer=4)
    assert not torch.equal(y, y_layer)
    dy = torch.ones_like(xc)
    dx, _ = ops.instnorm_prelu_bwd(dy, xc, stats, alpha, dropout=p, seed=42, layer=3)
    assert torch.isfinite(dx).all()
    # the mask is a property of the element index: bf16 storage draws the same one
    yb = ops.instnorm_prelu_fwd(to_cl(x, torch.bfloat16, dev), stats, alpha, dropout=p, seed=42, layer=3)
    assert torch.equal(yb == 0, dropped)
    # every image, channel and position of the four-element hash group drops at the same rate, and
    # neighbours are uncorrelated
    m = from_cl(dropped.float())                     # [nb, c, h, w]
    assert (m.mean(dim=(1, 2, 3)) - p).abs().max() < 0.01
    assert (m.mean(dim=(0, 2, 3)) - p).abs().max() < 0.02
    assert not torch.equal(m[0], m[1])
    for a, b in ((m[:, :-1], m[:, 1:]), (m[..., :-1], m[..., 1:]), (m[:, :, :-1], m[:, :, 1:]), (m[:, 0::4], m[:, 2::4])):
        cov = float((a * b).mean() - a.mean() * b.mean())
        assert abs(cov) < 2e-3, cov


def test_dropout_rate_follows_probability(dev):
    nb, c, sp = 1, 16, (128, 128)
    x = torch.randn(nb, c, *sp, generator=torch.Generator().manual_seed(6))
    stats = torch.zeros(nb, 8, c, 2, dtype=torch.float64)
    stats[:, 0, :, 0], stats[:, 0, :, 1] = x.flatten(2).sum(-1), (x * x).flatten(2).sum(-1)
    stats = stats.to(dev)
    alpha = torch.tensor([0.25], device=dev)
    xc = to_cl(x, torch.float32, dev)
    for p in (0.02, 0.1, 0.5, 0.9):
        y = ops.instnorm_prelu_fwd(xc, stats, alpha, dropout=p, seed=7, layer=0)
        frac = float((y == 0).float().mean())
        assert abs(frac - p) < 0.01, (p, frac)
    # one-channel planar tensors use the same generator
    x1 = torch.randn(2, 1, 128, 128, generator=torch.Generator().manual_seed(8))
    st1 = torch.zeros(2, 8, 1, 2, dtype=torch.float64)
    st1[:, 0, :, 0], st1[:, 0, :, 1] = x1.flatten(2).sum(-1), (x1 * x1).flatten(2).sum(-1)
    y1 = ops.instnorm_prelu_fwd(x1.to(dev).reshape(2, 128, 128, 1), st1.to(dev), alpha, dropout=0.1, seed=7, layer=0)
    assert abs(float((y1 == 0).float().mean()) - 0.1) < 0.01
