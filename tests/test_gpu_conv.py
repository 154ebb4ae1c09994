"""Implicit-GEMM kernels (cvg_conv_fwd / cvg_conv_wgrad / cvg_weight_pack) against
torch's CPU convolution on the same seeded inputs, through the C ABI.
Tolerances: f32 mode uses exact-f32 MFMA, so only summation order differs from the
CPU (<= 2e-5 of the tensor's max); bf16 mode rounds inputs to 8 significant bits
(<= 2e-2 of the tensor's max for these reduction lengths)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def _mods():
    from cvae_gan_amd import ops
    from cvae_gan_amd.engine import _desc, _wmap
    return ops, _desc, _wmap


def to_rows(x, dtype):            # NCHW f32 cpu -> [B*H*W][C] device
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().view(-1, x.shape[1])


def from_rows(t, B, H, W, C):
    return t.float().view(B, H, W, C).permute(0, 3, 1, 2).contiguous().cpu()


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def pack(ops, wmap, w, desc, dtype):
    packed = torch.zeros(desc.w_rows * desc.w_ld, dtype=dtype, device="cuda")
    ops.weight_pack(desc, wmap, w.cuda().contiguous(), packed)
    return packed


def ws_for(ops, desc):
    n = max(ops.conv_fwd_workspace(desc), ops.conv_wgrad_workspace(desc), 1024)
    return torch.zeros(n // 4 + 16, dtype=torch.float32, device="cuda")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,cin,cout,stride", [(3, 12, 16, 24, 1), (3, 12, 16, 32, 1), (2, 16, 64, 64, 1),
                                                 (1, 8, 128, 256, 1), (3, 12, 16, 40, 2), (3, 12, 8, 16, 2),
                                                 (2, 16, 64, 128, 2), (3, 40, 32, 128, 2)])
def test_conv3x3_fwd_dgrad_wgrad(dtype, B, H, cin, cout, stride):
    ops, _desc, _wmap = _mods()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + cin + stride)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    if dtype == torch.bfloat16:   # compare against the same rounded operands
        x, w = x.bfloat16().float(), w.bfloat16().float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, stride, 1)
    dy = torch.randn(y.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    y.backward(dy)
    OH = H // stride
    # forward
    d = _desc(dtype, B, H, H, cin, OH, OH, cout, 3, 9, stride=stride, off=(-1, -1))
    wm = _wmap(1, cin * 9, 0, cin, 1, 9, cout, cin, 9)
    ws = ws_for(ops, d)
    xr = to_rows(x.detach(), dtype)
    yr = torch.zeros(B * OH * OH, cout, dtype=dtype, device="cuda")
    ops.conv_fwd(d, xr, pack(ops, wm, w.detach(), d, dtype), yr, ws=ws)
    assert relmax(from_rows(yr, B, OH, OH, cout), y.detach()) < TOL[dtype]
    # weight gradient
    dyr = to_rows(dy, dtype)
    gw = torch.full(w.shape, 7.0, device="cuda")
    ops.conv_wgrad(d, xr, dyr, cout, wm, gw, ws)
    assert relmax(gw.cpu(), w.grad) < TOL[dtype]
    # input gradient (its k-extent per tap is cout, which the kernels need to be a power of two)
    if cout & (cout - 1):
        return
    dxr = torch.zeros(B * H * H, cin, dtype=dtype, device="cuda")
    if stride == 1:
        dd = _desc(dtype, B, H, H, cout, H, H, cin, 3, 9, off=(-1, -1))
        dm = _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, 9, tap_mode=1)
        ops.conv_fwd(dd, dyr, pack(ops, dm, w.detach(), dd, dtype), dxr, ws=ws_for(ops, dd))
    else:
        for py in (0, 1):
            for px in (0, 1):
                khs = [1] if py == 0 else [0, 2]
                kws = [1] if px == 0 else [0, 2]
                table = [kh * 3 + kw for kh in khs for kw in kws]
                dd = _desc(dtype, B, OH, OH, cout, OH, OH, cin, len(kws), len(table), tstep=(-1, -1),
                           off=(1 if py else 0, 1 if px else 0), o_mul=(2, 2), o_off=(py, px), y_dims=(H, H))
                dm = _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, len(table), tap_mode=2, table=table)
                ops.conv_fwd(dd, dyr, pack(ops, dm, w.detach(), dd, dtype), dxr, ws=ws_for(ops, dd))
    torch.cuda.synchronize()
    assert relmax(from_rows(dxr, B, H, H, cin), x.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_bias_residual_gate(dtype):
    """Epilogue: bias, per-class bias, residual scaled by the LeakyReLU derivative."""
    ops, _desc, _wmap = _mods()
    g = torch.Generator().manual_seed(5)
    B, H, cin, cout = 5, 8, 32, 48
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / 17
    bias = torch.randn(cout, generator=g)
    cb = torch.randn(7, cout, generator=g)
    labels = torch.randint(0, 7, (B,), generator=g)
    res = torch.randn(B, cout, H, H, generator=g)
    gate = torch.randn(B, cout, H, H, generator=g)
    if dtype == torch.bfloat16:
        x, w, res, gate = [t.bfloat16().float() for t in (x, w, res, gate)]
    ref = F.conv2d(x, w, bias, 1, 1) + cb[labels][:, :, None, None] + res * torch.where(gate > 0, 1.0, 0.2)
    d = _desc(dtype, B, H, H, cin, H, H, cout, 3, 9, off=(-1, -1))
    wm = _wmap(1, cin * 9, 0, cin, 1, 9, cout, cin, 9)
    yr = torch.zeros(B * H * H, cout, dtype=dtype, device="cuda")
    ops.conv_fwd(d, to_rows(x, dtype), pack(ops, wm, w, d, dtype), yr, bias=bias.cuda(), cbias=cb.cuda(),
                 labels=labels.int().cuda(), res=to_rows(res, dtype), gate=to_rows(gate, dtype), ws=ws_for(ops, d))
    assert relmax(from_rows(yr, B, H, H, cout), ref) < TOL[dtype]


@pytest.mark.parametrize("B,H,cin,cout", [(130, 32, 64, 128), (131, 64, 64, 64), (67, 64, 128, 64), (150, 48, 64, 64)])
def test_walking_workgroups_bit_identical(B, H, cin, cout):
    """CVG_CONV_ALONE: workgroups of patch_fwd keep their tile position and walk over several image groups
    (bias preloaded, residual fetched after the staging): eight-wave workgroups of the 128-channel form with the
    output staged behind the patch buffers, four-wave workgroups of the 64-channel form with it staged over the
    second patch buffer.  Same products in the same order: the output must equal the default form's bit for bit,
    with bias, residual and gate, and with group counts that the walks do not divide."""
    ops, _desc, _wmap = _mods()
    from cvae_gan_amd._lib import CONV_ALONE
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, cin, H, H, generator=g).bfloat16().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) / 24).bfloat16().float()
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, H, generator=g).bfloat16().float()
    gate = torch.randn(B, cout, H, H, generator=g).bfloat16().float()
    ref = F.conv2d(x, w, bias, 1, 1) + res * torch.where(gate > 0, 1.0, 0.2)
    d = _desc(dtype, B, H, H, cin, H, H, cout, 3, 9, off=(-1, -1))
    wm = _wmap(1, cin * 9, 0, cin, 1, 9, cout, cin, 9)
    xr, rr, gr, pk = to_rows(x, dtype), to_rows(res, dtype), to_rows(gate, dtype), pack(ops, wm, w, d, dtype)
    out = {}
    for flags in (0, CONV_ALONE):
        d.flags = flags
        for with_res in (False, True):
            yr = torch.zeros(B * H * H, cout, dtype=dtype, device="cuda")
            ops.conv_fwd(d, xr, pk, yr, bias=bias.cuda(), res=rr if with_res else None, gate=gr if with_res else None,
                         ws=ws_for(ops, d))
            torch.cuda.synchronize()
            out[flags, with_res] = yr
    assert relmax(from_rows(out[CONV_ALONE, True], B, H, H, cout), ref) < TOL[dtype]
    assert torch.equal(out[0, True], out[CONV_ALONE, True])
    assert torch.equal(out[0, False], out[CONV_ALONE, False])
    # the inference epilogue (folded BatchNorm, residual, LeakyReLU) on both forms
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    folded = {}
    for flags in (0, CONV_ALONE):
        d.flags = flags
        for with_res in (False, True):
            yr = torch.zeros(B * H * H, cout, dtype=dtype, device="cuda")
            ops.conv_bn_act_fwd(d, xr, pk, yr, scale, shift, bias=bias.cuda(), res=rr if with_res else None)
            torch.cuda.synchronize()
            folded[flags, with_res] = yr
    want = F.leaky_relu((F.conv2d(x, w, bias, 1, 1)) * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1) + res, 0.2)
    assert relmax(from_rows(folded[CONV_ALONE, True], B, H, H, cout), want) < TOL[dtype]
    assert torch.equal(folded[0, True], folded[CONV_ALONE, True])
    assert torch.equal(folded[0, False], folded[CONV_ALONE, False])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_output_gate(dtype):
    """gate without res: the output is multiplied by the LeakyReLU derivative of `gate` (gather
    kernel, here a stride-2 conv); a layer whose weights are packed for the patch kernel refuses it."""
    ops, _desc, _wmap = _mods()
    g = torch.Generator().manual_seed(6)
    B, H, cin, cout = 3, 12, 16, 40
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / 12
    gate = torch.randn(B, cout, H // 2, H // 2, generator=g)
    if dtype == torch.bfloat16:
        x, w, gate = [t.bfloat16().float() for t in (x, w, gate)]
    ref = F.conv2d(x, w, None, 2, 1) * torch.where(gate > 0, 1.0, 0.2)
    d = _desc(dtype, B, H, H, cin, H // 2, H // 2, cout, 3, 9, stride=2, off=(-1, -1))
    wm = _wmap(1, cin * 9, 0, cin, 1, 9, cout, cin, 9)
    yr = torch.zeros(B * (H // 2) ** 2, cout, dtype=dtype, device="cuda")
    ops.conv_fwd(d, to_rows(x, dtype), pack(ops, wm, w, d, dtype), yr, gate=to_rows(gate, dtype), ws=ws_for(ops, d))
    assert relmax(from_rows(yr, B, H // 2, H // 2, cout), ref) < TOL[dtype]
    d1 = _desc(dtype, 2, 8, 8, 32, 8, 8, 64, 3, 9, off=(-1, -1))          # stride 1, 32 channels: patch kernel
    if ops.conv_fwd_kernel(d1).startswith("patch_fwd"):
        wm1 = _wmap(1, 32 * 9, 0, 32, 1, 9, 64, 32, 9)
        w1 = torch.randn(64, 32, 3, 3, generator=g)
        y1 = torch.zeros(2 * 64, 64, dtype=dtype, device="cuda")
        x1 = torch.zeros(2 * 64, 32, dtype=dtype, device="cuda")
        with pytest.raises(RuntimeError, match="gather kernel only"):
            ops.conv_fwd(d1, x1, pack(ops, wm1, w1, d1, dtype), y1, gate=y1, ws=ws_for(ops, d1))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,cin,cout", [(3, 12, 16, 32), (2, 16, 64, 128), (5, 8, 128, 256), (9, 4, 32, 32),
                                          (70, 2, 64, 64), (3, 40, 72, 64), (2, 64, 64, 128)])
def test_s2_input_gradient_one_launch(dtype, B, H, cin, cout):
    """CVG_FORM_S2_TRANSPOSED: the input gradient of a 3x3 stride-2 convolution with all four output-parity classes
    in one launch, against autograd, with and without the LeakyReLU gate on the output; tiles that hold several
    images, ragged tiles (40 -> 20 x 20 grid), channel counts off the 64 / 128 tile and a 1 x 1 dY grid."""
    ops, _desc, _wmap = _mods()
    from cvae_gan_amd._lib import FORM_S2_TRANSPOSED
    g = torch.Generator().manual_seed(B * 100 + H + cin)
    x = torch.randn(B, cin, H, H, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    if dtype == torch.bfloat16:
        w = w.bfloat16().float()
    OH = H // 2
    y = F.conv2d(x, w, None, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    gate = torch.randn(B, cin, H, H, generator=g)
    if dtype == torch.bfloat16:
        dy, gate = dy.bfloat16().float(), gate.bfloat16().float()
    y.backward(dy)
    d = _desc(dtype, B, OH, OH, cout, OH, OH, cin, 3, 9, y_dims=(H, H))
    d.form = FORM_S2_TRANSPOSED
    assert ops.conv_fwd_kernel(d) == "patch_dgrad_s2"
    wm = _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, 9)
    pk = pack(ops, wm, w, d, dtype)
    dyr = to_rows(dy, dtype)
    for with_gate in (False, True):
        dxr = torch.full((B * H * H, cin), 7.0, dtype=dtype, device="cuda")
        ops.conv_fwd(d, dyr, pk, dxr, gate=to_rows(gate, dtype) if with_gate else None)
        torch.cuda.synchronize()
        ref = x.grad * (torch.where(gate > 0, 1.0, 0.2) if with_gate else 1.0)
        assert relmax(from_rows(dxr, B, H, H, cin), ref) < TOL[dtype]


def test_s2_input_gradient_rejects_unsupported():
    ops, _desc, _wmap = _mods()
    from cvae_gan_amd._lib import FORM_S2_TRANSPOSED
    d = _desc(torch.bfloat16, 2, 4, 4, 16, 4, 4, 8, 3, 9, y_dims=(8, 8))       # 16 dY channels: not a multiple of 32
    d.form = FORM_S2_TRANSPOSED
    assert ops.conv_fwd_kernel(d) == "unsupported"
    t = torch.zeros(4096, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="not supported"):
        ops.conv_fwd(d, t, t, t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,h,cin,cout", [(3, 5, 16, 24), (2, 8, 64, 32), (5, 1, 32, 8), (2, 2, 8, 16), (1, 16, 128, 64)])
def test_upconv_backward_at_low_resolution(dtype, B, h, cin, cout):
    """deconv3x3 (bilinear x2, ReflectionPad2d(1), 3x3 conv; networks.py:274-281): input and weight gradient through
    cvg_upsample_taps_bwd and two one-tap GEMMs over the low-resolution pixels, against autograd of the up-sampled
    form; 1 x 1 and 2 x 2 inputs exercise the edge clamp and the mirror rows together."""
    ops, _desc, _wmap = _mods()
    g = torch.Generator().manual_seed(B * 100 + h * 10 + cin)
    x = torch.randn(B, cin, h, h, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    dy = torch.randn(B, cout, 2 * h, 2 * h, generator=g)
    if dtype == torch.bfloat16:
        x, w, dy = x.bfloat16().float(), w.bfloat16().float(), dy.bfloat16().float()
    x.requires_grad_(True)
    w.requires_grad_(True)
    u = F.pad(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), (1, 1, 1, 1), mode="reflect")
    F.conv2d(u, w).backward(dy)
    dyr = to_rows(dy, dtype)
    z = torch.full((B * h * h, 9 * cout), 7.0, dtype=dtype, device="cuda")
    ops.upsample_taps_bwd(dyr, z, B, h, h, cout)
    dd = _desc(dtype, B, h, h, 9 * cout, h, h, cin, 1, 1)
    dm = _wmap(1, 9, 0, cout, 1, cin * 9, cin, cout, 9)
    dxr = torch.zeros(B * h * h, cin, dtype=dtype, device="cuda")
    ops.conv_fwd(dd, z, pack(ops, dm, w.detach(), dd, dtype), dxr, ws=ws_for(ops, dd))
    wd = _desc(dtype, B, h, h, cin, h, h, 9 * cout, 1, 1)
    wm = _wmap(cout, 1, cin * 9, cin, 0, 9, 9 * cout, cin, 1)
    gw = torch.full(w.shape, 7.0, device="cuda")
    ops.conv_wgrad(wd, to_rows(x.detach(), dtype), z, 9 * cout, wm, gw, ws_for(ops, wd))
    torch.cuda.synchronize()
    tol = TOL[dtype] if dtype == torch.float32 else 3e-2        # bf16: z is rounded once more than the up-sampled form
    assert relmax(from_rows(dxr, B, h, h, cin), x.grad) < tol
    assert relmax(gw.cpu(), w.grad) < tol
    # forward the same way: v = x . W' (9 cout columns), then the spatial operator and the bias
    bias = torch.randn(cout, generator=g)
    y_ref = F.conv2d(u, w, bias).detach()
    vd = _desc(dtype, B, h, h, cin, h, h, 9 * cout, 1, 1)
    v = torch.full((B * h * h, 9 * cout), 7.0, dtype=dtype, device="cuda")
    ops.conv_fwd(vd, to_rows(x.detach(), dtype), pack(ops, wm, w.detach(), vd, dtype), v, ws=ws_for(ops, vd))
    yr = torch.full((B * 4 * h * h, cout), 7.0, dtype=dtype, device="cuda")
    ops.upsample_taps_fwd(v, bias.cuda(), yr, B, h, h, cout)
    torch.cuda.synchronize()
    assert relmax(from_rows(yr, B, 2 * h, 2 * h, cout), y_ref) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reflect_7x7_to3(dtype):
    """Decoder tail: ReflectionPad2d(3) + Conv2d(C, 3, 7) with 3 of 4 stored channels valid."""
    ops, _desc, _wmap = _mods()
    from cvae_gan_amd._lib import PAD_REFLECT
    g = torch.Generator().manual_seed(9)
    B, P, C = 2, 16, 16
    x = torch.randn(B, C, P, P, generator=g)
    w = torch.randn(3, C, 7, 7, generator=g) / 28
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w)
    d = _desc(dtype, B, P, P, C, P, P, 4, 7, 49, off=(-3, -3), pad_mode=PAD_REFLECT)
    wm = _wmap(1, C * 49, 0, C, 1, 49, 3, C, 49)
    yr = torch.zeros(B * P * P, 4, dtype=dtype, device="cuda")
    ops.conv_fwd(d, to_rows(x, dtype), pack(ops, wm, w, d, dtype), yr, ws=ws_for(ops, d))
    out = yr.float().view(B, P, P, 4)
    assert float(out[..., 3].abs().max()) == 0.0
    assert relmax(out[..., :3].permute(0, 3, 1, 2).cpu(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(6, 1024, 32), (256, 8192, 128), (37, 136, 528)])
def test_linear_split_k(dtype, M, K, N):
    """Linear layers run as 1x1 'convolutions'; long K with few tiles takes the split-K path."""
    ops, _desc, _wmap = _mods()
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.linear(x, w, b)
    d = _desc(dtype, M, 1, 1, K, 1, 1, N, 1, 1)
    wm = _wmap(1, K, 0, K, 0, 1, N, K, 1)
    yr = torch.zeros(M, N, dtype=dtype, device="cuda")
    ops.conv_fwd(d, x.to(dtype).cuda(), pack(ops, wm, w, d, dtype), yr, bias=b.cuda(), ws=ws_for(ops, d))
    assert relmax(yr.float().cpu(), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,P,ngf", [(3, 16, 32), (2, 32, 64)])
def test_tail_weight_gradient_through_windows(dtype, B, P, ngf):
    """cvg_conv_wgrad_strided: the 7x7, 3-output-channel tail's weight gradient as 7 row taps with
    the (window pixel, channel) pairs of a zero-padded 4-channel dpre image as output channels,
    against torch's conv2d weight gradient (ReflectionPad2d(3) + Conv2d(ngf, 3, 7), networks.py:146)."""
    ops, _desc, _wmap = _mods()
    from cvae_gan_amd._lib import PAD_REFLECT
    g = torch.Generator().manual_seed(B * 100 + P + ngf)
    x = torch.randn(B, ngf, P, P, generator=g)
    dpre = torch.randn(B, 3, P, P, generator=g)
    if dtype == torch.bfloat16:
        x, dpre = x.bfloat16().float(), dpre.bfloat16().float()
    w = torch.zeros(3, ngf, 7, 7, requires_grad=True)
    F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w).backward(dpre)
    ref = w.grad
    # window image [B][P+12][P+14][4], interior at (6, 6)
    win = torch.zeros(B, P + 12, P + 14, 4)
    win[:, 6:6 + P, 6:6 + P, :3] = dpre.permute(0, 2, 3, 1)
    win = win.to(dtype).cuda()
    Wp = P + 14
    desc = _desc(dtype, B, P, P, ngf, P, P + 6, 32, 1, 7, off=(-3, -3), pad_mode=PAD_REFLECT,
                 y_pix=4, y_row=Wp * 4, y_img=(P + 12) * Wp * 4)
    wmap = _wmap(4, -1, ngf * 49, ngf, 1, 49, 28, ngf, 7, tap_mode=2, table=[kh * 7 + 6 for kh in range(7)],
                 nr_valid=3)
    assert ops.conv_wgrad_kernel(desc).startswith("patch_wgrad")
    grad = torch.full((3, ngf, 7, 7), 7.0, device="cuda")
    ops.conv_wgrad_strided(desc, to_rows(x, dtype), win[:, 6:], wmap, grad, ws_for(ops, desc))
    torch.cuda.synchronize()
    assert relmax(grad.cpu(), ref) <= TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,P,ngf", [(3, 16, 32), (2, 32, 64)])
def test_tail_forward_through_column_partials(dtype, B, P, ngf):
    """7-row-tap convolution with (kernel column, channel) output pairs + cvg_tail_sum_tanh against
    ReflectionPad2d(3) + Conv2d(ngf, 3, 7) + Tanh (networks.py:146-149)."""
    ops, _desc, _wmap = _mods()
    from cvae_gan_amd._lib import PAD_REFLECT
    g = torch.Generator().manual_seed(B * 100 + P + ngf + 1)
    x = torch.randn(B, ngf, P, P, generator=g)
    w = torch.randn(3, ngf, 7, 7, generator=g) / (7 * ngf ** 0.5)
    bias = torch.randn(3, generator=g) * 0.1
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref_pre = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w, bias)
    desc = _desc(dtype, B, P, P, ngf, P, P + 6, 32, 1, 7, off=(-3, -3), pad_mode=PAD_REFLECT)
    wmap = _wmap(4, 1, ngf * 49, ngf, 7, 49, 28, ngf, 7, nr_valid=3)
    assert ops.conv_fwd_kernel(desc).startswith("patch_fwd")
    packed = pack(ops, wmap, w, desc, dtype)
    v = torch.zeros(B * P * (P + 6), 32, dtype=dtype, device="cuda")
    ops.conv_fwd(desc, to_rows(x, dtype), packed, v, ws=ws_for(ops, desc))
    pre = torch.zeros(B * P * P, 4, dtype=dtype, device="cuda")
    dec = torch.zeros(B, 3, P, P, device="cuda")
    ops.tail_sum_tanh(v, bias.cuda(), pre, dec, B, P)
    torch.cuda.synchronize()
    got_pre = from_rows(pre, B, P, P, 4)[:, :3]
    assert relmax(got_pre, ref_pre) <= TOL[dtype]
    assert float((dec.cpu() - torch.tanh(ref_pre)).abs().max()) <= (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("M,C,mean,std,dtype", [(4, 64, 1.0, 0.1, torch.float32), (16, 64, 3.0, 0.5, torch.bfloat16),
                                                 (4096, 64, 100.0, 1.0, torch.float32), (262144, 128, 10.0, 1.0, torch.bfloat16),
                                                 (1048576, 64, 0.0, 1.0, torch.float32), (3, 8, -50.0, 0.01, torch.float32)])
def test_batchnorm_statistics_accuracy(M, C, mean, std, dtype):
    """cvg_bn_stats against float64 moments of the same values, also where |mean| >> std and for a handful of
    rows: the sums are taken about the first row, so the variance keeps its digits (sums of raw values measure
    2e-4 relative at 4 rows with mean / std = 10 and 1e-3 at mean / std = 100).  Bounds: mean within 1e-6 of
    (|mean| + std), variance within 5e-6 relative (1e-4 at mean / std = 5000)."""
    from cvae_gan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + C)
    x = (mean + std * torch.randn(M, C, device="cuda", generator=g)).to(dtype)
    xd = x.double()
    m_ref, v_ref = xd.mean(0), xd.var(0, unbiased=False)
    stats = torch.zeros(2, C, device="cuda")
    ws = torch.zeros(2 * ops.bn_stats_blocks(M, C) * C + 64, device="cuda")
    ops.bn_stats(x, M, C, stats, None, None, None, ws)
    torch.cuda.synchronize()
    assert float(((stats[0].double() - m_ref).abs() / (m_ref.abs() + std)).max()) <= 1e-6
    var_got = 1.0 / stats[1].double() ** 2 - 1e-5
    tol = 1e-4 if abs(mean) / std > 1000 else 5e-6
    assert float(((var_got - v_ref).abs() / v_ref).max()) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,with_res", [(3000, 64, False), (3000, 64, True), (517, 128, False), (96, 256, True)])
def test_batchnorm_leakyrelu_forward_backward(dtype, M, C, with_res):
    """cvg_bn_stats + cvg_bn_act_fwd + cvg_bn_act_bwd against torch autograd on the CPU.
    Without a residual input the backward is run twice: with the gate read from y and with the
    gate recomputed from x (y = NULL); the two must agree bit for bit.  A third run hands over dy
    already gated (y = NULL, beta = NULL)."""
    ops, _, _ = _mods()
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 1.5 + 0.3
    res = torch.randn(M, C, generator=g) if with_res else None
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    dy = torch.randn(M, C, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
        res = res.bfloat16().float() if with_res else None
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    pre = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    yr = F.leaky_relu(pre + res if with_res else pre, 0.2)
    yr.backward(dy)

    dev = lambda t: t.to(dtype).cuda().contiguous()
    f32 = lambda t: t.float().cuda().contiguous()
    xd, dyd = dev(x), dev(dy)
    yd, dxd = torch.empty_like(xd), torch.empty_like(xd)
    stats = torch.zeros(2 * C, device="cuda")
    rm, rv, nbt = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    ws = torch.zeros(2 * 1024 * 1024, device="cuda")
    gd, bd = f32(gamma), f32(beta)
    ops.bn_stats(xd, M, C, stats, rm, rv, nbt, ws)
    ops.bn_act_fwd(xd, dev(res) if with_res else None, yd, M, C, stats, gd, bd)
    tol = TOL[dtype]
    assert relmax(yd.float().cpu(), yr.detach()) <= tol
    # gradients are compared with the gate torch used, so feed the backward torch's y
    yd = dev(yr.detach())
    outs = []
    for y_arg in ([yd] if with_res else [yd, None]):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dxd.zero_()
        ops.bn_act_bwd(dyd, y_arg, xd, dxd, M, C, stats, gd, dg, db, ws, beta=bd)
        torch.cuda.synchronize()
        outs.append((dxd.clone(), dg.clone(), db.clone()))
        assert relmax(dg.cpu(), gr.grad) <= max(tol, 1e-4)
        assert relmax(db.cpu(), br.grad) <= max(tol, 1e-4)
        assert relmax(dxd.float().cpu(), xr.grad) <= max(tol, 1e-4)
    if not with_res and dtype == torch.float32:
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    # third form: the caller hands over dy already multiplied by the activation's derivative
    # (y = NULL and no beta); in f32 it must equal the gated-from-y result bit for bit
    gated = dy * torch.where(yr.detach() > 0, 1.0, 0.2)
    if dtype == torch.bfloat16:
        gated = gated.bfloat16().float()
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dxd.zero_()
    ops.bn_act_bwd(dev(gated), None, xd, dxd, M, C, stats, gd, dg, db, ws, beta=None)
    torch.cuda.synchronize()
    assert relmax(dg.cpu(), gr.grad) <= max(tol, 1e-4)
    assert relmax(db.cpu(), br.grad) <= max(tol, 1e-4)
    assert relmax(dxd.float().cpu(), xr.grad) <= max(tol, 1e-4)
    if dtype == torch.float32:
        for a, b in zip((dxd, dg, db), outs[0]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,cin,cout", [(5, 20, 32, 64), (3, 16, 64, 128), (2, 12, 32, 40), (7, 9, 128, 256)])
def test_conv_epilogue_statistics(dtype, B, H, cin, cout):
    """cvg_conv_fwd_stats + cvg_bn_stats_finalize: the stored output equals cvg_conv_fwd's bit for bit, and
    the batch statistics (mean, 1/sqrt(var + eps), running statistics, batch counter) equal those of
    cvg_bn_stats over that output up to f32 summation order (ragged tiles, several channel tiles)."""
    ops, _desc, _wmap = _mods()
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    bias = torch.randn(cout, generator=g).cuda()
    d = _desc(dtype, B, H, H, cin, H, H, cout, 3, 9, off=(-1, -1))
    wm = _wmap(1, cin * 9, 0, cin, 1, 9, cout, cin, 9)
    nblk = ops.conv_stats_partials(d)
    assert nblk > 0 and ops.conv_fwd_kernel(d).startswith("patch_fwd")
    xr, wp = to_rows(x, dtype), pack(ops, wm, w, d, dtype)
    M = B * H * H
    y0 = torch.zeros(M, cout, dtype=dtype, device="cuda")
    y1 = torch.zeros_like(y0)
    ops.conv_fwd(d, xr, wp, y0, bias=bias, ws=ws_for(ops, d))
    part = torch.full((2 * nblk * cout,), float("nan"), device="cuda")
    ops.conv_fwd_stats(d, xr, wp, y1, part, bias=bias)
    out = []
    for fused in (False, True):
        stats = torch.zeros(2, cout, device="cuda")
        rm, rv = torch.full((cout,), 0.3, device="cuda"), torch.full((cout,), 1.7, device="cuda")
        nbt = torch.full((), 4, dtype=torch.long, device="cuda")
        if fused:
            ops.bn_stats_finalize(part, nblk, M, cout, stats, rm, rv, nbt)
        else:
            ws = torch.zeros(2 * ops.bn_stats_blocks(M, cout) * cout, device="cuda")
            ops.bn_stats(y0, M, cout, stats, rm, rv, nbt, ws)
        torch.cuda.synchronize()
        out.append((stats, rm, rv, int(nbt)))
    assert torch.equal(y0, y1)
    assert torch.isfinite(part).all()
    (s0, m0, v0, n0), (s1, m1, v1, n1) = out
    assert n0 == n1 == 5
    assert torch.allclose(s0[0], s1[0], atol=1e-5, rtol=1e-5) and torch.allclose(s0[1], s1[1], rtol=1e-5)
    assert torch.allclose(m0, m1, atol=1e-6, rtol=1e-5) and torch.allclose(v0, v1, rtol=1e-5)
    # the gather kernel's layers report that they have no such epilogue
    d2 = _desc(dtype, B, H, H, cin, H // 2, H // 2, cout, 3, 9, stride=2, off=(-1, -1)) if H % 2 == 0 else None
    if d2 is not None:
        assert ops.conv_stats_partials(d2) == 0


def test_resident_row_tap_kernel_bit_identical():
    """The stem's shape class (7 row taps over 8-pixel x 4-channel windows, K = 224) runs on a kernel of its own
    (resident workgroups, raw-pixel staging); CVG_NO_LOWK=1 (read once per process, hence the child process) keeps it
    on patch_fwd.  Same summation order: the outputs, ragged 70 x 70 grid and class bias included, are bit-identical.
    Likewise the tail's forward shape and the stem's image gradient (patch_fwd with resident weights; CVG_NO_RES_ROWS=1)."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from cvae_gan_amd import ops
from cvae_gan_amd.engine import _desc, _wmap
torch.manual_seed(0)
T = torch.bfloat16
outs = []
for B, P, OH, pad, ngf, y in ((6, 64, 64, (6, 8), 64, 5), (3, 70, 70, (6, 8), 64, 0), (5, 12, 12, (6, 8), 64, 3), (9, 8, 8, (6, 8), 128, 4)):
    Hp, Wp = OH + pad[0], OH + pad[1]
    table = [(kh * 7 + kw) if kw < 7 else -1 for kh in range(7) for kw in range(8)]
    d = _desc(T, B, Hp, Wp, 32, OH, OH, ngf, 1, 7, x_pix=4)
    wm = _wmap(1, (3 + y) * 49, 0, 4, 1, 49, ngf, 3, 56, tap_mode=2, table=table)
    x = torch.randn(B * Hp * Wp * 4 + 64, device="cuda").to(T)
    w = torch.randn(ngf, 3 + y, 7, 7, device="cuda") * 0.05
    out = torch.zeros(B * OH * OH * ngf, device="cuda", dtype=T)
    packed = torch.zeros(d.w_rows * d.w_ld, dtype=T, device="cuda")
    ops.weight_pack(d, wm, w, packed)
    ws = torch.zeros(max(ops.conv_fwd_workspace(d), 1024) // 4 + 16, device="cuda")
    bias = torch.randn(ngf, device="cuda")
    kw = {}
    if y:
        kw = dict(cbias=torch.randn(y, ngf, device="cuda"), labels=torch.randint(0, y, (B,), device="cuda", dtype=torch.int32))
    ops.conv_fwd(d, x, packed, out, bias=bias, ws=ws, **kw)
    torch.cuda.synchronize()
    outs.append(out.float().cpu())
# the tail's forward shape (7 row taps x 64 channels, reflect padding) and the stem's image gradient of the same shape:
# weights resident in registers, workgroups walking over the image groups; ragged batches and sizes included
from cvae_gan_amd.engine import PAD_REFLECT
names = []
for B, P, ngf in ((100, 64, 64), (64, 32, 64), (66, 40, 64)):
    for d, wm, wshape in ((_desc(T, B, P, P, ngf, P, P + 6, 32, 1, 7, off=(-3, -3), pad_mode=PAD_REFLECT),
                           _wmap(4, 1, ngf * 49, ngf, 7, 49, 28, ngf, 7, nr_valid=3), (3, ngf, 7, 7)),
                          (_desc(T, B, P, P, ngf, P + 6, P, 32, 1, 7, off=(-6, 0)),
                           _wmap(4, 1, 49, ngf, 7, 8 * 49, 28, ngf, 7, tap_mode=1, nr_valid=3), (ngf, 8, 7, 7))):
        x = torch.randn(B * P * P * ngf, device="cuda").to(T)
        w = torch.randn(*wshape, device="cuda") * 0.05
        out = torch.zeros(B * (P + 6) * P * 32, device="cuda", dtype=T)
        packed = torch.zeros(d.w_rows * d.w_ld, dtype=T, device="cuda")
        ops.weight_pack(d, wm, w, packed)
        ws = torch.zeros(max(ops.conv_fwd_workspace(d), 1024) // 4 + 16, device="cuda")
        ops.conv_fwd(d, x, packed, out, ws=ws)
        torch.cuda.synchronize()
        outs.append(out.float().cpu())
        names.append(ops.conv_fwd_kernel(d))
torch.save(outs, sys.argv[1])
print(" ".join(sorted(set(n.replace(" ", "_") for n in names))))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    got = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, env in (("a", {}), ("b", {"CVG_NO_LOWK": "1", "CVG_NO_RES_ROWS": "1"})):
            path = os.path.join(tmp, name + ".pt")
            p = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env), capture_output=True, text=True,
                               timeout=280)
            assert p.returncode == 0, p.stderr[-2000:]
            got.append(torch.load(path, weights_only=True))
            assert ("resident_weights" in p.stdout) == (name == "a"), p.stdout
    for i, (a, b) in enumerate(zip(*got)):
        assert float(a.abs().max()) > 0
        assert torch.equal(a, b), "case %d: %d of %d elements differ, largest difference %g" % (
            i, int((a != b).sum()), a.numel(), float((a - b).abs().max()))



def test_mfma_rate_probe_runs():
    """cvg_mfma_rate_probe (measurement aid of bench.py): launches, writes one finite value per thread and reports the
    FLOPs it executed."""
    ops, _desc, _wmap = _mods()
    out = torch.full((8 * 256,), float("nan"), device="cuda")
    flops = ops.mfma_rate_probe(out, 8, 10)
    torch.cuda.synchronize()
    assert flops == 8 * 4 * 10 * 32 * 32768
    assert torch.isfinite(out).all()
