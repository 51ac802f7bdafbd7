"""GPU parity tests of the individual kernels, called through the C ABI (ctypes) on cuda:0.

conv: against torch-CPU fp32 conv2d on the same bf16-rounded inputs and weights; the only
differences are fp32 accumulation order and the final bf16 rounding of the output:
relative L2 <= 4e-3 and max-abs <= 1.5e-2 * max|y| are enforced.
layout / pool / upsample kernels: exact.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def _act(y, act):
    if act == 1:
        return y * torch.sigmoid(y)
    if act == 2:
        return F.leaky_relu(y, 0.1)
    if act == 3:
        return y * torch.tanh(F.softplus(y))
    if act == 4:
        return F.relu(y)
    return y


def _run_conv(dev, x, w, b, k, s, act, res=None, out_f32=False, in_extra=0, out_extra=0, tiling=None, res_scale=None):
    """x (B,Cin,H,W) fp32 (already bf16-representable); returns (B,Cout,Ho,Wo) fp32 from the HIP kernel.
    in_extra/out_extra: additional channels around the tensor inside a wider buffer (cstride test)."""
    from yolo_amd import lib as L
    Bn, cin, H, W = x.shape
    cout = w.shape[0]
    cin_st = (cin + 15) // 16 * 16
    d = L.ConvDesc()
    L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), k, s, cin_st, cout, act, int(out_f32)))
    if tiling is not None:
        L.check(L.LIB.yamd_conv_desc_tune(ctypes.byref(d), *tiling))
    assert d.residual_scale == 1.0
    if res_scale is not None:
        d.residual_scale = res_scale
    wp = torch.empty(d.packed_w_elems, dtype=torch.int16)
    bp = torch.empty(d.cout_pad, dtype=torch.float32)
    wc, bc = w.contiguous(), b.contiguous()
    L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), wc.data_ptr(), cin, bc.data_ptr(), wp.data_ptr(), bp.data_ptr()))
    xin = torch.zeros((Bn, H, W, cin_st + in_extra), dtype=torch.bfloat16)
    xin[..., in_extra:in_extra + cin] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    if in_extra:
        xin[..., :in_extra] = 7.0     # neighbours in the concat buffer must not leak in
    xin = xin.to(dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    ydt = torch.float32 if out_f32 else torch.bfloat16
    y = torch.full((Bn, Ho, Wo, cout + out_extra), -3.0, dtype=ydt, device=dev)
    rp, rcs = None, 0
    if res is not None:
        rt = res.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous().to(dev)
        rp, rcs = rt.data_ptr(), cout
    wd, bd = wp.to(dev), bp.to(dev)
    L.check(L.LIB.yamd_conv2d_bias_act(ctypes.byref(d), xin.data_ptr() + in_extra * 2, xin.shape[-1], wd.data_ptr(),
                                       bd.data_ptr(), rp, rcs, y.data_ptr() + out_extra * y.element_size(),
                                       y.shape[-1], Bn, H, W, L.current_stream()), "conv")
    torch.cuda.synchronize()
    yc = y.cpu().float()
    if out_extra:
        assert torch.all(yc[..., :out_extra] == -3.0), "conv wrote outside its channel slice"
    return yc[..., out_extra:].permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # k, s, cin, cout, H, W, act, residual, out_f32   (covers every built tiling, both tile widths, ragged edges)
    (1, 1, 32, 32, 16, 16, 1, False, False),
    (1, 1, 32, 64, 20, 20, 1, False, False),
    (1, 1, 32, 128, 9, 13, 1, False, False),
    (1, 1, 64, 32, 32, 32, 1, False, False),
    (1, 1, 128, 64, 40, 40, 1, False, False),
    (1, 1, 256, 128, 20, 20, 1, True, False),
    (1, 1, 1024, 512, 20, 20, 1, False, False),
    (1, 1, 128, 75, 16, 16, 0, False, True),
    (1, 1, 256, 75, 20, 20, 0, False, True),
    (1, 1, 512, 171, 10, 10, 0, False, True),
    (3, 1, 12, 32, 32, 32, 1, False, False),
    (3, 1, 16, 128, 20, 20, 2, False, False),
    (3, 1, 32, 32, 24, 40, 1, True, False),
    (3, 1, 64, 64, 17, 19, 1, True, False),
    (3, 1, 128, 128, 40, 40, 1, True, False),
    (3, 1, 256, 256, 20, 20, 3, False, False),
    (3, 2, 32, 64, 32, 32, 1, False, False),
    (3, 2, 64, 128, 40, 40, 1, False, False),
    (3, 2, 128, 256, 23, 31, 4, False, False),
    (3, 2, 256, 512, 40, 40, 1, False, False),
    # input widths that are multiples of 16 but not of 32 (m-size networks): 16-channel chunks
    (1, 1, 48, 48, 20, 20, 1, True, False),
    (1, 1, 48, 96, 17, 23, 1, False, False),
    (1, 1, 144, 160, 20, 20, 1, False, False),
    (3, 1, 48, 48, 24, 24, 1, True, False),
    (3, 1, 80, 80, 20, 20, 1, False, False),
    (3, 2, 48, 96, 32, 32, 1, False, False),
    (1, 1, 80, 75, 20, 20, 0, False, True),
    (1, 1, 96, 75, 16, 16, 0, False, True),
    # narrow float heads (box sides, one class score, landmarks): one 32-column group, element-parallel slice stores
    (1, 1, 64, 4, 20, 20, 0, False, True),
    (1, 1, 64, 1, 17, 23, 0, False, True),
    (1, 1, 128, 10, 40, 40, 0, False, True),
    (1, 1, 48, 32, 10, 10, 0, False, True),
    (1, 1, 96, 17, 12, 20, 0, False, True),
]


@pytest.mark.parametrize("k,s,cin,cout,H,W,act,use_res,out_f32", CONV_CASES)
def test_conv_parity(gpu_device, k, s, cin, cout, H, W, act, use_res, out_f32):
    g = torch.Generator().manual_seed(k * 1000 + cin + cout + H)
    Bn = 3
    x = _bf16_round(torch.randn((Bn, cin, H, W), generator=g))
    w = _bf16_round(torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn((cout,), generator=g) * 0.5
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = _bf16_round(torch.randn((Bn, cout, Ho, Wo), generator=g)) if use_res else None
    want = _act(F.conv2d(x, w, b, stride=s, padding=k // 2), act)
    if res is not None:
        want = want + res
    got = _run_conv(gpu_device, x, w, b, k, s, act, res, out_f32, in_extra=16 if cin % 16 == 0 else 0,
                    out_extra=8 if not out_f32 else 3)
    assert got.shape == want.shape
    err = (got - want).double()
    rel_l2 = float(err.norm() / want.double().norm())
    max_abs = float(err.abs().max())
    tol_l2, tol_abs = (1e-4, 1e-3) if out_f32 else (4e-3, 1.5e-2)
    assert rel_l2 <= tol_l2, f"rel L2 {rel_l2}"
    assert max_abs <= tol_abs * float(want.abs().max()), f"max abs {max_abs} vs {float(want.abs().max())}"


@pytest.mark.parametrize("ca,cb,cout,H,W,a_half,b_half,tiling,split", [
    (256, 256, 256, 40, 40, 1, 0, None, 128),            # v5s conv3_for_up_sample1: cat([up(p5), feat2]) into cv2+cv1
    (128, 128, 128, 80, 80, 1, 0, (1, 4, 64, 8, 1), 64),   # v5s conv3_for_up_sample2 with its tiling
    (512, 256, 256, 40, 40, 1, 0, (2, 4, 64, 4, 1), 0),    # v8s conv3_for_up_sample1.cv1
    (64, 128, 64, 24, 36, 0, 1, (2, 2, 64, 4, 1), 0),      # the upsampled part second; sizes that leave partial tiles
    (64, 64, 96, 20, 20, 0, 0, (2, 4, 32, 4, 1), 0),       # a plain cat of two full-resolution tensors
    (128, 64, 128, 18, 22, 1, 1, (2, 4, 64, 4, 2), 0),     # both parts upsampled; two-stage pipelined kernel
    (64, 192, 128, 40, 40, 1, 0, (1, 4, 64, 8, 2), 64),    # two-stage kernel, split outputs
])
def test_conv_over_cat_with_upsample(gpu_device, ca, cb, cout, H, W, a_half, b_half, tiling, split):
    """yamd_conv2d_bias_act_cat2 = 1x1 conv over torch.cat((a, b), 1) with either part behind a nearest 2x upsample.
    Against torch on the materialised concatenation (bf16 output tolerance) and, bit for bit, against
    yamd_conv2d_bias_act on the same materialised tensor: the two read the same values in the same order."""
    from yolo_amd import lib as L
    g = torch.Generator().manual_seed(ca * 3 + cb + H)
    Bn = 3
    dev = gpu_device

    def part(c, half):
        h, w = (H // 2, W // 2) if half else (H, W)
        t = _bf16_round(torch.randn((Bn, c, h, w), generator=g))
        full = F.interpolate(t, scale_factor=2, mode="nearest") if half else t
        return t, full

    a, a_full = part(ca, a_half)
    b, b_full = part(cb, b_half)
    xcat = torch.cat((a_full, b_full), 1)
    cin = ca + cb
    w = _bf16_round(torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5)
    bias = torch.randn((cout,), generator=g) * 0.5
    want = _act(F.conv2d(xcat, w, bias), 1)
    d = L.ConvDesc()
    L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), 1, 1, cin, cout, 1, 0))
    if tiling is not None:
        L.check(L.LIB.yamd_conv_desc_tune(ctypes.byref(d), *tiling))
    wp = torch.empty(d.packed_w_elems, dtype=torch.int16)
    bp = torch.empty(d.cout_pad, dtype=torch.float32)
    L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), w.contiguous().data_ptr(), cin, bias.data_ptr(), wp.data_ptr(), bp.data_ptr()))
    wd, bd = wp.to(dev), bp.to(dev)
    nhwc = lambda t, extra: torch.cat((t.permute(0, 2, 3, 1), torch.zeros(t.shape[0], t.shape[2], t.shape[3], extra)), -1).to(torch.bfloat16).contiguous().to(dev)
    ad, bdv, xd = nhwc(a, 16), nhwc(b, 8), nhwc(xcat, 0)
    st = L.current_stream()
    c1 = split if split else cout
    y = torch.full((Bn, H, W, c1 + 8), -3.0, dtype=torch.bfloat16, device=dev)
    y2 = torch.full((Bn, H, W, cout - c1 + 8), -3.0, dtype=torch.bfloat16, device=dev) if split else None
    L.check(L.LIB.yamd_conv2d_bias_act_cat2(ctypes.byref(d), ad.data_ptr(), ad.shape[-1], ca, a_half, bdv.data_ptr(), bdv.shape[-1],
                                            b_half, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), y.shape[-1],
                                            None if y2 is None else y2.data_ptr(), 0 if y2 is None else y2.shape[-1], split, Bn, H, W,
                                            st), "conv_cat2")
    ref = torch.empty((Bn, H, W, cout), dtype=torch.bfloat16, device=dev)
    L.check(L.LIB.yamd_conv2d_bias_act(ctypes.byref(d), xd.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(), None, 0, ref.data_ptr(),
                                       cout, Bn, H, W, st), "conv")
    torch.cuda.synchronize()
    assert torch.all(y[..., c1:] == -3.0) and (y2 is None or torch.all(y2[..., cout - c1:] == -3.0))
    got = y[..., :c1] if y2 is None else torch.cat((y[..., :c1], y2[..., :cout - c1]), -1)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    err = (got.cpu().float().permute(0, 3, 1, 2) - want).double()
    assert float(err.norm() / want.double().norm()) <= 4e-3
    # refused, not mis-computed: a first part that is not a multiple of the staged chunk, odd sizes under an upsample
    assert L.LIB.yamd_conv2d_bias_act_cat2(ctypes.byref(d), ad.data_ptr(), ad.shape[-1], ca - 8, a_half, bdv.data_ptr(), bdv.shape[-1],
                                           b_half, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), y.shape[-1], None, 0, 0, Bn, H, W, st) != 0
    if a_half or b_half:
        assert L.LIB.yamd_conv2d_bias_act_cat2(ctypes.byref(d), ad.data_ptr(), ad.shape[-1], ca, a_half, bdv.data_ptr(),
                                               bdv.shape[-1], b_half, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), y.shape[-1], None, 0,
                                               0, Bn, H - 1, W, st) != 0


@pytest.mark.parametrize("k,cin,cout,H,W,act,alpha", [(3, 64, 64, 20, 20, 4, 0.6), (3, 128, 128, 17, 23, 1, -1.25),
                                                       (1, 96, 96, 40, 40, 4, 0.0)])
def test_conv_weighted_residual(gpu_device, k, cin, cout, H, W, act, alpha):
    """y = act(conv(x)) + alpha * residual: the learned shortcut weight of YOLOv6's BottleRep."""
    g = torch.Generator().manual_seed(77 + cin + H)
    x = _bf16_round(torch.randn((2, cin, H, W), generator=g))
    w = _bf16_round(torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn((cout,), generator=g) * 0.5
    res = _bf16_round(torch.randn((2, cout, H, W), generator=g))
    want = _act(F.conv2d(x, w, b, padding=k // 2), act) + alpha * res
    got = _run_conv(gpu_device, x, w, b, k, 1, act, res, res_scale=alpha)
    err = (got - want).double()
    assert float(err.norm() / want.double().norm()) <= 4e-3
    assert float(err.abs().max()) <= 1.5e-2 * float(want.abs().max())


@pytest.mark.parametrize("cin,cout,H,W", [(128, 75, 80, 80), (64, 75, 24, 20), (64, 21, 13, 40), (64, 75, 19, 19),
                                          (32, 171, 16, 36)])
def test_conv_f32_dense_output(gpu_device, cin, cout, H, W):
    """Raw head logits written densely (pixel stride == cout): the flat store path, with 16-byte stores when the
    width is a multiple of 4 and scalar stores otherwise."""
    g = torch.Generator().manual_seed(cin + cout + H + W)
    x = _bf16_round(torch.randn((3, cin, H, W), generator=g))
    w = _bf16_round(torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5)
    b = torch.randn((cout,), generator=g)
    want = F.conv2d(x, w, b)
    got = _run_conv(gpu_device, x, w, b, 1, 1, 0, None, True)
    err = (got - want).double()
    assert float(err.norm() / want.double().norm()) <= 1e-4
    assert float(err.abs().max()) <= 1e-3 * float(want.abs().max())


@pytest.mark.parametrize("k,cin,cout,H,W,extra", [(1, 64, 20, 80, 80, 64), (1, 64, 64, 40, 40, 4), (1, 32, 4, 17, 23, 8),
                                                  (1, 128, 132, 20, 20, 12), (1, 64, 20, 9, 7, 2)])
def test_conv_f32_channel_slice(gpu_device, k, cin, cout, H, W, extra):
    """fp32 logits written into a channel slice of a wider pixel (the anchor-free heads): 16-byte pieces when the slice
    is aligned (extra % 4 == 0), scalar stores otherwise; the neighbouring channels stay untouched."""
    g = torch.Generator().manual_seed(cin + cout + H + W + extra)
    x = _bf16_round(torch.randn((2, cin, H, W), generator=g))
    w = _bf16_round(torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn((cout,), generator=g)
    want = F.conv2d(x, w, b, padding=k // 2)
    got = _run_conv(gpu_device, x, w, b, k, 1, 0, None, True, out_extra=extra)
    err = (got - want).double()
    assert float(err.norm() / want.double().norm()) <= 1e-4
    assert float(err.abs().max()) <= 1e-3 * float(want.abs().max())


TILING_CASES = [
    # k, s, cin, cout, H, W, residual, (mrep, nrep, ck, nwaves): every tiling the per-layer tuner may pick
    (1, 1, 64, 128, 40, 40, True, (1, 4, 32, 4)),
    (1, 1, 64, 128, 23, 17, False, (1, 4, 64, 8)),
    (1, 1, 128, 128, 40, 40, True, (2, 4, 64, 8)),
    (1, 1, 64, 64, 32, 32, False, (1, 2, 64, 8)),
    (1, 1, 64, 64, 20, 20, True, (2, 2, 64, 8)),
    (1, 1, 64, 128, 20, 20, False, (1, 4, 32, 8)),
    (1, 1, 64, 32, 20, 20, False, (1, 1, 64, 4)),
    (3, 1, 64, 128, 40, 40, True, (2, 4, 16, 8)),
    (3, 1, 64, 128, 19, 21, False, (1, 4, 16, 8)),
    (3, 1, 64, 64, 32, 32, True, (2, 2, 32, 8)),
    (3, 1, 64, 64, 20, 20, False, (1, 2, 32, 8)),
    (3, 1, 32, 32, 40, 40, True, (1, 1, 32, 8)),
    (3, 1, 32, 32, 16, 16, False, (2, 1, 16, 8)),
    (3, 1, 64, 64, 20, 20, False, (1, 2, 32, 4)),
    (3, 2, 64, 128, 40, 40, False, (1, 4, 16, 8)),
    (3, 2, 64, 64, 31, 23, False, (1, 2, 16, 8)),
    (3, 2, 64, 64, 32, 32, False, (1, 2, 32, 8)),
    (3, 2, 64, 64, 40, 40, False, (1, 2, 16, 4)),
    (3, 2, 64, 128, 40, 40, False, (2, 4, 16, 8)),
    (3, 2, 64, 128, 37, 29, False, (2, 4, 16, 4)),
    (3, 2, 32, 64, 64, 64, False, (2, 2, 16, 8)),
    (3, 2, 32, 64, 23, 41, False, (2, 2, 16, 4)),
]
TILING_CASES = [c[:7] + (c[7] + (1,),) for c in TILING_CASES]


def _second_set_cases():
    """One case per kernel of the second instantiation set (two LDS stages; 32-channel chunks with 128 couts), read
    from the table the kernels are generated from: 2 or 3 input-channel chunks (both parities of the stage ping-pong),
    ragged image edges, a partly filled last cout tile on every other case."""
    import json
    import os
    from conftest import ROOT
    with open(os.path.join(ROOT, "yolo_amd", "csrc", "conv_variants.json")) as f:
        tables = json.load(f)
    cases = []
    for key, (k, s) in (("k1b", (1, 1)), ("k3s1b", (3, 1)), ("k3s2b", (3, 2))):
        for i, (mrep, nrep, ck, f32, nwaves, stages) in enumerate(tables[key]):
            cin = ck * (3 if i % 2 else 2)
            cout = 32 * nrep + (40 if i % 2 else 0)
            H, W = ((37, 29) if i % 2 else (40, 40))
            cases.append((k, s, cin, cout, H, W, bool(i % 3 == 0), (mrep, nrep, ck, nwaves, stages)))
    return cases


TILING_CASES += _second_set_cases()
# one cout tile per workgroup (stages 3) for each kernel family, with a residual, ragged edges and a partial last tile
TILING_CASES += [(3, 2, 64, 512, 40, 40, False, (1, 4, 16, 8, 3)), (1, 1, 128, 296, 20, 20, True, (2, 4, 64, 4, 3)),
                 (3, 1, 32, 256, 23, 17, True, (2, 4, 16, 8, 3)), (3, 1, 64, 296, 40, 40, True, (2, 4, 16, 4, 3))]


@pytest.mark.parametrize("k,s,cin,cout,H,W,use_res,tiling", TILING_CASES)
def test_conv_tilings(gpu_device, k, s, cin, cout, H, W, use_res, tiling):
    g = torch.Generator().manual_seed(k * 77 + cin + cout + H + sum(tiling))
    Bn = 3
    x = _bf16_round(torch.randn((Bn, cin, H, W), generator=g))
    w = _bf16_round(torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn((cout,), generator=g) * 0.5
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = _bf16_round(torch.randn((Bn, cout, Ho, Wo), generator=g)) if use_res else None
    want = _act(F.conv2d(x, w, b, stride=s, padding=k // 2), 1)
    if res is not None:
        want = want + res
    got = _run_conv(gpu_device, x, w, b, k, s, 1, res, False, in_extra=16, out_extra=8, tiling=tiling)
    err = (got - want).double()
    assert float(err.norm() / want.double().norm()) <= 4e-3
    assert float(err.abs().max()) <= 1.5e-2 * float(want.abs().max())


@pytest.mark.parametrize("k,s,cin,cout,H,W,use_res,tiling", [
    (3, 1, 64, 128, 20, 20, True, (1, 4, 16, 4, 1)),      # 128-pixel tiles of 6 rows over 5 stacked 20-row images
    (3, 1, 64, 128, 19, 19, False, (2, 4, 16, 4, 1)),
    (3, 1, 64, 128, 19, 19, True, (2, 4, 32, 8, 1)),      # 512-pixel tiles: every tile spans two images
    (3, 1, 32, 64, 10, 12, True, (1, 2, 32, 8, 1)),
    (3, 2, 64, 128, 40, 40, False, (2, 4, 16, 4, 1)),     # stride 2: two zero rows between the stacked images
    (3, 2, 64, 128, 38, 38, False, (1, 4, 16, 4, 1)),
    (3, 2, 32, 64, 12, 20, False, (1, 2, 16, 8, 1)),
    # two LDS images: one workgroup walks the three cout tiles (the last one partly filled) of its pixel tile
    (3, 1, 80, 320, 19, 19, True, (2, 4, 16, 8, 2)),
    (3, 1, 48, 296, 26, 22, False, (2, 4, 16, 4, 2)),
    (1, 1, 96, 328, 20, 20, True, (2, 4, 32, 8, 2)),
    (3, 2, 48, 200, 38, 38, False, (1, 4, 16, 4, 2)),
])
def test_conv_small_maps_stacked_images(gpu_device, k, s, cin, cout, H, W, use_res, tiling):
    """Small feature maps: the launcher tiles the whole batch as one tall image (zero rows between the images), so
    tiles straddle images; every image, the first and last rows of each in particular, must still equal the
    per-image convolution."""
    g = torch.Generator().manual_seed(k * 91 + cin + cout + H + W + sum(tiling))
    Bn = 5
    x = _bf16_round(torch.randn((Bn, cin, H, W), generator=g))
    w = _bf16_round(torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn((cout,), generator=g) * 0.5
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = _bf16_round(torch.randn((Bn, cout, Ho, Wo), generator=g)) if use_res else None
    want = _act(F.conv2d(x, w, b, stride=s, padding=k // 2), 1)
    if res is not None:
        want = want + res
    got = _run_conv(gpu_device, x, w, b, k, s, 1, res, False, in_extra=16, out_extra=8, tiling=tiling)
    for i in range(Bn):
        err = (got[i] - want[i]).double()
        assert float(err.norm() / want[i].double().norm()) <= 4e-3, f"image {i}"
        assert float(err.abs().max()) <= 1.5e-2 * float(want.abs().max()), f"image {i}"


def test_conv_identity_asymmetric(gpu_device):
    """1x1 identity weights with an asymmetric input: catches any row/column or k-order permutation exactly."""
    cin = cout = 64
    x = torch.arange(2 * cin * 5 * 7, dtype=torch.float32).reshape(2, cin, 5, 7) % 251 - 125
    w = torch.eye(cout).reshape(cout, cin, 1, 1)
    got = _run_conv(gpu_device, x, w, torch.zeros(cout), 1, 1, 0)
    assert torch.equal(got, x)
    # 3x3 with a single off-centre tap = pure spatial shift (checks tap order and zero padding)
    w3 = torch.zeros(cout, cin, 3, 3)
    w3[torch.arange(cout), torch.arange(cin), 0, 2] = 1.0
    got = _run_conv(gpu_device, x, w3, torch.zeros(cout), 3, 1, 0)
    want = F.conv2d(x, w3, padding=1)
    assert torch.equal(got, want)
    got = _run_conv(gpu_device, x, w3, torch.zeros(cout), 3, 2, 0)
    assert torch.equal(got, F.conv2d(x, w3, stride=2, padding=1))


def test_focus_pack(gpu_device):
    from yolo_amd import lib as L
    img = torch.rand(2, 3, 12, 20)
    y = torch.full((2, 6, 10, 16), 9.0, dtype=torch.bfloat16, device=gpu_device)
    d = img.to(gpu_device)
    L.check(L.LIB.yamd_focus_pack(d.data_ptr(), y.data_ptr(), 16, 2, 12, 20, L.current_stream()))
    torch.cuda.synchronize()
    want = torch.cat([img[..., ::2, ::2], img[..., 1::2, ::2], img[..., ::2, 1::2], img[..., 1::2, 1::2]], 1)
    want = want.permute(0, 2, 3, 1).to(torch.bfloat16)
    got = y.cpu()
    assert torch.equal(got[..., :12], want) and torch.all(got[..., 12:] == 0)


def test_upsample_copy_maxpool(gpu_device):
    from yolo_amd import lib as L
    st = L.current_stream()
    x = torch.randn(2, 5, 7, 24).to(torch.bfloat16)
    xd = x.to(gpu_device)
    y = torch.zeros((2, 10, 14, 40), dtype=torch.bfloat16, device=gpu_device)
    L.check(L.LIB.yamd_upsample2x(xd.data_ptr() + 8 * 2, 24, y.data_ptr() + 16 * 2, 40, 2, 5, 7, 16, st))
    z = torch.zeros((2, 5, 7, 32), dtype=torch.bfloat16, device=gpu_device)
    L.check(L.LIB.yamd_copy_channels(xd.data_ptr(), 24, z.data_ptr() + 8 * 2, 32, 2 * 5 * 7, 24, st))
    # SPP: x in slice 0 of a 4-slice buffer, pools 5/9/13 into the others
    H = W = 20
    c = 16
    cat = torch.zeros((3, H, W, 4 * c), dtype=torch.bfloat16)
    src = torch.randn(3, H, W, c).to(torch.bfloat16)
    cat[..., :c] = src
    cd = cat.to(gpu_device)
    p = cd.data_ptr()
    L.check(L.LIB.yamd_maxpool5_cascade(p, 4 * c, p + c * 2, p + 2 * c * 2, p + 3 * c * 2, 4 * c, 3, H, W, c, st))
    torch.cuda.synchronize()
    up = x[..., 8:24].float().permute(0, 3, 1, 2)
    want_up = F.interpolate(up, scale_factor=2, mode="nearest").permute(0, 2, 3, 1).to(torch.bfloat16)
    got = y.cpu()
    assert torch.equal(got[..., 16:32], want_up) and torch.all(got[..., :16] == 0) and torch.all(got[..., 32:] == 0)
    assert torch.equal(z.cpu()[..., 8:32], x)
    s = src.float().permute(0, 3, 1, 2)
    got = cd.cpu()
    assert torch.equal(got[..., :c], src)
    for i, k in enumerate((5, 9, 13)):
        want = F.max_pool2d(s, k, 1, k // 2).permute(0, 2, 3, 1).to(torch.bfloat16)
        assert torch.equal(got[..., (i + 1) * c:(i + 2) * c], want), f"pool {k}"
    # the 3x3 cascade: pools 3/5/7 of the face / pose SPPF
    cd3 = cat.to(gpu_device)
    p3 = cd3.data_ptr()
    L.check(L.LIB.yamd_maxpool3_cascade(p3, 4 * c, p3 + c * 2, p3 + 2 * c * 2, p3 + 3 * c * 2, 4 * c, 3, H, W, c, st))
    got = cd3.cpu()
    assert torch.equal(got[..., :c], src)
    for i, k in enumerate((3, 5, 7)):
        want = F.max_pool2d(s, k, 1, k // 2).permute(0, 2, 3, 1).to(torch.bfloat16)
        assert torch.equal(got[..., (i + 1) * c:(i + 2) * c], want), f"pool {k}"


@pytest.mark.parametrize("H,W,c,kernel", [(20, 20, 256, 5), (13, 17, 64, 5), (40, 40, 128, 5), (64, 64, 8, 5), (19, 19, 24, 3),
                                            (20, 20, 96, 3), (12, 40, 32, 5)])
def test_maxpool_cascade_channel_groups(gpu_device, H, W, c, kernel):
    """The pool cascade for plane sizes and channel counts that take 8, 4, 2 and 1 pieces of 8 channels per workgroup:
    x in slice 1 of a 5-slice buffer, the three pools into slices 2..4, slice 0 untouched; exact against torch."""
    from yolo_amd import lib as L
    g = torch.Generator().manual_seed(H * 31 + c)
    Bn = 2
    cat = torch.zeros((Bn, H, W, 5 * c), dtype=torch.bfloat16)
    cat[..., :c] = 7.0
    src = torch.randn((Bn, H, W, c), generator=g).to(torch.bfloat16)
    cat[..., c:2 * c] = src
    cd = cat.to(gpu_device)
    p = cd.data_ptr()
    fn = L.LIB.yamd_maxpool5_cascade if kernel == 5 else L.LIB.yamd_maxpool3_cascade
    L.check(fn(p + c * 2, 5 * c, p + 2 * c * 2, p + 3 * c * 2, p + 4 * c * 2, 5 * c, Bn, H, W, c, L.current_stream()))
    torch.cuda.synchronize()
    got = cd.cpu()
    assert torch.all(got[..., :c] == 7.0) and torch.equal(got[..., c:2 * c], src)
    s = src.float().permute(0, 3, 1, 2)
    for i in range(3):
        k = kernel + i * (kernel - 1)
        want = F.max_pool2d(s, k, 1, k // 2).permute(0, 2, 3, 1).to(torch.bfloat16)
        assert torch.equal(got[..., (i + 2) * c:(i + 3) * c], want), f"pool {k}"


def test_maxpool2_and_conv_transpose(gpu_device):
    """MaxPool2d(2,2) exact; ConvTranspose2d(2,2)+bias = 1x1 conv to 4c channels + depth-to-space, through the
    engine ops, against torch fp32 on bf16-rounded operands (rel L2 <= 1e-2: one bf16 rounding of the output)."""
    from yolo_amd import lib as L
    from yolo_amd.engine import Program
    st = L.current_stream()
    x = torch.randn(2, 6, 10, 24).to(torch.bfloat16)
    xd = x.to(gpu_device)
    y = torch.zeros((2, 3, 5, 32), dtype=torch.bfloat16, device=gpu_device)
    L.check(L.LIB.yamd_maxpool2(xd.data_ptr() + 8 * 2, 24, y.data_ptr() + 8 * 2, 32, 2, 6, 10, 16, st))
    torch.cuda.synchronize()
    want = F.max_pool2d(x[..., 8:24].float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).to(torch.bfloat16)
    got = y.cpu()
    assert torch.equal(got[..., 8:24], want) and torch.all(got[..., :8] == 0) and torch.all(got[..., 24:] == 0)

    # transposed conv through a tiny program: image16 -> 3x3 conv to 32 ch -> ConvTranspose2d(32, 32, 2, 2)
    torch.manual_seed(3)
    P = Program(16, 24)
    w0 = torch.randn(32, 3, 3, 3) * 0.3
    wt = torch.randn(32, 32, 2, 2) * 0.2
    bt = torch.randn(32)
    a = P.conv("c0", P.image16(3), w0, None, 1, L.ACT_NONE)
    P.outputs["a"] = a
    P.outputs["up"] = P.conv_transpose2x("t", a, wt, bt)
    img = torch.rand(2, 3, 16, 24)
    out = P.run(img.to(gpu_device))
    av = out["a"].float().cpu().permute(0, 3, 1, 2)                      # the bf16 activations the GPU fed forward
    want = F.conv_transpose2d(av, _bf16_round(wt), bt, stride=2)
    got = out["up"].float().cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape
    rel = torch.linalg.norm(got - want) / torch.linalg.norm(want)
    assert rel <= 1e-2, float(rel)


@pytest.mark.parametrize("k,s,cin,cout,split,H,W", [(1, 1, 64, 64, 32, 40, 40), (1, 1, 32, 128, 96, 17, 23), (3, 1, 32, 64, 8, 20, 20),
                                                    (3, 2, 64, 128, 64, 32, 32)])
def test_conv_split_output(gpu_device, k, s, cin, cout, split, H, W):
    """yamd_conv2d_bias_act_split: couts [0, split) land in one channel slice, the rest densely in a second buffer;
    values are those of the plain conv, bit for bit, and nothing outside the two targets is written"""
    from yolo_amd import lib as L
    g = torch.Generator().manual_seed(cin + cout + split)
    Bn = 2
    x = _bf16_round(torch.randn((Bn, cin, H, W), generator=g))
    w = _bf16_round(torch.randn((cout, cin, k, k), generator=g) * (2.0 / (cin * k * k)) ** 0.5)
    b = torch.randn((cout,), generator=g) * 0.5
    want = _run_conv(gpu_device, x, w, b, k, s, 1)
    d = L.ConvDesc()
    L.check(L.LIB.yamd_conv_desc_init(ctypes.byref(d), k, s, cin, cout, 1, 0))
    wp = torch.empty(d.packed_w_elems, dtype=torch.int16)
    bp = torch.empty(d.cout_pad, dtype=torch.float32)
    L.check(L.LIB.yamd_conv_pack_weights(ctypes.byref(d), w.contiguous().data_ptr(), cin, b.contiguous().data_ptr(), wp.data_ptr(),
                                         bp.data_ptr()))
    xin = x.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous().to(gpu_device)
    Ho, Wo = want.shape[2], want.shape[3]
    y1 = torch.full((Bn, Ho, Wo, split + 16), -3.0, dtype=torch.bfloat16, device=gpu_device)      # slice [8, 8+split)
    y2 = torch.full((Bn, Ho, Wo, cout - split + 8), -5.0, dtype=torch.bfloat16, device=gpu_device)  # slice [0, cout-split)
    wd, bd = wp.to(gpu_device), bp.to(gpu_device)
    L.check(L.LIB.yamd_conv2d_bias_act_split(ctypes.byref(d), xin.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(),
                                             y1.data_ptr() + 8 * 2, y1.shape[-1], y2.data_ptr(), y2.shape[-1], split, Bn, H, W,
                                             L.current_stream()), "conv split")
    torch.cuda.synchronize()
    a, c = y1.cpu().float(), y2.cpu().float()
    assert torch.all(a[..., :8] == -3.0) and torch.all(a[..., 8 + split:] == -3.0) and torch.all(c[..., cout - split:] == -5.0)
    got = torch.cat((a[..., 8:8 + split], c[..., :cout - split]), -1).permute(0, 3, 1, 2)
    assert torch.equal(got, want)
    # bad split points are rejected
    assert L.LIB.yamd_conv2d_bias_act_split(ctypes.byref(d), xin.data_ptr(), cin, wd.data_ptr(), bd.data_ptr(), y1.data_ptr(),
                                            y1.shape[-1], y2.data_ptr(), y2.shape[-1], split + 4, Bn, H, W, L.current_stream()) != 0


@pytest.mark.parametrize("mode,c1,c2,H,W,act1,act2", [
    ("focus", 32, 64, 64, 128, 1, 1),      # YOLOv5-s: Focus stem, then dark2.0; 16 x 32 outputs = whole 4 x 16 tiles
    ("focus", 32, 64, 72, 88, 1, 1),       # 18 x 22 outputs: tiles cut by the bottom and right edges
    ("k3s2", 32, 64, 96, 160, 1, 1),       # YOLOv8-s: 3x3 stride-2 stem
    ("k3s2", 32, 128, 40, 72, 2, 4),       # four cout fragments (two per wavefront), other activations
    ("k3s2", 32, 40, 52, 36, 1, 0),        # second conv's last cout fragment partly filled, no activation
    ("focus", 64, 64, 44, 68, 1, 3),       # 64-channel stem output (4 k-steps per tap)
    ("k3s2", 64, 24, 36, 40, 4, 1),        # a single cout fragment: half of the wavefronts have no second-conv work
    ("k3s1", 32, 64, 38, 76, 2, 2),        # the stride-1 stems: DarkNet53 (LeakyReLU), CSPDarkNet (Mish), v7 (SiLU)
    ("k3s1", 32, 64, 64, 32, 3, 3),
    ("k3s1", 32, 64, 20, 44, 1, 1),
    ("k3s1", 32, 104, 18, 34, 1, 4),
])
def test_stem_and_next_conv_fused(gpu_device, mode, c1, c2, H, W, act1, act2):
    """yamd_stem_conv2 against torch: stem from the fp32 image, its output rounded to bf16 (what the unfused path
    stores), then Conv2d(c1, c2, 3, 2, 1); the output goes into a channel slice of a wider buffer."""
    from yolo_amd import lib as L
    from yolo_amd.engine import StemConv2Layer, StemLayer
    g = torch.Generator().manual_seed(H * 13 + c1 + c2)
    Bn = 3
    cin = 12 if mode == "focus" else 3
    x = torch.rand((Bn, 3, H, W), generator=g)
    w1 = _bf16_round(torch.randn((c1, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5)
    b1 = torch.randn((c1,), generator=g) * 0.5
    w2 = _bf16_round(torch.randn((c2, c1, 3, 3), generator=g) * (2.0 / (c1 * 9)) ** 0.5)
    b2 = torch.randn((c2,), generator=g) * 0.5
    xb = _bf16_round(x)
    if mode == "focus":
        xf = torch.cat((xb[..., ::2, ::2], xb[..., 1::2, ::2], xb[..., ::2, 1::2], xb[..., 1::2, 1::2]), 1)
        mid = F.conv2d(xf, w1, b1, stride=1, padding=1)
    else:
        mid = F.conv2d(xb, w1, b1, stride=1 if mode == "k3s1" else 2, padding=1)
    mid = _bf16_round(_act(mid, act1))
    want = _act(F.conv2d(mid, w2, b2, stride=2, padding=1), act2)
    stem = StemLayer("t", mode, w1, b1, act1)
    assert StemConv2Layer.supported(stem, c2) == (mode != "k3s1")      # the plan fuses stride-2 stems only (the library takes both)
    layer = StemConv2Layer(stem, "t2", w2, b2, act2)
    layer.to(gpu_device)
    Ho, Wo = want.shape[2], want.shape[3]
    y = torch.full((Bn, Ho, Wo, c2 + 16), -3.0, dtype=torch.bfloat16, device=gpu_device)
    xd = x.to(gpu_device)
    L.check(L.LIB.yamd_stem_conv2(stem.mode, xd.data_ptr(), Bn, H, W, stem.w_dev.data_ptr(), stem.b_dev.data_ptr(), c1, act1,
                                  layer.w_dev.data_ptr(), layer.b_dev.data_ptr(), c2, act2, y.data_ptr() + 8 * 2, y.shape[-1],
                                  L.current_stream()), "stem_conv2")
    torch.cuda.synchronize()
    yc = y.cpu().float()
    assert torch.all(yc[..., :8] == -3.0) and torch.all(yc[..., 8 + c2:] == -3.0)
    got = yc[..., 8:8 + c2].permute(0, 3, 1, 2)
    err = (got - want).double()
    assert float(err.norm() / want.double().norm()) <= 6e-3
    # a stem value that lands on the other side of a bf16 rounding boundary than torch's moves one product by 2^-8
    assert float(err.abs().max()) <= 3e-2 * float(want.abs().max())
    # shapes the fused form does not take are refused, not mis-computed
    assert L.LIB.yamd_stem_conv2_packed_elems(48, 96) == 0 and L.LIB.yamd_stem_conv2_packed_elems(64, 128) == 0
    # (a stride-1 stem with 64 channels, an unknown mode, an odd image)
    for bad_mode, bad_c1, bad_h in ((0, 64, H), (7, c1, H), (stem.mode, c1, H + 1)):
        assert L.LIB.yamd_stem_conv2(bad_mode, xd.data_ptr(), Bn, bad_h, W, stem.w_dev.data_ptr(), stem.b_dev.data_ptr(), bad_c1,
                                     act1, layer.w_dev.data_ptr(), layer.b_dev.data_ptr(), c2 if bad_c1 == c1 else 64, act2,
                                     y.data_ptr(), y.shape[-1], L.current_stream()) != 0


@pytest.mark.parametrize("mode,cout,H,W,act", [("k3s1", 32, 40, 72, 1), ("k3s1", 64, 33, 35, 2), ("k3s2", 32, 64, 96, 1),
                                               ("k3s2", 48, 30, 70, 4), ("k3s2", 16, 18, 34, 3), ("focus", 32, 64, 128, 1),
                                               ("focus", 64, 36, 44, 1)])
def test_stem_conv_from_image(gpu_device, mode, cout, H, W, act):
    """yamd_stem_conv against torch: 3x3 (stride 1 / 2) on the 3-channel image and the Focus variant, bf16-rounded
    image and weights, fp32 accumulation; output written into a channel slice of a wider buffer"""
    from yolo_amd import lib as L
    from yolo_amd.engine import StemLayer
    g = torch.Generator().manual_seed(H * 7 + cout)
    Bn = 3
    cin = 12 if mode == "focus" else 3
    x = torch.rand((Bn, 3, H, W), generator=g)
    w = _bf16_round(torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5)
    b = torch.randn((cout,), generator=g) * 0.5
    xb = _bf16_round(x)
    if mode == "focus":
        xf = torch.cat((xb[..., ::2, ::2], xb[..., 1::2, ::2], xb[..., ::2, 1::2], xb[..., 1::2, 1::2]), 1)
        want = F.conv2d(xf, w, b, stride=1, padding=1)
    else:
        want = F.conv2d(xb, w, b, stride=1 if mode == "k3s1" else 2, padding=1)
    want = _act(want, act)
    layer = StemLayer("t", mode, w, b, act)
    layer.to(gpu_device)
    Ho, Wo = want.shape[2], want.shape[3]
    y = torch.full((Bn, Ho, Wo, cout + 16), -3.0, dtype=torch.bfloat16, device=gpu_device)
    xd = x.to(gpu_device)
    L.check(L.LIB.yamd_stem_conv(layer.mode, xd.data_ptr(), Bn, H, W, layer.w_dev.data_ptr(), layer.b_dev.data_ptr(), cout, act,
                                 y.data_ptr() + 8 * 2, y.shape[-1], L.current_stream()), "stem")
    torch.cuda.synchronize()
    yc = y.cpu().float()
    assert torch.all(yc[..., :8] == -3.0) and torch.all(yc[..., 8 + cout:] == -3.0)
    got = yc[..., 8:8 + cout].permute(0, 3, 1, 2)
    err = (got - want).abs()
    tol = 2e-2 + 1e-2 * want.abs()          # bf16 output rounding
    assert (err <= tol).all(), f"max err {err.max()} at {want.flatten()[err.argmax()]}"


@pytest.mark.parametrize("cin,cmid,H,W,act,shortcut,batch", [
    (32, 32, 160, 160, 1, 1, 2), (64, 64, 80, 80, 1, 1, 2), (128, 128, 40, 40, 1, 1, 3), (64, 64, 80, 80, 1, 0, 1),
    (64, 32, 37, 45, 2, 1, 2), (128, 64, 19, 23, 3, 1, 2), (32, 32, 17, 16, 4, 0, 3), (128, 128, 20, 20, 1, 1, 5),
    (64, 64, 5, 3, 1, 1, 2),
])
def test_bottleneck_fused(gpu_device, cin, cmid, H, W, act, shortcut, batch):
    """yamd_bottleneck (1x1 conv, 3x3 conv and the shortcut in one launch, the 1x1 output kept in LDS) against the two
    yamd_conv2d_bias_act calls it replaces, the second one with 16-channel chunks (the fused kernel's order of
    accumulation): bit-identical, also at image borders (the 3x3 conv pads the 1x1 conv's output, not its input), with
    partial tiles, and when input and output are channel slices of wider buffers."""
    from yolo_amd import lib as L
    g = torch.Generator().manual_seed(cin * 1000 + cmid + H)
    cout = cin
    x = _bf16_round(torch.randn(batch, cin, H, W, generator=g))
    w1 = torch.randn(cmid, cin, 1, 1, generator=g) / cin ** 0.5
    b1 = torch.randn(cmid, generator=g) * 0.5
    w2 = torch.randn(cout, cmid, 3, 3, generator=g) / (3 * cmid ** 0.5)
    b2 = torch.randn(cout, generator=g) * 0.5
    t = _run_conv(gpu_device, x, w1, b1, 1, 1, act)
    want = _run_conv(gpu_device, t, w2, b2, 3, 1, act, res=x if shortcut else None, tiling=(1, 1, 16, 4, 1))

    n = L.LIB.yamd_bottleneck_packed_elems(cin, cmid, cout)
    assert n == cmid * cin + cout * cmid * 9
    wp = torch.empty(n, dtype=torch.int16)
    bp = torch.empty(cmid + cout, dtype=torch.float32)
    L.check(L.LIB.yamd_bottleneck_pack_weights(cin, cmid, cout, w1.contiguous().data_ptr(), b1.data_ptr(), w2.contiguous().data_ptr(),
                                               b2.data_ptr(), wp.data_ptr(), bp.data_ptr()))
    wd, bd = wp.to(gpu_device), bp.to(gpu_device)
    in_extra, out_extra = 8, 16
    xin = torch.full((batch, H, W, cin + in_extra), 7.0, dtype=torch.bfloat16)
    xin[..., in_extra:] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    xin = xin.to(gpu_device)
    y = torch.full((batch, H, W, cout + out_extra), -3.0, dtype=torch.bfloat16, device=gpu_device)
    L.check(L.LIB.yamd_bottleneck(cin, cmid, cout, act, xin.data_ptr() + in_extra * 2, xin.shape[-1], wd.data_ptr(), bd.data_ptr(),
                                  shortcut, y.data_ptr() + out_extra * 2, y.shape[-1], batch, H, W, L.current_stream()), "bottleneck")
    torch.cuda.synchronize()
    yc = y.cpu().float()
    assert torch.all(yc[..., :out_extra] == -3.0), "wrote outside its channel slice"
    got = yc[..., out_extra:].permute(0, 3, 1, 2)
    assert torch.equal(got, want), (float((got - want).abs().max()), int((got != want).sum()))

    # refusals: a shape that is not built, and an output that overlaps the input
    assert L.LIB.yamd_bottleneck_packed_elems(256, 256, 256) == 0
    assert L.LIB.yamd_bottleneck(256, 256, 256, act, xin.data_ptr(), 256, wd.data_ptr(), bd.data_ptr(), 0, y.data_ptr(), 256,
                                 1, 4, 4, L.current_stream()) != 0
    assert L.LIB.yamd_bottleneck(cin, cmid, cout, act, xin.data_ptr() + in_extra * 2, xin.shape[-1], wd.data_ptr(), bd.data_ptr(),
                                 0, xin.data_ptr() + in_extra * 2, xin.shape[-1], batch, H, W, L.current_stream()) != 0


@pytest.mark.parametrize("H,W,shortcut,batch", [(160, 160, 1, 2), (24, 40, 1, 3), (17, 19, 0, 2), (8, 16, 1, 1), (5, 3, 1, 2),
                                                (33, 160, 0, 1)])
def test_c3_block_fused(gpu_device, H, W, shortcut, batch):
    """yamd_c3_fused (a whole C3 block with one bottleneck in one launch) against the launches it replaces -- the 1x1 conv
    to both branches, yamd_bottleneck, the 1x1 conv over [bottleneck output | other branch] -- and against fp32 torch
    on the CPU with every intermediate tensor rounded to bf16.  The fused kernel accumulates in another order (channel
    order of the chained layers), so the bound is the conv bound, not bit identity; image borders (the 3x3 conv pads
    the inner 1x1 conv's output), partial tiles, images smaller than one tile and channel slices of wider buffers."""
    from yolo_amd import lib as L
    cin, ch, cout, act = 64, 32, 64, 1
    g = torch.Generator().manual_seed(H * 131 + W)
    x = _bf16_round(torch.randn(batch, cin, H, W, generator=g))
    w_cv1 = torch.randn(ch, cin, 1, 1, generator=g) / cin ** 0.5
    w_cv2 = torch.randn(ch, cin, 1, 1, generator=g) / cin ** 0.5
    w_m1 = torch.randn(ch, ch, 1, 1, generator=g) / ch ** 0.5
    w_m2 = torch.randn(ch, ch, 3, 3, generator=g) / (3 * ch ** 0.5)
    w_cv3 = torch.randn(cout, 2 * ch, 1, 1, generator=g) / (2 * ch) ** 0.5
    b_cv1, b_cv2, b_m1, b_m2 = (torch.randn(ch, generator=g) * 0.5 for _ in range(4))
    b_cv3 = torch.randn(cout, generator=g) * 0.5

    # fp32 reference, intermediates rounded to bf16 as both GPU paths store / keep them
    rb = _bf16_round
    conv = lambda t, w, b, k: _act(F.conv2d(t, rb(w), b, padding=k // 2), act)      # noqa: E731
    a, b_ = rb(conv(x, w_cv1, b_cv1, 1)), rb(conv(x, w_cv2, b_cv2, 1))
    u = conv(rb(conv(a, w_m1, b_m1, 1)), w_m2, b_m2, 3)
    m = rb(a + u if shortcut else u)
    want = conv(torch.cat((m, b_), 1), w_cv3, b_cv3, 1)

    # the launches the fused one replaces
    a_g = _run_conv(gpu_device, x, w_cv1, b_cv1, 1, 1, act)
    b_g = _run_conv(gpu_device, x, w_cv2, b_cv2, 1, 1, act)
    t_g = _run_conv(gpu_device, a_g, w_m1, b_m1, 1, 1, act)
    m_g = _run_conv(gpu_device, t_g, w_m2, b_m2, 3, 1, act, res=a_g if shortcut else None)
    apart = _run_conv(gpu_device, torch.cat((m_g, b_g), 1), w_cv3, b_cv3, 1, 1, act)

    n = L.LIB.yamd_c3_packed_elems(cin, ch, cout)
    assert n > 0 and L.LIB.yamd_c3_packed_elems(128, 64, 128) == 0
    wp = torch.empty(n, dtype=torch.int16)
    bp = torch.empty(4 * ch + cout, dtype=torch.float32)
    ws = [t.contiguous() for t in (w_cv1, b_cv1, w_cv2, b_cv2, w_m1, b_m1, w_m2, b_m2, w_cv3, b_cv3)]
    L.check(L.LIB.yamd_c3_pack_weights(cin, ch, cout, *[t.data_ptr() for t in ws], wp.data_ptr(), bp.data_ptr()))
    wd, bd = wp.to(gpu_device), bp.to(gpu_device)
    in_extra, out_extra = 8, 16
    xin = torch.full((batch, H, W, cin + in_extra), 7.0, dtype=torch.bfloat16)
    xin[..., in_extra:] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    xin = xin.to(gpu_device)
    y = torch.full((batch, H, W, cout + out_extra), -3.0, dtype=torch.bfloat16, device=gpu_device)
    L.check(L.LIB.yamd_c3_fused(cin, ch, cout, act, xin.data_ptr() + in_extra * 2, xin.shape[-1], wd.data_ptr(), bd.data_ptr(),
                                shortcut, y.data_ptr() + out_extra * 2, y.shape[-1], batch, H, W, L.current_stream()), "c3_fused")
    torch.cuda.synchronize()
    yc = y.cpu().float()
    assert torch.all(yc[..., :out_extra] == -3.0), "wrote outside its channel slice"
    got = yc[..., out_extra:].permute(0, 3, 1, 2)
    rel = lambda p_, q_: float((p_ - q_).norm() / q_.norm())      # noqa: E731
    assert rel(got, want) <= 6e-3, rel(got, want)
    assert rel(got, apart) <= 6e-3, rel(got, apart)
    assert float((got - want).abs().max()) <= 3e-2 * float(want.abs().max())
    # borders separately: an indexing slip there hides in the total
    for sl in (np.s_[:, :, 0], np.s_[:, :, -1], np.s_[:, :, :, 0], np.s_[:, :, :, -1]):
        assert rel(got[sl], want[sl]) <= 8e-3
    # refusals: other activations, and an output that overlaps the input
    assert L.LIB.yamd_c3_fused(cin, ch, cout, 2, xin.data_ptr(), cin + in_extra, wd.data_ptr(), bd.data_ptr(), 0, y.data_ptr(),
                               y.shape[-1], batch, H, W, L.current_stream()) != 0
    assert L.LIB.yamd_c3_fused(cin, ch, cout, act, xin.data_ptr() + in_extra * 2, xin.shape[-1], wd.data_ptr(), bd.data_ptr(), 0,
                               xin.data_ptr() + in_extra * 2, xin.shape[-1], batch, H, W, L.current_stream()) != 0
