"""TEST INFRASTRUCTURE ONLY -- fp32 CPU reference of ONE recorded op of a compiled layer program, computed from the
GPU's own input of that op, so that the difference to the GPU's output is that launch's bf16 rounding alone.

What an op restates from the reference:
  conv / conv_cat2   ``act(bn(conv(x)))`` of the Conv blocks (luoyang/yolov5/YoloV5Body.py:26-51,
                     yolov8/YoloV8Body.py:26-63, yolov7_pose/YoloV7PoseBody.py:24-48, yolov6_face/YoloV6FaceBody.py:98-134,
                     yolov4/YoloV4BackBone.py:18-43, yolov3/DarkNet53.py:22-37) with the BatchNorm already folded into
                     weight and bias (utils/torch_utils.py:344-370), the shortcut add of the residual units
                     (YoloV5Body.py:80-93, DarkNet53.py:10-47; weighted by ``alpha`` in yolov6/YoloV6Body.py:451-472), and
                     ``torch.cat((nn.Upsample(2)(a), b), 1)`` in front of a 1x1 conv (YoloV5Body.py:232-240)
  bneck              Bottleneck cv1 -> cv2 (+ x) (YoloV5Body.py:80-93)
  c3                 C3.forward with one Bottleneck (YoloV5Body.py:96-111)
  stem / stem2       Focus + conv (YoloV5Body.py:54-77) or the first 3x3 conv, and the 3x3 stride-2 conv after it
  mp5 / mp2 / up2    nn.MaxPool2d(k, 1, k // 2) chains of SPP / SPPF (YoloV5Body.py:114-132, yolov8/YoloV8Body.py:102-114),
                     MP (YoloV7PoseBody.py:76-82), nn.Upsample(scale_factor=2, mode="nearest")
  d2s                the scatter half of nn.ConvTranspose2d(k=2, s=2) (YoloV6FaceBody.py:475-492)

Weights are the layer's folded fp32 weights rounded to bf16 (what the GPU multiplies with); accumulation, bias,
activation and residual in fp32; intermediate tensors that the GPU keeps in bf16 are rounded to bf16 here too.
"""
from typing import Callable, Dict, Tuple

import torch
import torch.nn.functional as F

ACT_NONE, ACT_SILU, ACT_LEAKY, ACT_MISH, ACT_RELU = 0, 1, 2, 3, 4


def act(y: torch.Tensor, kind: int) -> torch.Tensor:
    if kind == ACT_SILU:
        return y * torch.sigmoid(y)
    if kind == ACT_LEAKY:
        return F.leaky_relu(y, 0.1)
    if kind == ACT_MISH:
        return y * torch.tanh(F.softplus(y))
    if kind == ACT_RELU:
        return F.relu(y)
    return y


def bf16(x: torch.Tensor) -> torch.Tensor:
    return x.bfloat16().float()


def _conv(x, w, b, stride, kind):
    k = w.shape[-1]
    return act(F.conv2d(x, bf16(w), b, stride=stride, padding=k // 2), kind)


def focus(images: torch.Tensor) -> torch.Tensor:
    return torch.cat((images[..., ::2, ::2], images[..., 1::2, ::2], images[..., ::2, 1::2], images[..., 1::2, 1::2]), 1)


def stem_ref(layer, images: torch.Tensor) -> torch.Tensor:
    """First conv of the network from the fp32 image (mode 0: 3x3 stride 1, 1: 3x3 stride 2, 2: Focus + 3x3)."""
    x = bf16(images)
    if layer.mode == 2:
        return _conv(focus(x), layer._w, layer._b, 1, layer.act)
    return _conv(x, layer._w, layer._b, 1 if layer.mode == 0 else 2, layer.act)


def op_reference(op, read: Callable, images: torch.Tensor) -> Dict[str, Tuple[torch.Tensor, torch.Tensor, bool]]:
    """{label: (want, got, exact)} for one recorded op.  ``read(view)`` returns the GPU's current contents of an
    activation view as float32 NCHW on the CPU; ``images`` are the program's input images (CPU, fp32)."""
    kind = op[0]
    if kind == "conv":
        _, layer, x, out, res, out2 = op
        k, s, _cin, _cout, a, f32 = layer._args
        y = _conv(read(x)[:, :layer.cin_real], layer._w, layer._b, s, a)
        if res is not None:
            y = y + layer.residual_scale * read(res)
        got = read(out) if out2 is None else torch.cat((read(out), read(out2)), 1)
        return {f"conv {layer.name}": (y, got, False)}
    if kind == "conv_cat2":
        _, layer, xa, a_half, xb, b_half, out, out2 = op
        ta, tb = read(xa), read(xb)
        if a_half:
            ta = F.interpolate(ta, scale_factor=2, mode="nearest")
        if b_half:
            tb = F.interpolate(tb, scale_factor=2, mode="nearest")
        y = _conv(torch.cat((ta, tb), 1)[:, :layer.cin_real], layer._w, layer._b, 1, layer._args[4])
        got = read(out) if out2 is None else torch.cat((read(out), read(out2)), 1)
        return {f"conv_cat2 {layer.name}": (y, got, False)}
    if kind == "bneck":
        _, layer, x, out, shortcut = op
        xin = read(x)
        l1, l2 = layer.first, layer.second
        t = bf16(_conv(xin[:, :l1.cin_real], l1._w, l1._b, 1, l1._args[4]))
        y = _conv(t, l2._w, l2._b, 1, l2._args[4])
        if shortcut:
            y = y + xin
        return {f"bneck {layer.name}": (y, read(out), False)}
    if kind == "c3":
        _, layer, x, out, shortcut = op
        le, lu, lx = layer.entry, layer.unit, layer.exit
        ch = layer.ch
        ab = bf16(_conv(read(x)[:, :le.cin_real], le._w, le._b, 1, le._args[4]))       # channels [cv2 | cv1]
        b, a = ab[:, :ch], ab[:, ch:]
        t = bf16(_conv(a, lu.first._w, lu.first._b, 1, lu.first._args[4]))
        u = _conv(t, lu.second._w, lu.second._b, 1, lu.second._args[4])
        m = bf16(a + u if shortcut else u)
        y = _conv(torch.cat((m, b), 1), lx._w, lx._b, 1, lx._args[4])
        return {f"c3 {layer.name}": (y, read(out), False)}
    if kind == "stem":
        _, layer, out = op
        return {f"stem {layer.name}": (stem_ref(layer, images), read(out), False)}
    if kind == "stem2":
        _, layer, out = op
        t = bf16(stem_ref(layer.stem, images))
        return {f"stem2 {layer.name}": (_conv(t, layer._w, layer._b, 2, layer.act), read(out), False)}
    if kind == "mp5":
        _, x, y1, y2, y3, kernel = op
        t, res = read(x), {}
        for i, yv in enumerate((y1, y2, y3)):
            t = F.max_pool2d(t, kernel, 1, kernel // 2)
            res[f"maxpool{kernel} #{i + 1}"] = (t, read(yv), True)
        return res
    if kind == "mp2":
        return {"maxpool2": (F.max_pool2d(read(op[1]), 2, 2), read(op[2]), True)}
    if kind == "up2":
        return {"upsample2x": (F.interpolate(read(op[1]), scale_factor=2, mode="nearest"), read(op[2]), True)}
    if kind == "copy":
        return {"copy": (read(op[1]), read(op[2]), True)}
    if kind == "d2s":
        t = read(op[1])
        b, c4, h, w = t.shape
        c = c4 // 4
        want = t.view(b, 2, 2, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(b, c, 2 * h, 2 * w)
        return {"depth_to_space2": (want, read(op[2]), True)}
    return {}
