"""Shared pieces of the model descriptions: parameter holders with the reference's attribute names,
the recorder that turns them into fused conv ops, and the TaskLayer-shaped base class.

A model module (``yolov5``, ``yolov8`` ...) builds a tree of bare ``nn.Module`` holders whose
``state_dict`` keys equal the reference's, then walks it once per image size to record an
``engine.Program``.  BatchNorm folding, RepVGG re-parameterisation and bf16 packing all happen at
that point ("compile"), never per batch.
"""
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import lib as _lib
from . import torch_utils
from .engine import Program, View


def conv_bn(cin: int, cout: int, k: int = 1, s: int = 1, eps: float = 1e-3, momentum: float = 0.03,
            conv_name: str = "conv", bn_name: str = "bn") -> nn.Module:
    """Holder of a bias-free conv followed by BatchNorm under the given attribute names."""
    m = nn.Module()
    setattr(m, conv_name, nn.Conv2d(cin, cout, (k, k), (s, s), k // 2, bias=False))
    setattr(m, bn_name, nn.BatchNorm2d(cout, eps=eps, momentum=momentum))
    return m


def folded(m: nn.Module, conv_name: str = "conv", bn_name: str = "bn"):
    """(weight, bias, stride) of a conv(+bn) holder with the BatchNorm folded in fp64; a holder whose bn was
    removed by ``fuse()`` carries its bias on the conv."""
    conv = getattr(m, conv_name)
    bn = getattr(m, bn_name, None)
    if isinstance(bn, nn.BatchNorm2d):
        w, b = torch_utils.fold_conv_bn(conv.weight, conv.bias, bn)
    else:
        w = conv.weight.detach()
        b = None if conv.bias is None else conv.bias.detach()
    return w, b, conv.stride[0]


class Recorder:
    """Records conv(+bn)+activation holders into a Program."""

    def __init__(self, prog: Program, act: int):
        self.p = prog
        self.act = act

    def cba(self, name: str, m: nn.Module, x: View, out: Optional[View] = None, residual: Optional[View] = None,
            act: Optional[int] = None, conv_name: str = "conv", bn_name: str = "bn") -> View:
        w, b, s = folded(m, conv_name, bn_name)
        return self.p.conv(name, x, w, b, s, self.act if act is None else act, out=out, residual=residual)

    def head(self, name: str, conv: nn.Conv2d, x: View, out: Optional[View] = None) -> View:
        """Bias-only 1x1 conv producing float32 logits."""
        return self.p.conv(name, x, conv.weight.detach(), None if conv.bias is None else conv.bias.detach(), 1,
                           _lib.ACT_NONE, out=out, out_f32=True)


class TaskModel(nn.Module):
    """Plan cache + the parts of the reference's TaskLayer protocol that do not depend on the model."""

    def __init__(self):
        super().__init__()
        self._program: Optional[Program] = None
        self._program_key = None

    def invalidate(self) -> None:
        """Drop the compiled plan (call after changing weights in place)."""
        self._program = None

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def program(self) -> Program:
        key = (self.param.img_size,)
        if self._program is None or self._program_key != key:
            if self.param.img_size % max(self.param.strides):
                raise _lib.YamdError(f"img_size must be a multiple of {max(self.param.strides)}")
            self._program = self._compile(self.param.img_size)
            self._program_key = key
        return self._program

    def _compile(self, size: int) -> Program:      # pragma: no cover - abstract
        raise NotImplementedError

    def fuse_conv_bn_holders(self, conv_name: str = "conv", bn_name: str = "bn") -> None:
        """Fold every (conv, bn) holder pair into a conv with bias and delete the bn, which is what the
        reference's per-model ``fuse()`` leaves in the ``state_dict``."""
        for m in self.modules():
            if isinstance(getattr(m, bn_name, None), nn.BatchNorm2d) and isinstance(getattr(m, conv_name, None), nn.Conv2d):
                setattr(m, conv_name, torch_utils.fuse_conv_and_bn(getattr(m, conv_name), getattr(m, bn_name)))
                delattr(m, bn_name)
        self.invalidate()

    def to_onnx(self, onnx_path: str):
        raise NotImplementedError("ONNX export is outside the accelerated path; use the reference for export")
