"""Layer program for the conv graph: a recorded list of C-ABI calls over NHWC bf16 buffers.

A model description (e.g. ``yolo_amd.yolov5``) walks its parameter tree once and records ops here;
``Program.run`` then replays them for a batch.  Every activation is a *view* ``(buffer, channel
offset, channels)`` so that ``torch.cat`` of the reference becomes "producers write their slice of a
wider buffer" and costs nothing.  All arithmetic happens inside libyolo_amd.so; torch only owns the
device memory and the stream.
"""
import ctypes
import os
from typing import Dict, List, NamedTuple, Optional, Tuple

import torch

from . import lib as _lib

LIB = _lib.LIB


class View(NamedTuple):
    buf: str      # buffer name
    coff: int     # first channel inside the buffer
    c: int        # channels of the view
    h: int
    w: int

    def slice(self, coff: int, c: int) -> "View":
        assert 0 <= coff and coff + c <= self.c
        return View(self.buf, self.coff + coff, c, self.h, self.w)


# Which fused forms Program.conv may substitute while a model is recorded (all on; tests and layer-by-layer inspection
# switch one off before building a model).  None of them changes a result beyond the bf16 rounding both forms share.
FUSE = {"stem": True,          # first conv straight from the fp32 image (yamd_stem_conv)
        "stem_conv2": True,    # stride-2 stem + the 3x3 stride-2 conv after it as one launch (yamd_stem_conv2)
        "upsample_cat": True,  # 1x1 convs read torch.cat((up(a), b), 1) from a and b themselves (yamd_conv2d_bias_act_cat2)
        "bottleneck": True,    # 1x1 conv + 3x3 conv (+ shortcut) of a narrow residual unit as one launch (yamd_bottleneck)
        "c3": True}            # a whole narrow C3 block with one bottleneck as one launch (yamd_c3_fused)

_PACK_GENERATION = [0]      # bumped whenever a layer's packed weights or descriptor change: compiled C plans are stale then


class _CPlan:
    """Owner of a ``yamd_plan`` handle (the recorded program with every argument resolved, replayed by one C call)."""

    def __init__(self, ops: "ctypes.Array", batch: int, img_h: int, img_w: int, names: List[str], keep: tuple):
        self.handle = ctypes.c_void_p()
        _lib.check(LIB.yamd_plan_create(ops, len(ops), batch, img_h, img_w, ctypes.byref(self.handle)), "plan_create")
        self.names = names
        self.keep = keep              # the tensors whose addresses the plan holds
        self.generation = _PACK_GENERATION[0]
        self.failed = ctypes.c_int(-1)

    def run(self, images_ptr: int, stream: int) -> None:
        rc = LIB.yamd_plan_run(self.handle, images_ptr, stream, ctypes.byref(self.failed))
        if rc != 0:
            k = self.failed.value
            _lib.check(rc, self.names[k] if 0 <= k < len(self.names) else "plan_run")

    def __del__(self):
        if getattr(self, "handle", None):
            LIB.yamd_plan_destroy(self.handle)
            self.handle = None


class ConvLayer:
    """One fused conv: descriptor + packed weights (host copies kept so any device can be fed)."""

    def __init__(self, name: str, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, act: int,
                 cin_stored: int, out_f32: bool, tiling: Optional[Tuple[int, ...]] = None, residual_scale: float = 1.0):
        w = weight.detach().to(torch.float32).cpu().contiguous()
        cout, cin_real, k, k2 = w.shape
        assert k == k2
        self.name = name
        self.cin_real = cin_real
        self.macs_per_pixel = cout * cin_real * k * k
        self._w = w
        self._b = None if bias is None else bias.detach().to(torch.float32).cpu().contiguous()
        self._args = (k, stride, cin_stored, cout, act, int(out_f32))
        self.residual_scale = float(residual_scale)
        self.pack(tiling)

    def pack(self, tiling: Optional[Tuple[int, ...]] = None) -> None:
        """(Re)build the descriptor and the packed weights; ``tiling`` = (mrep, nrep, ck, nwaves[, stages]) overrides
        the default."""
        name = self.name
        self.desc = _lib.ConvDesc()
        _lib.check(LIB.yamd_conv_desc_init(ctypes.byref(self.desc), *self._args), f"conv_desc_init[{name}]")
        if tiling is not None:
            t = tuple(int(v) for v in tiling)
            t = t + (4,) * (4 - len(t)) if len(t) < 4 else t           # older tables: no nwaves / stages column
            t = t + (1,) if len(t) == 4 else t
            _lib.check(LIB.yamd_conv_desc_tune(ctypes.byref(self.desc), *t), f"conv_desc_tune[{name}]")
        self.desc.residual_scale = self.residual_scale
        wp = torch.empty(self.desc.packed_w_elems, dtype=torch.int16)
        bp = torch.empty(self.desc.cout_pad, dtype=torch.float32)
        _lib.check(LIB.yamd_conv_pack_weights(ctypes.byref(self.desc), self._w.data_ptr(), self.cin_real,
                                              None if self._b is None else self._b.data_ptr(), wp.data_ptr(),
                                              bp.data_ptr()), f"conv_pack_weights[{name}]")
        self.w_host, self.b_host = wp, bp
        self.w_dev: Optional[torch.Tensor] = None
        self.b_dev: Optional[torch.Tensor] = None
        _PACK_GENERATION[0] += 1

    @property
    def tiling(self) -> Tuple[int, int, int, int, int]:
        return (self.desc.mrep, self.desc.nrep, self.desc.ck, self.desc.nwaves, self.desc.stages)

    def to(self, device) -> None:
        if self.w_dev is None or self.w_dev.device != torch.device(device):
            self.w_dev = self.w_host.to(device)
            self.b_dev = self.b_host.to(device)


class StemLayer:
    """First conv of a network, computed straight from the fp32 NCHW image (``yamd_stem_conv``)."""

    MODES = {"k3s1": 0, "k3s2": 1, "focus": 2}

    def __init__(self, name: str, mode: str, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int):
        w = weight.detach().to(torch.float32).cpu().contiguous()
        cout, cin_real, k, _ = w.shape
        assert k == 3 and cin_real == (12 if mode == "focus" else 3)
        self.name, self.mode, self.act, self.cout, self.cin_real = name, self.MODES[mode], act, cout, cin_real
        self.macs_per_pixel = cout * cin_real * 9
        b = None if bias is None else bias.detach().to(torch.float32).cpu().contiguous()
        self._w, self._b = w, b
        n = LIB.yamd_stem_packed_elems(self.mode, cout)
        if n <= 0:
            raise _lib.YamdError(f"stem[{name}]: cout={cout} not supported")
        self.w_host = torch.empty(n, dtype=torch.int16)
        self.b_host = torch.empty(64, dtype=torch.float32)
        _lib.check(LIB.yamd_stem_pack_weights(self.mode, w.data_ptr(), None if b is None else b.data_ptr(), cout,
                                              self.w_host.data_ptr(), self.b_host.data_ptr()), f"stem_pack_weights[{name}]")
        self.w_dev: Optional[torch.Tensor] = None
        self.b_dev: Optional[torch.Tensor] = None

    def to(self, device) -> None:
        if self.w_dev is None or self.w_dev.device != torch.device(device):
            self.w_dev = self.w_host.to(device)
            self.b_dev = self.b_host.to(device)


class StemConv2Layer:
    """A stem (stride 2, or 3x3 stride 1 with 32 output channels) and the 3x3 stride-2 conv that follows it, run as one
    launch (``yamd_stem_conv2``): the stem's output is never written to memory."""

    def __init__(self, stem: StemLayer, name: str, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int):
        w = weight.detach().to(torch.float32).cpu().contiguous()
        cout, cin, k, _ = w.shape
        assert k == 3 and cin == stem.cout
        self.stem, self.name, self.act, self.cout, self.cin_real = stem, name, act, cout, cin
        self.macs_per_pixel = cout * cin * 9
        b = None if bias is None else bias.detach().to(torch.float32).cpu().contiguous()
        self._w, self._b = w, b
        n = LIB.yamd_stem_conv2_packed_elems(cin, cout)
        if n <= 0:
            raise _lib.YamdError(f"stem_conv2[{name}]: cin={cin} cout={cout} not supported")
        self.w_host = torch.empty(n, dtype=torch.int16)
        self.b_host = torch.empty(128, dtype=torch.float32)
        _lib.check(LIB.yamd_stem_conv2_pack_weights(w.data_ptr(), None if b is None else b.data_ptr(), cin, cout,
                                                    self.w_host.data_ptr(), self.b_host.data_ptr()),
                   f"stem_conv2_pack_weights[{name}]")
        self.w_dev: Optional[torch.Tensor] = None
        self.b_dev: Optional[torch.Tensor] = None

    @staticmethod
    def supported(stem: StemLayer, cout: int) -> bool:
        # (the library also takes a 3x3 stride-1 stem with 32 channels; measured slower than the two separate kernels at
        # 608 x 608 -- v3 conv stage 50.2 against 47.0 ms -- because its 4 x 16 output tiles are too small for a second conv
        # that runs at half the image resolution, so the plan does not select it)
        return stem.mode in (StemLayer.MODES["k3s2"], StemLayer.MODES["focus"]) and \
            LIB.yamd_stem_conv2_packed_elems(stem.cout, cout) > 0

    def to(self, device) -> None:
        self.stem.to(device)
        if self.w_dev is None or self.w_dev.device != torch.device(device):
            self.w_dev = self.w_host.to(device)
            self.b_dev = self.b_host.to(device)


class BottleneckLayer:
    """A 1x1 conv and the 3x3 conv that follows it (plus the unit's shortcut) run as one launch (``yamd_bottleneck``):
    the 1x1 conv's output is never written to memory."""

    # (cin = cout, cmid) the plan fuses.  The library also builds (128, 128); at that width the unit is bound by
    # arithmetic, not memory, and the two tuned launches are faster (207 against 243 us at 40 x 40, batch 256).
    SHAPES = ((32, 32), (64, 64), (64, 32), (128, 64))

    def __init__(self, first: ConvLayer, second: ConvLayer):
        cmid, cin = first._w.shape[:2]
        cout = second._w.shape[0]
        self.name = f"{first.name}+{second.name}"
        self.first, self.second = first, second
        self.cin, self.cmid, self.cout, self.act = cin, cmid, cout, first._args[4]
        self.macs_per_pixel = first.macs_per_pixel + second.macs_per_pixel
        n = LIB.yamd_bottleneck_packed_elems(cin, cmid, cout)
        if n <= 0:
            raise _lib.YamdError(f"bottleneck[{self.name}]: {cin} > {cmid} > {cout} channels not supported")
        self.w_host = torch.empty(n, dtype=torch.int16)
        self.b_host = torch.empty(cmid + cout, dtype=torch.float32)
        zeros = lambda k: torch.zeros(k, dtype=torch.float32)
        b1 = first._b if first._b is not None else zeros(cmid)
        b2 = second._b if second._b is not None else zeros(cout)
        _lib.check(LIB.yamd_bottleneck_pack_weights(cin, cmid, cout, first._w.data_ptr(), b1.data_ptr(), second._w.data_ptr(),
                                                    b2.data_ptr(), self.w_host.data_ptr(), self.b_host.data_ptr()),
                   f"bottleneck_pack_weights[{self.name}]")
        self.w_dev: Optional[torch.Tensor] = None
        self.b_dev: Optional[torch.Tensor] = None

    def to(self, device) -> None:
        if self.w_dev is None or self.w_dev.device != torch.device(device):
            self.w_dev = self.w_host.to(device)
            self.b_dev = self.b_host.to(device)


class C3Layer:
    """A whole C3 block with one bottleneck run as one launch (``yamd_c3_fused``): ``entry`` is the 1x1 conv that computes
    both branches (output channels [cv2 | cv1]), ``unit`` the fused bottleneck on the cv1 branch, ``exit`` the 1x1 conv
    over [bottleneck output | cv2 branch].  None of the tensors between them is written to memory."""

    def __init__(self, entry: ConvLayer, unit: BottleneckLayer, exit_: ConvLayer):
        ch = unit.cin
        cin, cout = entry._w.shape[1], exit_._w.shape[0]
        self.name = f"{entry.name}+{unit.name}+{exit_.name}"
        self.entry, self.unit, self.exit = entry, unit, exit_
        self.cin, self.ch, self.cout, self.act = cin, ch, cout, entry._args[4]
        self.macs_per_pixel = entry.macs_per_pixel + unit.macs_per_pixel + exit_.macs_per_pixel
        n = LIB.yamd_c3_packed_elems(cin, ch, cout)
        if n <= 0:
            raise _lib.YamdError(f"c3[{self.name}]: {cin} > {ch} > {cout} channels not supported")
        self.w_host = torch.empty(n, dtype=torch.int16)
        self.b_host = torch.empty(4 * ch + cout, dtype=torch.float32)
        zeros = lambda k: torch.zeros(k, dtype=torch.float32)
        be = entry._b if entry._b is not None else zeros(2 * ch)
        w_cv2, w_cv1 = entry._w[:ch].contiguous(), entry._w[ch:].contiguous()
        b_cv2, b_cv1 = be[:ch].contiguous(), be[ch:].contiguous()
        l1, l2 = unit.first, unit.second
        b_m1 = l1._b if l1._b is not None else zeros(ch)
        b_m2 = l2._b if l2._b is not None else zeros(ch)
        b3 = exit_._b if exit_._b is not None else zeros(cout)
        self._keep = (w_cv1, w_cv2, b_cv1, b_cv2, b_m1, b_m2, b3)
        _lib.check(LIB.yamd_c3_pack_weights(cin, ch, cout, w_cv1.data_ptr(), b_cv1.data_ptr(), w_cv2.data_ptr(), b_cv2.data_ptr(),
                                            l1._w.data_ptr(), b_m1.data_ptr(), l2._w.data_ptr(), b_m2.data_ptr(),
                                            exit_._w.data_ptr(), b3.data_ptr(), self.w_host.data_ptr(), self.b_host.data_ptr()),
                   f"c3_pack_weights[{self.name}]")
        self.w_dev: Optional[torch.Tensor] = None
        self.b_dev: Optional[torch.Tensor] = None

    def to(self, device) -> None:
        if self.w_dev is None or self.w_dev.device != torch.device(device):
            self.w_dev = self.w_host.to(device)
            self.b_dev = self.b_host.to(device)


class Program:
    def __init__(self, img_h: int, img_w: int):
        self.img_h, self.img_w = img_h, img_w
        self.buffers: Dict[str, Tuple[int, int, int, torch.dtype]] = {}
        self.ops: List[tuple] = []
        self.convs: List[ConvLayer] = []
        self.stems: List[StemLayer] = []
        self.outputs: Dict[str, View] = {}
        self.taps: Dict[str, View] = {}                 # named intermediate activations (verification aid, see read_view)
        self.alias: Dict[str, Tuple[str, int]] = {}     # buffer -> (wider buffer it lives in, first channel)
        self._alloc: Dict[Tuple[int, str], Dict[str, torch.Tensor]] = {}
        self._cplan: Dict[Tuple[int, str], _CPlan] = {}
        self.use_c_plan = True                   # False: run() issues one foreign call per op from Python (same calls, same order)
        self._n = 0
        self.span_sink: Optional[list] = None    # when set, every run() appends its conv spans here (see run)
        self.fused_away: set = set()             # buffers whose producer was merged into its consumer: never written
        self.unwritten: List[Tuple[str, int, int]] = []   # (buffer, first channel, end): slices no op writes any more
        self._finalized = False

    # ---- recording -------------------------------------------------------------------------------
    def buffer(self, h: int, w: int, c: int, dtype=torch.bfloat16, name: Optional[str] = None) -> View:
        self._n += 1
        name = name or f"t{self._n}"
        assert name not in self.buffers
        self.buffers[name] = (h, w, c, dtype)
        return View(name, 0, c, h, w)

    def focus(self) -> View:
        """fp32 NCHW images -> Focus space-to-depth, NHWC bf16, 12 channels padded to 16."""
        y = self.buffer(self.img_h // 2, self.img_w // 2, 16, name="focus")
        self.ops.append(("focus", y))
        return y

    def image16(self, cin: int = 3) -> View:
        y = self.buffer(self.img_h, self.img_w, 16, name="image16")
        self.ops.append(("nhwc16", y, cin))
        return y

    def conv(self, name: str, x: View, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, act: int,
             out: Optional[View] = None, residual: Optional[View] = None, out_f32: bool = False,
             out2: Optional[View] = None, residual_scale: float = 1.0) -> View:
        """``residual_scale``: the output is act(conv) + residual_scale * residual.
        ``out2``: the last ``out2.c`` output channels go to that view instead, the first ``cout - out2.c`` to ``out``
        (one conv computing two branches whose consumers want each branch densely packed)."""
        k = weight.shape[-1]
        cout = weight.shape[0]
        pad = k // 2
        ho = (x.h + 2 * pad - k) // stride + 1
        wo = (x.w + 2 * pad - k) // stride + 1
        if (x.buf in ("focus", "image16") and self.ops and self.ops[-1][0] in ("focus", "nhwc16") and self.ops[-1][1].buf == x.buf
                and k == 3 and cout <= 64 and cout % 8 == 0 and residual is None and out2 is None and not out_f32
                and (x.buf == "image16" or stride == 1) and FUSE["stem"]):
            # the network's first conv reads the image itself: no packed copy of the image is written
            self.ops.pop()
            del self.buffers[x.buf]
            mode = "focus" if x.buf == "focus" else ("k3s1" if stride == 1 else "k3s2")
            layer = StemLayer(name, mode, weight, bias, act)
            if out is None:
                out = self.buffer(ho, wo, cout)
            assert (out.h, out.w, out.c) == (ho, wo, cout), (name, out, ho, wo, cout)
            self.stems.append(layer)
            self.ops.append(("stem", layer, out))
            return out
        if (self.ops and self.ops[-1][0] == "stem" and self.ops[-1][2] == x and k == 3 and stride == 2 and residual is None
                and out2 is None and not out_f32 and StemConv2Layer.supported(self.ops[-1][1], cout)
                and self.img_h % 2 == 0 and self.img_w % 2 == 0 and FUSE["stem_conv2"]):
            # the stem's only reader follows it directly: both run as one launch and the stem's output, the largest
            # activation of the network, is never written (a later reader of that buffer would find it unwritten, so the
            # buffer is dropped: recording such a reader fails in resolve())
            _, stem_layer, stem_out = self.ops.pop()
            del self.buffers[stem_out.buf]
            self.fused_away.add(stem_out.buf)
            layer = StemConv2Layer(stem_layer, name, weight, bias, act)
            self.stems.remove(stem_layer)       # uploaded through the fused layer
            if out is None:
                out = self.buffer(ho, wo, cout)
            assert (out.h, out.w, out.c) == (ho, wo, cout), (name, out, ho, wo, cout)
            self.stems.append(layer)
            self.ops.append(("stem2", layer, out))
            return out
        from . import tuning
        layer = ConvLayer(name, weight, bias, stride, act, x.c, out_f32,
                          None if out_f32 else tuning.lookup(k, stride, x.c, cout, ho, wo), residual_scale)
        if out2 is not None:
            assert out is not None and residual is None and not out_f32
            assert (out.h, out.w, out2.h, out2.w) == (ho, wo, ho, wo) and out.c + out2.c == cout and out.c % 8 == 0
        else:
            if out is None:
                out = self.buffer(ho, wo, cout, torch.float32 if out_f32 else torch.bfloat16)
            assert (out.h, out.w, out.c) == (ho, wo, cout), (name, out, ho, wo, cout)
        if residual is not None:
            assert (residual.h, residual.w, residual.c) == (ho, wo, cout)
        self.convs.append(layer)
        self.ops.append(("conv", layer, x, out, residual, out2))
        return out

    def upsample2x(self, x: View, out: Optional[View] = None) -> View:
        if out is None:
            out = self.buffer(2 * x.h, 2 * x.w, x.c)
        assert (out.h, out.w, out.c) == (2 * x.h, 2 * x.w, x.c)
        self.ops.append(("up2", x, out))
        return out

    def maxpool5_cascade(self, x: View, y1: View, y2: View, y3: View, kernel: int = 5) -> None:
        """y1 = mp(x), y2 = mp(y1), y3 = mp(y2) with kernel x kernel stride-1 pools (5: SPP k 5/9/13; 3: k 3/5/7)."""
        assert kernel in (3, 5)
        for y in (y1, y2, y3):
            assert (y.h, y.w, y.c) == (x.h, x.w, x.c) and y.buf == y1.buf
        self.ops.append(("mp5", x, y1, y2, y3, kernel))

    def copy(self, x: View, out: View) -> View:
        assert (out.h, out.w, out.c) == (x.h, x.w, x.c)
        self.ops.append(("copy", x, out))
        return out

    def maxpool2(self, x: View, out: Optional[View] = None) -> View:
        """MaxPool2d(kernel 2, stride 2)."""
        assert x.h % 2 == 0 and x.w % 2 == 0
        if out is None:
            out = self.buffer(x.h // 2, x.w // 2, x.c)
        assert (out.h, out.w, out.c) == (x.h // 2, x.w // 2, x.c)
        self.ops.append(("mp2", x, out))
        return out

    def conv_transpose2x(self, name: str, x: View, weight: torch.Tensor, bias: Optional[torch.Tensor],
                         out: Optional[View] = None) -> View:
        """nn.ConvTranspose2d(kernel 2, stride 2): weight (cin, cout, 2, 2).  Runs as a 1x1 conv to 4*cout channels
        ordered (dy, dx, cout) followed by a depth-to-space scatter."""
        cin, cout, kh, kw = weight.shape
        assert (kh, kw) == (2, 2) and cin == x.c
        w1 = weight.detach().permute(2, 3, 1, 0).reshape(4 * cout, cin, 1, 1)
        b1 = None if bias is None else bias.detach().repeat(4)
        t = self.conv(name, x, w1, b1, 1, _lib.ACT_NONE)
        if out is None:
            out = self.buffer(2 * x.h, 2 * x.w, cout)
        assert (out.h, out.w, out.c) == (2 * x.h, 2 * x.w, cout)
        self.ops.append(("d2s", t, out))
        return out

    def cat(self, views: List[View], name: Optional[str] = None) -> View:
        """torch.cat(views, dim=1) without moving data where possible: a source that is a whole buffer of its
        own is re-homed into the concat buffer (its producers then write their channel slice in place, also
        the ones recorded earlier); a source that already lives elsewhere (a slice, or a tensor that is part
        of another concat) is copied."""
        assert views and all((v.h, v.w) == (views[0].h, views[0].w) for v in views)
        merged: List[View] = []
        for v in views:                                   # adjacent slices of one buffer count as one source
            if merged and merged[-1].buf == v.buf and merged[-1].coff + merged[-1].c == v.coff:
                merged[-1] = View(v.buf, merged[-1].coff, merged[-1].c + v.c, v.h, v.w)
            else:
                merged.append(v)
        total = sum(v.c for v in merged)
        out = self.buffer(views[0].h, views[0].w, total, name=name)
        off = 0
        for v in merged:
            h, w, c, dt = self.buffers[v.buf]
            whole = v.coff == 0 and v.c == c and v.buf not in self.alias and dt == torch.bfloat16
            if whole and v.buf not in {o.buf for o in self.outputs.values()}:
                self.alias[v.buf] = (out.buf, off)
            else:
                self.copy(v, out.slice(off, v.c))
            off += v.c
        return out

    def resolve(self, v: View) -> Tuple[str, int]:
        """(root buffer, first channel inside it) of a view after following the re-homing chain."""
        buf, coff = v.buf, v.coff
        while buf in self.alias:
            buf, o = self.alias[buf]
            coff += o
        return buf, coff

    # ---- per-layer tiling search ------------------------------------------------------------------
    def autotune(self, batch: int, device, reps: int = 5, log=None, skip=()) -> Dict[str, Tuple[int, ...]]:
        """Times every built tiling of every bf16 conv on this program's own buffers and keeps the fastest.
        Returns {tuning key: (mrep, nrep, ck)} for ``tuning.save``; layer shapes whose key is in ``skip`` keep their tiling."""
        from . import tuning
        bufs = self._buffers_for(batch, device)
        st = _lib.current_stream()
        best: Dict[str, Tuple[int, ...]] = {}
        for op in self.ops:
            if op[0] != "conv" or op[1].desc.out_f32:
                continue
            _, layer, x, out, res, out2 = op
            k, s = layer._args[0], layer._args[1]
            cout = out.c + (out2.c if out2 is not None else 0)
            key = tuning.key(k, s, x.c, cout, out.h, out.w)
            if key in skip:
                continue
            if key in best:
                layer.pack(best[key])
                layer.to(device)
                continue
            xp, xcs = self._ptr(bufs, x)
            yp, ycs = self._ptr(bufs, out)
            rp, rcs = self._ptr(bufs, res) if res is not None else (None, 0)
            results = []
            for cand in tuning.candidates(k, s, x.c, cout):
                try:
                    layer.pack(cand)
                except _lib.YamdError:
                    continue
                layer.to(device)

                def run():
                    if out2 is not None:
                        y2p, y2cs = self._ptr(bufs, out2)
                        _lib.check(LIB.yamd_conv2d_bias_act_split(ctypes.byref(layer.desc), xp, xcs, layer.w_dev.data_ptr(),
                                                                  layer.b_dev.data_ptr(), yp, ycs, y2p, y2cs, out.c, batch,
                                                                  x.h, x.w, st))
                    else:
                        _lib.check(LIB.yamd_conv2d_bias_act(ctypes.byref(layer.desc), xp, xcs, layer.w_dev.data_ptr(),
                                                            layer.b_dev.data_ptr(), rp, rcs, yp, ycs, batch, x.h, x.w, st))
                try:
                    run()
                except _lib.YamdError:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                results.append((e0.elapsed_time(e1) / reps, cand))
            results.sort()
            best[key] = results[0][1]
            layer.pack(best[key])
            layer.to(device)
            if log is not None:
                log(f"{layer.name:34s} {key:28s} " + "  ".join(f"{c}={t:.4f}" for t, c in results[:5]))
        return best

    def autotune_in_situ(self, images: torch.Tensor, reps: int = 7, log=None, only: str = "") -> Dict[str, Tuple[int, ...]]:
        """Like ``autotune``, but every candidate tiling is timed inside a run of the whole program (events around the
        layers of that shape), so its inputs come from where the previous layer left them (HBM or L2) and not from a
        cache warmed by running the same layer back to back.  Starts from the tilings the layers currently have."""
        import statistics
        from . import tuning
        groups: Dict[str, list] = {}
        for op in self.ops:
            if op[0] != "conv" or op[1].desc.out_f32:
                continue
            _, layer, x, out, res, out2 = op
            cout = out.c + (out2.c if out2 is not None else 0)
            k, s = layer._args[0], layer._args[1]
            groups.setdefault(tuning.key(k, s, x.c, cout, out.h, out.w), []).append((layer, k, s, x.c, cout))
        device = images.device

        def measure(names) -> float:
            ts = []
            for _ in range(reps):
                t: list = []
                self.run(images, timings=t)
                torch.cuda.synchronize()
                ts.append(sum(e0.elapsed_time(e1) for kind, name, e0, e1, _f, _b in t if kind == "conv" and name in names))
            return statistics.median(ts)

        best: Dict[str, Tuple[int, ...]] = {}
        self.run(images)
        for key, members in groups.items():
            if only and not key.startswith(only):
                continue
            names = {m[0].name for m in members}
            _, k, s, cin, cout = members[0]
            start = members[0][0].tiling
            results = []
            for cand in tuning.candidates(k, s, cin, cout):
                try:
                    for m in members:
                        m[0].pack(cand)
                        m[0].to(device)
                    self.run(images)
                except _lib.YamdError:
                    continue
                results.append((measure(names), cand))
            results.sort()
            best[key] = results[0][1] if results else start
            for m in members:
                m[0].pack(best[key])
                m[0].to(device)
            if log is not None:
                log(f"{key:28s} x{len(members)} " + "  ".join(f"{c}={t:.4f}" for t, c in results[:5])
                    + "".join(f"  [{c}={t:.4f}]" for t, c in results[5:] if c[0] == 4))
        return best

    # ---- accounting ------------------------------------------------------------------------------
    def read_view(self, v: View, batch: int, device) -> torch.Tensor:
        """The current contents of an activation view as a float32 NCHW tensor (after a ``run`` with this batch)."""
        bufs = self._buffers_for(batch, device)
        buf, coff = self.resolve(v)
        if buf in self.fused_away:
            raise _lib.YamdError(f"activation '{buf}' is not materialised: its layer runs fused with the next one (set "
                                 "yolo_amd.engine.FUSE['stem_conv2'] / ['bottleneck'] / ['c3'] = False before building the model to inspect it)")
        if any(b == buf and c0 < coff + v.c and c1 > coff for b, c0, c1 in self.unwritten):
            raise _lib.YamdError(f"activation '{buf}'[{coff}:{coff + v.c}] is not materialised: its readers take the upsampled "
                                 "tensor from the low-resolution one (set yolo_amd.engine.FUSE['upsample_cat'] = False before "
                                 "building the model to inspect it)")
        return bufs[buf][..., coff:coff + v.c].permute(0, 3, 1, 2).float().contiguous()

    def n_conv_launches(self) -> int:
        return sum(1 for op in self.ops if op[0] in ("conv", "conv_cat2", "stem", "stem2", "bneck", "c3"))

    def conv_launch_signature(self) -> List[str]:
        """One string per conv launch of a pass, in launch order: kind, layer name, shape and tiling.  Two plans with the
        same list launch the same kernels on the same shapes (what a stored per-launch measurement has to match)."""
        sig = []
        for op in self.ops:
            kind = op[0]
            if kind in ("conv", "conv_cat2"):
                d = op[1].desc
                o = op[3] if kind == "conv" else op[6]
                sig.append(f"{kind} {op[1].name} k{d.ksize}s{d.stride} {d.cin}>{d.cout} @{o.h}x{o.w} tiling={op[1].tiling}")
            elif kind == "bneck":
                sig.append(f"bneck {op[1].name} {op[1].cin}>{op[1].cmid}>{op[1].cout} @{op[3].h}x{op[3].w}")
            elif kind == "c3":
                sig.append(f"c3 {op[1].name} {op[1].cin}>{op[1].ch}>{op[1].cout} @{op[3].h}x{op[3].w}")
            elif kind in ("stem", "stem2"):
                sig.append(f"{kind} {op[1].name} @{op[2].h}x{op[2].w}")
        return sig

    def conv_macs_per_image(self) -> int:
        total = 0
        for op in self.ops:
            if op[0] == "conv":
                total += op[1].macs_per_pixel * op[3].h * op[3].w
            elif op[0] == "conv_cat2":
                total += op[1].macs_per_pixel * op[6].h * op[6].w
            elif op[0] in ("bneck", "c3"):      # the MACs of the layers, not of the fused kernel (which recomputes halos)
                total += op[1].macs_per_pixel * op[3].h * op[3].w
            elif op[0] == "stem":
                total += op[1].macs_per_pixel * op[2].h * op[2].w
            elif op[0] == "stem2":      # the MACs of the two layers, not of the fused kernel (which recomputes tile overlaps)
                sdiv = 1 if op[1].stem.mode == StemLayer.MODES["k3s1"] else 2
                total += op[1].macs_per_pixel * op[2].h * op[2].w + op[1].stem.macs_per_pixel * (self.img_h // sdiv) * (self.img_w // sdiv)
        return total

    def activation_bytes_per_image(self) -> int:
        return sum(h * w * c * (2 if dt == torch.bfloat16 else 4) for n, (h, w, c, dt) in self.buffers.items()
                   if n not in self.alias)

    def op_cost(self, op, batch: int) -> Tuple[float, float]:
        """(algorithmic flops, algorithmic bytes) of one recorded op for a batch: every input and output
        element counted once (bf16 = 2 bytes), weights once."""
        kind = op[0]
        if kind == "conv":
            _, layer, x, out, res, out2 = op
            px = batch * out.h * out.w
            flops = 2.0 * layer.macs_per_pixel * px
            ob = 4 if layer.desc.out_f32 else 2
            cout = out.c + (out2.c if out2 is not None else 0)
            byts = batch * x.h * x.w * layer.cin_real * 2 + px * cout * ob + layer.macs_per_pixel * 2
            if res is not None:
                byts += px * out.c * 2
            return flops, float(byts)
        if kind == "conv_cat2":
            _, layer, xa, _ah, xb, _bh, out, out2 = op
            px = batch * out.h * out.w
            cout = out.c + (out2.c if out2 is not None else 0)
            byts = batch * (xa.h * xa.w * xa.c + xb.h * xb.w * xb.c) * 2 + px * cout * (4 if layer.desc.out_f32 else 2) + layer.macs_per_pixel * 2
            return 2.0 * layer.macs_per_pixel * px, float(byts)
        if kind in ("bneck", "c3"):
            _, layer, x, out, _sc = op
            px = batch * out.h * out.w
            return 2.0 * layer.macs_per_pixel * px, float(px * (x.c + out.c) * 2 + layer.macs_per_pixel * 2)
        if kind == "stem":
            _, layer, out = op
            px = batch * out.h * out.w
            return 2.0 * layer.macs_per_pixel * px, float(batch * self.img_h * self.img_w * 3 * 4 + px * out.c * 2)
        if kind == "stem2":
            _, layer, out = op
            px = batch * out.h * out.w
            sdiv = 1 if layer.stem.mode == StemLayer.MODES["k3s1"] else 2
            flops = 2.0 * layer.macs_per_pixel * px + 2.0 * layer.stem.macs_per_pixel * batch * (self.img_h // sdiv) * (self.img_w // sdiv)
            return flops, float(batch * self.img_h * self.img_w * 3 * 4 + px * out.c * 2 + layer.macs_per_pixel * 2)
        if kind == "focus":
            return 0.0, float(batch * self.img_h * self.img_w * 3 * 4 + batch * op[1].h * op[1].w * 16 * 2)
        if kind == "nhwc16":
            return 0.0, float(batch * self.img_h * self.img_w * (op[2] * 4 + 32))
        if kind == "up2":
            return 0.0, float(batch * op[1].h * op[1].w * op[1].c * 2 * 5)
        if kind == "mp5":
            return 0.0, float(batch * op[1].h * op[1].w * op[1].c * 2 * 4)
        if kind == "copy":
            return 0.0, float(batch * op[1].h * op[1].w * op[1].c * 2 * 2)
        if kind == "mp2":
            return 0.0, float(batch * op[1].h * op[1].w * op[1].c * 2 * 1.25)
        if kind == "d2s":
            return 0.0, float(batch * op[1].h * op[1].w * op[1].c * 2 * 2)
        return 0.0, 0.0

    # ---- execution -------------------------------------------------------------------------------
    @staticmethod
    def _inputs_of(op) -> List[View]:
        kind = op[0]
        if kind == "conv":
            return [v for v in (op[2], op[4]) if v is not None]
        if kind == "conv_cat2":
            return [op[2], op[4]]
        if kind in ("bneck", "c3"):
            return [op[2]]
        if kind in ("up2", "copy", "d2s", "mp2", "mp5"):
            return [op[1]]
        return []

    def fuse_upsample_cat(self) -> int:
        """Removes every 2x upsampling whose output is one end of a concatenation that only 1x1 convs read: those convs
        then read the low-resolution tensor and the other part themselves (``conv_cat2`` ops), and neither the
        upsampled tensor nor its slice of the concatenation is ever written.  Returns the number of upsamplings
        removed.  Runs once, before the first buffers are allocated."""
        def rng(v: View):
            buf, coff = self.resolve(v)
            return buf, coff, coff + v.c

        removed = 0
        for up in [op for op in self.ops if op[0] == "up2"]:
            _, low, out = up
            obuf, o0, o1 = rng(out)
            if any(rng(v)[0] == obuf and rng(v)[1] < o1 and rng(v)[2] > o0 for v in self.outputs.values()):
                continue
            readers, ok = [], True
            for j, op in enumerate(self.ops):
                if op is up:
                    continue
                for v in self._inputs_of(op):
                    b, c0, c1 = rng(v)
                    if b != obuf or c0 >= o1 or c1 <= o0:
                        continue
                    layer = op[1] if op[0] == "conv" else None
                    # a 1x1 conv without residual whose input is exactly [upsampled part | other part] or the reverse
                    fits = (op[0] == "conv" and v is op[2] and op[4] is None and layer._args[0] == 1 and c0 <= o0 and o1 <= c1
                            and (c0 == o0) != (c1 == o1) and (o1 - o0) % layer.desc.ck == 0
                            and (c1 - c0 - (o1 - o0)) % layer.desc.ck == 0)
                    if not fits:
                        ok = False
                    else:
                        readers.append((j, c0, c1))
            if not ok or not readers:
                continue
            for j, c0, c1 in readers:
                _, layer, x, y, _res, y2 = self.ops[j]
                first = c0 == o0                     # the upsampled part comes first in the concatenation
                other = View(x.buf, x.coff + ((o1 - o0) if first else 0), (c1 - c0) - (o1 - o0), x.h, x.w)
                xa, xb = (low, other) if first else (other, low)
                self.ops[j] = ("conv_cat2", layer, xa, int(first), xb, int(not first), y, y2)
            self.ops.remove(up)
            self.unwritten.append((obuf, o0, o1))
            removed += 1
        return removed

    def fuse_bottlenecks(self) -> int:
        """Replaces every 1x1 conv directly followed by the 3x3 stride-1 conv that is its only reader (same activation,
        shortcut from the 1x1 conv's input or none) by one ``bneck`` op when the library builds the fused kernel for
        those widths; the tensor between them is dropped.  Returns the number of pairs fused.  Runs once, before the
        first buffers are allocated."""
        def rng(v: View):
            buf, coff = self.resolve(v)
            return buf, coff, coff + v.c

        def overlaps(a, b):
            return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]

        fused, j = 0, 0
        while j + 1 < len(self.ops):
            a, b = self.ops[j], self.ops[j + 1]
            j += 1
            if a[0] != "conv" or b[0] != "conv":
                continue
            _, l1, x, t, res1, t2 = a
            _, l2, t_in, y, res2, y2 = b
            k1, s1, cin1, cmid, act1, f1 = l1._args
            k2, s2, _cin2, cout, act2, f2 = l2._args
            if not (k1 == 1 and k2 == 3 and s1 == 1 and s2 == 1 and act1 == act2 and not f1 and not f2 and res1 is None
                    and t2 is None and y2 is None and rng(t_in) == rng(t) and l1.cin_real == x.c and l2.cin_real == t.c
                    and x.c == cout and (x.c, cmid) in BottleneckLayer.SHAPES and l2.residual_scale == 1.0
                    and (res2 is None or rng(res2) == rng(x))):
                continue
            tr = rng(t)
            if t.buf in self.alias or self.buffers[t.buf][2] != t.c or any(overlaps(rng(v), tr) for v in self.outputs.values()):
                continue
            if any(overlaps(rng(v), tr) for k, op in enumerate(self.ops) if k != j for v in self._inputs_of(op)):
                continue      # (k == j is the 3x3 conv itself; ops[j - 1] has no input in t)
            if overlaps(rng(x), rng(y)):
                continue
            layer = BottleneckLayer(l1, l2)
            self.ops[j - 1:j + 1] = [("bneck", layer, x, y, int(res2 is not None))]
            self.convs.remove(l1)
            self.convs.remove(l2)
            self.stems.append(layer)
            del self.buffers[t.buf]
            self.fused_away.add(t.buf)
            fused += 1
        return fused

    def fuse_c3(self) -> int:
        """Replaces [1x1 conv to two branches, fused bottleneck on one of them, 1x1 conv over (bottleneck output | other
        branch)] -- a C3 block with one bottleneck after ``fuse_bottlenecks`` -- by one ``c3`` op when the library builds
        the fused kernel for those widths and nothing else reads the tensors in between.  Returns the number of blocks
        fused.  Runs once, before the first buffers are allocated."""
        def rng(v: View):
            buf, coff = self.resolve(v)
            return buf, coff, coff + v.c

        def overlaps(a, b):
            return a[0] == b[0] and a[1] < b[2] and b[1] < a[2]

        fused, j = 0, 0
        while j + 2 < len(self.ops):
            e, u, x3 = self.ops[j], self.ops[j + 1], self.ops[j + 2]
            j += 1
            if e[0] != "conv" or u[0] != "bneck" or x3[0] != "conv":
                continue
            _, le, x, b_out, res_e, a_out = e
            _, lu, u_in, m_out, shortcut = u
            _, lx, cat_in, y, res_x, y2 = x3
            if a_out is None or res_e is not None or res_x is not None or y2 is not None:
                continue
            ke, se, _cin, _cout, act_e, f_e = le._args
            kx, sx, _c2, _c3, act_x, f_x = lx._args
            ch = lu.cin
            if not (ke == 1 and kx == 1 and se == 1 and sx == 1 and not f_e and not f_x and act_e == act_x == lu.act == _lib.ACT_SILU
                    and lu.cmid == ch and lu.cout == ch and b_out.c == ch and a_out.c == ch and le.cin_real == x.c
                    and lx.cin_real == 2 * ch and LIB.yamd_c3_packed_elems(x.c, ch, y.c) > 0
                    and le.residual_scale == 1.0 and lx.residual_scale == 1.0):
                continue
            # the bottleneck reads the cv1 branch, and the exit conv reads exactly [bottleneck output | cv2 branch]
            ra, rb, rm, rcat = rng(a_out), rng(b_out), rng(m_out), rng(cat_in)
            if rng(u_in) != ra or rm[0] != rb[0] or rm[2] != rb[1] or rcat != (rm[0], rm[1], rb[2]):
                continue
            inner = (ra, rb, rm)
            if any(overlaps(rng(v), r) for v in self.outputs.values() for r in inner):
                continue
            if any(overlaps(rng(v), r) for k, op in enumerate(self.ops) if k not in (j, j + 1) for v in self._inputs_of(op) for r in inner):
                continue      # (k == j: the bottleneck, k == j + 1: the exit conv)
            if overlaps(rng(x), rng(y)) or a_out.buf in self.alias or m_out.buf in self.alias:
                continue
            if self.buffers[a_out.buf][2] != ch or self.buffers[m_out.buf][2] != 2 * ch:
                continue
            layer = C3Layer(le, lu, lx)
            self.ops[j - 1:j + 2] = [("c3", layer, x, y, shortcut)]
            self.convs.remove(le)
            self.convs.remove(lx)
            self.stems.remove(lu)
            self.stems.append(layer)
            for buf in (a_out.buf, m_out.buf):
                del self.buffers[buf]
                self.fused_away.add(buf)
            fused += 1
        return fused

    def _buffers_for(self, batch: int, device) -> Dict[str, torch.Tensor]:
        if not self._finalized:
            self._finalized = True
            if FUSE["upsample_cat"]:
                self.fuse_upsample_cat()
            if FUSE["bottleneck"]:
                self.fuse_bottlenecks()
                if FUSE["c3"]:
                    self.fuse_c3()
        key = (batch, str(device))
        if key not in self._alloc:
            self._alloc.clear()   # one live batch size at a time: the arena is large
            self._cplan.clear()
            self._alloc[key] = {n: torch.empty((batch, h, w, c), dtype=dt, device=device)
                                for n, (h, w, c, dt) in self.buffers.items() if n not in self.alias}
            for layer in self.convs + self.stems:
                layer.to(device)
        return self._alloc[key]

    def _ptr(self, bufs: Dict[str, torch.Tensor], v: View) -> Tuple[int, int]:
        buf, coff = self.resolve(v)
        t = bufs[buf]
        return t.data_ptr() + coff * t.element_size(), t.shape[-1]

    def c_plan(self, batch: int, device) -> _CPlan:
        """The recorded ops as a ``yamd_plan`` for this batch size and device (built once; rebuilt when a layer was
        re-packed or the buffers were re-allocated)."""
        key = (batch, str(device))
        plan = self._cplan.get(key)
        if plan is not None and plan.generation == _PACK_GENERATION[0]:
            return plan
        bufs = self._buffers_for(batch, device)
        for layer in self.convs + self.stems:
            layer.to(device)
        P = self._ptr
        ops = (_lib.PlanOp * len(self.ops))()
        names: List[str] = []
        for o, op in zip(ops, self.ops):
            kind = op[0]

            def fill(k, ints, ptrs):
                o.kind = k
                for j, v in enumerate(ints):
                    o.i[j] = int(v)
                for j, v in ptrs.items():
                    o.p[j] = v

            if kind == "conv":
                _, layer, x, out, res, out2 = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                o.desc = layer.desc
                if out2 is not None:
                    y2p, y2cs = P(bufs, out2)
                    fill(_lib.OP_CONV_SPLIT, (xcs, 0, ycs, x.h, x.w, y2cs, out.c),
                         {0: xp, 1: layer.w_dev.data_ptr(), 2: layer.b_dev.data_ptr(), 4: yp, 5: y2p})
                else:
                    rp, rcs = P(bufs, res) if res is not None else (None, 0)
                    fill(_lib.OP_CONV, (xcs, rcs, ycs, x.h, x.w),
                         {0: xp, 1: layer.w_dev.data_ptr(), 2: layer.b_dev.data_ptr(), 3: rp, 4: yp})
                names.append(f"conv[{layer.name}]")
            elif kind == "conv_cat2":
                _, layer, xa, a_half, xb, b_half, out, out2 = op
                xap, xacs = P(bufs, xa)
                xbp, xbcs = P(bufs, xb)
                yp, ycs = P(bufs, out)
                y2p, y2cs = P(bufs, out2) if out2 is not None else (None, 0)
                o.desc = layer.desc
                fill(_lib.OP_CONV_CAT2, (xacs, xbcs, ycs, out.h, out.w, y2cs, out.c if out2 is not None else 0, xa.c, a_half | (b_half << 1)),
                     {0: xap, 1: layer.w_dev.data_ptr(), 2: layer.b_dev.data_ptr(), 3: xbp, 4: yp, 5: y2p})
                names.append(f"conv[{layer.name}]")
            elif kind == "bneck":
                _, layer, x, out, shortcut = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                fill(_lib.OP_BOTTLENECK, (xcs, layer.cin, ycs, out.h, out.w, layer.cmid, layer.cout, layer.act, shortcut),
                     {0: xp, 1: layer.w_dev.data_ptr(), 2: layer.b_dev.data_ptr(), 4: yp})
                names.append(f"bottleneck[{layer.name}]")
            elif kind == "c3":
                _, layer, x, out, shortcut = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                fill(_lib.OP_C3, (xcs, layer.cin, ycs, out.h, out.w, layer.ch, layer.cout, layer.act, shortcut),
                     {0: xp, 1: layer.w_dev.data_ptr(), 2: layer.b_dev.data_ptr(), 4: yp})
                names.append(f"c3[{layer.name}]")
            elif kind == "stem":
                _, layer, out = op
                yp, ycs = P(bufs, out)
                fill(_lib.OP_STEM, (layer.mode, layer.cout, layer.act, ycs),
                     {1: layer.w_dev.data_ptr(), 2: layer.b_dev.data_ptr(), 4: yp})
                names.append(f"stem[{layer.name}]")
            elif kind == "stem2":
                _, layer, out = op
                yp, ycs = P(bufs, out)
                st1 = layer.stem
                fill(_lib.OP_STEM_CONV2, (st1.mode, st1.cout, st1.act, ycs, layer.cout, layer.act),
                     {1: st1.w_dev.data_ptr(), 2: st1.b_dev.data_ptr(), 4: yp, 5: layer.w_dev.data_ptr(), 6: layer.b_dev.data_ptr()})
                names.append(f"stem_conv2[{layer.name}]")
            elif kind == "focus":
                yp, ycs = P(bufs, op[1])
                fill(_lib.OP_FOCUS, (ycs,), {4: yp})
                names.append("focus")
            elif kind == "nhwc16":
                yp, ycs = P(bufs, op[1])
                fill(_lib.OP_NHWC16, (ycs, op[2]), {4: yp})
                names.append("nhwc16")
            elif kind == "mp5":
                _, x, y1, y2, y3, kernel = op
                xp, xcs = P(bufs, x)
                p1, ycs = P(bufs, y1)
                fill(_lib.OP_MAXPOOL5_CASCADE if kernel == 5 else _lib.OP_MAXPOOL3_CASCADE, (xcs, ycs, x.h, x.w, x.c),
                     {0: xp, 4: p1, 5: P(bufs, y2)[0], 6: P(bufs, y3)[0]})
                names.append("maxpool_cascade")
            elif kind in ("up2", "copy", "d2s", "mp2"):
                _, x, out = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                code = {"up2": _lib.OP_UPSAMPLE2X, "copy": _lib.OP_COPY_CHANNELS, "d2s": _lib.OP_DEPTH_TO_SPACE2,
                        "mp2": _lib.OP_MAXPOOL2}[kind]
                fill(code, (xcs, ycs, x.h, x.w, out.c if kind == "d2s" else x.c), {0: xp, 4: yp})
                names.append(kind)
            else:
                raise AssertionError(kind)
        keep = tuple(bufs.values()) + tuple(t for layer in self.convs + self.stems for t in (layer.w_dev, layer.b_dev))
        plan = _CPlan(ops, batch, self.img_h, self.img_w, names, keep)
        self._cplan[key] = plan
        return plan

    def run(self, images: torch.Tensor, timings: Optional[list] = None, conv_spans: Optional[list] = None
            ) -> Dict[str, torch.Tensor]:
        """images: (B,3,H,W) float32 NCHW on the GPU.  Returns {output name: tensor view}; raw head
        outputs are float32 NHWC ``(B,h,w,c)``.  Work is enqueued on torch's current stream.
        If ``timings`` is a list, every op is bracketed by events on that stream and
        ``(kind, name, start_event, end_event, flops, bytes)`` tuples are appended (profiling aid).
        If ``conv_spans`` is a list, every maximal run of consecutive conv launches is bracketed by one event pair on
        that stream and ``(start_event, end_event)`` is appended: the conv kernels' time inside a running pipeline
        at the cost of a dozen event records per step."""
        _lib.require_gpu(images, torch.float32, "images")
        if images.dim() != 4 or images.shape[1] != 3:
            raise _lib.YamdError(f"images must be (batch, 3, H, W), got {tuple(images.shape)}")
        B, cin, H, W = images.shape
        if (H, W) != (self.img_h, self.img_w):
            raise _lib.YamdError(f"program was built for {self.img_h}x{self.img_w} images, got {H}x{W}")
        bufs = self._buffers_for(B, images.device)
        st = _lib.current_stream()
        P = self._ptr
        span_open = None
        if conv_spans is None:
            conv_spans = self.span_sink
        if timings is None and conv_spans is None and self.use_c_plan:
            # the usual case: the whole list replayed by one C call (arguments were resolved when the plan was made)
            self.c_plan(B, images.device).run(images.data_ptr(), st)
            return self._outputs(bufs)
        for op in self.ops:
            kind = op[0]
            if conv_spans is not None:
                is_conv = kind in ("conv", "conv_cat2", "stem", "stem2", "bneck", "c3")
                if is_conv and span_open is None:
                    span_open = torch.cuda.Event(enable_timing=True)
                    span_open.record()
                elif not is_conv and span_open is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    conv_spans.append((span_open, ev))
                    span_open = None
            if timings is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            if kind == "conv":
                _, layer, x, out, res, out2 = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                if out2 is not None:
                    y2p, y2cs = P(bufs, out2)
                    _lib.check(LIB.yamd_conv2d_bias_act_split(ctypes.byref(layer.desc), xp, xcs, layer.w_dev.data_ptr(),
                                                              layer.b_dev.data_ptr(), yp, ycs, y2p, y2cs, out.c, B, x.h, x.w, st),
                               f"conv[{layer.name}]")
                else:
                    rp, rcs = P(bufs, res) if res is not None else (None, 0)
                    _lib.check(LIB.yamd_conv2d_bias_act(ctypes.byref(layer.desc), xp, xcs, layer.w_dev.data_ptr(),
                                                        layer.b_dev.data_ptr(), rp, rcs, yp, ycs, B, x.h, x.w, st),
                               f"conv[{layer.name}]")
            elif kind == "conv_cat2":
                _, layer, xa, a_half, xb, b_half, out, out2 = op
                xap, xacs = P(bufs, xa)
                xbp, xbcs = P(bufs, xb)
                yp, ycs = P(bufs, out)
                y2p, y2cs = P(bufs, out2) if out2 is not None else (None, 0)
                _lib.check(LIB.yamd_conv2d_bias_act_cat2(ctypes.byref(layer.desc), xap, xacs, xa.c, a_half, xbp, xbcs, b_half,
                                                         layer.w_dev.data_ptr(), layer.b_dev.data_ptr(), yp, ycs, y2p, y2cs,
                                                         out.c if out2 is not None else 0, B, out.h, out.w, st), f"conv[{layer.name}]")
            elif kind == "bneck":
                _, layer, x, out, shortcut = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                _lib.check(LIB.yamd_bottleneck(layer.cin, layer.cmid, layer.cout, layer.act, xp, xcs, layer.w_dev.data_ptr(),
                                               layer.b_dev.data_ptr(), shortcut, yp, ycs, B, out.h, out.w, st), f"bottleneck[{layer.name}]")
            elif kind == "c3":
                _, layer, x, out, shortcut = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                _lib.check(LIB.yamd_c3_fused(layer.cin, layer.ch, layer.cout, layer.act, xp, xcs, layer.w_dev.data_ptr(),
                                             layer.b_dev.data_ptr(), shortcut, yp, ycs, B, out.h, out.w, st), f"c3[{layer.name}]")
            elif kind == "stem":
                _, layer, out = op
                yp, ycs = P(bufs, out)
                _lib.check(LIB.yamd_stem_conv(layer.mode, images.data_ptr(), B, H, W, layer.w_dev.data_ptr(),
                                              layer.b_dev.data_ptr(), layer.cout, layer.act, yp, ycs, st), f"stem[{layer.name}]")
            elif kind == "stem2":
                _, layer, out = op
                yp, ycs = P(bufs, out)
                st1 = layer.stem
                _lib.check(LIB.yamd_stem_conv2(st1.mode, images.data_ptr(), B, H, W, st1.w_dev.data_ptr(), st1.b_dev.data_ptr(),
                                               st1.cout, st1.act, layer.w_dev.data_ptr(), layer.b_dev.data_ptr(), layer.cout,
                                               layer.act, yp, ycs, st), f"stem_conv2[{layer.name}]")
            elif kind == "focus":
                yp, ycs = P(bufs, op[1])
                _lib.check(LIB.yamd_focus_pack(images.data_ptr(), yp, ycs, B, H, W, st), "focus")
            elif kind == "nhwc16":
                yp, ycs = P(bufs, op[1])
                _lib.check(LIB.yamd_nchw_to_nhwc16(images.data_ptr(), yp, ycs, B, op[2], H, W, st), "nhwc16")
            elif kind == "up2":
                _, x, out = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                _lib.check(LIB.yamd_upsample2x(xp, xcs, yp, ycs, B, x.h, x.w, x.c, st), "upsample2x")
            elif kind == "mp5":
                _, x, y1, y2, y3, kernel = op
                xp, xcs = P(bufs, x)
                p1, ycs = P(bufs, y1)
                p2, _ = P(bufs, y2)
                p3, _ = P(bufs, y3)
                fn = LIB.yamd_maxpool5_cascade if kernel == 5 else LIB.yamd_maxpool3_cascade
                _lib.check(fn(xp, xcs, p1, p2, p3, ycs, B, x.h, x.w, x.c, st), "maxpool_cascade")
            elif kind == "copy":
                _, x, out = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                _lib.check(LIB.yamd_copy_channels(xp, xcs, yp, ycs, B * x.h * x.w, x.c, st), "copy")
            elif kind == "d2s":
                _, x, out = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                _lib.check(LIB.yamd_depth_to_space2(xp, xcs, yp, ycs, B, x.h, x.w, out.c, st), "depth_to_space2")
            elif kind == "mp2":
                _, x, out = op
                xp, xcs = P(bufs, x)
                yp, ycs = P(bufs, out)
                _lib.check(LIB.yamd_maxpool2(xp, xcs, yp, ycs, B, x.h, x.w, x.c, st), "maxpool2")
            else:
                raise AssertionError(kind)
            if timings is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                if kind in ("stem", "stem2", "conv_cat2", "bneck", "c3"):          # reported with the convs
                    timings.append(("conv", op[1].name, ev0, ev1) + self.op_cost(op, B))
                else:
                    timings.append((kind, op[1].name if kind == "conv" else kind, ev0, ev1) + self.op_cost(op, B))
        if span_open is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            conv_spans.append((span_open, ev))
        return self._outputs(bufs)

    def _outputs(self, bufs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        res = {}
        for name, v in self.outputs.items():
            buf, coff = self.resolve(v)
            t = bufs[buf]
            res[name] = t if (coff == 0 and v.c == t.shape[-1]) else t[..., coff:coff + v.c]
        return res
