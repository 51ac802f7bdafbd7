"""yolo_amd -- MI355X-native inference hot path for the luoyang YOLO family.

Host side (this package) mirrors the reference's interfaces for the path
conv backbone/neck -> head decode -> confidence filter + NMS; all arithmetic runs in
hand-written gfx950 kernels behind the C ABI of ``include/yolo_amd.h``.
"""
__all__ = ["param", "torch_utils"]
