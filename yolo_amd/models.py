"""Name -> (model class, parameter factory) for the model families built on the HIP path; the names are
the ones tests, goldens and the benchmark use."""
from typing import Callable, Dict, Tuple

from . import param as _param


def registry() -> Dict[str, Tuple[type, Callable[[], _param.Param]]]:
    from .yolov5 import YoloV5
    from .yolov8 import YoloV8
    return {
        "v5s": (YoloV5, lambda: _param.Yolo5Param(m_type="s").set_img_size(640)),
        "v8s": (YoloV8, lambda: _param.Yolo8Param(m_type="s")),
    }


def create(name: str, img_size: int = 0):
    cls, mk = registry()[name]
    p = mk()
    if img_size:
        p.set_img_size(img_size)
    return p, cls(p)
