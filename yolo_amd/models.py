"""Name -> model class, parameter factory and seeded-initialiser options for the model families built on the
HIP path; the names are the ones tests, goldens and the benchmark use."""
from typing import Any, Dict

from . import param as _param


def registry() -> Dict[str, Dict[str, Any]]:
    from .yolov3 import YoloV3
    from .yolov4 import YoloV4
    from .yolov5 import YoloV5
    from .yolov6_face import YoloV6Face
    from .yolov7_pose import YoloV7Pose
    from .yolov8 import YoloV8
    return {
        "v5s": dict(cls=YoloV5, param=lambda: _param.Yolo5Param(m_type="s").set_img_size(640), init={}),
        "v8s": dict(cls=YoloV8, param=lambda: _param.Yolo8Param(m_type="s"), init={}),
        # 23 un-normalised residual adds: damp the branch output so random-init activations stay in range, and the
        # head convs so that the logits (w,h go through exp) have a std of 1-2
        "v3": dict(cls=YoloV3, param=lambda: _param.Yolo3Param().set_img_size(608),
                   init=dict(scale={".conv.3.weight": 0.25, ".6.weight": 0.05})),
        "v4": dict(cls=YoloV4, param=lambda: _param.Yolo4Param().set_img_size(608),
                   init=dict(scale={".block.1.conv.weight": 0.25, "yolo_head1.1.weight": 0.05,
                                    "yolo_head2.1.weight": 0.05, "yolo_head3.1.weight": 0.05})),
        "v7pose": dict(cls=YoloV7Pose, param=lambda: _param.Yolo7PoseParam(m_type="l"), init={}),
        # RepVGG blocks sum three un-normalised branches ~40 times in a row: scale the branch BatchNorms so that the
        # sum keeps the activation variance; box/landmark logits (grid units, multiplied by the stride) get a std near 1
        "v6face": dict(cls=YoloV6Face, param=lambda: _param.Yolo6FaceParam(m_type="s"),
                       init=dict(scale={".rbr_dense.bn.weight": 0.55, ".rbr_1x1.bn.weight": 0.55,
                                        ".rbr_identity.weight": 0.55, "reg_predict_list": 0.3,
                                        "reg_point_predict_list": 0.3})),
    }


def init_options(name: str) -> Dict[str, Any]:
    """Keyword arguments for ``init.seeded_state_dict`` that go with this model's goldens."""
    return dict(registry()[name]["init"])


def create(name: str, img_size: int = 0):
    e = registry()[name]
    p = e["param"]()
    if img_size:
        p.set_img_size(img_size)
    return p, e["cls"](p)
