"""Name -> model class, parameter factory and seeded-initialiser options for the model families built on the
HIP path; the names are the ones tests, goldens and the benchmark use."""
from typing import Any, Dict

from . import param as _param


def registry() -> Dict[str, Dict[str, Any]]:
    from .yolov3 import YoloV3
    from .yolov4 import YoloV4
    from .yolov5 import YoloV5
    from .yolov5_face import YoloV5Face, YoloV5Pose
    from .yolov6_face import YoloV6, YoloV6Face, YoloV6Pose
    from .yolov7 import YoloV7, YoloV7Face
    from .yolov7_pose import YoloV7Pose
    from .yolov8 import YoloV8
    return {
        # at 640x640 the random-init features of this graph reach the heads with a std near 10: damp the head convs
        # so that the logits stay in the range of a trained network
        "v5face": dict(cls=YoloV5Face, param=lambda: _param.Yolo5FaceParam(m_type="s"), init=dict(scale={"yolo_head_P": 0.1})),
        "v5pose": dict(cls=YoloV5Pose, param=lambda: _param.Yolo5PoseParam(m_type="s"), init=dict(scale={"yolo_head_P": 0.1})),
        # m / x widths of the same graphs (24- and 40-channel stem conv); bottleneck damping as for v5m below
        "v5facem": dict(cls=YoloV5Face, param=lambda: _param.Yolo5FaceParam(m_type="m"),
                        init=dict(scale={f"m.{i}.cv2.bn.weight": 0.5 for i in range(6)})),
        "v5posex": dict(cls=YoloV5Pose, param=lambda: _param.Yolo5PoseParam(m_type="x"),
                        init=dict(scale={f"m.{i}.cv2.bn.weight": 0.5 for i in range(12)})),
        "v5s": dict(cls=YoloV5, param=lambda: _param.Yolo5Param(m_type="s").set_img_size(640), init={}),
        "v8s": dict(cls=YoloV8, param=lambda: _param.Yolo8Param(m_type="s"), init={}),
        # other widths of the two benchmark families (channel counts that are multiples of 16 but not of 32)
        # 2 / 6 / 6 bottlenecks per C3 with un-normalised residual adds: damp every bottleneck's last BatchNorm so that
        # random-init logits keep a std near 1 (they reach 20-50 otherwise)
        "v5m": dict(cls=YoloV5, param=lambda: _param.Yolo5Param(m_type="m").set_img_size(640),
                    init=dict(scale={f"m.{i}.cv2.bn.weight": 0.5 for i in range(6)})),
        "v8m": dict(cls=YoloV8, param=lambda: _param.Yolo8Param(m_type="m"), init={}),
        "v8n": dict(cls=YoloV8, param=lambda: _param.Yolo8Param(m_type="n"), init={}),
        # 23 un-normalised residual adds: damp the branch output so random-init activations stay in range, and the
        # head convs so that the logits (w,h go through exp) have a std of 1-2
        "v3": dict(cls=YoloV3, param=lambda: _param.Yolo3Param().set_img_size(608),
                   init=dict(scale={".conv.3.weight": 0.25, ".6.weight": 0.05})),
        "v4": dict(cls=YoloV4, param=lambda: _param.Yolo4Param().set_img_size(608),
                   init=dict(scale={".block.1.conv.weight": 0.25, "yolo_head1.1.weight": 0.05,
                                    "yolo_head2.1.weight": 0.05, "yolo_head3.1.weight": 0.05})),
        "v7pose": dict(cls=YoloV7Pose, param=lambda: _param.Yolo7PoseParam(m_type="l"), init={}),
        "v7": dict(cls=YoloV7, param=lambda: _param.Yolo7Param(m_type="l"), init={}),
        "v7x": dict(cls=YoloV7, param=lambda: _param.Yolo7Param(m_type="x"), init={}),
        "v7face": dict(cls=YoloV7Face, param=lambda: _param.Yolo7FaceParam(m_type="l"), init={}),
        # RepVGG blocks sum three un-normalised branches ~40 times in a row: scale the branch BatchNorms so that the
        # sum keeps the activation variance; box/landmark logits (grid units, multiplied by the stride) get a std near 1
        "v6": dict(cls=YoloV6, param=lambda: _param.Yolo6Param(m_type="s"),
                   init=dict(scale={".rbr_dense.bn.weight": 0.55, ".rbr_1x1.bn.weight": 0.55,
                                    ".rbr_identity.weight": 0.55, "reg_predict_list": 0.3})),
        # m/l: BottleRep pairs with a learned shortcut weight; alpha away from 1 so that the scaling is exercised and the
        # residual stacks keep their range; DFL logits with a std near 2 give varied box sizes
        "v6m": dict(cls=YoloV6, param=lambda: _param.Yolo6Param(m_type="m"),
                    init=dict(scale={".rbr_dense.bn.weight": 0.55, ".rbr_1x1.bn.weight": 0.55,
                                     ".rbr_identity.weight": 0.55, "reg_predict_list": 2.0, ".alpha": 0.6,
                                     ".cv3.block.bn.weight": 0.5})),
        "v6l": dict(cls=YoloV6, param=lambda: _param.Yolo6Param(m_type="l"),
                    init=dict(scale={"reg_predict_list": 2.0, ".alpha": 0.6, ".cv3.block.bn.weight": 0.5})),
        "v6facem": dict(cls=YoloV6Face, param=lambda: _param.Yolo6FaceParam(m_type="m"),
                        init=dict(scale={".rbr_dense.bn.weight": 0.55, ".rbr_1x1.bn.weight": 0.55,
                                         ".rbr_identity.weight": 0.55, "reg_predict_list": 2.0, ".alpha": 0.6,
                                         ".cv3.block.bn.weight": 0.5, "reg_point_predict_list": 0.3})),
        "v6posel": dict(cls=YoloV6Pose, param=lambda: _param.Yolo6PoseParam(m_type="l"),
                        init=dict(scale={"reg_predict_list": 2.0, ".alpha": 0.6, ".cv3.block.bn.weight": 0.5,
                                         "reg_point_predict_list": 0.3})),
        "v6pose": dict(cls=YoloV6Pose, param=lambda: _param.Yolo6PoseParam(m_type="s"),
                       init=dict(scale={".rbr_dense.bn.weight": 0.55, ".rbr_1x1.bn.weight": 0.55,
                                        ".rbr_identity.weight": 0.55, "reg_predict_list": 0.3,
                                        "reg_point_predict_list": 0.3})),
        "v6face": dict(cls=YoloV6Face, param=lambda: _param.Yolo6FaceParam(m_type="s"),
                       init=dict(scale={".rbr_dense.bn.weight": 0.55, ".rbr_1x1.bn.weight": 0.55,
                                        ".rbr_identity.weight": 0.55, "reg_predict_list": 0.3,
                                        "reg_point_predict_list": 0.3})),
        "v6facen": dict(cls=YoloV6Face, param=lambda: _param.Yolo6FaceParam(m_type="n"),
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


def from_state_dict(sd, img_size: int, output_size: int = 0, point_num: int = 0):
    """Builds the model a checkpoint belongs to from its key names and tensor shapes (what an ONNX export carries
    implicitly) and loads it.  ``output_size`` is the per-anchor column count of ``_forward`` (5 + classes [+ points]);
    0 derives the class count from the head convolutions."""
    from .yolov3 import YoloV3
    from .yolov4 import YoloV4
    from .yolov5 import YoloV5
    from .yolov6_face import YoloV6Face
    from .yolov7_pose import YoloV7Pose
    from .yolov8 import YoloV8
    keys = sd.keys()

    def has(k):
        return k in keys

    def width_type(w, table):
        for t, v in table.items():
            if v == w:
                return t
        raise ValueError(f"unrecognised network width {w}")

    def with_classes(p, nc):
        if nc != p.class_num:
            p.class_names = [str(i) for i in range(nc)]
            p.class_num = nc
        return p

    if has("body_net.darknet53.conv1.0.weight"):
        nc = sd["body_net.last_layer0.6.weight"].shape[0] // 3 - 5
        p, cls = with_classes(_param.Yolo3Param(), nc), YoloV3
    elif has("body_net.backbone.stages.0.downsample_conv.conv.weight"):
        nc = sd["body_net.yolo_head3.1.weight"].shape[0] // 3 - 5
        p, cls = with_classes(_param.Yolo4Param(), nc), YoloV4
    elif has("body_net.backbone.stem.stem_1.conv.weight"):
        # v5 face / pose share one graph; the head width tells them apart: 5 + 3*17 + classes (pose, one confidence
        # per keypoint) when that leaves at least one class, else 5 + 2*points + classes (face)
        from .yolov5_face import YoloV5Face, YoloV5Pose
        t = width_type(sd["body_net.backbone.stem.stem_1.conv.weight"].shape[0], {"s": 32, "m": 48, "l": 64, "x": 80})
        oc = sd["body_net.yolo_head_P3.weight"].shape[0] // 3
        if point_num in (0, 17) and oc - 5 - 3 * 17 >= 1:
            p, cls = with_classes(_param.Yolo5PoseParam(m_type=t), oc - 5 - 3 * 17), YoloV5Pose
        else:
            p = _param.Yolo5FaceParam(m_type=t)
            p.point_num = point_num or p.point_num
            p, cls = with_classes(p, oc - 5 - 2 * p.point_num), YoloV5Face
    elif has("body_net.backbone.stem.conv.conv.weight"):
        t = width_type(sd["body_net.backbone.stem.conv.conv.weight"].shape[0], {"s": 32, "m": 48, "l": 64, "x": 80})
        nc = sd["body_net.yolo_head_P3.weight"].shape[0] // 3 - 5
        p, cls = with_classes(_param.Yolo5Param(m_type=t), nc), YoloV5
    elif has("head.proj") and has("body_net.backbone.stem.conv.weight"):
        w0 = sd["body_net.backbone.stem.conv.weight"].shape[0]
        n_bottle = sum(1 for k in keys if k.startswith("body_net.backbone.dark2.1.m.") and k.endswith("cv1.conv.weight"))
        t = {(16, 1): "n", (32, 1): "s", (48, 2): "m", (64, 3): "l", (80, 3): "x"}[(w0, n_bottle)]
        nc = sd["body_net.cv3.0.2.weight"].shape[0]
        p, cls = with_classes(_param.Yolo8Param(m_type=t), nc), YoloV8
    elif has("body_net.backbone.stem.0.conv.weight") or has("body_net.backbone.stem.0.conv.bias"):
        # the three v7 models share one graph; the head width (and the caller's point count) tells them apart:
        # pose 5 + 3 x 17 + classes, face 5 + 2 x 5 + classes, detect 5 + classes
        from .yolov7 import YoloV7, YoloV7Face
        t = width_type(sd["body_net.backbone.stem.0.conv.weight"].shape[0], {"l": 32, "x": 40})
        oc = sd["body_net.yolo_head_P3.weight"].shape[0] // 3
        if point_num >= 10 or (point_num == 0 and oc - 5 - 3 * 17 >= 1):
            p = _param.Yolo7PoseParam(m_type=t)
            p.point_num = point_num or p.point_num
            p, cls = with_classes(p, oc - 5 - 3 * p.point_num), YoloV7Pose
        elif point_num > 0 or oc == 16:
            p = _param.Yolo7FaceParam(m_type=t)
            p.point_num = point_num or p.point_num
            p, cls = with_classes(p, oc - 5 - 2 * p.point_num), YoloV7Face
        else:
            p, cls = with_classes(_param.Yolo7Param(m_type=t), oc - 5), YoloV7
    elif any(k.startswith("yolo_body.back_bone_network.stem.") for k in keys):
        k = "yolo_body.head.stems.0.block.conv.weight"
        t = width_type(sd[k].shape[0], {"n": 32, "s": 64, "m": 96, "l": 128})
        from .yolov6_face import YoloV6
        pk = "yolo_body.head.reg_point_predict_list.0.weight"
        from .yolov6_face import YoloV6Pose
        if pk in keys and (point_num >= 10 or (point_num == 0 and sd[pk].shape[0] // 2 >= 10)):
            # many keypoints: the pose model, whose class conv also carries one confidence per keypoint
            p, cls = _param.Yolo6PoseParam(m_type=t), YoloV6Pose
            p.point_num = sd[pk].shape[0] // 2
        elif pk in keys:                                      # landmark branch present: the face model
            p, cls = _param.Yolo6FaceParam(m_type=t), YoloV6Face
            p.point_num = sd[pk].shape[0] // 2
        else:
            p, cls = _param.Yolo6Param(m_type=t), YoloV6
        p.num_layers = sum(1 for q in keys if q.startswith("yolo_body.head.stems.") and q.endswith("conv.weight"))
        p.strides = [8, 16, 32, 64][:p.num_layers]
        p = with_classes(p, sd["yolo_body.head.cls_predict_list.0.weight"].shape[0]
                         - (p.point_num if getattr(p, "point_conf", False) else 0))
    else:
        raise ValueError("checkpoint does not belong to a model family built on the HIP path")
    p.set_img_size(img_size)
    m = cls(p)
    if set(keys) != set(m.state_dict().keys()):
        try:
            m.fuse()                  # a checkpoint saved after the reference's fuse() has that layout
        except NotImplementedError:
            pass
    m.load_state_dict(sd, strict=True)
    width = m._forward_width() if hasattr(m, "_forward_width") else 0
    if output_size and width and width != output_size:
        raise ValueError(f"checkpoint decodes to {width} columns per anchor, caller expects {output_size}")
    return p, m
