"""YOLOv7 detect and YOLOv7 face on the HIP path.

The reference's three v7 bodies are one graph with different head widths (luoyang/yolov7/YoloV7Body.py and
luoyang/yolov7_face/YoloV7FaceBody.py equal luoyang/yolov7_pose/YoloV7PoseBody.py up to ``output_size`` :449), so
both classes reuse the graph of ``yolov7_pose.YoloV7Pose``.  Mirrors luoyang/yolov7/YoloV7.py (``_forward`` :45-71,
``predict`` :124-150, ``fuse`` :201-209; head = luoyang/yolov7/YoloV7Head.py:23-60, the v5 decode) and
luoyang/yolov7_face/YoloV7Face.py (``_forward`` :60-88, ``predict`` :130-166 with clamped landmarks, ``fuse`` :217-225;
head = luoyang/yolov7_face/YoloV7FaceHead.py:25-77).
"""
from . import lib as _lib
from .param import Yolo7FaceParam, Yolo7Param
from .yolov7_pose import YoloV7Pose, _tree


class YoloV7(YoloV7Pose):
    """``YoloV7(param)``: rows [x1, y1, x2, y2, obj, cls_conf, cls_id]."""

    DECODE_KIND = _lib.DECODE_V5
    POINT_NUM = 0
    POINT_CONF = False

    def __init__(self, param: Yolo7Param, anchor_num: int = 3):
        super(YoloV7Pose, self).__init__()
        self.param = param
        self.body_output_size = 5 + param.class_num
        self.body_net = _tree(param, anchor_num, self.body_output_size)
        self.eval()


class YoloV7Face(YoloV7Pose):
    """``YoloV7Face(param)``: rows [x1, y1, x2, y2, obj, cls_conf, 5 x (lx, ly), cls_id]."""

    DECODE_KIND = _lib.DECODE_FACE
    POINT_CONF = False
    CLAMP_POINTS_IN_PREDICT = True

    def __init__(self, param: Yolo7FaceParam, anchor_num: int = 3):
        super(YoloV7Pose, self).__init__()
        self.param = param
        self.POINT_NUM = param.point_num
        self.body_output_size = 5 + 2 * param.point_num + param.class_num
        self.body_net = _tree(param, anchor_num, self.body_output_size)
        self.eval()
