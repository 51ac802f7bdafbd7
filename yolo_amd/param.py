"""Per-model hyper-parameter objects with the attribute names the reference's callers use
(luoyang/param/Param.py: Param :5-45, Yolo3Param :48-94, Yolo4Param :97-140, Yolo5Param :143-196,
Yolo7PoseParam :437-468, Yolo8Param :505-535, Yolo6FaceParam :254-288).

Only the inference-relevant fields are kept; training schedules are out of scope.
``feature_size`` is derived from ``img_size`` and ``strides`` (the reference hard-codes it and asks
callers to keep both in sync), but it stays a plain attribute so scripts that assign it still work.
"""
from typing import List, Optional, Tuple

VOC_NAMES = ("aeroplane bicycle bird boat bottle bus car cat chair cow diningtable dog horse motorbike person "
             "pottedplant sheep sofa train tvmonitor").split()

_ANCHORS_V3 = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]
_ANCHORS_V7 = [(12, 16), (19, 36), (40, 28), (36, 75), (76, 55), (72, 146), (142, 110), (192, 243), (459, 401)]
_ANCHORS_FACE = [(4, 5), (8, 10), (13, 16), (23, 29), (43, 55), (73, 105), (146, 217), (231, 300), (335, 433)]


class Param(object):
    """Common inference fields (reference defaults: conf 0.1, nms 0.3, 20 VOC classes)."""

    def __init__(self, batch_size: int = 1):
        self.batch_size = batch_size
        self.class_names: List[str] = list(VOC_NAMES)
        self.class_num: int = len(self.class_names)
        self.strides: List[int] = [8, 16, 32]
        self.anchors: Optional[List[Tuple[int, int]]] = None
        self.anchor_index = [(0, 1, 2), (3, 4, 5), (6, 7, 8)]
        self.anchor_num = 3
        self.img_size: int = 416
        self.feature_size: List[int] = [52, 26, 13]
        self.conf_threshold: float = 0.1
        self.nms_threshold: float = 0.3
        self.ap_iou_threshold: float = 0.3
        self.image_mean = None
        self.image_std = None

    def set_img_size(self, img_size: int) -> "Param":
        """Set ``img_size`` and the matching ``feature_size`` together."""
        self.img_size = int(img_size)
        self.feature_size = [self.img_size // s for s in self.strides]
        return self

    def _names(self, names):
        self.class_names = list(names)
        self.class_num = len(self.class_names)


class Yolo3Param(Param):
    def __init__(self, batch_size: int = 12, darknet53_weight_path: Optional[str] = None):
        super().__init__(batch_size)
        self.darknet53_weight_path = darknet53_weight_path
        self.dark_blocks = [1, 2, 8, 8, 4]
        self.darknet53_output_filters = [64, 128, 256, 512, 1024]
        self.anchors = list(_ANCHORS_V3)


class Yolo4Param(Param):
    def __init__(self, batch_size: int = 6):
        super().__init__(batch_size)
        self.darknet_csp_weight_path = None
        self.dark_blocks = [1, 2, 8, 8, 4]
        self.anchors = list(_ANCHORS_V3)


class Yolo5Param(Param):
    def __init__(self, batch_size: int = 6, m_type: str = "l"):
        super().__init__(batch_size)
        if m_type not in ("s", "m", "l", "x"):
            raise NotImplementedError(m_type)
        self.m_type = m_type
        self.darknet_csp_weight_path = None
        self.anchors = list(_ANCHORS_V3)


class Yolo7Param(Param):
    def __init__(self, batch_size: int = 8, m_type: str = "l"):
        super().__init__(batch_size)
        self.m_type = m_type
        self.pretrain_path = None
        self.anchors = list(_ANCHORS_V7)
        self.set_img_size(640)


class Yolo7PoseParam(Yolo7Param):
    def __init__(self, batch_size: int = 12, m_type: str = "l"):
        super().__init__(batch_size, m_type)
        self._names(["person"])
        self.point_num = 17
        self.conf_threshold = 0.3


class Yolo7FaceParam(Yolo7Param):
    def __init__(self, batch_size: int = 8, m_type: str = "l"):
        super().__init__(batch_size, m_type)
        self._names(["face"])
        self.anchors = list(_ANCHORS_FACE)
        self.point_num = 5
        self.conf_threshold = 0.3


class Yolo5FaceParam(Yolo5Param):
    def __init__(self, batch_size: int = 12, m_type: str = "l"):
        super().__init__(batch_size, m_type)
        self._names(["face"])
        self.anchors = list(_ANCHORS_FACE)
        self.point_num = 5
        self.conf_threshold = 0.3
        self.set_img_size(640)


class Yolo5PoseParam(Yolo5Param):
    def __init__(self, batch_size: int = 12, m_type: str = "s"):
        super().__init__(batch_size, m_type)
        self._names(["person"])
        self.anchors = list(_ANCHORS_FACE)
        self.point_num = 17
        self.conf_threshold = 0.3
        self.set_img_size(640)


class Yolo8Param(Param):
    def __init__(self, batch_size: int = 8, m_type: str = "l"):
        super().__init__(batch_size)
        self.m_type = m_type
        self.pretrain_path = None
        self.set_img_size(640)


class Yolo6Param(Param):
    def __init__(self, batch_size: int = 6, m_type: str = "s"):
        super().__init__(batch_size)
        assert m_type in ("n", "s", "m", "l")
        self.m_type = m_type
        self.use_dfl = m_type in ("m", "l")
        self.pretrain_path = None
        self.num_layers = 4
        self.strides = [8, 16, 32, 64]
        self.set_img_size(640)


class Yolo6PoseParam(Param):
    def __init__(self, batch_size: int = 6, m_type: str = "s"):
        super().__init__(batch_size)
        assert m_type in ("n", "s", "m", "l")
        self.m_type = m_type
        self.use_dfl = m_type in ("m", "l")
        self.pretrain_path = None
        self._names(["person"])
        self.num_layers = 4
        self.strides = [8, 16, 32, 64]
        self.point_num = 17
        self.point_conf = True          # the head carries one confidence per keypoint
        self.set_img_size(640)


class Yolo6FaceParam(Param):
    def __init__(self, batch_size: int = 6, m_type: str = "s"):
        super().__init__(batch_size)
        assert m_type in ("n", "s", "m", "l")
        self.m_type = m_type
        self.use_dfl = m_type in ("m", "l")
        self.pretrain_path = None
        self._names(["person"])
        self.num_layers = 4
        self.strides = [8, 16, 32, 64]
        self.point_num = 5
        self.conf_threshold = 0.3
        self.set_img_size(640)
