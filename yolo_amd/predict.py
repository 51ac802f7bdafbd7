"""Deployment entry points with the reference's signatures (luoyang/predict/Predict.py:10-32,
luoyang/predict/YoloPredict.py:8-115).

The reference wraps an onnxruntime session over an ONNX export of ``TaskLayer._forward`` and runs
``torch_utils.non_max_suppression*`` on its output.  Here the "session" is the compiled HIP plan of the same
model, so ``onnx_path`` names the checkpoint the export would have been made from (a ``state_dict`` saved by
``torch.save``, luoyang/model/Model.py:348-349); the architecture is recognised from its keys.  Alternatively a
ready model object can be passed with ``model=``.  ``predict`` keeps the whole path on the GPU: one H2D copy of
the images, one packed D2H copy of the kept rows.
"""
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import models as _models
from . import torch_utils


class Predict(object):
    def __init__(self, onnx_path: Optional[str], device: str, input_size: Tuple[int, int] = (640, 640),
                 output_size: int = 0, point_num: int = 0, model=None):
        self.onnx_path = onnx_path
        self.device = device
        if model is None:
            sd = torch.load(onnx_path, map_location="cpu", weights_only=True)
            _, model = _models.from_state_dict(sd, int(input_size[0]), output_size, point_num)
        self.session = model
        self.outputs = NotImplemented

    def _device_forward(self, feed_dict: Dict[str, np.ndarray]) -> torch.Tensor:
        images = feed_dict["images"]
        if not isinstance(images, torch.Tensor):
            images = torch.from_numpy(np.ascontiguousarray(images, dtype=np.float32))
        return self.session._forward(images.to(self.device, dtype=torch.float32).contiguous())[0]

    def sess_run(self, outputs: List[str], feed_dict: Dict[str, np.ndarray]) -> List[np.ndarray]:
        """The exported graph has the single output ``head_predicts`` = ``_forward(images)[0]``."""
        if list(outputs) != ["head_predicts"]:
            raise KeyError(f"the exported graph has one output, 'head_predicts'; asked for {outputs}")
        return [self._device_forward(feed_dict).cpu().numpy()]

    def io_biding_run(self):
        raise NotImplementedError

    def predict(self, feed_dict: Dict[str, np.ndarray]):
        raise NotImplementedError

    def sess_predict(self, feed_dict: Dict[str, np.ndarray]) -> np.ndarray:
        return self.sess_run(outputs=self.outputs, feed_dict=feed_dict)[0]


class _NmsPredict(Predict):
    N_EXTRA_PER_POINT = 0

    def __init__(self, onnx_path: Optional[str], device: str, point_num: int, input_size: Tuple[int, int],
                 output_size: int, conf_threshold: float, nms_threshold: float, outputs: Union[None, List[str]], model):
        super().__init__(onnx_path, device, input_size, output_size, point_num, model)
        self.point_num = point_num
        self.output_size = output_size
        self.conf_threshold = conf_threshold
        self.nms_threshold = nms_threshold
        self.input_size = input_size
        self.outputs = ["head_predicts"] if outputs is None else outputs

    def predict(self, feed_dict: Dict[str, np.ndarray]) -> List[Optional[np.ndarray]]:
        pred = self._device_forward(feed_dict)
        rows, counts, _ = torch_utils.nms_batched(pred, self.input_size[0], "cxcywh", self.conf_threshold,
                                                  self.nms_threshold, self.N_EXTRA_PER_POINT * self.point_num)
        return torch_utils.rows_to_host(rows, counts)


class YoloPredict(_NmsPredict):
    """``YoloPredict(onnx_path, device, input_size, output_size, conf_threshold=0.1, nms_threshold=0.3, outputs=None)``
    (YoloPredict.py:8-39); rows [x1,y1,x2,y2,obj,cls_conf,cls_id]."""

    def __init__(self, onnx_path: Optional[str], device: str, input_size: Tuple[int, int], output_size: int,
                 conf_threshold: float = 0.1, nms_threshold: float = 0.3, outputs: Union[None, List[str]] = None,
                 model=None):
        super().__init__(onnx_path, device, 0, input_size, output_size, conf_threshold, nms_threshold, outputs, model)


class YoloFacePredict(_NmsPredict):
    """YoloPredict.py:42-77: ``non_max_suppression_with_point``; rows [box, obj, cls_conf, 2P landmarks, cls_id]."""
    N_EXTRA_PER_POINT = 2

    def __init__(self, onnx_path: Optional[str], device: str, point_num: int, input_size: Tuple[int, int],
                 output_size: int, conf_threshold: float = 0.1, nms_threshold: float = 0.3,
                 outputs: Union[None, List[str]] = None, model=None):
        super().__init__(onnx_path, device, point_num, input_size, output_size, conf_threshold, nms_threshold, outputs,
                         model)


class YoloPosePredict(_NmsPredict):
    """YoloPredict.py:80-115: ``non_max_suppression_with_point_conf``; rows [box, obj, cls_conf, 2P keypoints,
    P keypoint confidences, cls_id]."""
    N_EXTRA_PER_POINT = 3

    def __init__(self, onnx_path: Optional[str], device: str, point_num: int, input_size: Tuple[int, int],
                 output_size: int, conf_threshold: float = 0.1, nms_threshold: float = 0.3,
                 outputs: Union[None, List[str]] = None, model=None):
        super().__init__(onnx_path, device, point_num, input_size, output_size, conf_threshold, nms_threshold, outputs,
                         model)
