"""The batching-service entry of the reference (luoyang/predict/Predict.py:35-65 ``ManagePredict``,
luoyang/predict/ServicePredict.py:12-93 ``YoloPredict``) on the HIP path.

The reference class is a ``service_streamer.ManagedModel``: the streamer creates one instance per worker process,
calls ``init_model(model_dir=...)`` once and then ``predict(feed_dicts)`` with the requests it has batched.  The
streamer itself (process management, request queues) is a third-party package and is not rebuilt here; this module
keeps the part that belongs to the model: reading ``<model_dir>/config.json`` with the reference's keys

    onnx_name, input_size, output_size, conf_threshold, nms_threshold, outputs, nms, lables, mean, std

building the session, and ``predict(feed_dicts)``: decode -> letterbox -> normalise -> network -> NMS -> boxes back in
the original image -> ``{"predict_boxs": [[x1,y1,x2,y2], ...], "predict_label": [names]}`` per request, which is what
the block at ServicePredict.py:80-93 produces.  ``onnx_name`` names the exported graph; the session here is the
compiled plan of the checkpoint the export was made from, looked up next to it (same stem, ``.pth``).
"""
import base64
import json
import os
from io import BytesIO
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from . import models as _models
from . import torch_utils
from .transformer import RestoreSize, letterbox_batch


class ManagePredict(object):
    """Predict.py:35-65 without the streamer base class."""

    def __init__(self, gpu_id: Union[int, None] = None):
        self.gpu_id = gpu_id
        self.session = NotImplemented
        self.outputs = NotImplemented
        self.config_dict: Dict[str, Any] = NotImplemented
        self.restore_size = RestoreSize()
        self.device = torch.device("cuda", self.device_ordinal(gpu_id))

    @staticmethod
    def device_ordinal(gpu_id: Union[int, None]) -> int:
        """The device ordinal inside this process for the streamer's ``gpu_id``.  ``service_streamer.ManagedModel``
        workers select their GPU by masking visibility (``CUDA_VISIBLE_DEVICES = gpu_id`` before ``init_model``), after
        which the one visible device is ordinal 0; a caller that masks nothing addresses the device by ``gpu_id``."""
        if gpu_id in (None, -1):
            return 0
        for var in ("CUDA_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
            if os.environ.get(var, "").strip() == str(int(gpu_id)):
                return 0
        return int(gpu_id)

    def _on_device(self):
        """Makes ``self.device`` current for the calling thread: the library launches on the current device's stream and
        looks its per-device state up by the current device, so every call into it runs inside this context."""
        return torch.cuda.device(self.device)

    def _upload(self, array) -> torch.Tensor:
        return torch.as_tensor(array, dtype=torch.float32).to(self.device).contiguous()

    @staticmethod
    def checkpoint_path(model_dir: str, onnx_name: str) -> str:
        stem = onnx_name[:-5] if onnx_name.endswith(".onnx") else os.path.splitext(onnx_name)[0]
        for cand in (onnx_name, stem + ".pth", stem + ".pt"):
            path = os.path.join(model_dir, cand)
            if not cand.endswith(".onnx") and os.path.exists(path):
                return path
        raise FileNotFoundError(f"no checkpoint for '{onnx_name}' in {model_dir} (looked for {stem}.pth / {stem}.pt)")

    def init_model(self, *args, **kwargs):
        model_dir = kwargs["model_dir"]
        if self.gpu_id == -1:
            raise RuntimeError("gpu_id -1 asks for the CPU provider: yolo_amd has no CPU path")
        with open("{0}/config.json".format(model_dir), encoding="utf-8") as f:
            config_dict = json.load(f)
        path = self.checkpoint_path(model_dir, config_dict["onnx_name"])
        sd = torch.load(path, map_location="cpu", weights_only=True)
        with self._on_device():
            _, self.session = _models.from_state_dict(sd, int(config_dict["input_size"]), int(config_dict.get("output_size", 0)),
                                                      int(kwargs.get("point_num", 0)))
        self.config_dict = config_dict

    def predict(self, feed_dicts: List[Dict[str, Any]]):
        raise NotImplementedError

    def sess_run(self, outputs: List[str], feed_dict: Dict[str, np.ndarray]) -> List[np.ndarray]:
        if list(outputs) != ["head_predicts"]:
            raise KeyError(f"the exported graph has one output, 'head_predicts'; asked for {outputs}")
        with self._on_device():
            return [self.session._forward(self._upload(feed_dict["images"]))[0].cpu().numpy()]

    def sess_predict(self, feed_dict: Dict[str, np.ndarray]):
        return self.sess_run(outputs=self.outputs, feed_dict=feed_dict)[0]


def _decode_image(image) -> np.ndarray:
    """DecodeImage (transformer.py:124-131) + the ndarray view ResizeImage works on: base64 string, PIL image or an
    (H, W, 3) uint8 array -> (H, W, 3) uint8 RGB."""
    if isinstance(image, str):
        from PIL import Image
        image = Image.open(BytesIO(base64.b64decode(image)))
    if not isinstance(image, (np.ndarray, torch.Tensor)):
        image = np.asarray(image.convert("RGB"))
    return image


class YoloPredict(ManagePredict):
    """ServicePredict.py:12-93."""

    def __init__(self, gpu_id: Union[int, None] = None):
        super().__init__(gpu_id=gpu_id)
        self.input_size = NotImplemented
        self.output_size = NotImplemented
        self.conf_threshold = NotImplemented
        self.nms_threshold = NotImplemented
        self.nms = NotImplemented
        self.labels: List[str] = NotImplemented
        self.mean: Optional[torch.Tensor] = None
        self.std: Optional[torch.Tensor] = None

    def init_model(self, *args, **kwargs):
        super().init_model(*args, **kwargs)
        c = self.config_dict
        self.input_size = (c["input_size"], c["input_size"])
        self.output_size = c["output_size"]
        self.conf_threshold = c["conf_threshold"]
        self.nms_threshold = c["nms_threshold"]
        self.outputs = c.get("outputs", ["head_predicts"])
        self.nms = c.get("nms", True)
        self.labels = c["lables"]
        mean, std = c.get("mean", None), c.get("std", None)
        if mean is not None and std is not None:          # ImageNorm normalises only when both are given (:213-218)
            with self._on_device():
                self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device).view(1, 3, 1, 1)
                self.std = torch.tensor(std, dtype=torch.float32, device=self.device).view(1, 3, 1, 1)
        # warm-up request, as the reference sends one (its outcome is only reported there, :55-62)
        n = int(kwargs.get("batch_size", 1))
        self.sess_predict(feed_dict={"images": np.zeros((n, 3, self.input_size[0], self.input_size[0]), dtype=np.float32)})

    def predict(self, feed_dicts: List[Dict[str, Any]]) -> List[Any]:
        if not self.nms:
            raise NotImplementedError("config.json has nms = false: the reference has no such path either")
        with self._on_device():
            images, metas = letterbox_batch([_decode_image(d["image"]) for d in feed_dicts], self.input_size[0],
                                            self.input_size[1], device=self.device)
            if self.mean is not None:
                images = (images - self.mean) / self.std
            pred = self.session._forward(images)[0]
            rows, counts, _ = torch_utils.nms_batched(pred, self.input_size[0], "cxcywh", self.conf_threshold,
                                                      self.nms_threshold, 0)
            host_rows = torch_utils.rows_to_host(rows, counts)
        outputs = []
        for feed_dict, meta, nms_predicts in zip(feed_dicts, metas, host_rows):
            feed_dict.update(meta)
            feed_dict["predict"] = nms_predicts
            self.restore_size.transformer(data_dict=feed_dict)
            boxes = feed_dict["predict_boxs"]
            outputs.append({"predict_boxs": boxes.tolist() if isinstance(boxes, np.ndarray) else list(boxes),
                            "predict_label": [self.labels[i] for i in feed_dict["predict_label"]]})
        return outputs
