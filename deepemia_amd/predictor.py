"""Drop-in for Detectron2's ``DefaultPredictor`` + ``Instances`` as the reference uses them.

Reference contract:
  * built by ``load_model(cfg, path, dataset_name)`` -> ``DefaultPredictor(cfg)``
    (``src/data/models.py:103-107``);
  * called as ``outputs = predictor(image)`` with an (H, W, 3) uint8 BGR array
    (``src/functions/inference.py:1395,1398,1507,1669``);
  * consumed as ``outputs["instances"].to("cpu")._fields["pred_masks"].numpy()``
    (``inference.py:1401-1403``), ``outputs["instances"].pred_classes.cpu().numpy()``
    (``1514-1516``), ``len(outputs["instances"])`` (``1509``);
  * ``predictor.model.eval()`` / ``predictor.model.parameters()`` must exist
    (``inference.py:113-116``).

MI355X-native difference: masks stay on the GPU bit-packed; the (N, H, W) bool tensor
Detectron2 would have produced is materialised lazily, only if a caller asks for
``pred_masks`` (the native pipeline in ``functions/inference.py`` never does).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from .engine import MaskRCNNEngine, RawDetections


class _LazyFields(dict):
    """``Instances._fields`` whose ``pred_masks`` entry is produced on first access."""

    def __init__(self):
        super().__init__()
        self._thunk = None

    def _force(self):
        if self._thunk is not None:
            thunk, self._thunk = self._thunk, None
            dict.__setitem__(self, "pred_masks", thunk())

    def __getitem__(self, k):
        if k == "pred_masks":
            self._force()
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return dict.__contains__(self, k) or (k == "pred_masks" and self._thunk is not None)

    def items(self):
        self._force()
        return dict.items(self)

    def keys(self):
        self._force()
        return dict.keys(self)

    def values(self):
        self._force()
        return dict.values(self)


class Instances:
    """Minimal ``detectron2.structures.Instances`` work-alike (fields + ``to`` + ``len``)."""

    def __init__(self, image_size, engine: Optional[MaskRCNNEngine] = None, **fields):
        self.__dict__["_image_size"] = tuple(image_size)
        self.__dict__["_engine"] = engine
        self.__dict__["_fields"] = _LazyFields()
        self.__dict__["_packed"] = None
        for k, v in fields.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def set(self, name: str, value) -> None:
        dict.__setitem__(self._fields, name, value)

    def __setattr__(self, name, value):
        if name.startswith("_"):
            self.__dict__[name] = value
        else:
            self.set(name, value)

    def has(self, name: str) -> bool:
        return name in self._fields

    def get(self, name: str):
        return self._fields[name]

    def get_fields(self):
        self._fields._force()
        return self._fields

    def __getattr__(self, name: str):
        if name.startswith("_") or name not in self.__dict__["_fields"]:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self.__dict__["_fields"][name]

    def set_packed_masks(self, packed: torch.Tensor) -> None:
        """Attach bit-packed masks; ``pred_masks`` (N, H, W) bool is unpacked on first access."""
        self.__dict__["_packed"] = packed
        h, w = self._image_size
        eng = self._engine
        self._fields._thunk = lambda: eng.unpack(packed, h, w)

    def to(self, device) -> "Instances":
        out = Instances(self._image_size, self._engine)
        for k, v in self._fields.items():  # forces the mask unpack, as Detectron2's .to("cpu") would copy it
            out.set(k, v.to(device) if hasattr(v, "to") else v)
        return out

    def __len__(self) -> int:
        f = self.__dict__["_fields"]
        for k in ("scores", "pred_classes", "pred_boxes"):
            if dict.__contains__(f, k):
                return len(dict.__getitem__(f, k))
        return 0

    @property
    def packed_masks(self) -> Optional[torch.Tensor]:
        """[N, H, W/32] int32 device tensor, bit (x & 31) of word (x >> 5) = mask[y][x]."""
        return self.__dict__["_packed"]


class _ModelShim(torch.nn.Module):
    """``predictor.model`` as the reference touches it: ``eval()`` and ``parameters()``."""

    def __init__(self, engine: MaskRCNNEngine):
        super().__init__()
        self._engine = engine
        params = [engine.stem_w]
        for stage in engine.blocks:
            for blk in stage:
                params += [l.w for l in blk.values()]
        self._params = [torch.nn.Parameter(p, requires_grad=False) for p in params[:4]]
        for i, p in enumerate(self._params):
            self.register_parameter(f"w{i}", p)

    def forward(self, *a, **k):  # pragma: no cover - the shim is never called
        raise RuntimeError("use the predictor, not predictor.model")


class Predictor:
    """``predictor(image_bgr_u8) -> {"instances": Instances}`` on one MI355X."""

    def __init__(self, engine: MaskRCNNEngine):
        self.engine = engine
        self.model = _ModelShim(engine)
        self.input_format = "BGR"

    def instances_from_raw(self, raw: RawDetections, b: int) -> Instances:
        n = int(raw.count[b].item())
        valid = raw.valid[b, :n].bool()
        inst = Instances((raw.height, raw.width), self.engine)
        if bool(valid.all()):
            sel = slice(0, n)
            inst.set_packed_masks(raw.packed[b, :n])
        else:
            sel = valid.nonzero().flatten()
            inst.set_packed_masks(raw.packed[b, :n][sel].contiguous())
        inst.set("pred_boxes", raw.boxes[b, :n][sel])
        inst.set("scores", raw.scores[b, :n][sel])
        inst.set("pred_classes", raw.classes[b, :n][sel].to(torch.int64))
        return inst

    @torch.no_grad()
    def __call__(self, original_image: np.ndarray) -> Dict[str, Instances]:
        if original_image.ndim != 3 or original_image.shape[2] != 3 or original_image.dtype != np.uint8:
            raise ValueError("predictor expects an (H, W, 3) uint8 BGR image")
        x = torch.from_numpy(np.ascontiguousarray(original_image))[None].to(self.engine.device, non_blocking=True)
        raw = self.engine.forward(x)
        return {"instances": self.instances_from_raw(raw, 0)}

    @torch.no_grad()
    def predict_batch(self, images: torch.Tensor) -> List[Dict[str, Instances]]:
        """[B, H, W, 3] uint8 device tensor -> list of outputs (tiles share one launch sequence)."""
        raw = self.engine.forward(images)
        return [{"instances": self.instances_from_raw(raw, b)} for b in range(images.shape[0])]
