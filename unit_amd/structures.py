"""Minimal stand-ins for detectron2.structures.{Boxes, Instances, ImageList} and detectron2.utils.registry.Registry
with the attribute names the reference's plugin code uses (detectron2 is absent on the GPU box).
They are plain containers -- no arithmetic lives here."""
import torch


def _detectron2_registry(name):
    """the real Detectron2 registry object of that name when `import detectron2` succeeds, else None. d2's `build_model` /
    `build_roi_heads` / ... look classes up in THESE objects (call site in the reference: modeling/roi_heads/fast_rcnn.py:587-589
    for UniT's own two registries, which have no Detectron2 counterpart and stay local). Anything that is not a registry INSTANCE
    with a callable `register` (e.g. a placeholder class of a partial stub) is ignored."""
    where = {"META_ARCH": ("detectron2.modeling", "META_ARCH_REGISTRY"), "BACKBONE": ("detectron2.modeling", "BACKBONE_REGISTRY"),
             "PROPOSAL_GENERATOR": ("detectron2.modeling", "PROPOSAL_GENERATOR_REGISTRY"),
             "ROI_HEADS": ("detectron2.modeling", "ROI_HEADS_REGISTRY"), "ROI_BOX_HEAD": ("detectron2.modeling", "ROI_BOX_HEAD_REGISTRY"),
             "ROI_MASK_HEAD": ("detectron2.modeling", "ROI_MASK_HEAD_REGISTRY")}.get(name)
    if where is None:
        return None
    try:
        import importlib
        reg = getattr(importlib.import_module(where[0]), where[1])
    except Exception:       # detectron2 absent (this image) or too old to have the registry
        return None
    if isinstance(reg, type) or not callable(getattr(reg, "register", None)) or not hasattr(reg, "__contains__"):
        return None
    return reg


_ALL_REGISTRIES = []


class Registry:
    def __init__(self, name):
        self._name, self._map = name, {}
        _ALL_REGISTRIES.append(self)

    def register(self, obj=None):
        def deco(o):
            self._map[o.__name__] = o
            return o
        return deco(obj) if obj is not None else deco

    def get(self, name):
        if name not in self._map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._map[name]

    def __contains__(self, name):
        return name in self._map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
ROI_MASK_HEAD_REGISTRY = Registry("ROI_MASK_HEAD")
FAST_RCNN_REGISTRY = Registry("FAST_RCNN_REGISTRY")
WEAK_DETECTOR_FAST_RCNN_REGISTRY = Registry("WEAK_DETECTOR_FAST_RCNN")


def register_into_detectron2(overwrite=False):
    """EXPLICIT opt-in: mirror every class registered here into Detectron2's own registries, through their public
    `register` only. A name Detectron2 (or an already imported UniT.modeling) holds is left alone and reported, unless
    `overwrite=True`, in which case the existing entry is replaced for this process. Returns {"registered": [...], "skipped": [...]};
    raises if detectron2 is not importable."""
    import warnings
    done, skipped, found = [], [], False
    for r in _ALL_REGISTRIES:
        d2 = _detectron2_registry(r._name)
        if d2 is None:
            continue
        found = True
        for name, obj in r._map.items():
            if name in d2:
                if not overwrite:
                    skipped.append(f"{r._name}.{name}")
                    continue
                m = getattr(d2, "_obj_map", None)
                if not isinstance(m, dict):
                    skipped.append(f"{r._name}.{name}")
                    continue
                m.pop(name)
            d2.register(obj)
            done.append(f"{r._name}.{name}")
    if not found:
        raise RuntimeError("register_into_detectron2: detectron2's registries are not importable")
    if skipped:
        warnings.warn("unit_amd: names already registered in Detectron2 were left alone: " + ", ".join(skipped)
                      + " (import unit_amd.modeling INSTEAD of UniT.modeling, or pass overwrite=True)")
    return {"registered": done, "skipped": skipped}


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class Boxes:
    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        self.tensor = tensor

    def __len__(self):
        return self.tensor.shape[0]

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        return Boxes(self.tensor[item])

    def to(self, device):
        return Boxes(self.tensor.to(device))

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def cat(boxes_list):
        return Boxes(torch.cat([b.tensor for b in boxes_list], dim=0))


class Instances:
    def __init__(self, image_size, **kwargs):
        self._image_size = image_size
        self._fields = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return self._fields[name]

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, device):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v.to(device) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item):
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0


class ImageList:
    """tensor is NHWC (channels padded) in this framework; image_sizes keeps the un-padded (h, w) like detectron2."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    def __len__(self):
        return len(self.image_sizes)


class PolygonMasks:
    """Detectron2 `structures.masks.PolygonMasks` (the ground-truth mask container the reference's COCO-segm configuration trains on:
    INPUT.MASK_FORMAT is left at "polygon", data/dataset_mapper.py:104-106): per instance a list of polygons, each a flat float64 array
    [x0, y0, x1, y1, ...] in image coordinates. Only what the training path touches: len, indexing, `to`, and `crop_and_resize`
    (mask_head.py:34 -> mask_rcnn_loss), which runs the polygon rasteriser of the HIP library (unit_mask_targets_polygon)."""

    def __init__(self, polygons):
        import numpy as np
        self.polygons = [[np.asarray(p, dtype="float64").reshape(-1) for p in inst] for inst in polygons]
        for inst in self.polygons:
            for p in inst:
                if len(p) % 2 != 0 or len(p) < 6:
                    raise ValueError(f"a polygon needs at least 3 (x, y) points, got {len(p)} numbers")

    def __len__(self):
        return len(self.polygons)

    def __getitem__(self, item):
        import torch
        if isinstance(item, int):
            return PolygonMasks([self.polygons[item]])
        if isinstance(item, slice):
            return PolygonMasks(self.polygons[item])
        if torch.is_tensor(item):
            item = item.nonzero().flatten().tolist() if item.dtype == torch.bool else item.tolist()
        return PolygonMasks([self.polygons[i] for i in item])

    def to(self, *args, **kwargs):
        return self

    def crop_and_resize(self, boxes, mask_size):
        """boxes [N, 4] (device tensor, one per instance) -> bool [N, mask_size, mask_size]"""
        import torch
        from .modeling.mask_head import mask_targets_polygon
        n = len(self)
        assert boxes.shape[0] == n
        packed = PackedPolygons.pack([self], boxes.device, max(n, 1))
        rois = torch.cat([torch.zeros((n, 1), dtype=torch.float32, device=boxes.device), boxes.float()], 1).contiguous()
        idx = torch.arange(n, dtype=torch.int32, device=boxes.device)
        out = mask_targets_polygon(packed, rois, idx, torch.zeros(n, dtype=torch.int32, device=boxes.device), 1, mask_size)
        return out.bool()


class PackedPolygons:
    """device form of a batch's polygon ground truth (PackedBatch.gt_masks for MASK_FORMAT "polygon"): xy fp64 [Vcap, 2] every vertex,
    poly_start int32 [Pcap + 1] vertex ranges, inst_start int32 [B * Mcap + 1] polygon ranges of the flat instances (image b's instance j =
    b * Mcap + j), image_inst0 int32 [B]. Capacities are rounded up (vertices to 2048, polygons to 128) so that batches share shapes
    (= batch keys of a captured / recorded step); the padding is empty ranges."""

    def __init__(self, xy, poly_start, inst_start, image_inst0):
        self.xy, self.poly_start, self.inst_start, self.image_inst0 = xy, poly_start, inst_start, image_inst0

    @staticmethod
    def pack(per_image, device, mcap):
        import numpy as np
        import torch
        verts, pstart, istart = [], [0], [0]
        for pm in per_image:
            assert len(pm) <= mcap
            for j in range(mcap):
                if j < len(pm):
                    for poly in pm.polygons[j]:
                        verts.append(poly.reshape(-1, 2))
                        pstart.append(pstart[-1] + len(poly) // 2)
                istart.append(len(pstart) - 1)
        v = np.concatenate(verts, 0) if verts else np.zeros((0, 2))
        vcap = max(2048, (len(v) + 2047) // 2048 * 2048)
        pcap = max(128, (len(pstart) - 1 + 127) // 128 * 128)
        xy = torch.zeros((vcap, 2), dtype=torch.float64)
        xy[: len(v)] = torch.from_numpy(np.ascontiguousarray(v))
        ps = torch.full((pcap + 1,), pstart[-1], dtype=torch.int32)
        ps[: len(pstart)] = torch.tensor(pstart, dtype=torch.int32)
        return PackedPolygons(xy.to(device), ps.to(device), torch.tensor(istart, dtype=torch.int32).to(device),
                              (torch.arange(len(per_image), dtype=torch.int32) * mcap).to(device))

    @property
    def shape(self):
        return ("polygons", self.xy.shape[0], self.poly_start.shape[0], self.inst_start.shape[0])

    def clone(self):
        return PackedPolygons(self.xy.clone(), self.poly_start.clone(), self.inst_start.clone(), self.image_inst0.clone())

    def copy_(self, other, non_blocking=False):
        for a, b in zip((self.xy, self.poly_start, self.inst_start, self.image_inst0), (other.xy, other.poly_start, other.inst_start, other.image_inst0)):
            a.copy_(b, non_blocking=non_blocking)
        return self
