"""DtoidNet -- the detector object scripts/online_learning.py drives (reference: models/dtoid/__init__.py:23-260).

Same constructor argument (a config with .model.{img_h,img_w,heatmap_h,heatmap_w,lam_*,...}), same methods
(forward(dict)->dict, forwardTestTime(dict)->dict, clearCache, load_pretrained_state_dict, configure_optimizers),
same dict keys in and out, same state_dict keys ("model.<submodule>...", :32). It is a plain nn.Module: the
LightningModule plumbing of the reference (training_step / logging / wandb) belongs to offline pre-training, which is
out of scope.

Differences that matter on the GPU: the template-feature cache stays in HBM (the reference moves it to the host and
back on every frame, :107-115); the three input-range asserts that force host syncs (:181-183) are folded into one
optional check; IoU metrics are computed with tensor ops (pl.metrics is not a dependency).
"""
import torch
import torch.nn as nn

from ..hostutil import to_np  # noqa: F401  (re-exported for callers that use the reference's helper)
from . import ops
from .loss import DetectionLoss
from .network import BBoxTransform, ClipBoxes, Network

_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


_NORM_CONST = {}


def normalizeImageRange(img):
    """ImageNet mean/std normalisation of [B,3,H,W] in [0,1] (reference: utils/__init__.py:33-39). The two constant
    vectors are created once per (device, dtype): no host->device copy per call, so the op is graph-capturable."""
    key = (img.device, img.dtype)
    if key not in _NORM_CONST:
        _NORM_CONST[key] = (torch.tensor(_MEAN, dtype=img.dtype, device=img.device).view(1, 3, 1, 1),
                            torch.tensor(_STD, dtype=img.dtype, device=img.device).view(1, 3, 1, 1))
    mean, std = _NORM_CONST[key]
    return (img - mean) / std


def binary_iou(pred, target):
    """IoU of the foreground class of two boolean masks, per leading item; empty union counts as 0
    (pl.metrics.functional.classification.iou(..., ignore_index=0) on a 2-class problem)."""
    p, t = pred.bool().flatten(1), target.bool().flatten(1)
    inter = (p & t).sum(1).float()
    union = (p | t).sum(1).float()
    return torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(inter))


def non_max_sup(classifications, boxes, iou_thresh=0.5, score_thresh=0.05):
    """Per-image, per-foreground-class NMS of raw head output (reference: models/dtoid/utils.py:5-47)."""
    B, N, C = classifications.shape
    scores_out, class_out, box_out = [], [], []
    for b in range(B):
        s_l, c_l, b_l = [], [], []
        for c in range(1, C):
            sc = classifications[b, :, c]
            sel = sc > score_thresh
            if not bool(sel.any()):
                continue
            sc, bx = sc[sel], boxes[b, sel]
            keep = ops.nms(bx, sc, iou_thresh)
            s_l.append(sc[keep])
            c_l.append(torch.full((keep.numel(),), float(c), device=sc.device))
            b_l.append(bx[keep])
        empty = classifications.new_zeros(0)
        scores_out.append(torch.cat(s_l) if s_l else empty)
        class_out.append(torch.cat(c_l) if c_l else empty)
        box_out.append(torch.cat(b_l) if b_l else empty)
    return scores_out, class_out, box_out


class DtoidNet(nn.Module):
    TEMPLATE_CHUNK = 120   # templates per compute_template_local call (reference :92)
    TOP_K = 500            # boxes kept at test time (reference :117)

    def __init__(self, config):
        super().__init__()
        self.cfg = config.model
        self.img_size = (self.cfg.img_h, self.cfg.img_w)
        self.heatmap_size = (self.cfg.heatmap_h, self.cfg.heatmap_w)
        self.model = Network(img_size=self.img_size, heatmap_size=self.heatmap_size)
        if getattr(self.cfg, "use_pretrained_dtoid", False):
            ckpt = torch.load(self.cfg.pretrained_dtoid_path, map_location="cpu")
            self.load_pretrained_state_dict(ckpt["state_dict"])
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()
        self.det_loss_func = DetectionLoss()
        self.center_loss_func = nn.L1Loss()
        self.seg_loss_func = nn.BCELoss()
        self.check_input_range = False          # the reference's asserts (:181-183); each one is a host sync
        self.template_feature_cache = {}

    @property
    def device(self):
        return next(self.parameters()).device

    def load_pretrained_state_dict(self, state_dict):
        self.model.load_state_dict(state_dict)

    def clearCache(self):
        self.template_feature_cache = {}

    def _template_features(self, input, obj_id, device):
        """(local feature chunks, [global feature]) of an object's templates; computed once, kept on the device."""
        if obj_id not in self.template_feature_cache:
            template = normalizeImageRange(input["limg"][0])
            template = torch.cat([template, input["lmask"][0]], dim=1)             # [n_t,4,h,w]
            with torch.no_grad():
                glob = [self.model.compute_template_global(template[0:1])]
                local = [self.model.compute_template_local(c) for c in torch.split(template, self.TEMPLATE_CHUNK, 0)]
            self.template_feature_cache[obj_id] = (local, glob)
        local, glob = self.template_feature_cache[obj_id]
        if local[0].device != device:
            local, glob = [t.to(device) for t in local], [t.to(device) for t in glob]
            self.template_feature_cache[obj_id] = (local, glob)
        return local, glob

    def forwardTestTime(self, input):
        image = input["img"]
        assert len(image) == 1, "test time handles one image and one object at a time (reference :64)"
        raw = image.is_cuda                     # on the GPU normalizeImageRange is fused into the stem's gather (D1)
        if not raw:
            image = normalizeImageRange(image)
        obj_id = int(input["obj_id"][0])
        local, glob = self._template_features(input, obj_id, image.device)
        with torch.no_grad():
            scores, boxes, tids, seg, heat = self.model.forward_all_templates(image, local, glob, topk=self.TOP_K,
                                                                               seg_sigmoid=True, raw_image=raw)   # :147
            if "template_z_values" in input and getattr(self.cfg, "filter_z", False):
                z = input["template_z_values"].to(boxes.device)[0, tids[:, 0].long()]
                size = torch.maximum(boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1])
                pred_z = (124.0 / size) * -z
                ok = torch.nonzero((pred_z > 0.4) & (pred_z < 2)).flatten()
                if ok.numel() == 0:
                    ok = torch.zeros(1, dtype=torch.long, device=boxes.device)
                scores, boxes, tids, seg, heat = scores[ok], boxes[ok], tids[ok], seg[ok], heat[ok]
            tids = tids[:, 0]
        out = {"pred_bbox": boxes, "pred_scores": scores, "pred_template_ids": tids, "segmentation": seg.unsqueeze(1),
               "heat_map": heat.unsqueeze(1), "final_bbox": [boxes], "final_score": [scores]}
        if "heatmap" in input:
            iou = binary_iou((seg[0] > 0.5)[None], input["mask"][0, 0][None] > 0)[0]
            out["seg_IoU"] = iou
            out["seg_IoU_50"] = (iou > 0.5).float()
        return out

    def forwardTestTimeBatch(self, input):
        """ADDITIVE API (BASELINE configs[2] "batch=32 ... with 21 templates"): forwardTestTime for a
        batch of images `img [B,3,H,W]` that all look for the SAME object (`limg [1,n_t,3,h,w]`, `lmask`, `obj_id` as in
        forwardTestTime). The reference asserts B = 1 (models/dtoid/__init__.py:64) and would be called B times; here the
        image backbone runs once on the whole batch. Returns a list of B dicts with forwardTestTime's keys."""
        image = input["img"]
        raw = image.is_cuda
        if not raw:
            image = normalizeImageRange(image)
        obj_id = int(input["obj_id"][0])
        local, glob = self._template_features(input, obj_id, image.device)
        res = self.model.forward_all_templates_batch(image, local, glob, topk=self.TOP_K, seg_sigmoid=True, raw_image=raw)
        outs = []
        for scores, boxes, tids, seg, heat in res:
            outs.append({"pred_bbox": boxes, "pred_scores": scores, "pred_template_ids": tids[:, 0],
                         "segmentation": seg.unsqueeze(1), "heat_map": heat.unsqueeze(1), "final_bbox": [boxes],
                         "final_score": [scores]})
        return outs

    def forward(self, input):
        image, template, template_mask = input["img"], input["limg"], input["lmask"]
        global_template, global_template_mask = input["gimg"], input["gmask"]
        if self.check_input_range:
            for t in (image, template, global_template):
                assert float(t.max()) <= 1 and float(t.min()) >= 0
        image_n = normalizeImageRange(image)
        cls, reg, anchors, heat_map, seg_logit = self.model(
            image_n, normalizeImageRange(template), template_mask, normalizeImageRange(global_template),
            global_template_mask)
        fused_tail = seg_logit.is_cuda and "heatmap" in input
        if fused_tail:
            # sigmoid + BCE + per-image IoU in one pass (loss.SegBceIou); the clipped boxes -- an output for metrics / NMS, not
            # part of any loss (reference :206-208) -- by the decode kernel of the test-time path instead of ~15 torch ops
            from .loss import SegBceIou
            segmentation, loss_seg, iou = SegBceIou.apply(seg_logit, input["mask"])
            boxes = ops.decode_clip_boxes(anchors, reg, image_n.shape[3], image_n.shape[2])
        else:
            segmentation = torch.sigmoid(seg_logit)
            boxes = self.clipBoxes(self.regressBoxes(anchors, reg), image_n)
        out = {"classifications": cls, "regressions": reg, "anchors": anchors, "heat_map": heat_map,
               "segmentation": segmentation, "transformed_anchors": boxes}
        if "heatmap" in input:
            dev = cls.device
            loss_cls, loss_reg = self.det_loss_func(cls, reg, anchors, input["bbox_gt"].to(dev))
            loss_center = self.center_loss_func(input["heatmap"].to(dev), heat_map)   # float64 target promotes (:214)
            if not fused_tail:
                loss_seg = self.seg_loss_func(segmentation, input["mask"].to(dev))
            out["loss_seg"] = self.cfg.lam_seg * loss_seg
            out["loss_center"] = self.cfg.lam_center * loss_center
            out["loss_cls"] = self.cfg.lam_cls * loss_cls
            out["loss_reg"] = self.cfg.lam_reg * loss_reg
            out["loss"] = out["loss_seg"] + out["loss_center"] + out["loss_cls"] + out["loss_reg"]
            with torch.no_grad():
                if not fused_tail:
                    iou = binary_iou(segmentation.detach()[:, 0] > 0.5, input["mask"].to(dev)[:, 0] > 0)
                out["seg_IoU"] = iou.mean()
                out["seg_IoU_50"] = (iou > 0.5).float().mean()
                if getattr(self, "compute_train_nms", False):   # metrics-only pass (:235); off the finetune hot path
                    fs, fc, fb = non_max_sup(cls.detach(), boxes.detach())
                    out["final_score"], out["final_class"], out["final_bbox"] = fs, fc, fb
        return out

    def configure_optimizers(self):
        opt = torch.optim.Adam(self.parameters(), lr=self.cfg.learning_rate, weight_decay=self.cfg.weight_decay,
                               amsgrad=True)
        return [opt], [torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[20, 40], gamma=0.1)]


class DtoidConfig:
    """Minimal stand-in for the OmegaConf object of dtoid_conf_{lmo,ycbv}.yaml (conf/model/dtoid.yaml,
    conf/dataset/dtoid_bop.yaml:24-27): cfg.model.* with the reference's defaults."""

    class _NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    def __init__(self, img_h=480, img_w=640, heatmap_h=29, heatmap_w=39, **kw):
        m = dict(name="dtoid", lam_seg=20, lam_center=20, lam_cls=1, lam_reg=1, learning_rate=1e-4, weight_decay=1e-6,
                 nms_iou_thresh=0.5, img_h=img_h, img_w=img_w, heatmap_h=heatmap_h, heatmap_w=heatmap_w, filter_z=False,
                 valid_all_templates=False, use_pretrained_dtoid=False, pretrained_dtoid_path="")
        m.update(kw)
        self.model = DtoidConfig._NS(**m)
