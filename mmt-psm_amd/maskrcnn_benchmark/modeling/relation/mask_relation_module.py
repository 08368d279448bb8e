"""IR-Net mask relation (reference: modeling/relation/mask_relation_module.py:16-242,
relation_mask_feature_extractor.py:10-48) -- SURVEY.md row a26.

Per image and per class the instances are sorted by objectness; their first-pass mask probability (28x28 ->
max-pooled 14x14) is concatenated to the 256-channel ROI feature, pushed through 3 x conv3x3(256) + conv3x3(16),
mixed across instances by the cross-instance channel attention CIAM, then deconv(16) + 1x1(3) give the second
mask logits.  Parameter names as in the reference (mask_heads.mask.mask_relation_module.*).  The convolutions run
on the MFMA implicit GEMM (the 257-channel input is carried as 272 channels, weights zero-padded on the fly);
CIAM is one launch forward, two backward (csrc/relation.hip: mmt_ciam_fwd / mmt_ciam_bwd) and nothing else: CPU tensors and
batches beyond the kernel's capacity raise; the formulation with two batched library GEMMs it is tested against lives in
tests/tensor_formulations.py."""
import torch
from torch import nn
import torch.nn.functional as F

from maskrcnn_benchmark import _hip as _H
from maskrcnn_benchmark.layers import Conv2d, ConvTranspose2d, fused
from maskrcnn_benchmark.structures.boxlist_ops import cat_boxlist



class RoiAlignMaskFeatureExtractor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg.MODEL.RELATION_MASK.EXTRACTOR_CHANNEL == 1:
            raise NotImplementedError("EXTRACTOR_CHANNEL=1 variant is not the shipped configuration")
        self.mask_fcn1 = Conv2d(257, 256, 3, 1, 1)
        self.mask_fcn2 = Conv2d(256, 256, 3, 1, 1)
        self.mask_fcn3 = Conv2d(256, 256, 3, 1, 1)
        self.conv5_mask = Conv2d(256, cfg.MODEL.RELATION_MASK.EXTRACTOR_CHANNEL, 3, 1, 1)
        for l in (self.mask_fcn1, self.mask_fcn2, self.mask_fcn3, self.conv5_mask):
            nn.init.kaiming_normal_(l.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(l.bias, 0)

    def forward(self, x):
        x, mask = x
        from maskrcnn_benchmark.layers import fused
        pool = F.max_pool2d(mask, kernel_size=2, stride=2)
        # 257 -> 272 channels (a multiple of 16: the DMA-fed split-bf16 kernel takes the layer; zero channels add zeros)
        x = torch.cat((x, pool, pool.new_zeros((pool.shape[0], 15, pool.shape[2], pool.shape[3]))), 1)
        w1 = F.pad(self.mask_fcn1.weight, (0, 0, 0, 0, 0, 15))
        x = fused.conv(x, w1.contiguous(memory_format=torch.channels_last), self.mask_fcn1.bias, 1, 1, True, False)
        x = self.mask_fcn2(x, relu=True, input_relu=True)
        x = self.mask_fcn3(x, relu=True, input_relu=True)
        # the ReLU after conv5_mask feeds CIAM (library ops, not fused nodes) -> a plain elementwise ReLU here
        return F.relu(self.conv5_mask(x, relu=False, input_relu=True))


class CIAM_Module(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg.MODEL.RELATION_MASK.NORM in (1, 2) or cfg.MODEL.RELATION_MASK.PRE_NORM:
            raise NotImplementedError("CIAM NORM / PRE_NORM variants are off in the shipped configuration")
        self.gamma = nn.Parameter(torch.zeros(1))
        self.topk = cfg.MODEL.RELATION_MASK.TOPK

    def forward(self, x, group=None):
        """x (n, C, H, W).  `group` (n,) int: the attention runs inside every group of equal ids separately -- what the
        reference gets by calling the module once per (image, class) slice (mask_relation_module.py:82-92), here as ONE pass
        with the cross-group entries masked out (exp(-inf) = 0 exactly: the same softmax over the same values), so that no
        group size ever has to reach the host."""
        n, C, Hh, Ww = x.size()
        if group is None:
            group = torch.zeros((n,), dtype=torch.int64, device=x.device)     # one group: the reference's per-slice call
        if not (x.is_cuda and 0 < n <= _H.CIAM_MAX_N and C <= 16 and 4 * (C * Hh * Ww + (C + 1) * n) <= 65536):
            # (the kernel's LDS: the instance's feature + its energy rows)
            raise RuntimeError("CIAM: GPU tensors, 1..%d instances per batch, <= 16 channels (csrc/relation.hip is the only "
                               "implementation)" % _H.CIAM_MAX_N)
        # one forward launch, two backward launches for all groups of the batch (csrc/relation.hip); `group` is sorted by
        # the caller (forward_batch orders the instances by (image, class)): equal ids are contiguous
        return fused.CIAMFn.apply(x, group, self.gamma)


class MaskRelationRefineNet(nn.Module):
    def __init__(self, cfg, predictor=None):
        super().__init__()
        self.cfg = cfg.clone()
        rm = cfg.MODEL.RELATION_MASK
        self.fg_class = cfg.MODEL.ROI_BOX_HEAD.NUM_CLASSES - 1
        self.appearance_feature_extractor = RoiAlignMaskFeatureExtractor(cfg)
        ch = rm.EXTRACTOR_CHANNEL
        self.classifier = Conv2d(ch, 3, 1, 1, 0)
        self.deconv_1 = ConvTranspose2d(ch, ch, 2, 2, 0)
        if rm.TYPE != "CIAM" or rm.SAME_PREDICTOR:
            raise NotImplementedError("mask relation: only TYPE 'CIAM' with its own predictor is built")
        self.relation_module = CIAM_Module(cfg)

    def forward(self, x):
        """(ROI features (P,256,14,14), first mask logits (P,3,28,28), BoxList, target) of ONE image
        -> (second mask logits in class-sorted order, [sorted BoxList], target, None)"""
        feat_roi, mask_logits, proposal, target = x
        rel, props = self.forward_batch(feat_roi, mask_logits, [proposal])
        return rel, props, target, None

    def forward_batch(self, feat_roi, mask_logits, proposals):
        """all images of the batch in one pass (the reference loops over images, mask_head.py:98-127): instances are
        ordered by (image, class, objectness descending), the appearance extractor runs once over all of them and CIAM mixes
        inside every (image, class) group.  No tensor size depends on device data: nothing is read back.
        -> (second logits in sorted order, [sorted BoxList per image])"""
        sizes = [len(p) for p in proposals]
        labels = torch.cat([p.get_field("labels") for p in proposals]) if len(proposals) > 1 else proposals[0].get_field("labels")
        obj = torch.cat([p.get_field("objectness") for p in proposals]) if len(proposals) > 1 else proposals[0].get_field("objectness")
        dev = labels.device
        img = torch.cat([torch.full((n,), i, dtype=torch.int64, device=dev) for i, n in enumerate(sizes)]) \
            if len(sizes) > 1 else torch.zeros((sizes[0],), dtype=torch.int64, device=dev)
        group = img * (self.fg_class + 1) + labels
        # stable sort by objectness (descending), then stable by group: = per (image, class) `idx[sort(obj[idx], desc, stable)]`
        o1 = torch.sort(obj, descending=True, stable=True)[1]
        order = o1[torch.sort(group[o1], stable=True)[1]]
        sorted_mask = mask_logits[order]
        sel = sorted_mask[torch.arange(order.numel(), device=dev), labels[order]]
        feat = self.appearance_feature_extractor((feat_roi[order], torch.sigmoid(sel)[:, None, :, :]))
        rel = self.relation_module(feat, group[order])
        rel = self.deconv_1(rel, relu=True, input_relu=False)
        rel = self.classifier(rel, relu=False, input_relu=True)
        out, st = [], 0
        for p, n in zip(proposals, sizes):
            oi = order[st:st + n] - st      # the image's instances stay together (group is image-major)
            out.append(p.copy_with_fields([f for f in p.fields() if f != "mask"])[oi])
            st += n
        return rel, out


