"""IR-Net mask relation (reference: modeling/relation/mask_relation_module.py:16-242,
relation_mask_feature_extractor.py:10-48) -- SURVEY.md row a26.

Per image and per class the instances are sorted by objectness; their first-pass mask probability (28x28 ->
max-pooled 14x14) is concatenated to the 256-channel ROI feature, pushed through 3 x conv3x3(256) + conv3x3(16),
mixed across instances by the cross-instance channel attention CIAM, then deconv(16) + 1x1(3) give the second
mask logits.  Parameter names as in the reference (mask_heads.mask.mask_relation_module.*).  The convolutions run
on the MFMA implicit GEMM (the 257-channel input is carried as 272 channels, weights zero-padded on the fly);
CIAM is two batched library GEMMs over <= 128 instances."""
import torch
from torch import nn
import torch.nn.functional as F

from maskrcnn_benchmark.layers import Conv2d, ConvTranspose2d
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

    def forward(self, x):
        n, C, Hh, Ww = x.size()
        cw = x.permute(1, 0, 2, 3).reshape(C, n, -1)
        energy = torch.bmm(cw, cw.permute(0, 2, 1))
        ne = torch.max(energy, -1, keepdim=True)[0] - energy
        att = F.softmax(torch.mean(ne, 0), dim=-1)
        out = torch.mm(att, x.reshape(n, -1)).view(n, C, Hh, Ww)
        return self.gamma * out + x


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
        labels, obj = proposal.get_field("labels"), proposal.get_field("objectness")
        order, cls_len = [], []
        for c in range(self.fg_class):
            idx = torch.nonzero(labels == (c + 1))[:, 0]
            idx = idx[torch.sort(obj[idx], descending=True, stable=True)[1]]
            order.append(idx)
            cls_len.append(int(idx.numel()))
        order = torch.cat(order)
        sorted_mask = mask_logits[order]
        sel = sorted_mask[torch.arange(order.numel(), device=order.device), labels[order]]
        feat = self.appearance_feature_extractor((feat_roi[order], torch.sigmoid(sel)[:, None, :, :]))
        rel = torch.cat([self.relation_module(f) for f in torch.split(feat, cls_len) if f.shape[0] != 0])
        rel = self.deconv_1(rel, relu=True, input_relu=False)
        rel = self.classifier(rel, relu=False, input_relu=True)
        sorted_fields = proposal.copy_with_fields([f for f in proposal.fields() if f != "mask"])[order]
        return rel, [sorted_fields], target, None
