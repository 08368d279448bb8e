"""ImageList / to_image_list (reference: structures/image_list.py:7-89): a zero-padded batch tensor plus the
unpadded (H, W) of every image.  hflip() mutates in place exactly like the reference (extract_aug_feat
relies on it, detector/generalized_rcnn.py:201-208)."""
import math

import torch


class ImageList(object):
    def __init__(self, tensors, image_sizes):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)

    def hflip(self):
        self.tensors = torch.flip(self.tensors, (3,))

    def vflip(self):
        self.tensors = torch.flip(self.tensors, (2,))

    def flip(self):
        self.tensors = torch.flip(self.tensors, (2, 3))


def to_image_list(tensors, size_divisible=0):
    if isinstance(tensors, torch.Tensor) and size_divisible > 0:
        tensors = [tensors] if tensors.dim() == 3 else list(tensors)
    if isinstance(tensors, ImageList) or (hasattr(tensors, "tensors") and hasattr(tensors, "image_sizes")):
        return tensors
    if isinstance(tensors, torch.Tensor):
        assert tensors.dim() == 4
        return ImageList(tensors, [tuple(t.shape[-2:]) for t in tensors])
    if isinstance(tensors, (tuple, list)):
        c = tensors[0].shape[0]
        h = max(t.shape[1] for t in tensors)
        w = max(t.shape[2] for t in tensors)
        if size_divisible > 0:
            h = int(math.ceil(h / size_divisible) * size_divisible)
            w = int(math.ceil(w / size_divisible) * size_divisible)
        batch = tensors[0].new_zeros((len(tensors), c, h, w))
        for img, slot in zip(tensors, batch):
            slot[:, :img.shape[1], :img.shape[2]].copy_(img)
        return ImageList(batch, [tuple(t.shape[-2:]) for t in tensors])
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))


def cat_image_list(lists):
    return ImageList(torch.cat([l.tensors for l in lists]), [s for l in lists for s in l.image_sizes])
