"""Input transforms on the device (reference: data/transforms/transforms.py:10-205) -- same class names, same
`transform(image, target) -> (image, target)` protocol, same random draws (`random` / `numpy.random`, in the reference's
order, so a seeded run augments exactly like the reference), but the image is a `DeviceImage`: uint8 RGB pixels in HBM plus
the colour operations recorded so far.  Nothing is computed until `Normalize`, which runs the whole chain
flip -> brightness -> contrast -> hue -> erasing -> to_tensor -> BGR*255 - mean as ONE pass over the pixels
(`mmt_aug_views` + `mmt_aug_erase`), bit-exact with the Pillow / torchvision arithmetic the reference uses.
`augment_views` does the K views of an unlabeled image (data/datasets/Pap.py:818-830) in one launch."""
import math
import random

import numpy as np
import torch

from maskrcnn_benchmark import _hip as H


class DeviceImage(object):
    """uint8 (H, W, 3) RGB tensor on the GPU + pending operations; `.size` is (width, height) like PIL's"""

    def __init__(self, pixels):
        if isinstance(pixels, np.ndarray):
            pixels = torch.from_numpy(np.ascontiguousarray(pixels))
        if pixels.dtype != torch.uint8 or pixels.dim() != 3 or pixels.shape[2] != 3:
            raise ValueError("DeviceImage wants uint8 (H, W, 3) RGB pixels")
        self.pixels = pixels.cuda().contiguous() if not pixels.is_cuda else pixels.contiguous()
        self.flip = False
        self.brightness, self.contrast, self.hue = None, None, None
        self.rects, self.fills = [], []
        self.as_tensor = False

    @property
    def size(self):
        return self.pixels.shape[1], self.pixels.shape[0]

    def clone(self):  # copy.deepcopy of the PIL image in Pap.py:826: pixels are immutable here, share them
        c = DeviceImage.__new__(DeviceImage)
        c.__dict__.update(self.__dict__)
        c.rects, c.fills = list(self.rects), list(self.fills)
        return c

    __deepcopy__ = lambda self, memo: self.clone()

    def _pending_colour(self):
        return self.brightness is not None or self.contrast is not None or self.hue is not None or self.rects


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target=None):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join("\n    {0}".format(t) for t in self.transforms) + "\n)"


_COEFFS = {}


def _resample_tables(insz, outsz, device):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c (BILINEAR / triangle filter), cached per size"""
    key = (insz, outsz, str(device))
    if key not in _COEFFS:
        scale = insz / outsz
        fscale = max(scale, 1.0)
        support = 1.0 * fscale
        ksize = int(math.ceil(support)) * 2 + 1
        bounds = np.zeros((outsz, 2), np.int32)
        kk = np.zeros((outsz, ksize), np.int32)
        for xx in range(outsz):
            center = (xx + 0.5) * scale
            ss = 1.0 / fscale
            xmin = max(int(center - support + 0.5), 0)
            xmax = min(int(center + support + 0.5), insz) - xmin
            k = []
            for x in range(xmax):
                a = (x + xmin - center + 0.5) * ss
                k.append(1.0 - abs(a) if abs(a) < 1.0 else 0.0)
            ww = sum(k)
            if ww != 0.0:
                k = [v / ww for v in k]
            for x, v in enumerate(k):
                kk[xx, x] = int(0.5 + v * (1 << 22)) if v >= 0 else int(-0.5 + v * (1 << 22))
            bounds[xx] = (xmin, xmax)
        _COEFFS[key] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device))
    return _COEFFS[key]


class Resize(object):
    def __init__(self, min_size, max_size):
        self.min_size, self.max_size = min_size, max_size

    def get_size(self, image_size):  # transforms.py:34-58
        w, h = image_size
        size, max_size = self.min_size, self.max_size
        if max_size is not None:
            if w == max_size and h == max_size:
                return (h, w)
            mn, mx = float(min((w, h))), float(max((w, h)))
            if mx / mn * size > max_size:
                size = int(round(max_size * mn / mx))
        if (w <= h and w == size) or (h <= w and h == size):
            return (h, w)
        if w < h:
            return (int(size * h / w), size)
        return (size, int(size * w / h))

    def __call__(self, image, target):
        if image._pending_colour() or image.flip:
            raise NotImplementedError("Resize comes first in the reference's pipelines (build.py:20-50)")
        oh, ow = self.get_size(image.size)
        w, h = image.size
        px = image.pixels
        if ow != w:  # Pillow: horizontal pass first, 8-bit intermediate
            px = H.resample_u8(px, ow, True, *_resample_tables(w, ow, px.device))
        if oh != h:
            px = H.resample_u8(px, oh, False, *_resample_tables(h, oh, px.device))
        if px is not image.pixels:
            image = DeviceImage(px)
        if target is not None:
            target = target.resize(image.size)
        return image, target


class RandomHorizontalFlip(object):
    def __init__(self, prob=0.5):
        self.prob = prob

    def __call__(self, image, target):
        if random.random() < self.prob:
            if image._pending_colour():
                raise NotImplementedError("the flip precedes the colour transforms in the reference's pipelines")
            image = image.clone()
            image.flip = not image.flip
            if target is not None:
                target = target.transpose(0)
        return image, target


def _set_once(image, name, value):
    if getattr(image, name) is not None:
        raise NotImplementedError("one %s adjustment per pipeline (build.py:27-31)" % name)
    image = image.clone()
    setattr(image, name, value)
    return image


class AdjustBrightness(object):
    def __init__(self, bf):
        self.bf = bf

    def __call__(self, img, target):
        if img.contrast is not None or img.hue is not None or img.rects:
            raise NotImplementedError("order of the reference: brightness, contrast, hue, erasing")
        return _set_once(img, "brightness", random.uniform(1 - self.bf, 1 + self.bf)), target


class AdjustContrast(object):
    def __init__(self, cf):
        self.cf = cf

    def __call__(self, img, target):
        if img.hue is not None or img.rects:
            raise NotImplementedError("order of the reference: brightness, contrast, hue, erasing")
        return _set_once(img, "contrast", random.uniform(1 - self.cf, 1 + self.cf)), target


class AdjustHue(object):
    def __init__(self, hue):
        self.hue = hue

    def __call__(self, img, target):
        if img.rects:
            raise NotImplementedError("order of the reference: brightness, contrast, hue, erasing")
        return _set_once(img, "hue", random.uniform(-self.hue, self.hue)), target


class RandomErasing(object):
    """transforms.py:146-205: up to 10 attempts, each erasing (probability `prob`) a small rectangle with per-pixel
    uniform noise.  Positions and noise are drawn on the host with numpy exactly as the reference does."""

    def __init__(self, prob, s_l=0.001, s_h=0.004, r_1=0.2, r_2=1 / 0.2, v_l=0, v_h=255):
        self.prob = prob
        self.p = (s_l, s_h, r_1, r_2, v_l, v_h)

    def __call__(self, img, target):
        s_l, s_h, r_1, r_2, v_l, v_h = self.p
        img_w, img_h = img.size
        img = img.clone()
        for _ in range(random.randint(0, 10)):
            if np.random.rand() > self.prob:
                continue
            while True:
                s = np.random.uniform(s_l, s_h) * img_h * img_w
                r = np.random.uniform(r_1, r_2)
                w, h = int(np.sqrt(s / r)), int(np.sqrt(s * r))
                left, top = np.random.randint(0, img_w), np.random.randint(0, img_h)
                if left + w <= img_w and top + h <= img_h:
                    break
            img.rects.append((top, left, h, w))
            img.fills.append(np.random.uniform(v_l, v_h, (h, w, 3)).astype(np.uint8))  # assignment into a uint8 array
        return img, target


class ToTensor(object):
    def __call__(self, image, target):
        image = image.clone()
        image.as_tensor = True
        return image, target


def _materialize(images, mean, size_divisible=0):
    """images: DeviceImages sharing the same pixels and flip (the K views of one sample) -> fp32 (V,3,Hp,Wp) tensor in
    NHWC memory (channels_last), zero-padded to `size_divisible`"""
    base = images[0]
    if any(i.pixels is not base.pixels or i.flip != base.flip for i in images):
        raise ValueError("views of one call must share the base image and the flip")
    dev = base.pixels.device
    Hh, Ww = base.pixels.shape[:2]
    Hp, Wp = Hh, Ww
    if size_divisible > 0:
        Hp, Wp = int(math.ceil(Hh / size_divisible) * size_divisible), int(math.ceil(Ww / size_divisible) * size_divisible)
    V = len(images)
    out = torch.zeros((V, Hp, Wp, 3), dtype=torch.float32, device=dev)
    # Pillow's enhance() takes python floats and casts to C float; F.adjust_hue casts hue*255 to uint8 (C wrap-around)
    params = np.array([[1.0 if i.brightness is None else i.brightness, 1.0 if i.contrast is None else i.contrast]
                       for i in images], dtype=np.float32)
    hue = np.array([0 if i.hue is None else int(float(i.hue) * 255) & 255 for i in images], dtype=np.int32)
    p = torch.from_numpy(params).to(dev)
    # rectangles may overlap and the reference writes them one after the other (later wins): round j of the erase launches
    # takes the j-th rectangle of every view, so rectangles of one view are never in flight together
    rounds, fills, off = [], [], 0
    for v, i in enumerate(images):
        j = 0
        for (t, l, h, w), f in zip(i.rects, i.fills):
            if h * w == 0:
                continue
            while len(rounds) <= j:
                rounds.append(([], []))
            rounds[j][0].append((v, t, l, h, w))
            rounds[j][1].append(off)
            fills.append(np.ascontiguousarray(f).reshape(-1))
            off += h * w * 3
            j += 1
    plain = [i.brightness is None and i.contrast is None and i.hue is None for i in images]
    if any(i.brightness is None or i.contrast is None or i.hue is None for i, pl in zip(images, plain) if not pl):
        raise NotImplementedError("either the full colour chain (brightness, contrast, hue) or none (test-time pipeline)")
    hue = np.where(np.array(plain), -1, hue).astype(np.int32)
    H.aug_views(base.pixels, base.flip, p[:, 0].contiguous(), p[:, 1].contiguous(), torch.from_numpy(hue).to(dev), mean, out)
    if rounds:
        fill_dev = torch.from_numpy(np.concatenate(fills)).to(dev)
        for rects, offs in rounds:
            H.aug_erase(out, torch.tensor(rects, dtype=torch.int32, device=dev),
                        torch.tensor(offs, dtype=torch.int64, device=dev), fill_dev, mean)
    return out.permute(0, 3, 1, 2)


class Normalize(object):
    def __init__(self, mean, std, to_bgr255=True):
        if not to_bgr255 or any(float(s) != 1.0 for s in std):
            raise NotImplementedError("the path's configuration: TO_BGR255 True, PIXEL_STD 1 (config/defaults.py:48-52)")
        self.mean, self.std, self.to_bgr255 = mean, std, to_bgr255

    def __call__(self, image, target):
        if not image.as_tensor:
            raise RuntimeError("Normalize follows ToTensor")
        return _materialize([image], self.mean)[0], target


def augment_views(base, transform, k, size_divisible=0):
    """the K views of an unlabeled sample (Pap.py:824-829) with ONE pair of launches: `transform` is the second part of
    build_transforms(domain='no_label'); its random draws happen view by view in the reference's order"""
    ops = transform.transforms
    if not ops or not isinstance(ops[-1], Normalize):
        raise ValueError("augment_views wants a Compose ending in Normalize")
    pending = []
    for _ in range(k):
        img = base.clone()
        for t in ops[:-1]:
            img, _ = t(img, None)
        pending.append(img)
    return _materialize(pending, ops[-1].mean, size_divisible)
