"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU oracle of the input augmentation pipeline, SURVEY.md 8(f) rank 3
(reference: data/transforms/transforms.py:28-205, data/transforms/build.py:20-50, data/datasets/Pap.py:818-830).

Two layers:
  * `pil_*`  -- what the reference executes.  Its transforms call torchvision.transforms.functional on PIL images;
    torchvision is a third-party dependency that is absent from /root/reference and from this image (the reference pins
    no version; the PIL code paths of F.resize / hflip / adjust_brightness / adjust_contrast / adjust_hue / to_tensor /
    normalize have been the same thin wrappers since torchvision 0.2: Image.resize(BILINEAR), transpose(FLIP_LEFT_RIGHT),
    ImageEnhance.Brightness / Contrast .enhance, HSV split + uint8 wrap-around add, /255, (x - mean) / std).  They are
    restated here on top of Pillow itself, which IS installed: the pixel arithmetic is Pillow's own C code.
  * `np_*`   -- plain-numpy restatement of those Pillow algorithms (ImagingResample 8-bit path, ImagingBlend,
    rgb2hsv / hsv2rgb of Convert.c, convert('L')), operation for operation what csrc/augment.hip does.
PINNED: tests/test_oracle_golden.py::test_transforms_* check np_* == pil_* exhaustively for the colour-space maps (all
2^24 triples, both directions) and on random images / factors / sizes for the rest, and both against tests/golden/transforms.npz.
"""
import math

import numpy as np

PIXEL_MEAN = (102.9801, 115.9465, 122.7717)  # config/defaults.py:48 (BGR order after to_bgr255)


# ------------------------------------------------------------------ reference behaviour through Pillow
def pil_resize(img, oh, ow):  # transforms.py:62 F.resize(image, (h, w)) -> Image.resize((w, h), BILINEAR)
    from PIL import Image
    return np.array(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))


def pil_color(img, brightness, contrast, hue):
    """AdjustBrightness -> AdjustContrast -> AdjustHue (build.py:27-31 order), factors already drawn"""
    from PIL import Image, ImageEnhance
    im = Image.fromarray(img, "RGB")
    im = ImageEnhance.Brightness(im).enhance(brightness)   # F.adjust_brightness
    im = ImageEnhance.Contrast(im).enhance(contrast)       # F.adjust_contrast
    h, s, v = im.convert("HSV").split()                    # F.adjust_hue
    nh = np.array(h, dtype=np.uint8)
    nh += np.uint8(hue_shift(hue))                         # `np_h += np.uint8(hue_factor * 255)`: wrap-around add
    im = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
    return np.array(im)


def hue_shift(hue):
    """np.uint8(hue_factor * 255) of F.adjust_hue: C float -> uint8 conversion, i.e. truncation toward zero, modulo 256
    (numpy >= 2 refuses the negative scalar; the numpy of the reference's era wrapped it)"""
    return int(float(hue) * 255) & 255


def erase(img, rects, fills):
    """RandomErasing.eraser (transforms.py:165-192) with the rectangle list and fill arrays already drawn:
    rects [(top, left, h, w)], fills: float arrays (h, w, 3) in [0, 255) -> assigned into the uint8 image (truncation)"""
    out = img.copy()
    for (t, l, h, w), c in zip(rects, fills):
        out[t:t + h, l:l + w, :] = np.asarray(c).astype(np.uint8)  # numpy float -> uint8 assignment truncates
    return out


def to_tensor_normalize(img, mean=PIXEL_MEAN):
    """ToTensor + Normalize(to_bgr255=True, std=1) (transforms.py:84-99): (3,H,W) float32"""
    t = np.transpose(img, (2, 0, 1)).astype(np.float32) / np.float32(255.0)   # F.to_tensor
    t = t[[2, 1, 0]] * np.float32(255.0)
    return (t - np.asarray(mean, np.float32)[:, None, None]) / np.float32(1.0)


# ------------------------------------------------------------------ restated algorithms (what the HIP kernels mirror)
PRECISION_BITS = 32 - 8 - 2  # Resample.c


def resample_coeffs(insz, outsz):
    """precompute_coeffs + normalize_coeffs_8bpc of Pillow's Resample.c for the BILINEAR (triangle) filter"""
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
        k, ww = [], 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            w = 1.0 - abs(a) if abs(a) < 1.0 else 0.0
            k.append(w)
            ww += w
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS)) if v >= 0 else int(-0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, outsz, axis):
    bounds, kk = resample_coeffs(img.shape[axis], outsz)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((outsz,) + src.shape[1:], np.uint8)
    for xx in range(outsz):
        xmin, xmax = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def np_resize(img, oh, ow):  # horizontal pass first, then vertical, 8-bit intermediate (ImagingResample)
    t = img
    if ow != img.shape[1]:
        t = _resample_axis(t, ow, 1)
    if oh != img.shape[0]:
        t = _resample_axis(t, oh, 0)
    return t


def np_blend(deg, img, alpha):
    """ImagingBlend (Blend.c): float arithmetic; plain (UINT8) cast inside [0,1], clipped cast outside"""
    a = np.float32(alpha)
    i1, i2 = deg.astype(np.int32), img.astype(np.int32)
    tmp = (i1.astype(np.float32) + a * (i2 - i1).astype(np.float32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return tmp.astype(np.uint8)
    return np.where(tmp <= 0, 0, np.where(tmp >= 255, 255, tmp.astype(np.int32))).astype(np.uint8)


def np_luma(img):  # convert('L'): ITU-R 601-2, Convert.c L24 macro
    i = img.astype(np.uint32)
    return (i[..., 0] * 19595 + i[..., 1] * 38470 + i[..., 2] * 7471 + 0x8000) >> 16


def np_rgb2hsv(rgb):  # Convert.c rgb2hsv_row: float for the ratios, double for the hue wrap and the x255 scaling
    f32, f64 = np.float32, np.float64
    r, g, b = [rgb[..., i].astype(np.int32) for i in range(3)]
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = cr / maxc.astype(f32)
        rc, gc, bc = (maxc - r).astype(f32) / cr, (maxc - g).astype(f32) / cr, (maxc - b).astype(f32) / cr
        h = np.where(r == maxc, (bc - gc).astype(f64),
                     np.where(g == maxc, 2.0 + rc.astype(f64) - bc, 4.0 + gc.astype(f64) - rc)).astype(f32)
        h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
        uh = np.clip((h.astype(f64) * 255.0).astype(np.int32), 0, 255)
        us = np.clip((s.astype(f64) * 255.0).astype(np.int32), 0, 255)
    gray = maxc == minc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def np_hsv2rgb(hsv):  # Convert.c hsv2rgb_row: double arithmetic, round half up
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    fh = h.astype(np.float64) * 6.0 / 255.0
    fs = s.astype(np.float64) / 255.0
    i = np.floor(fh).astype(np.int32)
    f = fh - i
    mv = v.astype(np.float64)

    def r8(x):
        return np.clip(np.floor(x + 0.5).astype(np.int32), 0, 255)

    p, q, t = r8(mv * (1.0 - fs)), r8(mv * (1.0 - fs * f)), r8(mv * (1.0 - fs * (1.0 - f)))
    vv, i6 = v.astype(np.int32), i % 6
    sel = [i6 == k for k in range(6)]
    r = np.select(sel, [vv, q, p, p, t, vv])
    g = np.select(sel, [t, vv, vv, q, p, p])
    b = np.select(sel, [p, p, t, vv, vv, q])
    gray = s == 0
    return np.stack([np.where(gray, vv, r), np.where(gray, vv, g), np.where(gray, vv, b)], -1).astype(np.uint8)


def np_color(img, brightness, contrast, hue):
    x = np_blend(np.zeros_like(img), img, brightness)
    lum = np_luma(x)
    mean = int(int(lum.sum()) / lum.size + 0.5)          # int(ImageStat.Stat(L).mean[0] + 0.5)
    x = np_blend(np.full_like(x, mean), x, contrast)
    hsv = np_rgb2hsv(x)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift(hue)) & 255
    return np_hsv2rgb(hsv)


def view(img, flip, brightness, contrast, hue, rects, fills, resize_to=None, restated=True):
    """one augmented view of the no_label / source pipeline (build.py:20-50): Resize, flip, colour, erasing, tensor"""
    x = img
    if resize_to is not None:
        x = (np_resize if restated else pil_resize)(x, *resize_to)
    if flip:
        x = x[:, ::-1, :]
    x = (np_color if restated else pil_color)(np.ascontiguousarray(x), brightness, contrast, hue)
    x = erase(x, rects, fills)
    return x, to_tensor_normalize(x)


# ------------------------------------------------------------------ the reference's transform classes, RNG draws included
def get_size(w, h, min_size, max_size):
    """Resize.get_size (transforms.py:34-58) -> (oh, ow)"""
    size = min_size
    if max_size is not None:
        if w == max_size and h == max_size:
            return h, w
        mn, mx = float(min(w, h)), float(max(w, h))
        if mx / mn * size > max_size:
            size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def draw_color(rnd):
    """AdjustBrightness(0.15), AdjustContrast(0.15), AdjustHue(0.05) (transforms.py:120-143): three random.uniform draws"""
    return rnd.uniform(1 - 0.15, 1 + 0.15), rnd.uniform(1 - 0.15, 1 + 0.15), rnd.uniform(-0.05, 0.05)


def draw_erasing(rnd, nprnd, prob, img_h, img_w, img_c=3, s_l=0.001, s_h=0.004, r_1=0.2, r_2=1 / 0.2, v_l=0, v_h=255):
    """RandomErasing.__call__ + eraser (transforms.py:160-205): `random.randint(0, 10)` attempts, numpy draws inside"""
    rects, fills = [], []
    for _ in range(rnd.randint(0, 10)):
        if nprnd.rand() > prob:
            continue
        while True:
            s = nprnd.uniform(s_l, s_h) * img_h * img_w
            r = nprnd.uniform(r_1, r_2)
            w, h = int(np.sqrt(s / r)), int(np.sqrt(s * r))
            left, top = nprnd.randint(0, img_w), nprnd.randint(0, img_h)
            if left + w <= img_w and top + h <= img_h:
                break
        rects.append((top, left, h, w))
        fills.append(nprnd.uniform(v_l, v_h, (h, w, img_c)))
    return rects, fills


def pipeline(img, domain, n_views, min_size, max_size, rnd, nprnd, restated=True):
    """build_transforms(cfg, is_train=True, domain) applied as data/datasets/Pap.py:818-830 does (no_label: Resize + flip
    once, then `n_views` x [colour, erasing, tensor]) or as a labeled sample (source: one view).  `rnd` = the `random`
    module (or random.Random), `nprnd` = numpy.random (or RandomState): the draw ORDER is the reference's.
    -> list of (uint8 HWC view, float32 CHW tensor)"""
    h, w = img.shape[:2]
    oh, ow = get_size(w, h, min_size, max_size)
    size = None if (oh, ow) == (h, w) else (oh, ow)
    flip = rnd.random() < 0.5
    out = []
    for _ in range(n_views if domain == "no_label" else 1):
        b, c, hu = draw_color(rnd)
        rects, fills = draw_erasing(rnd, nprnd, 0.9 if domain == "no_label" else 0.7, oh, ow)
        out.append(view(img, flip, b, c, hu, rects, fills, size, restated))
    return out
