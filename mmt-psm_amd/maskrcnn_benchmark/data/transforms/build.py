"""build_transforms (reference: data/transforms/build.py:5-80) over the device transforms"""
from . import transforms as T


def build_transforms(cfg, is_train=True, domain="source"):
    tta = cfg.TEST.TTA
    if is_train:
        min_size, max_size, flip_prob = cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, 0.5
    else:
        min_size, max_size, flip_prob = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, 0
    normalize_transform = T.Normalize(mean=cfg.INPUT.PIXEL_MEAN, std=cfg.INPUT.PIXEL_STD, to_bgr255=cfg.INPUT.TO_BGR255)

    def two_part():  # build.py:22-34 / :58-70
        return [T.Compose([T.Resize(min_size, max_size), T.RandomHorizontalFlip(flip_prob)]),
                T.Compose([T.AdjustBrightness(0.15), T.AdjustContrast(0.15), T.AdjustHue(0.05), T.RandomErasing(0.9),
                           T.ToTensor(), normalize_transform])]

    if is_train and not tta:
        if domain == "no_label":
            return two_part()
        if domain == "source":
            return T.Compose([T.Resize(min_size, max_size), T.RandomHorizontalFlip(flip_prob), T.AdjustBrightness(0.15),
                              T.AdjustContrast(0.15), T.AdjustHue(0.05), T.RandomErasing(0.7), T.ToTensor(),
                              normalize_transform])
        print("domain is invalid, no transform is built")
        return None
    if not tta:
        return T.Compose([T.Resize(min_size, max_size), T.RandomHorizontalFlip(flip_prob), T.ToTensor(), normalize_transform])
    return two_part()
