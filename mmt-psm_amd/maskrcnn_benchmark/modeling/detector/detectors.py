from .generalized_rcnn import GeneralizedRCNN


def build_detection_model(cfg, is_teacher=False, is_student=False):
    """detector/detectors.py:5-7"""
    if cfg.MODEL.META_ARCHITECTURE != "GeneralizedRCNN":
        raise KeyError(cfg.MODEL.META_ARCHITECTURE)
    return GeneralizedRCNN(cfg, is_teacher, is_student)
