from .transforms import (Compose, Resize, RandomHorizontalFlip, ToTensor, Normalize, AdjustBrightness, AdjustContrast,
                         AdjustHue, RandomErasing, DeviceImage)
from .build import build_transforms

__all__ = ["Compose", "Resize", "RandomHorizontalFlip", "ToTensor", "Normalize", "AdjustBrightness", "AdjustContrast",
           "AdjustHue", "RandomErasing", "DeviceImage", "build_transforms"]
