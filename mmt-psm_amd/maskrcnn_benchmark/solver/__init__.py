from .build import make_optimizer, make_lr_scheduler  # noqa: F401
