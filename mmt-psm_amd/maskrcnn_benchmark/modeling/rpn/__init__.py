from .rpn import build_rpn  # noqa: F401
