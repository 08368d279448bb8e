from .defaults import cfg, CfgNode, make_default_cfg  # noqa: F401
