from .box_head import FastRCNNLossComputation, make_roi_box_loss_evaluator, sharpen  # noqa: F401
