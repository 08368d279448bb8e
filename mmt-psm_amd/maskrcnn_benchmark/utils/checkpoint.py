"""Checkpoint I/O (reference: utils/checkpoint.py:13-205).

Same file format as the reference -- `torch.save({"model": state_dict, "optimizer": ..., "scheduler": ..., **extra})`
plus a `last_checkpoint` tag file in the save directory -- and the same state-dict KEYS, so the `model` entry
of a checkpoint moves between the reference and this build in both directions (the one layout difference inside this build,
fc6's column order, is converted at the state-dict boundary: modeling/roi_heads/box_head/box_head.py).  The `optimizer` entry is
this build's own (FlatSGD: momentum by parameter name), not torch.optim.SGD's state/param_groups: the fork never restores it
anyway (below), and `FlatSGD.load_state_dict` says so when handed the other format.

Loading follows the FORK's `Checkpointer.load` (utils/checkpoint.py:69-117), which differs from upstream maskrcnn-benchmark:
  * `load(f, test=True)` -- what tools/train_mean_teacher.py:42-43 calls for both models: plain suffix-matched load;
  * `load(f)`: a file name containing 'e2e_mask_rcnn_R_50_FPN_1x.pth' means transfer learning (the predictor layers
    cls_score / bbox_pred / mask_fcn_logits keep their fresh initialisation, `iteration` = -1); otherwise the directory's
    `last_checkpoint`, if there is one, overrides `f`.  The fork DELETES the optimizer and scheduler entries before it
    looks for them, so neither is ever restored (and MTtrainer starts at iteration 0, MTtrainer.py:120): reproduced --
    they are dropped here too.  Two crashes of the fork are not reproduced: `load(None)` (TypeError on `'...' in None`)
    and checkpoints without optimizer / scheduler entries (KeyError on `del`) behave as in upstream: "no checkpoint
    found" / entries simply absent.
Not built: Caffe2 .pkl conversion, catalog:// and http:// sources (weight import paths, out of scope by SURVEY 2 row 16);
they raise."""
import logging
import os

import torch

from maskrcnn_benchmark.utils.model_serialization import (align_and_update_state_dicts, load_state_dict,
                                                         refresh_derived, strip_prefix_if_present)

TRANSFER_TAG = "e2e_mask_rcnn_R_50_FPN_1x.pth"


class Checkpointer(object):
    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.save_dir, self.save_to_disk = save_dir, save_to_disk
        self.logger = logger if logger is not None else logging.getLogger(__name__)
        self.transfer_learning = False

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        if self.optimizer is not None:
            data["optimizer"] = self.optimizer.state_dict()
        if self.scheduler is not None:
            data["scheduler"] = self.scheduler.state_dict()
        data.update(kwargs)
        save_file = os.path.join(self.save_dir, "{}.pth".format(name))
        self.logger.info("Saving checkpoint to {}".format(save_file))
        torch.save(data, save_file)
        self.tag_last_checkpoint(save_file)

    def load_extra_data(self, f=None):
        checkpoint = {}
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
        elif TRANSFER_TAG in f:
            self.transfer_learning = True
            checkpoint["iteration"] = -1
        return checkpoint

    def load(self, f=None, test=False):
        if test:
            self.logger.info("Loading checkpoint from {}".format(f))
            checkpoint = self._load_file(f)
            load_state_dict(self.model, checkpoint.pop("model"))
            return checkpoint
        self.transfer_learning = bool(f) and TRANSFER_TAG in f
        if not self.transfer_learning and self.has_checkpoint():
            f = self.get_checkpoint_file()  # an existing checkpoint of this run overrides the argument
        if not f:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        self.logger.info("Loading checkpoint from {}".format(f))
        checkpoint = self._load_file(f)
        checkpoint.pop("scheduler", None)  # the fork drops both before it would restore them (:90)
        checkpoint.pop("optimizer", None)
        self._load_model(checkpoint)
        if self.transfer_learning:
            checkpoint["iteration"] = -1
        return checkpoint

    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint"), "r") as f:
                last_saved = f.read()
        except IOError:  # deleted by another process in the meantime
            last_saved = ""
        return last_saved.strip("\n")

    def tag_last_checkpoint(self, last_filename):
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(last_filename)

    def _load_file(self, f):
        return torch.load(f, map_location=torch.device("cpu"))

    def _load_model(self, checkpoint):
        if not self.transfer_learning:
            load_state_dict(self.model, checkpoint.pop("model"))
            return
        # transfer learning: everything that matches, except the class-count dependent predictors (:148-160)
        pretrained = strip_prefix_if_present(checkpoint.pop("model"), prefix="module.")
        model_state_dict = self.model.state_dict()
        align_and_update_state_dicts(model_state_dict, pretrained)
        model_state_dict = {k: v for k, v in model_state_dict.items()
                            if "cls_score" not in k and "bbox_pred" not in k and "mask_fcn_logits" not in k}
        self.model.load_state_dict(model_state_dict, strict=False)
        refresh_derived(self.model)


class DetectronCheckpointer(Checkpointer):
    def __init__(self, cfg, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=None, logger=None):
        super().__init__(model, optimizer, scheduler, save_dir, save_to_disk, logger)
        self.cfg = cfg.clone()

    def _load_file(self, f):
        if f.startswith("catalog://") or f.startswith("http") or f.endswith(".pkl"):
            raise NotImplementedError("catalog / URL / Caffe2 .pkl weight sources are outside the MI355X hot-path build; "
                                      "convert to a .pth state dict with the reference's tools first: %s" % f)
        loaded = super()._load_file(f)
        if "model" not in loaded:
            loaded = dict(model=loaded)
        return loaded

    def load_optimizer(self, checkpoint):
        self.logger.info("Loading optimizer from ckpt")
        self.optimizer.load_state_dict(torch.load(checkpoint, map_location=torch.device("cpu")).pop("optimizer"))
