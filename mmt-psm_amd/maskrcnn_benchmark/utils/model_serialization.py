"""State-dict loading with suffix matching (reference: utils/model_serialization.py:10-80).

`load_state_dict(model, loaded)` gives every key of the model the loaded tensor whose name is the LONGEST suffix of it
(so `backbone.body.layer1.0.conv1.weight` takes `layer1.0.conv1.weight` over `conv1.weight`), after removing a
`module.` prefix that DataParallel / DistributedDataParallel put on every key; keys without a match keep the model's
own value; the result is loaded strictly.  For a model on flat parameter storage (engine/flat.py) the copy lands in the
flat buffer and the packed bf16 weight planes are refreshed."""
import logging
from collections import OrderedDict

import torch


def align_and_update_state_dicts(model_state_dict, loaded_state_dict):
    current, loaded = sorted(model_state_dict.keys()), sorted(loaded_state_dict.keys())
    logger = logging.getLogger(__name__)
    width = max([len(k) for k in current], default=1)
    width_l = max([len(k) for k in loaded], default=1)
    for key in current:
        # longest loaded name that is a suffix of this key; among equally long candidates there can be only one (it is
        # the same string).  First maximum in sorted order, as the reference's match-matrix argmax picks it.
        best, best_len = None, 0
        for cand in loaded:
            if len(cand) > best_len and key.endswith(cand):
                best, best_len = cand, len(cand)
        if best is None:
            continue
        model_state_dict[key] = loaded_state_dict[best]
        logger.info("{: <{}} loaded from {: <{}} of shape {}".format(key, width, best, width_l,
                                                                    tuple(loaded_state_dict[best].shape)))


def strip_prefix_if_present(state_dict, prefix):
    keys = sorted(state_dict.keys())
    if not all(k.startswith(prefix) for k in keys):
        return state_dict
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k.replace(prefix, "")] = v  # every occurrence, as the reference's str.replace does
    return out


def load_state_dict(model, loaded_state_dict):
    model_state_dict = model.state_dict()
    loaded_state_dict = strip_prefix_if_present(loaded_state_dict, prefix="module.")
    align_and_update_state_dicts(model_state_dict, loaded_state_dict)
    model.load_state_dict(model_state_dict)  # strict
    refresh_derived(model)


def refresh_derived(model):
    """after parameters were rewritten: re-pack the bf16 weight planes of a flattened model (engine/flat.py)"""
    m = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
    flat = getattr(m, "_flat", None)
    if flat is not None:
        flat.refresh_planes()
