"""Evaluation dispatch (reference: data/datasets/evaluation/__init__.py:7-34).  Only the PAP metrics are built (SURVEY.md
8f-4); the dataset classes themselves are out of scope (private data), so the dispatch goes by what the dataset
offers -- the three members `prepare_for_pap_segmentation` reads -- instead of by class."""
from .pap import pap_evaluation


def evaluate(dataset, predictions, output_folder, **kwargs):
    if all(hasattr(dataset, a) for a in ("id_to_img_map", "get_ground_truth", "contiguous_category_id_to_json_id", "maxWS")):
        kwargs.setdefault("box_only", False)
        kwargs.setdefault("iou_types", ("segm",))
        return pap_evaluation(dataset=dataset, predictions=predictions, output_folder=output_folder, **kwargs)
    raise NotImplementedError("Unsupported dataset type {}.".format(dataset.__class__.__name__))
