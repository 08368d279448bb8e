"""Evaluation loop (reference: engine/inference.py:16-125): eval-mode forward over a data loader, predictions
moved to the host and keyed by image id, gathered over ranks, optionally saved as predictions.pth and handed to an
evaluator.

The reference's PAP metrics (data/datasets/evaluation/pap/pap_eval.py: AJI / F1 / DSC / mAP with `iouIntUni`) are built in
`data/datasets/evaluation/pap/` (round 3; pinned to the reference's own evaluator, tests/test_pap_eval.py): `inference` calls
the `evaluator` argument, else `dataset.evaluate`, else `data.datasets.evaluation.evaluate` when the dataset offers what the
PAP evaluator reads, and otherwise returns the predictions (the dataset classes are private data).  Differences,
on purpose: a failing batch raises instead of being skipped by a bare `except: continue` (:39-40), and predictions of
other ranks travel through `torch.distributed.all_gather_object` instead of a temporary directory (utils/comm.py:81-147)."""
import datetime
import logging
import os
import time

import torch
import torch.distributed as dist


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def is_main_process():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def compute_on_dataset(model, data_loader, device, tta=False):
    """-> {image_id: BoxList on the host}; batches are (images, targets, image_ids) as the collators deliver them"""
    if tta:
        raise NotImplementedError("test-time augmentation is not on the MI355X hot path")
    model.eval()
    results, cpu = {}, torch.device("cpu")
    for images, _targets, image_ids in data_loader:
        with torch.no_grad():
            output = model(images.to(device))
        results.update({i: o.to(cpu) for i, o in zip(image_ids, output)})
    return results


def _accumulate_predictions_from_multiple_gpus(predictions_per_gpu):
    if _world() > 1:
        parts = [None] * _world()
        dist.all_gather_object(parts, predictions_per_gpu)
    else:
        parts = [predictions_per_gpu]
    if not is_main_process():
        return None
    predictions = {}
    for p in parts:
        predictions.update(p)
    ids = sorted(predictions.keys())
    if ids and all(isinstance(i, int) for i in ids) and len(ids) != ids[-1] + 1:
        logging.getLogger("maskrcnn_benchmark.inference").warning(
            "Number of images that were gathered from multiple processes is not a contiguous set. "
            "Some images might be missing from the evaluation")
    return predictions


def inference(model, data_loader, dataset_name, iou_types=("bbox",), box_only=False, device="cuda", expected_results=(),
              expected_results_sigma_tol=4, output_folder=None, generate_data=False, visual_num=0, evaluator=None):
    device = torch.device(device)
    logger = logging.getLogger("maskrcnn_benchmark.inference")
    dataset = getattr(data_loader, "dataset", None)
    n_img = len(dataset) if dataset is not None else None
    logger.info("Start evaluation on {} dataset({} images).".format(dataset_name, n_img))
    t0 = time.time()
    predictions = compute_on_dataset(model, data_loader, device)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    if _world() > 1:
        dist.barrier()
    total = time.time() - t0
    logger.info("Total inference time: {} ({} s / img per device, on {} devices)".format(
        str(datetime.timedelta(seconds=total)), total * _world() / max(n_img or len(predictions), 1), _world()))
    predictions = _accumulate_predictions_from_multiple_gpus(predictions)
    if not is_main_process():
        return None
    if output_folder:
        torch.save(predictions, os.path.join(output_folder, "predictions.pth"))
    evaluate = evaluator if evaluator is not None else getattr(dataset, "evaluate", None)
    if evaluate is None and dataset is not None:
        from maskrcnn_benchmark.data.datasets import evaluation as _ev
        if all(hasattr(dataset, a) for a in ("id_to_img_map", "get_ground_truth", "contiguous_category_id_to_json_id", "maxWS")):
            return _ev.evaluate(dataset, predictions, output_folder, box_only=box_only, iou_types=iou_types, visual_num=visual_num)
    if evaluate is None:
        return predictions
    return evaluate(predictions=predictions, output_folder=output_folder, box_only=box_only, iou_types=iou_types,
                    expected_results=expected_results, expected_results_sigma_tol=expected_results_sigma_tol,
                    visual_num=visual_num)
