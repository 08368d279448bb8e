"""SURVEY.md 8f-4 (evaluation): the PAP evaluator (AJI / F1 / DSC / TPRp / FNRo / FDRo / mAP / AP50 / AP75 / AP85 with
`iouIntUni`) and its run-length mask codec against outputs of the REFERENCE's own `Papeval` and vendored pycocotools on the
synthetic windows of tests/pap_inputs.py (tests/golden/pap_eval.json, written by `gen_golden.py pap`).  Host code; CPU test."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD
import pap_inputs


@pytest.fixture(scope="module")
def fixture():
    return json.load(open(os.path.join(GOLD, "pap_eval.json")))


@pytest.fixture(scope="module")
def data():
    from maskrcnn_benchmark.data.datasets.evaluation.pap import mask_rle as mu
    gts, dts = pap_inputs.make(7)
    for lst in (gts, dts):
        for x in lst:
            r = mu.encode(x["mask"])
            x["segmentation"] = {"size": r["size"], "counts": r["counts"].decode("ascii")}
    return gts, dts


def test_rle_codec_matches_pycocotools(fixture, data):
    """encode -> the compressed string pycocotools wrote for the same mask, byte for byte; decode inverts it; areas"""
    from maskrcnn_benchmark.data.datasets.evaluation.pap import mask_rle as mu
    gts, dts = data
    allm = gts + dts
    assert [x["segmentation"]["counts"] for x in allm] == fixture["rle_counts"]
    assert [int(mu.area(x["segmentation"])) for x in allm] == fixture["areas"]
    for x in allm[::7]:
        assert np.array_equal(mu.decode(x["segmentation"]), x["mask"])
    both = mu.decode([allm[0]["segmentation"], allm[1]["segmentation"]])
    assert both.shape == (pap_inputs.SIZE, pap_inputs.SIZE, 2)
    u = mu.merge([allm[0]["segmentation"], allm[1]["segmentation"]])
    assert int(mu.area(u)) == int(np.count_nonzero(allm[0]["mask"] | allm[1]["mask"]))
    # uncompressed counts (COCO's list form) decode too; an empty mask
    rl = {"size": [4, 3], "counts": [2, 3, 7]}
    assert mu.decode(rl).flatten(order="F").tolist() == [0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    z = mu.encode(np.zeros((5, 6), np.uint8))
    assert int(mu.area(z)) == 0 and mu.iouIntUni([z], [allm[0]["segmentation"]], [0])[0][0, 0] == 0


def test_iou_int_uni_matches_reference(fixture, data):
    from maskrcnn_benchmark.data.datasets.evaluation.pap import mask_rle as mu
    gts, dts = data
    img, cat = fixture["window"]["key"]
    key = lambda x: x["image_id"]["file_name"] + "_%d_%d" % tuple(x["image_id"]["location"])
    g = [x for x in gts if key(x) == img and x["category_id"] == cat]
    d = sorted([x for x in dts if key(x) == img and x["category_id"] == cat], key=lambda q: -q["score"])
    iou, inter, uni = mu.iouIntUni([x["segmentation"] for x in d], [x["segmentation"] for x in g], [0] * len(g))
    np.testing.assert_allclose(iou, np.array(fixture["window"]["iou"]), rtol=0, atol=1e-15)
    np.testing.assert_array_equal(np.where(iou > 0, inter, 0), np.array(fixture["window"]["inter"]))
    np.testing.assert_array_equal(np.where(iou > 0, uni, 0), np.array(fixture["window"]["union"]))
    assert mu.iouIntUni([], [x["segmentation"] for x in g], []) == []


def test_iouintuni_cells_of_disjoint_boxes_are_zero_by_contract():
    """D14: for a pair whose bounding boxes do not overlap the reference's C code writes iou = 0 and leaves
    the intersection / union cells of its malloc'ed arrays UNWRITTEN (/root/reference/pycoco/maskApi.c:238-259), yet
    caclulateMetrics divides the whole arrays (data/datasets/evaluation/pap/pap_eval.py:425-477): its per-window DSC / TPRp / FNRo /
    FDR depend on what the allocator returned.  The fixture pins the realisation of a fresh process -- zero pages -- and this
    build's iouIntUni returns exactly that BY CONTRACT: intersection 0 and union 0 wherever the boxes are disjoint, whatever ran
    before in the process (fresh arrays, poisoned heap, second call)."""
    from maskrcnn_benchmark.data.datasets.evaluation.pap import mask_rle as mu
    a = np.zeros((40, 40), np.uint8); a[2:8, 3:9] = 1          # three masks with pairwise disjoint boxes ...
    b = np.zeros((40, 40), np.uint8); b[20:30, 22:31] = 1
    c = np.zeros((40, 40), np.uint8); c[33:39, 1:5] = 1
    o = np.zeros((40, 40), np.uint8); o[5:25, 5:25] = 1        # ... and one that overlaps a and b
    e = np.zeros((40, 40), np.uint8)                           # an empty mask: box [0, 0, 0, 0]
    R = [mu.encode(np.asfortranarray(m)) for m in (a, b, c, o, e)]
    junk = [np.full((64, 64), 1e300) for _ in range(8)]        # freed float64 blocks of the result arrays' size class
    del junk
    for _ in range(2):
        iou, inter, uni = mu.iouIntUni(R, R[:4], [0, 0, 0, 0])
        disjoint = np.array([[0, 1, 1, 0], [1, 0, 1, 0], [1, 1, 0, 1], [0, 0, 1, 0], [1, 1, 1, 1]], bool)
        assert (iou[disjoint] == 0).all() and (inter[disjoint] == 0).all() and (uni[disjoint] == 0).all()
        assert inter[0, 0] == 36 and uni[0, 0] == 36 and iou[0, 0] == 1.0
        assert inter[3, 0] == 12 and uni[3, 0] == 36 + 400 - 12 and inter[3, 1] == 15
        assert inter[1, 3] == 15 and uni[1, 3] == 90 + 400 - 15


def test_papeval_statistics_match_reference(fixture, data):
    from maskrcnn_benchmark.data.datasets.evaluation.pap.pap_eval import evaluate_predictions_on_pap, PapResults
    gts, dts = data
    strip = lambda lst: [{k: v for k, v in x.items() if k != "mask"} for x in lst]
    ev = evaluate_predictions_on_pap(strip(gts), strip(dts), None, "segm")
    # per window
    assert len(ev.evalImgs) == len(fixture["per_window"])
    for own, ref in zip(ev.evalImgs, fixture["per_window"]):
        assert (own is None) == (ref is None)
        if ref is None:
            continue
        assert own["image_id"] == ref["image_id"] and own["category_id"] == ref["category_id"]
        assert float(own["AJI"][0, 0]) == pytest.approx(ref["AJI"], rel=1e-12, abs=1e-15)
        assert float(own["F1"]) == pytest.approx(ref["F1"], rel=1e-12)
        assert float(own["FNRo"]) == ref["FNRo"] and float(own["FDR"]) == ref["FDR"]
        np.testing.assert_allclose(np.asarray(own["DSC"], float), np.asarray(ref["DSC"], float), rtol=1e-12, atol=0)
        np.testing.assert_allclose(np.asarray(own["TPRp"], float), np.asarray(ref["TPRp"], float), rtol=1e-12, atol=0)
    assert list(ev.eval["precision"].shape) == fixture["precision_shape"]
    assert float(ev.eval["precision"].sum()) == pytest.approx(fixture["precision_sum"], rel=1e-12)
    np.testing.assert_allclose(ev.eval["recall"], np.array(fixture["recall"]), rtol=1e-12)
    # final statistics
    for m, per in fixture["stats"].items():
        for k, v in per.items():
            kk = k if k == "all" else int(k)
            own = ev.stats[m][kk]
            own = float(np.asarray(own).reshape(-1)[0])
            assert own == pytest.approx(v, rel=1e-12, abs=1e-15), (m, k, own, v)
    res = PapResults("segm")
    res.update(ev)
    assert set(res.results["segm"]) == set(fixture["stats"]) and isinstance(res.results["segm"]["AJI"][1], float)
    assert ev.stats["AJI"][1] > 0.3 and 0 < ev.stats["mAP"]["all"] < 1


def test_dataset_to_evaluator_plumbing(fixture, data):
    """prepare_for_pap_segmentation + the evaluation dispatch on a minimal dataset object: predictions as BoxLists with pasted
    masks, ground truth from `get_ground_truth` -- the statistics are those of the direct call above"""
    import torch
    from maskrcnn_benchmark.data.datasets.evaluation import evaluate
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    gts, dts = data
    keyf = lambda x: x["image_id"]["file_name"] + "_%d_%d" % tuple(x["image_id"]["location"])
    ids = {}
    for x in gts + dts:
        ids.setdefault(keyf(x), x["image_id"])
    order = sorted(ids)

    class DS(object):
        maxWS = pap_inputs.SIZE
        id_to_img_map = {i: ids[k] for i, k in enumerate(order)}
        contiguous_category_id_to_json_id = {1: 1, 2: 2}

        def get_ground_truth(self, original_id):
            g = [x for x in gts if x["image_id"] is original_id or x["image_id"] == original_id]
            b = BoxList(torch.zeros((len(g), 4)), (self.maxWS, self.maxWS), "xyxy")
            b.add_field("labels", torch.tensor([x["category_id"] for x in g], dtype=torch.int64))
            b.add_field("masks", [x["segmentation"] for x in g])
            return b

    preds = {}
    for i, k in enumerate(order):
        d = [x for x in dts if keyf(x) == k]
        b = BoxList(torch.zeros((len(d), 4)), (pap_inputs.SIZE, pap_inputs.SIZE), "xyxy")
        b.add_field("scores", torch.tensor([x["score"] for x in d], dtype=torch.float64))
        b.add_field("labels", torch.tensor([x["category_id"] for x in d], dtype=torch.int64))
        b.add_field("mask", torch.from_numpy(np.stack([x["mask"] for x in d])[:, None]) if d else torch.zeros((0, 1, 96, 96), dtype=torch.uint8))
        preds[i] = b
    results, pap_results = evaluate(DS(), preds, None, iou_types=("segm",), box_only=False)
    # windows whose prediction list is empty are skipped together with their ground truth (pap_eval.py:84-85), so compare on
    # the statistics of a run restricted to the same windows
    from maskrcnn_benchmark.data.datasets.evaluation.pap.pap_eval import evaluate_predictions_on_pap
    kept = {k for i, k in enumerate(order) if len(preds[i])}
    strip = lambda lst: [{q: v for q, v in x.items() if q != "mask"} for x in lst if keyf(x) in kept]
    ref = evaluate_predictions_on_pap(strip(gts), strip(dts), None, "segm")
    for m in ("AJI", "F1", "DSC", "mAP", "AP50"):
        for k, v in ref.stats[m].items():
            assert float(np.asarray(results.results["segm"][m][k]).reshape(-1)[0]) == pytest.approx(float(np.asarray(v).reshape(-1)[0]), rel=1e-9), (m, k)
    assert len(pap_results) == len(strip(dts))
