"""AJI / F1 / DSC / mAP evaluation of instance masks on the PAP windows (reference: data/datasets/evaluation/pap/
pap_eval.py:20-974).  Host code -- evaluation runs once per checkpoint and is "irrelevant to the throughput
metric" -- restated with the reference's outputs, including the behaviour that only shows on odd inputs:

  * masks are compared through `iouIntUni` (the fork's pycocotools addition, here `mask_rle.iouIntUni`): iou, intersection and
    union of every (detection, ground-truth) pair of an (image, category);
  * AJI (evaluateImg :569-633): ground truths in annotation order each take the unmatched detection of highest IoU >= 0.5
    (the LAST one among equals); matched intersections / unions are summed, the areas of unmatched ground truths and
    detections are added to the union.  "Matched" is recorded as the window's numeric id and tested with `> 0`, so on a
    window whose id is 0 a detection can be taken twice and is never counted as missed (:585, :607-612) -- kept;
  * F1 (compute_F1 :332-425): every ground truth claims its best-IoU detection, contested detections go to the claimant with
    the highest IoU and the losers look again; TP = claims with IoU > 0.5;
  * DSC / TPRp / FNRo / FDRo (caclulateMetrics :427-478): greedy one-to-one matching on the Dice matrix above 0.7;
  * mAP / AP50 / AP75 / AP85 (cal_MAP :480-510, accumulate :706-797): COCO matching at IoU 0.5:0.05:0.95 and the 101-point
    interpolated precision, no area ranges, no crowd / ignore flags.
Detections and ground truths are dicts {"image_id": {"file_name", "location": (x, y), "id"}, "category_id", "segmentation":
RLE, "score"}; a window's key is file_name_x_y (:222-225)."""
import copy
import os
import tempfile
import time
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import mask_rle as maskUtils


def _key(image_id):
    return image_id["file_name"] + "_%d_%d" % (image_id["location"][0], image_id["location"][1])


class Params(object):
    """pap_eval.py:945-974"""

    def __init__(self, iouType="segm"):
        if iouType not in ("segm", "bbox"):
            raise Exception("iouType not supported")
        self.imgIds, self.catIds = [], []
        self.maxDets = [200]
        self.areaRng = [[0 ** 2, 1e5 ** 2]]
        self.useCats = 1
        self.UseIOU = True
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.iouType = iouType


PR_THRESHOLDS = np.linspace(0.2, 0.9, 28)


class Papeval(object):
    def __init__(self, gts, dts, iou_type):
        self.iou_type = iou_type
        self.PapGt, self.PapDt = gts, dts
        self.params = Params(iouType=iou_type)
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for gt in gts:
            self._gts[_key(gt["image_id"]), gt["category_id"]].append(gt)
        for dt in dts:
            self._dts[_key(dt["image_id"]), dt["category_id"]].append(dt)
        self.evalImgs, self.eval = defaultdict(list), {}
        self.createIndex()
        self.params.imgIds = sorted(copy.deepcopy(self.img_id))
        self.params.catIds = sorted(self.catToImgGTs.keys())

    def createIndex(self):
        self.catToImgGTs, self.catToImgDTs = defaultdict(list), defaultdict(list)
        self.img_id = []
        for gt in self.PapGt:
            self.catToImgGTs[gt["category_id"]].append(gt["image_id"])
            self.img_id.append(_key(gt["image_id"]))
        for dt in self.PapDt:
            self.catToImgDTs[dt["category_id"]].append(dt["image_id"])

    # ------------------------------------------------------------------------------------------------ per window
    def _lists(self, imgId, catId):
        p = self.params
        if p.useCats:
            return self._gts[imgId, catId], self._dts[imgId, catId]
        return ([g for c in p.catIds for g in self._gts[imgId, c]], [d for c in p.catIds for d in self._dts[imgId, c]])

    def evaluate(self):
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        if p.useCats:
            p.catIds = list(np.unique(p.catIds))
        p.maxDets = sorted(p.maxDets)
        catIds = p.catIds if p.useCats else [-1]
        self.ious = {(i, c): self.computeIoU(i, c) for i in p.imgIds for c in catIds}
        maxDet = p.maxDets[-1]
        self.evalImgs = [self.evaluateImg(i, c, maxDet) for c in catIds for i in p.imgIds]
        self._paramsEval = copy.deepcopy(self.params)

    def computeIoU(self, imgId, catId):
        """-> (ious, intersection, union, gt_area, dsc), matrices (detections in score order) x (ground truths); ([], [], [])
        for an empty window (pap_eval.py:274-329)"""
        p = self.params
        gt, dt = self._lists(imgId, catId)
        if len(gt) == 0 and len(dt) == 0:
            return [], [], []
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in order][:p.maxDets[-1]]
        if p.iouType != "segm":
            raise Exception("unknown iouType for iou computation")
        g = [x["segmentation"] for x in gt]
        d = [x["segmentation"] for x in dt]
        gt_area = maskUtils.area(g) if len(g) else None
        res = maskUtils.iouIntUni(d, g, [0] * len(g))
        if len(d) == 0 or len(g) == 0:
            merged = maskUtils.merge(copy.deepcopy(d if len(d) else g), intersect=False)
            return [], [], [maskUtils.area(merged)], gt_area, []
        ious, inter, union = res
        inter[ious <= 0] = 0
        dsc = 2 * inter / (union + inter + 1e-10)
        return ious, inter, union, gt_area, dsc

    def compute_F1(self, gt_area, iou, intersection, UseIOU=True):
        """-> (precision list, recall list over 28 thresholds, F1, precision, recall at 0.5) (pap_eval.py:332-425)"""
        D, G = iou.shape
        cols = iou.T.tolist()                                 # per ground truth: its IoU with every detection
        claim = [c.index(max(c)) if max(c) > 0 else -1 for c in cols]
        value = [max(c) for c in cols]
        claimed = set(claim)
        claimed.discard(-1)
        while len(claim) - value.count(0) != len(claimed):
            # a detection claimed by several ground truths stays with the one of highest IoU; the others look again
            rivals, v = [], None
            for v in claimed:
                if claim.count(v) > 1:
                    rivals = [i for i, x in enumerate(claim) if x == v]
                    break
            best = [value[i] for i in rivals]
            del rivals[best.index(max(best))]
            for i in rivals:
                cols[i][v] = 0
                claim[i] = cols[i].index(max(cols[i])) if max(cols[i]) > 0 else -1
                value[i] = max(cols[i])
            claimed = set(claim)
            claimed.discard(-1)
        TP, tp_at = 0, [0] * len(PR_THRESHOLDS)
        for g, d in enumerate(claim):
            if d == -1:
                continue
            v = cols[g][d] if UseIOU else intersection[g, d] / gt_area[g]
            TP += v > 0.5
            for k, t in enumerate(PR_THRESHOLDS):
                tp_at[k] += v > t
        n_gt = len(gt_area)
        plist = [t / (t + (D - t)) for t in tp_at]
        rlist = [t / (t + (n_gt - t)) for t in tp_at]
        FN, FP = n_gt - TP, D - TP
        precision, recall = TP / (TP + FP), TP / (TP + FN)
        F1 = 0 if recall + precision == 0 else 2 * precision * recall / (precision + recall)
        return plist, rlist, F1, precision, recall

    def caclulateMetrics(self, ious, ints, areas, dsc, gt):
        """-> (Dice values of the matched ground truths, their pixel recalls, missed ground truths, unmatched detections):
        greedy one-to-one matching on the Dice matrix above 0.7, consumed in place (pap_eval.py:427-478)"""
        thr = 0.7
        try:
            D, G = ious.shape
        except AttributeError:
            return np.zeros(0), np.zeros(0), 0, 0             # no detections (or no ground truths): nothing counted (sic)
        tpr = ints / areas
        gtdsc, gttpr = np.zeros(G), np.zeros(G)
        while dsc.max() > thr:
            d, g = np.unravel_index(np.argmax(dsc), dsc.shape)
            gtdsc[g], gttpr[g] = dsc[d, g], tpr[d, g]
            dsc[d] = 0
            dsc[:, g] = 0
        hit = gtdsc > thr
        return gtdsc[hit], gttpr[hit], G - np.count_nonzero(gtdsc), D - np.count_nonzero(gtdsc)

    def cal_MAP(self, dt, gt, ious, thr=None):
        """COCO matching at every IoU threshold: detections in score order take the unmatched ground truth of highest IoU
        (the last among equals) (pap_eval.py:480-510)"""
        thr = self.params.iouThrs if thr is None else thr
        T, G, D = len(thr), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        if len(ious):
            for ti, t in enumerate(thr):
                for di in range(D):
                    best, m = min([t, 1 - 1e-10]), -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 or ious[di, gi] < best:
                            continue
                        best, m = ious[di, gi], gi
                    if m >= 0:
                        dtm[ti, di], gtm[ti, m] = m + 1, di + 1
        return dtm, gtm

    def evaluateImg(self, imgId, catId, maxDet):
        gt, dt = self._lists(imgId, catId)
        if len(gt) == 0 and len(dt) == 0:
            return None
        for g in gt:
            g["_ignore"] = 0
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in order[0:maxDet]]
        ious, inter, union, area, dsc = self.ious[imgId, catId]
        if len(gt) and len(dt):
            _, _, F1, _, _ = self.compute_F1(area, ious, inter, self.params.UseIOU)
        elif len(dt):
            F1 = 1                                             # detections on a window without ground truth (sic)
        else:
            F1 = 0
        mdsc, mtpr, FNR, FDR = self.caclulateMetrics(ious, inter, area, dsc, gt)
        ap_dtm, ap_gtm = self.cal_MAP(dt, gt, ious)
        # ---- AJI at IoU 0.5
        G, D = len(gt), len(dt)
        gtm, dtm = -np.ones((1, G)), -np.ones((1, D))
        gtIg = np.array([g["_ignore"] for g in gt])
        if len(ious):
            I = U = 0.0
            for gi, g in enumerate(gt):
                best, m, _i, _u = min([0.5, 1 - 1e-10]), -1, 0, 0
                for di in range(D):
                    if dtm[0, di] > 0 or ious[di, gi] < best:   # `> 0`: a match is stored as the window's numeric id (sic)
                        continue
                    best, _u, _i, m = ious[di, gi], union[di, gi], inter[di, gi], di
                if m == -1:
                    continue
                dtm[0, m] = g["image_id"]["id"]
                gtm[0, gi] = dt[m]["image_id"]["id"]
                I, U = I + _i, U + _u
            U += sum(maskUtils.area(gt[i]["segmentation"]) for i in np.argwhere(gtm == -1)[:, 1])
            U += sum(maskUtils.area(dt[i]["segmentation"]) for i in np.argwhere(dtm == -1)[:, 1])
            with np.errstate(divide="ignore", invalid="ignore"):
                AJI = np.divide(np.full((1, 1), I), np.full((1, 1), float(U)))
            gtIg = np.array([0 for _ in gt])
        else:
            AJI = np.zeros((1, 1))
        return {"image_id": imgId, "category_id": catId, "maxDet": maxDet, "dtMatches": dtm, "gtMatches": gtm,
                "map_dtMatches": ap_dtm, "map_gtMatches": ap_gtm, "dtScores": [d["score"] for d in dt], "AJI": AJI, "F1": F1,
                "DSC": mdsc, "TPRp": mtpr, "FNRo": FNR, "FDR": FDR, "num_G": G, "num_D": D, "gtIg": gtIg}

    # ------------------------------------------------------------------------------------------------ over the set
    def accumulate(self):
        """101-point interpolated precision per (IoU threshold, category) (pap_eval.py:706-797)"""
        p = self.params
        T, R = len(p.iouThrs), len(p.recThrs)
        K = len(p.catIds) if p.useCats else 1
        precision, recall, scores = -np.ones((T, R, K)), -np.ones((T, K)), -np.ones((T, R, K))
        pe = self._paramsEval
        setK, setI = set(pe.catIds if pe.useCats else [-1]), set(pe.imgIds)
        k_list = [n for n, k in enumerate(p.catIds) if k in setK]
        i_list = [n for n, i in enumerate(p.imgIds) if i in setI]
        I0 = len(pe.imgIds)
        for k, k0 in enumerate(k_list):
            E = [self.evalImgs[k0 * I0 + i] for i in i_list]
            E = [e for e in E if e is not None]
            if not E:
                continue
            dtScores = np.concatenate([e["dtScores"] for e in E])
            inds = np.argsort(-dtScores, kind="mergesort")
            sorted_scores = dtScores[inds]
            dtm = np.concatenate([e["map_dtMatches"] for e in E], axis=1)[:, inds]
            npig = np.count_nonzero(np.concatenate([e["gtIg"] for e in E]) == 0)
            tp_sum = np.cumsum(dtm > 0, axis=1).astype(float)
            fp_sum = np.cumsum(np.logical_not(dtm), axis=1).astype(float)
            for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                nd = len(tp)
                with np.errstate(divide="ignore", invalid="ignore"):
                    rc = tp / npig
                pr = (tp / (fp + tp + np.spacing(1))).tolist()
                recall[t, k] = rc[-1] if nd else 0
                for i in range(nd - 1, 0, -1):                 # precision envelope
                    if pr[i] > pr[i - 1]:
                        pr[i - 1] = pr[i]
                q, ss = [0.0] * R, np.zeros(R)
                for ri, pi in enumerate(np.searchsorted(rc, p.recThrs, side="left")):
                    if pi >= nd:                               # recall level never reached: this and all higher ones stay 0
                        break
                    q[ri], ss[ri] = pr[pi], sorted_scores[pi]
                precision[t, :, k], scores[t, :, k] = np.array(q), ss
        self.eval = {"params": p, "counts": [T, R, K], "precision": precision, "recall": recall, "scores": scores}

    def summarize(self):
        """-> self.stats: per category (and 'all' for the AP family) AJI, F1, DSC, TPRP, FNRo, FDRo, mAP, AP50, AP75, AP85
        (pap_eval.py:799-943)"""
        p = self.params

        def ap(cat=None, iouThr=None):
            s = self.eval["precision"]
            if iouThr is not None:
                s = s[np.where(iouThr == p.iouThrs)[0]]
            if cat is not None:
                s = s[:, :, cat]
            return -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])

        names = ("AJI", "F1", "DSC", "TPRP", "FNRo", "FDRo", "mAP", "AP50", "AP75", "AP85")
        st = {n: {} for n in names}
        for ci, cat in enumerate(self._paramsEval.catIds):
            rs = [r for r in self.evalImgs if r is not None and r["category_id"] == cat]
            n = len(rs)
            aji = np.zeros((len(self._paramsEval.iouThrs), 1))
            for r in rs:
                aji = aji + r["AJI"]
            dsc = [v for r in rs for v in list(r["DSC"])]
            tpr = [v for r in rs for v in list(r["TPRp"])]
            st["AJI"][cat] = np.divide(aji, n)
            st["F1"][cat] = sum(r["F1"] for r in rs) / n
            st["DSC"][cat] = sum(dsc) / (len(dsc) + 1e-10)
            st["TPRP"][cat] = sum(tpr) / (len(tpr) + 1e-10)
            st["FNRo"][cat] = sum(r["FNRo"] for r in rs) / sum(r["num_G"] for r in rs)
            st["FDRo"][cat] = sum(r["FDR"] for r in rs) / sum(r["num_D"] for r in rs)
            st["mAP"][cat], st["AP50"][cat], st["AP75"][cat], st["AP85"][cat] = ap(ci), ap(ci, .5), ap(ci, .75), ap(ci, .85)
        st["mAP"]["all"], st["AP50"]["all"], st["AP75"]["all"], st["AP85"]["all"] = ap(), ap(iouThr=.5), ap(iouThr=.75), ap(iouThr=.85)
        self.stats = st


class PapResults(object):
    """pap_eval.py:146-187"""
    METRICS = {"segm": ["AJI", "F1", "DSC", "TPRP", "FNRo", "FDRo", "mAP", "AP50", "AP75", "AP85"]}

    def __init__(self, *iou_types):
        assert all(t in ("segm",) for t in iou_types)
        self.results = OrderedDict((t, OrderedDict((m, -1) for m in self.METRICS[t])) for t in iou_types)

    def update(self, pap_eval):
        if pap_eval is None:
            return
        import numbers
        res = self.results[pap_eval.params.iouType]
        for m in self.METRICS[pap_eval.params.iouType]:
            st = pap_eval.stats[m]
            for k, v in st.items():   # AJI arrives as a (thresholds, 1) array of equal entries: its first one
                if not isinstance(v, (numbers.Number, list)):
                    st[k] = v[0, 0]
            res[m] = st

    def __repr__(self):
        return repr(self.results)


def evaluate_predictions_on_pap(pap_gts, pap_results, json_result_file=None, iou_type="segm"):
    """pap_eval.py:189-203"""
    if json_result_file:
        import json
        with open(json_result_file, "w") as f:
            json.dump(pap_results, f)
    ev = Papeval(pap_gts, pap_results, iou_type)
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    return ev


def prepare_for_pap_segmentation(predictions, dataset):
    """predictions {image index: BoxList with 'mask', 'scores', 'labels'} + the dataset's ground truth -> the evaluator's dict
    lists (pap_eval.py:79-143).  Masks that are not window-sized yet -- the M x M probabilities of MaskPostProcessor, as in a
    predictions.pth -- are pasted first, as the reference does (pap_eval.py:107-109: Masker(threshold=0.5, padding=1)); the
    paste is the device kernel `mmt_paste_mask_stack`, so that case needs the GPU (no host implementation: it raises)."""
    masker = None
    gts, dts = [], []
    for image_id, prediction in predictions.items():
        original_id = dataset.id_to_img_map[image_id]
        if len(prediction) == 0:
            continue
        target = dataset.get_ground_truth(original_id)
        labels = [dataset.contiguous_category_id_to_json_id[i] for i in target.get_field("labels").tolist()]
        boxes = target.bbox.tolist()
        for k, rle in enumerate(target.get_field("masks")):
            gts.append({"image_id": original_id, "category_id": labels[k], "segmentation": rle, "bbox": boxes[k]})
        prediction = prediction.resize((dataset.maxWS, dataset.maxWS))
        masks = prediction.get_field("mask")
        if tuple(masks.shape[-2:]) != (dataset.maxWS, dataset.maxWS):
            if not torch.cuda.is_available():
                raise RuntimeError("predictions carry %dx%d masks and pasting them into the %d-pixel window runs on the GPU "
                                   "(mmt_paste_mask_stack); no MI355X visible" % (masks.shape[-2], masks.shape[-1], dataset.maxWS))
            if masker is None:
                from maskrcnn_benchmark.modeling.roi_heads.mask_head.mask_head import Masker
                masker = Masker(threshold=0.5, padding=1)
            dev = torch.device("cuda", torch.cuda.current_device())
            masks = masker.forward_single_image(masks.to(dev), prediction.to(dev)).cpu()
        scores = prediction.get_field("scores").tolist()
        labels = [dataset.contiguous_category_id_to_json_id[i] for i in prediction.get_field("labels").tolist()]
        boxes = prediction.bbox.tolist()
        for k, m in enumerate(masks):
            rle = maskUtils.encode(np.asarray(m[0].cpu() if hasattr(m, "cpu") else m[0], dtype=np.uint8))
            rle["counts"] = rle["counts"].decode("utf-8")
            dts.append({"image_id": original_id, "category_id": labels[k], "segmentation": rle, "score": scores[k], "bbox": boxes[k]})
    return gts, dts


def do_pap_evaluation(dataset, predictions, output_folder, iou_types, logger, visual_num=0):
    """pap_eval.py:20-47 (the visualisation of :49-77 is out of scope)"""
    logger.info("Preparing results for Pap format")
    pap_gts, pap_results = prepare_for_pap_segmentation(predictions, dataset)
    results = PapResults(*iou_types)
    logger.info("Evaluating predictions")
    for iou_type in iou_types:
        with tempfile.NamedTemporaryFile() as f:
            path = os.path.join(output_folder, iou_type + ".json") if output_folder else f.name
            results.update(evaluate_predictions_on_pap(pap_gts, pap_results, path, iou_type))
    logger.info(results)
    if output_folder:
        torch.save(results, os.path.join(output_folder, "pap_results.pth"))
    return results, pap_results
