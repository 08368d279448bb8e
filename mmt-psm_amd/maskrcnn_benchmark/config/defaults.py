"""Configuration object for the hot path.

The reference builds every module from a yacs `CfgNode` (config/defaults.py:21-411, merged with
configs/pap/e2e_mask_rcnn_R_50_FPN_1x.yaml and the KEY VALUE list of scripts/train_mt.sh).  The image has
no yacs, so this is a small attribute-dict with the same surface (`clone`, `freeze`, `merge_from_file`,
`merge_from_list`) and the same KEY names; only keys read on the hot path carry defaults here, any other
key of the reference yaml is accepted and stored.  Values below are the EFFECTIVE ones for the shipped
PAP R-50-FPN mean-teacher recipe, IR-Net off.
"""
import copy


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError("cfg is frozen")
        self[k] = v

    def clone(self):
        c = copy.deepcopy(self)
        c.defrost()
        return c

    def _set_frozen(self, f):
        self.__dict__["_frozen"] = f
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(f)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    @staticmethod
    def _coerce(v):
        if isinstance(v, str):
            s = v.strip()
            if s.startswith("(") and s.endswith(")"):
                try:
                    return tuple(eval(s, {"__builtins__": {}}))
                except Exception:
                    return v
        if isinstance(v, list):
            return tuple(v)
        return v

    def _merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v)

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, lst):
        if len(lst) % 2:
            raise ValueError("merge_from_list expects KEY VALUE pairs")
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = CfgNode()
                node = node[p]
            old = node.get(parts[-1])
            v = self._coerce(v)
            if isinstance(v, str) and old is not None and not isinstance(old, str):
                try:
                    v = type(old)(eval(v, {"__builtins__": {}})) if not isinstance(old, bool) else v in ("True", "true", "1")
                except Exception:
                    pass
            node[parts[-1]] = v


_DEFAULTS = {
    "MODEL": {
        "DEVICE": "cuda", "META_ARCHITECTURE": "GeneralizedRCNN", "MASK_ON": True, "RPN_ONLY": False, "WEIGHT": "",
        "BACKBONE": {"CONV_BODY": "R-50-FPN", "FREEZE_CONV_BODY_AT": 2, "OUT_CHANNELS": 256},
        "RESNETS": {"NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True, "RES2_OUT_CHANNELS": 256,
                    "STEM_OUT_CHANNELS": 64, "TRANS_FUNC": "BottleneckWithFixedBatchNorm",
                    "STEM_FUNC": "StemWithFixedBatchNorm"},
        "RPN": {"USE_FPN": True, "ANCHOR_SIZES": (32, 64, 128, 256, 512), "ANCHOR_STRIDE": (4, 8, 16, 32, 64),
                "ASPECT_RATIOS": (0.5, 1.0, 2.0), "STRADDLE_THRESH": 0, "FG_IOU_THRESHOLD": 0.7,
                "BG_IOU_THRESHOLD": 0.3, "BATCH_SIZE_PER_IMAGE": 256, "POSITIVE_FRACTION": 0.5,
                "PRE_NMS_TOP_N_TRAIN": 2000, "PRE_NMS_TOP_N_TEST": 1000, "POST_NMS_TOP_N_TRAIN": 2000,
                "POST_NMS_TOP_N_TEST": 1000, "NMS_THRESH": 0.7, "MIN_SIZE": 0,
                "FPN_POST_NMS_TOP_N_TRAIN": 2000, "FPN_POST_NMS_TOP_N_TEST": 1000, "RPN_HEAD": "SingleConvRPNHead"},
        "ROI_HEADS": {"USE_FPN": True, "FG_IOU_THRESHOLD": 0.5, "BG_IOU_THRESHOLD": 0.5,
                      "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "BATCH_SIZE_PER_IMAGE": 512,
                      "POSITIVE_FRACTION": 0.25, "SCORE_THRESH": 0.05, "NMS": 0.5, "DETECTIONS_PER_IMG": 200},
        "ROI_BOX_HEAD": {"DO": 0.5, "K_HEAD": 1, "FEATURE_EXTRACTOR": "FPN2MLPFeatureExtractor",
                         "PREDICTOR": "FPNPredictor", "POOLER_RESOLUTION": 7, "POOLER_SAMPLING_RATIO": 2,
                         "POOLER_SCALES": (0.25, 0.125, 0.0625, 0.03125), "NUM_CLASSES": 3, "MLP_HEAD_DIM": 1024},
        "ROI_MASK_HEAD": {"FEATURE_EXTRACTOR": "MaskRCNNFPNFeatureExtractor", "PREDICTOR": "MaskRCNNC4Predictor",
                          "POOLER_RESOLUTION": 14, "POOLER_SAMPLING_RATIO": 2,
                          "POOLER_SCALES": (0.25, 0.125, 0.0625, 0.03125), "CONV_LAYERS": (256, 256, 256, 256),
                          "RESOLUTION": 28, "SHARE_BOX_FEATURE_EXTRACTOR": False, "POSTPROCESS_MASKS": False,
                          "POSTPROCESS_MASKS_THRESHOLD": 0.5},
        # IR-Net (config/defaults.py:247-305 + configs/pap/e2e_mask_rcnn_R_50_FPN_1x.yaml:35-85 + train_mt.sh);
        # the two USE_* switches default to off here (BASELINE configs 1-4), config 5 turns them on
        "RELATION_NMS": {"USE_RELATION_NMS": False, "LOSS": 1.0, "DO": 0.5, "FIRST_N": 90, "THREAD": (0.1,),
                         "ROI_FEAT_DIM": 1024, "APPEARANCE_FEAT_DIM": 128, "GEO_FEAT_DIM": 64, "FC_DIM": (64, 16),
                         "GROUP": 16, "HID_DIM": (1024, 1024, 128), "CLASS_AGNOSTIC": False, "MERGE_METHOD": 0,
                         "FG_THREAD": 0.1, "POS_NMS": 0.55, "CLS_WISE_RELATION": False, "MUTRELATION": False,
                         "TAG": "CIAM", "CONCAT": False, "TOPK": 40, "APPEARANCE_INTER": True, "USE_IOU": False,
                         "IOU_METHOD": "n", "WEIGHT": 1.0, "ALPHA": 0.2, "GAMMA": 1.0, "REG_IOU": True,
                         "REG_IOU_MSK": False, "D_LOSS": 0.0},
        "RELATION_MASK": {"USE_RELATION": False, "BINARY": False, "USE_PRE_FEATURE": False, "PRE_NORM": False,
                          "NORM": -1, "TYPE": "CIAM", "SAME_PREDICTOR": False, "DEEP_SUPER": True, "CAM": False,
                          "CIAM": True, "TRAIN_CENTER_ONLY": False, "PROTO": False, "ALPHA": 0.5, "CENTER_TOPK": 20,
                          "CENTER_PER_CLASS": 8, "APPEARANCE_FEAT_DIM": 128, "GEO_FEAT_DIM": 64, "FC_DIM": (64, 16),
                          "GROUP": 16, "HID_DIM": (1024, 1024), "TOPK": 128, "EXTRACTOR_CHANNEL": 16,
                          "FEATURE_EXTRACTOR": "RoiAlignMaskFeatureExtractor", "RANK": True, "CLSWIZE": True,
                          "XY_COOR": True, "IOU_COOR": False},
    },
    "INPUT": {"MIN_SIZE_TRAIN": 800, "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333,
              "PIXEL_MEAN": [102.9801, 115.9465, 122.7717], "PIXEL_STD": [1.0, 1.0, 1.0], "TO_BGR255": True},
    "TEST": {"TTA": False},
    "DATALOADER": {"SIZE_DIVISIBILITY": 32},
    "DATASETS": {"NO_LABEL": True, "SYN": False},
    "SOLVER": {"BASE_LR": 0.005, "BIAS_LR_FACTOR": 2, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0001,
               "WEIGHT_DECAY_BIAS": 0, "GAMMA": 0.1, "STEPS": (5000,), "MAX_ITER": 7000, "WARMUP_FACTOR": 1.0 / 3,
               "WARMUP_ITERS": 500, "WARMUP_METHOD": "linear", "CHECKPOINT_PERIOD": 50, "IMS_PER_BATCH": 4},
    "MT": {"ALPHA": 0.99, "ALPHA_RAMPUP": 0.99, "LAMBDA": 5.0, "RAMPUP_STEP": 250, "RAMPDOWN_STEP": 250,
           "CLS_LOSS": 0.2, "CLS_LOSS_TYPE": "bce", "TEMP": 0.5, "SHARPEN": True, "FLIP": True, "HARD_NEG": True,
           "CLS_BALANCE_WEIGHT": 1.5, "RANK_FILTER": 0.2, "FG_HINT": 1.0, "T_ADAPT": True, "AUG_K": 2, "AUG_S": 1,
           "START_MT": 1000, "N_STEP_UNLABEL": 1, "HINT": 0.0, "ODKD": False, "FFI": False,
           "RPN_BOOST_ALPHA": 0.5, "REG_LOSS_TYPE": "smooth_l1"},
}


def make_default_cfg():
    return CfgNode(copy.deepcopy(_DEFAULTS))


cfg = make_default_cfg()
