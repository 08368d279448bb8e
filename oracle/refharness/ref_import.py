"""Import the REAL reference (Amandaynzhou/MMT-PSM at /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (where /root/reference
exists); never on the GPU box; never imported by the product.

What it does (each step cites the reference defect it works
around):
  1. copies /root/reference/maskrcnn_benchmark to a scratch dir under /tmp (nothing
     of the reference is ever written under /root/repo);
  2. streams csrc/cpu/*.cpp through sed (`x.type()` -> `x.scalar_type()`, D12) and
     builds the reference's own `_C` CPU extension into oracle/_ref/ref_C.so
     (git-ignored build output);
  3. builds the vendored pycocotools C core (pycoco/maskApi.c, D4 stray `*/`
     removed) into the scratch dir;
  4. installs stub modules for packages the image lacks (yacs, cv2, skimage,
     torchvision, D1/D2 phantom modules) and numpy aliases (D12);
  5. patches the three hard-coded `cuda:0` hops (D6, D7) and the eval-mode mask
     list bug (D8) in the scratch copy.

`load_reference()` returns the imported `maskrcnn_benchmark` package and a cfg
builder.
"""
import os
import re
import shutil
import subprocess
import sys
import types

REF = "/root/reference"
SCRATCH = os.environ.get("MMTPSM_REF_SCRATCH", "/tmp/mmtpsm_ref_scratch")
HERE = os.path.dirname(os.path.abspath(__file__))
REF_OUT = os.path.normpath(os.path.join(HERE, "..", "_ref"))


def _sed_file(path, subs):
    with open(path) as f:
        s = f.read()
    for pat, rep in subs:
        s, n = re.subn(pat, rep, s)
        assert n > 0, (path, pat)
    with open(path, "w") as f:
        f.write(s)


def _prepare_scratch():
    marker = os.path.join(SCRATCH, ".ready")
    if os.path.exists(marker):
        return
    if os.path.exists(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    ref_pkg = os.path.join(SCRATCH, "ref")
    os.makedirs(ref_pkg)
    shutil.copytree(os.path.join(REF, "maskrcnn_benchmark"),
                    os.path.join(ref_pkg, "maskrcnn_benchmark"))
    shutil.copytree(os.path.join(REF, "configs"), os.path.join(SCRATCH, "configs"))
    csrc = os.path.join(SCRATCH, "csrc")
    shutil.copytree(os.path.join(REF, "maskrcnn_benchmark", "csrc"), csrc)
    shutil.rmtree(os.path.join(csrc, "cuda"))
    # D12: x.type() no longer converts to ScalarType in AT_DISPATCH
    _sed_file(os.path.join(csrc, "cpu", "ROIAlign_cpu.cpp"),
              [(r"AT_DISPATCH_FLOATING_TYPES\(input\.type\(\)", "AT_DISPATCH_FLOATING_TYPES(input.scalar_type()")])
    _sed_file(os.path.join(csrc, "cpu", "nms_cpu.cpp"),
              [(r"AT_DISPATCH_FLOATING_TYPES\(dets\.type\(\)", "AT_DISPATCH_FLOATING_TYPES(dets.scalar_type()")])
    pkg = os.path.join(ref_pkg, "maskrcnn_benchmark")
    # D6: NMS wrapper pinned to cuda:0
    _sed_file(os.path.join(pkg, "structures", "boxlist_ops.py"),
              [(r"boxes = boxes\.to\('cuda:0'\)", "pass"),
               (r"score = score\.to\('cuda:0'\)", "pass")])
    # D7: cuda:0/cuda:1 hop in both ROI feature extractors
    _sed_file(os.path.join(pkg, "modeling", "roi_heads", "box_head", "roi_box_feature_extractors.py"),
              [(r"device =x\[0\]\.device\.index", "device = 0")])
    _sed_file(os.path.join(pkg, "modeling", "roi_heads", "mask_head", "roi_mask_feature_extractors.py"),
              [(r"device = x\[0\]\.device\.index", "device = 0")])
    # D8: eval-mode mask head hands a tuple to the post-processor
    _sed_file(os.path.join(pkg, "modeling", "roi_heads", "mask_head", "mask_head.py"),
              [(r"mask_logits = mask_logits_1\n", "mask_logits = torch.cat(list(mask_logits_1))\n")])
    # IR-Net (config 5): .type(torch.cuda.FloatTensor)
    _sed_file(os.path.join(pkg, "modeling", "relation", "relation_module.py"),
              [(r"\.type\(torch\.cuda\.FloatTensor\)", ".float()")])
    # D12: add_(1, x) in the trainer
    _sed_file(os.path.join(pkg, "engine", "MTtrainer.py"),
              [(r"add_\(1, param\.data\)", "add_(param.data, alpha=1)"),
               (r"add_\(1 - alpha, param\.data\)", "add_(param.data, alpha=1 - alpha)")])
    # vendored pycocotools
    pyc = os.path.join(SCRATCH, "pyc")
    os.makedirs(os.path.join(pyc, "pycocotools"))
    os.makedirs(os.path.join(SCRATCH, "common"))
    for f in ("_mask.pyx", "mask.py", "maskApi.c", "maskApi.h"):
        shutil.copy(os.path.join(REF, "pycoco", f), os.path.join(pyc, "pycocotools", f))
    open(os.path.join(pyc, "pycocotools", "__init__.py"), "w").write("")
    for f in ("maskApi.c", "maskApi.h"):
        shutil.copy(os.path.join(REF, "pycoco", f), os.path.join(SCRATCH, "common", f))
    for p in (os.path.join(pyc, "pycocotools", "maskApi.c"), os.path.join(SCRATCH, "common", "maskApi.c")):
        lines = open(p).read().split("\n")
        # D4: stray "*/" on line 10 closes nothing
        assert lines[9].strip() == "*/", lines[9]
        del lines[9]
        open(p, "w").write("\n".join(lines))
    setup = (
        "from setuptools import setup, Extension\n"
        "from Cython.Build import cythonize\n"
        "import numpy as np\n"
        "setup(ext_modules=cythonize([Extension('pycocotools._mask', ['pycocotools/_mask.pyx'],\n"
        "      include_dirs=[np.get_include(), 'pycocotools'],\n"
        "      extra_compile_args=['-Wno-cpp', '-Wno-unused-function', '-std=c99'])]))\n")
    open(os.path.join(pyc, "setup.py"), "w").write(setup)
    subprocess.check_call([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=pyc,
                          stdout=subprocess.DEVNULL)
    open(marker, "w").write("ok")


def build_ref_C():
    """Build the reference's own CPU `_C` (nms + roi_align_forward) -> oracle/_ref/ref_C.so."""
    import torch.utils.cpp_extension as ext
    os.makedirs(REF_OUT, exist_ok=True)
    csrc = os.path.join(SCRATCH, "csrc")
    srcs = [os.path.join(csrc, "vision.cpp"),
            os.path.join(csrc, "cpu", "nms_cpu.cpp"),
            os.path.join(csrc, "cpu", "ROIAlign_cpu.cpp")]
    return ext.load("ref_C", srcs, extra_include_paths=[csrc], build_directory=REF_OUT,
                    extra_cflags=["-O2", "-w"], verbose=False)


class CfgNode(dict):
    """Minimal stand-in for yacs.config.CfgNode (the image has no yacs)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    @staticmethod
    def _coerce(v):
        if isinstance(v, str):
            s = v.strip()
            if s.startswith("(") and s.endswith(")"):
                return eval(s)
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
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = self._coerce(v)


def _install_stubs():
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    yacs = mod("yacs")
    yacs.config = mod("yacs.config", CfgNode=CfgNode)
    mod("cv2")
    sk = mod("skimage")
    sk.measure = mod("skimage.measure")
    tv = mod("torchvision")
    tv.models = mod("torchvision.models", VGG=object)
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional")


_CACHE = {}


def load_reference():
    """Returns (maskrcnn_benchmark module, make_cfg(extra_list) -> cfg)."""
    if "mb" in _CACHE:
        return _CACHE["mb"], _CACHE["make_cfg"]
    assert os.path.isdir(REF), "the reference exists only in the build container"
    _prepare_scratch()
    _install_stubs()
    ref_C = build_ref_C()
    sys.path.insert(0, os.path.join(SCRATCH, "pyc"))
    sys.path.insert(0, os.path.join(SCRATCH, "ref"))
    # make sure the PRODUCT's mirror package (same name) is not what gets imported here
    for k in [k for k in sys.modules if k == "maskrcnn_benchmark" or k.startswith("maskrcnn_benchmark.")]:
        del sys.modules[k]
    import maskrcnn_benchmark
    assert maskrcnn_benchmark.__file__.startswith(SCRATCH), maskrcnn_benchmark.__file__
    maskrcnn_benchmark._C = ref_C
    sys.modules["maskrcnn_benchmark._C"] = ref_C
    # D2 / D1 phantom modules
    m = types.ModuleType("maskrcnn_benchmark.utils.cuda_kmeans")
    m.lloyd = None
    sys.modules[m.__name__] = m
    m1 = types.ModuleType("maskrcnn_benchmark.modeling.roi_heads.maskiou_head")
    m2 = types.ModuleType("maskrcnn_benchmark.modeling.roi_heads.maskiou_head.maskiou_head")
    m2.build_roi_maskiou_head = None
    m1.maskiou_head = m2
    sys.modules[m1.__name__] = m1
    sys.modules[m2.__name__] = m2

    def make_cfg(extra=(), relation=False):
        from maskrcnn_benchmark.config import cfg as base
        cfg = base.clone()
        cfg.merge_from_file(os.path.join(SCRATCH, "configs", "pap", "e2e_mask_rcnn_R_50_FPN_1x.yaml"))
        # scripts/train_mt.sh:4-20
        lst = ["MT.CLS_LOSS", 0.2, "MT.FG_HINT", 1.0, "MT.T_ADAPT", True, "MT.SHARPEN", True,
               "MT.TEMP", 0.5, "MT.HARD_NEG", True, "MT.CLS_BALANCE_WEIGHT", 1.5,
               "MT.RANK_FILTER", 0.2, "MT.FLIP", True, "MT.AUG_K", 2, "MT.AUG_S", 1,
               "MT.LAMBDA", 5.0, "MT.START_MT", 1000, "MT.ALPHA", 0.99, "MT.ALPHA_RAMPUP", 0.99,
               "MT.RAMPUP_STEP", 250, "MT.RAMPDOWN_STEP", 250, "MT.CLS_LOSS_TYPE", "bce",
               "MODEL.ROI_BOX_HEAD.DO", 0.5, "MODEL.RELATION_NMS.DO", 0.5,
               "MODEL.RELATION_NMS.REG_IOU", True, "MODEL.RELATION_NMS.REG_IOU_MSK", False,
               "SOLVER.IMS_PER_BATCH", 4, "SOLVER.BASE_LR", 0.005, "MODEL.DEVICE", "cpu"]
        if not relation:
            lst += ["MODEL.RELATION_NMS.USE_RELATION_NMS", False,
                    "MODEL.RELATION_MASK.USE_RELATION", False]
        cfg.merge_from_list(lst + list(extra))
        return cfg

    _CACHE["mb"] = maskrcnn_benchmark
    _CACHE["make_cfg"] = make_cfg
    return maskrcnn_benchmark, make_cfg


if __name__ == "__main__":
    import torch
    mb, make_cfg = load_reference()
    C = mb._C
    # SURVEY.md Appendix C known answers
    keep = C.nms(torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60], [0, 0, 10, 10.5]]),
                 torch.tensor([.5, .9, .3, .8]), 0.5)
    print("nms KAT", keep.tolist())
    x = torch.arange(2 * 3 * 8 * 8).view(2, 3, 8, 8).float()
    o = C.roi_align_forward(x, torch.tensor([[0, 0, 0, 7, 7], [1, 2, 2, 5, 6.5]]), 0.5, 2, 2, 2)
    print("roi KAT", o.flatten()[:6].tolist())
    cfg = make_cfg()
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    torch.manual_seed(0)
    m = build_detection_model(cfg)
    print("params", sum(p.numel() for p in m.parameters()))
