"""BUILD CONTAINER ONLY (needs /root/reference): re-measures the reference-CPU table of BASELINE.md section 2 with the
protocol of BASELINE.md section 4 -- the REAL reference (imported through ref_import.py), fp32, torch.set_num_threads(8),
1 warm-up + 3 timed calls, median -- on the synthetic inputs of SURVEY.md 8(d) at 1000 x 1000, and writes
profiles/r02_reference_cpu.json.  The reference's CPU path is forward-only (csrc/ROIAlign.h:44).

    python oracle/refharness/time_reference.py [size]
"""
import importlib.util
import json
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from oracle.refharness.ref_import import load_reference  # noqa: E402


def _synth():
    spec = importlib.util.spec_from_file_location("synthetic", os.path.join(ROOT, "mmt-psm_amd", "synthetic.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    threads = 8
    torch.set_num_threads(threads)
    synth = _synth()
    mb, make_cfg = load_reference()
    from maskrcnn_benchmark.modeling.detector import build_detection_model
    from maskrcnn_benchmark.structures.image_list import to_image_list
    from maskrcnn_benchmark.structures.bounding_box import BoxList
    from maskrcnn_benchmark.structures.segmentation_mask import SegmentationMask
    cfg = make_cfg()
    torch.manual_seed(0)
    student = build_detection_model(cfg, is_student=True)
    teacher = build_detection_model(cfg, is_teacher=True)
    sd = synth.make_weights({k: tuple(v.shape) for k, v in student.state_dict().items()}, seed=0)
    student.load_state_dict(sd, strict=False)
    teacher.load_state_dict(sd, strict=False)
    student.train()
    teacher.eval()
    imgs, tgs = synth.make_labeled(2, size, 12, seed=1234)
    unl = synth.make_unlabeled(2, size, 3, seed=4321)
    targets = []
    for t in tgs:
        bl = BoxList(t["boxes"], t["size"], mode="xyxy")
        bl.add_field("labels", t["labels"])
        bl.add_field("masks", SegmentationMask([[p.tolist() for p in inst] for inst in t["polys"]], t["size"], mode="poly"))
        targets.append(bl)

    def sup():  # training-mode forward with autograd recording, as MTtrainer runs it (backward does not exist on CPU)
        return student(to_image_list(list(imgs), 32), targets)

    state = {}

    def teach():
        with torch.no_grad():
            state["tr"] = teacher.forward_teacher([to_image_list(list(u), 32) for u in unl[:2]])

    def stud():
        return student.forward_student([to_image_list(list(unl[-1]), 32)], state["tr"])

    out = {"host": "build container", "threads": threads, "cpu_count": os.cpu_count(), "size": size,
           "protocol": "1 warm-up + 3 timed calls, median; fp32; torch %s" % torch.__version__, "rows": {}}
    for name, fn in (("student_supervised_forward", sup), ("forward_teacher", teach), ("forward_student", stud)):
        fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        out["rows"][name] = {"median_s": round(statistics.median(ts), 3), "all_s": [round(t, 3) for t in ts]}
        print(name, out["rows"][name], flush=True)
    step = sum(r["median_s"] for r in out["rows"].values())
    out["forward_only_step_s"] = round(step, 3)
    out["forward_only_imgs_per_s"] = round(4.0 / step, 4)
    out["backward"] = "n/a: ROIAlign backward is 'Not implemented on the CPU' in the reference (csrc/ROIAlign.h:44)"
    path = os.path.join(ROOT, "profiles", "r02_reference_cpu.json" if size == 1000 else "r02_reference_cpu_%d.json" % size)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
