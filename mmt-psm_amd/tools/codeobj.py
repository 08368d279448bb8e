"""Kernel resource usage of the shipped library, read from its gfx950 code objects (no GPU): registers, spills, scratch, LDS.
libmmtpsm.so carries one offload bundle per translation unit back to back in .hip_fatbin; each is unbundled and its AMDGPU
metadata note parsed.  `python codeobj.py [substring]` prints the table; tests/test_kernel_resources.py asserts on it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
LIB = os.path.join(ROOT, "mmt-psm_amd", "libmmtpsm.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib=LIB, workdir=None):
    """-> paths of the gfx950 code objects inside `lib`, one per translation unit"""
    workdir = workdir or tempfile.mkdtemp(prefix="mmtco")
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, s in enumerate(starts):
        part = os.path.join(workdir, "bundle%d.bin" % i)
        open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(workdir, "dev%d.co" % i)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        if os.path.getsize(co):
            out.append(co)
    return out


def kernel_table(lib=LIB, workdir=None):
    """{mangled kernel name: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}}"""
    tab = {}
    for co in code_objects(lib, workdir):
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
        for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
            k = ".agpr_count" + k
            g = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, k).group(1))   # noqa: E731
            name = re.search(r"\.name:\s+(\S+)", k).group(1)
            tab[name] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), vgpr_spill=g("vgpr_spill_count"),
                             sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    return tab


def demangle(names):
    import shutil
    filt = shutil.which("c++filt")
    if filt is None:
        return {n: n for n in names}
    out = subprocess.check_output([filt], input="\n".join(names), text=True)
    return dict(zip(names, out.splitlines()))


if __name__ == "__main__":
    t = kernel_table()
    d = demangle(sorted(t))
    sub = sys.argv[1] if len(sys.argv) > 1 else ""
    print("%d kernels in %s" % (len(t), LIB))
    for n in sorted(t, key=lambda n: d[n]):
        if sub in d[n]:
            r = t[n]
            print("%-110s vgpr %3d agpr %3d sgpr %3d spill v %2d s %3d scratch %4d" % (
                re.sub(r"^void |\(anonymous namespace\)::", "", d[n]).split("(mmtconv")[0][:110], r["vgpr"], r["agpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch"]))
