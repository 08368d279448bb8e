"""The row-resident 1x1 kernel (csrc/conv_igemm.hip: conv1x1_rows_kernel) waits for the LDS-DMA copies of the next panel's
weight planes with a COUNTED `s_waitcnt vmcnt(n)`: n = the vector memory operations issued after those copies, which may
stay in flight (its residual / mask requests and its stores).  The copies are inline assembly the compiler does not see, so
nothing but this count orders the LDS reads of the next panel behind them.  A count larger than the operations really issued
would let a panel be multiplied before its planes have landed -- and only under load.  This test reads the count back from
the code object inside the shipped library (no GPU needed): every instantiation, every marked wait."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mmt-psm_amd", "libmmtpsm.so")
LLVM = "/opt/rocm/lib/llvm/bin"

VM_OP = re.compile(r"^\s*(buffer_(load|store|atomic)\w*|global_(load|store|atomic)\w*|flat_(load|store|atomic)\w*|scratch_(load|store)\w*)\s")


def disassemble(tmp_path):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM tools not present")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, LIB])
    subprocess.check_call([tools[1], "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.check_output([tools[2], "-d", "--no-show-raw-insn", co], text=True)


def parse_kernels(text):
    """{name: [(address, instruction text, branch target address or None)]} of the row-resident kernel's instantiations"""
    kernels, cur, base = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            cur = m.group(2) if "conv1x1_rows_kernel" in m.group(2) else None
            base = int(m.group(1), 16)
            if cur:
                kernels[cur] = []
            continue
        if not cur or not line.strip():
            continue
        m = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        addr = int(m.group(1), 16)
        ins = re.sub(r"//.*", "", line).strip()
        tgt = None
        if ins.startswith(("s_branch", "s_cbranch")):
            t = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", line)
            tgt = base + int(t.group(1), 16) if t else base
        kernels[cur].append((addr, ins, tgt))
    return kernels


def test_counted_waits_of_the_rows_kernel(tmp_path):
    kernels = parse_kernels(disassemble(tmp_path))
    assert len(kernels) >= 9, sorted(kernels)
    for name, ins in kernels.items():
        masked = "Lb1ELb1EEEv" in name   # <KT, NS, F16 = true, MASK = true>
        index = {a: k for k, (a, _, _) in enumerate(ins)}
        preds = [[] for _ in ins]
        for k, (_, text, tgt) in enumerate(ins):
            if text.startswith("s_endpgm"):
                continue
            if tgt is not None:
                preds[index[tgt]].append(k)
            if not text.startswith("s_branch") and k + 1 < len(ins):
                preds[k + 1].append(k)
        marks = [k for k in range(1, len(ins)) if ins[k][1].startswith("s_setprio 0") and ins[k - 1][1].startswith("s_waitcnt vmcnt(")]
        # the loop is unrolled by two; on the fp16 split (round 6) a panel ends in one of two waits -- with or without the eight
        # stores of the row-blocked output planes (a uniform branch on mmt_conv_args.y_rb), each with its own count
        f16 = "Lb1ELb" in name   # <KT, NS, F16 = true, MASK>
        counts = sorted(int(re.search(r"vmcnt\((\d+)\)", ins[k - 1][1]).group(1)) for k in marks)
        base = 50 if masked else 34
        assert counts == ([base, base, base + 8, base + 8] if f16 else [base, base]), (name, counts)
        for k in marks:
            n = int(re.search(r"vmcnt\((\d+)\)", ins[k - 1][1]).group(1))
            # Program order is fixed by the source (volatile assembly and compiler barriers): barrier, copies, MFMAs, epilogue,
            # wait.  Walk the control flow graph backwards from the wait to the barrier(s) it can be reached from and count the
            # vector memory operations on the way (the copies themselves excluded): every path must give exactly n.
            seen, stack, ends = {}, [(k - 1, 0)], []
            while stack:
                j, ops = stack.pop()
                text = ins[j][1]
                if text.startswith("s_barrier"):
                    ends.append(ops)
                    continue
                if VM_OP.match(text + " ") and not text.startswith("global_load_lds"):
                    ops += 1
                assert ops <= n, (name, "more operations than allowed in flight before a barrier is reached", ops)
                if j in seen:
                    assert seen[j] == ops, (name, "paths with different counts", seen[j], ops)
                    continue
                seen[j] = ops
                assert preds[j], (name, "reached the kernel entry without a barrier")
                for q in preds[j]:
                    stack.append((q, ops))
            assert ends and all(e == n for e in ends), (name, ends, n)
