"""Instruction mix of the loops of a gfx950 kernel, read from hipcc's assembly (runs on CPU: no GPU needed).

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S wgrad.hip -o /tmp/wgrad.s
    python tools/isa_loop_mix.py /tmp/wgrad.s wgrad3_kernelILi64ELi64ELi4ELb0

For every loop (a backward branch to a label) that holds MFMAs it prints the count of matrix, vector-ALU, scalar-ALU, LDS,
LDS-DMA / vector-memory instructions, waits and barriers in ONE trip of the loop body, and the vector + scalar
instructions per MFMA -- the figure the round-4 "instruction diet" of the weight-gradient and packed-rows kernels
was steered by: a v_mfma_f32_32x32x16_bf16 hides about five single-issue instructions, a 16x16x32 about two.
"""
import re
import sys
from collections import Counter


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op in ("s_waitcnt", "s_waitcnt_vscnt"):
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op in ("s_nop", "s_sleep", "s_setprio"):
        return "nop"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and False):
        return "dma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def kernel_body(lines, pattern):
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.split(";")[0].rstrip().endswith(":") and pattern in l:
            start = i
            break
    if start is None:
        raise SystemExit(f"no kernel matching {pattern!r}")
    end = start
    for i in range(start + 1, len(lines)):
        if lines[i].lstrip().startswith("s_endpgm"):
            end = i
    # the last s_endpgm before the next function
    for i in range(start + 1, len(lines)):
        if lines[i].startswith(".Lfunc_end"):
            end = i
            break
    return lines[start:end]


def main():
    path, pattern = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    body = kernel_body(lines, pattern)
    labels = {}
    insts = []          # (index in body, opcode, text)
    for i, l in enumerate(body):
        s = l.strip()
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        insts.append((i, op, s))
    print(f"{body[0].split(':')[0]}: {len(insts)} instructions")
    loops = []
    for k, (_, op, s) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= k:
                loops.append((labels[tgt], k, tgt))
    for a, b, tgt in sorted(loops, key=lambda t: t[1] - t[0]):
        c = Counter(classify(op) for _, op, _ in insts[a:b + 1])
        if not c["mfma"]:
            continue
        glds = sum(1 for _, op, s in insts[a:b + 1] if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in s))
        vm = c["vmem"] + c["dma"]
        per = (c["valu"] + c["salu"]) / c["mfma"]
        print(f"  loop {tgt} [{b - a + 1} insts]: mfma {c['mfma']}  valu {c['valu']}  salu {c['salu']}  lds {c['lds']}  "
              f"vmem {vm} (lds-dma {glds})  smem {c['smem']}  wait {c['wait']}  barrier {c['barrier']}  branch {c['branch']}  nop {c['nop']}"
              f"   -> {c['valu'] / c['mfma']:.2f} valu + {c['salu'] / c['mfma']:.2f} salu per mfma ({per:.2f})")
        if len(sys.argv) > 3 and sys.argv[3] == "-v":
            top = Counter(op for _, op, _ in insts[a:b + 1] if classify(op) in ("valu", "salu"))
            print("     ", ", ".join(f"{o} {n}" for o, n in top.most_common(24)))


if __name__ == "__main__":
    main()
