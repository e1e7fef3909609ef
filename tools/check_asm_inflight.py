"""Static check of the compiled kernels for the one hazard hand-issued LDS reads have: hipcc does not know that an
`asm volatile("ds_read ...")` returns its data later, so it may READ the destination registers (a register copy for an in/out
asm operand, a phi move) before the `s_waitcnt lgkmcnt` that covers the read.  For every kernel of the given .s files: walk the
instructions in order, keep the asm-issued ds_reads that are still in flight (LDS operations retire in order: `lgkmcnt(n)` leaves
the n youngest), and report any instruction whose SOURCE operands touch a register one of them will write.
    hipcc -S --cuda-device-only ... -o x.s ; python tools/check_asm_inflight.py x.s"""
import re
import sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def check(path):
    bad = 0
    kernel, in_asm, pending = None, False, []          # pending: (dest regs, line no) of asm ds_reads in flight, oldest first
    lgkm_others = 0
    for no, line in enumerate(open(path), 1):
        t = line.strip()
        if re.match(r"^_Z\S+:", t):
            kernel, pending = t.split(":")[0], []
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".")):
            if t.startswith(".LBB") or t.startswith(".Lfunc_end"):
                pass
            continue
        op = t.split()[0]
        body = t[len(op):].split(";")[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                pending = [] if n == 0 else pending[-n:]
            continue
        if op in ("s_endpgm",):
            pending = []
            continue
        if op.startswith("ds_"):
            # every LDS operation sits in the lgkmcnt queue (in order); only the asm-issued reads are the ones hipcc cannot see
            parts = [p.strip() for p in body.split(",")]
            is_read = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")) or "_rtn" in op
            pending.append((regs(parts[0]) if (is_read and in_asm) else set(), no))
            if not (is_read and in_asm):
                # its own sources may still touch an in-flight asm destination
                srcs = regs(",".join(parts[1:] if is_read else parts))
                hit = [ln for d, ln in pending[:-1] if d & srcs]
                if hit:
                    bad += 1
                    print(f"{path}:{no}: {(kernel or '?')[:70]}: `{t}` reads a register of the asm ds_read at line {hit[0]} still in flight")
            continue
        if op.startswith(("s_load", "s_buffer_load")):
            continue          # scalar loads return out of order: the compiler waits lgkmcnt(0) for them (which only helps here)
        if not pending:
            continue
        parts = [p.strip() for p in body.split(",")]
        srcs = regs(",".join(parts[1:])) if op.startswith("v_") and not op.startswith("v_cmp") else regs(body)
        if op.startswith(("v_mfma", "v_fmac", "v_mac")):
            srcs |= regs(parts[0])
        hit = [ln for d, ln in pending if d & srcs]
        if hit:
            bad += 1
            print(f"{path}:{no}: {(kernel or '?')[:70]}: `{t}` reads a register of the asm ds_read at line {hit[0]} still in flight")
    return bad


if __name__ == "__main__":
    total = sum(check(p) for p in sys.argv[1:])
    print("in-flight reads found:", total)
    sys.exit(1 if total else 0)
