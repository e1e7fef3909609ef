"""Static check of the compiled kernels for the one hazard hand-issued LDS reads have: hipcc does not know that an
`asm volatile("ds_read ...")` returns its data later, so it may READ the destination registers (a register copy for an in/out
asm operand, a phi move) before the `s_waitcnt lgkmcnt` that covers the read.  For every kernel of the given .s files: walk the
control-flow graph, keep the asm-issued ds_reads that are still in flight (LDS operations retire in order: `lgkmcnt(n)` leaves
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


def _parse(path):
    """-> [(kernel name, [block])], block = {label, ins: [(line no, text, inside asm)], succ: [label | None (fall through)]}."""
    kernels, blocks, cur, in_asm, kernel = [], None, None, False, None
    for no, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\S+):", t)
        if m:
            blocks = []
            kernels.append((m.group(1), blocks))
            cur = {"label": "entry", "ins": []}
            blocks.append(cur)
            continue
        if blocks is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = {"label": m.group(1), "ins": []}
            blocks.append(cur)
            continue
        if not t or t.startswith((";", ".")):
            if t.startswith(".Lfunc_end"):
                blocks = None
            continue
        cur["ins"].append((no, t, in_asm))
        if t.split()[0].startswith(("s_cbranch", "s_branch", "s_endpgm")):      # ends the block (the listing's next one is "; %bb.N")
            cur = {"label": None, "ins": []}
            blocks.append(cur)
    return kernels


def _step(pending, no, t, in_asm, report):
    """One instruction on the queue of LDS operations in flight (oldest first; (asm destination registers, line))."""
    op = t.split()[0]
    body = t[len(op):].split(";")[0]
    if op == "s_waitcnt":
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            pending = () if n == 0 else pending[-n:]
        return pending
    if op == "s_endpgm":
        return ()
    if op.startswith("ds_"):
        # every LDS operation sits in the lgkmcnt queue (in order); only the asm-issued reads are the ones hipcc cannot see
        parts = [p.strip() for p in body.split(",")]
        is_read = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")) or "_rtn" in op
        if not (is_read and in_asm):
            srcs = regs(",".join(parts[1:] if is_read else parts))      # its own sources may touch an in-flight asm destination
            hit = [ln for d, ln in pending if d & srcs]
            if hit:
                report(no, t, hit[0])
        return (pending + ((frozenset(regs(parts[0])) if (is_read and in_asm) else frozenset(), no),))[-64:]
    if op.startswith(("s_load", "s_buffer_load")) or not pending:
        return pending        # scalar loads return out of order: the compiler waits lgkmcnt(0) for them (which only helps here)
    parts = [p.strip() for p in body.split(",")]
    srcs = regs(",".join(parts[1:])) if op.startswith("v_") and not op.startswith("v_cmp") else regs(body)
    if op.startswith(("v_mfma", "v_fmac", "v_mac")):
        srcs |= regs(parts[0])
    hit = [ln for d, ln in pending if d & srcs]
    if hit:
        report(no, t, hit[0])
    return pending


def check(path):
    """Walks every kernel's control-flow graph (a block is re-examined for each distinct queue it can be entered with: loops
    rotated by the compiler enter a body in the middle of the listing, so the listing order says nothing)."""
    bad = set()
    for kernel, blocks in _parse(path):
        index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}

        def report(no, t, first, kernel=kernel):
            if no not in bad:
                bad.add(no)
                print(f"{path}:{no}: {kernel[:70]}: `{t}` reads a register of the asm ds_read at line {first} still in flight")

        seen, work = set(), [(0, ())]
        while work:
            bi, pending = work.pop()
            key = (bi, tuple(ln if d else 0 for d, ln in pending))      # (operations without an asm destination are alike)
            if key in seen:
                continue
            if len(seen) > 2000000:
                raise RuntimeError(f"{kernel}: too many queue states")
            seen.add(key)
            succ = [bi + 1]
            for no, t, in_asm in blocks[bi]["ins"]:
                pending = _step(pending, no, t, in_asm, report)
                op = t.split()[0]
                if op == "s_endpgm":
                    succ = []
                elif op == "s_branch":
                    succ = [index.get(t.split()[1])]
                elif op.startswith("s_cbranch"):
                    # (a conditional branch in the middle of a block does not occur in hipcc's listings: it ends the block)
                    succ = [index.get(t.split()[1]), bi + 1]
            for sidx in succ:
                if sidx is not None and sidx < len(blocks):
                    work.append((sidx, pending))
    return len(bad)


if __name__ == "__main__":
    total = sum(check(p) for p in sys.argv[1:])
    print("in-flight reads found:", total)
    sys.exit(1 if total else 0)
