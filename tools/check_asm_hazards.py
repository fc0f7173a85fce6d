"""Build-time check for kernels whose LDS reads / MFMAs are inline assembly (the compiler does not know that their
results arrive later): scans the ISA of the named kernels for
  (1) an instruction that reads (or overwrites) the destination of an inline-asm ds_read before the next
      s_waitcnt lgkmcnt that covers it -- e.g. a register copy the allocator placed right behind the read
      (per basic block: reads that are waited for in a later block are not followed);
  (2) a non-MFMA instruction touching the destination of an inline-asm MFMA within the MFMA's latency (N instructions).
Usage: check_asm_hazards.py file.s kernel_substring [...]      exit code 1 on a finding."""
import re
import sys


def regs(tok, cls):
    out = set()
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]" % cls, tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\b%s(\d+)\b" % cls, tok):
        out.add(int(m.group(1)))
    return out


def kernel_bodies(text, names):
    cur, body = None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if any(n in m.group(1) for n in names) else None
            body = []
        if cur:
            body.append(line)
            if "s_endpgm" in line:
                yield cur, body
                cur = None


def check(name, lines, mfma_window=10):
    bad = []
    pending = []          # [(regs, line_no, outstanding_after)] of asm ds_reads not yet covered by a wait
    in_asm = False
    recent_mfma = []      # [(dst regs (cls, set), countdown)]
    for i, raw in enumerate(lines):
        t = raw.strip()
        if "ASMSTART" in t:
            in_asm = True
            continue
        if "ASMEND" in t:
            in_asm = False
            continue
        if t.endswith(":") or (t.startswith(".LBB") and ":" in t):
            pending, recent_mfma = [], []          # block boundary: the scan is per basic block (layout is not linear)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        parts = t.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        rest = rest.split(";")[0]
        if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm"):
            pending, recent_mfma = [], []
            continue
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        touched_v = regs(rest, "v")
        if op.startswith("ds_read") and in_asm:
            dst = regs(rest.split(",")[0], "v")
            pending.append((dst, i))
            continue
        if op.startswith("ds_") or op.startswith("s_load"):
            pending.append((set(), i))   # any other LGKM operation occupies a counter slot
            continue
        for dst, at in pending:
            if dst & touched_v:
                bad.append(f"{name}: line {i}: `{t[:80]}` touches v{sorted(dst & touched_v)[0]} of the LDS read at "
                           f"line {at} before a wait")
        if op.startswith("v_mfma"):
            if in_asm:
                d0 = rest.split(",")[0]
                recent_mfma.append([regs(d0, "v"), regs(d0, "a"), mfma_window])
            continue
        if op.startswith("s_nop"):
            k = int(rest.strip() or 0) + 1
            for r in recent_mfma:
                r[2] -= max(1, k // 4)
        touched_a = regs(rest, "a")
        for r in recent_mfma:
            if (r[0] & touched_v) or (r[1] & touched_a):
                bad.append(f"{name}: line {i}: `{t[:80]}` touches the destination of an MFMA issued <= {mfma_window} "
                           f"instructions earlier")
            r[2] -= 1
        recent_mfma = [r for r in recent_mfma if r[2] > 0]
    return bad


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    findings = []
    n = 0
    for name, body in kernel_bodies(text, sys.argv[2:]):
        n += 1
        findings += check(name, body)
    for f in findings[:40]:
        print(f)
    print(f"{n} kernels checked, {len(findings)} findings")
    sys.exit(1 if findings or not n else 0)
