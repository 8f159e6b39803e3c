#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc -S --cuda-device-only) for the one thing the compiler cannot know about hand-placed
`s_waitcnt lgkmcnt(N)`: an inline-asm `ds_read` returns its data LATER, so until the counted wait that covers it its destination registers must
not be read, written, copied or spilled by anything else.  Per kernel: every instruction that touches a register with an LDS read still pending.
(Found in round 6: built for 256 registers, htsat_attn_big_kernel<192> spilled a prefetched fragment right behind the asm statement that
requested it -- garbage in the spill slot, and the late data landed in what had become an accumulator.  DESIGN 8.6.9.)
usage: check_pending_lds_regs.py listing.s [kernel-name-substring]"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(name, lines):
    pend, issues = [], []
    for i, l in enumerate(lines):
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        if t.startswith("s_waitcnt"):
            mm = re.search(r"lgkmcnt\((\d+)\)", t)
            if mm:
                n = int(mm.group(1))
                pend = pend[len(pend) - n:] if n > 0 else []
            continue
        parts = re.split(r"[ ,]+", t)
        op, ops = parts[0], parts[1:]
        used = set()
        for o in ops:
            used |= regs(o)
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            dst = regs(ops[0]) if (op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_swizzle")) else set()
            src = used - dst
            for r, j in pend:
                if r & (src | dst):
                    issues.append((i + 1, t, j))
                    break
            pend.append((dst, i + 1))
            continue
        for r, j in pend:
            if r & used:
                issues.append((i + 1, t, j))
                break
    print(f"{name[:90]}: {len(issues)} instruction(s) touch a register with an LDS read pending")
    for ln, t, j in issues[:6]:
        print(f"    line {ln}: {t}   (read requested at line {j})")
    return len(issues)


text = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
bad, cur, name = 0, None, None
for l in text:
    m = re.match(r"^(_Z\S+):\s*; @", l)
    if m:
        name, cur = m.group(1), []
        continue
    if cur is not None:
        if "; -- End function" in l:
            if pat in name:
                bad += check(name, cur)
            cur = None
        else:
            cur.append(l)
sys.exit(1 if bad else 0)
