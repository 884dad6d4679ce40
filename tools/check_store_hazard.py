#!/usr/bin/env python3
"""Build-time guard for the 16-byte store data hazard of the extraction kernels (kpe_bs.hip / kpe_pc.hip).

A buffer_store_dwordx4 reads its four data registers a moment after it issues.  The ISA manual and LLVM's hazard
recogniser require one wait state between a VMEM store of more than 64 bits and a VALU write of its data registers
— but exempt MUBUF stores whose soffset is an SGPR, which is exactly the form these kernels use, and there the
hardware was seen to pick up a register overwritten right behind the store (tools/cmp_kpe_paths.py, DESIGN.md
4.1e: the next store's lane offset showed up in dword 0 of the medians).  The kernels therefore keep every such
store inside two sched_barriers with an `s_nop 1` behind the pair; this script checks the compiler's output: in
the named kernels no instruction that writes a data register of a buffer_store_dwordx4 may follow it within WAIT
instruction slots unless an s_nop / s_waitcnt / another store's issue fills them.

usage: tools/check_store_hazard.py file.s kernel_substring [kernel_substring ...]   (exit 1 on a violation)"""
import re
import sys

WAIT = 2  # wait states demanded between the store and a write of its data registers


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def slots(ins):
    """wait states an instruction provides on top of its own issue slot"""
    m = re.match(r"s_nop\s+(\d+)", ins)
    if m:
        return int(m.group(1)) + 1
    return 1


def check(path, keys):
    s = open(path).read()
    bad = 0
    seen = 0
    for name in re.findall(r"^(\S+):\s*(?:;.*)?$", s, re.M):
        if name.startswith(".") or not any(k in name for k in keys):
            continue
        i = s.index("\n" + name + ":")
        j = s.index(".end_amdhsa_kernel", i)
        ins = []
        for l in s[i:j].split("\n"):
            t = l.split(";")[0].strip()
            if not t or t.startswith(".") or re.match(r"^[\w.$]+:", t):
                continue
            ins.append(t)
        for k, t in enumerate(ins):
            if not t.startswith("buffer_store_dwordx4"):
                continue
            seen += 1
            data = regs(t.split()[1].rstrip(","))
            waited = 0
            for u in ins[k + 1:k + 1 + WAIT + 1]:
                if waited >= WAIT:
                    break
                op = u.split()[0]
                if op.startswith("v_") or op.startswith("ds_read") or op.startswith("buffer_load") or op.startswith("global_load"):
                    dst = regs(u.split()[1].rstrip(","))
                    if dst & data:
                        print(f"{name}: '{u}' writes data registers of '{t}' after {waited} wait state(s)")
                        bad += 1
                        break
                waited += slots(u)
    print(f"{path}: {seen} 16-byte buffer stores checked in kernels matching {keys}, {bad} violation(s)")
    return bad


if __name__ == "__main__":
    sys.exit(1 if check(sys.argv[1], sys.argv[2:]) else 0)
