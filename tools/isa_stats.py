#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -save-temps assembly file (gfx950).
usage: tools/isa_stats.py file.s kernel_substring [--loops]
Prints registers, LDS, scratch and the opcode histogram of the whole kernel; --loops adds the histogram of every
basic-block range between a label and the backward branch to it (innermost loops first by size)."""
import re
import sys
from collections import Counter

SLOW = ("v_bitop3", "v_perm", "v_lshl", "v_lshr", "v_ashr", "v_pk_", "v_bfi", "v_alignb", "v_min3", "v_med3", "v_max3",
        "v_sad", "v_and_or", "v_or3", "v_lshl_or", "v_lshl_add", "v_add_lshl", "v_mul", "v_mad", "v_bfe", "v_cndmask", "v_readlane",
        "v_writelane", "v_bcnt", "v_ffb", "v_mbcnt", "v_cmp", "v_xad")


def main():
    path, key = sys.argv[1], sys.argv[2]
    s = open(path).read()
    names = re.findall(r"^(\S*%s\S*):\s*(?:;.*)?$" % re.escape(key), s, re.M)
    names = [n for n in names if not n.startswith(".")]
    if not names:
        sys.exit("no kernel matches " + key)
    for name in names:
        i = s.index("\n" + name + ":")
        j = s.index(".end_amdhsa_kernel", i)
        body = s[i:j]
        meta = dict(re.findall(r"\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset) (\d+)", body))
        spill = re.findall(r"; (SGPR|VGPR) spill count: (\d+)|ScratchSize: (\d+)", s[j:j + 3000])
        print(name, meta)
        extra = re.findall(r"; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|sgpr_spill_count|vgpr_spill_count): ?(\d+)", s[j:j + 4000])
        print("  ", dict(extra))
        code = body[:body.index(".section")] if ".section" in body else body
        lines = []
        for l in code.split("\n"):
            t = l.strip()
            t = t.split(";")[0].strip()
            if not t:
                continue
            if t.startswith("."):
                if re.match(r"^\.L\w+:", t):
                    lines.append(("L", t.split(":")[0]))
                continue
            if re.match(r"^[\w.$]+:", t):
                lines.append(("L", t.split(":")[0]))
                continue
            lines.append(("I", t))
        ins = [t for k, t in lines if k == "I"]

        def hist(ins):
            c = Counter(x.split()[0] for x in ins)
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            slow = sum(v for k, v in c.items() if k.startswith(SLOW) or "_dpp" in k or "_sdwa" in k)
            dpp = sum(1 for x in ins if " row_" in x or " wave_" in x or "quad_perm" in x)
            return c, valu, slow, dpp

        c, valu, slow, dpp = hist(ins)
        print("   instructions %d: valu %d (3-input/shift/perm class %d, dpp %d) salu %d ds %d vmem %d" % (
            len(ins), valu, slow, dpp, sum(v for k, v in c.items() if k.startswith("s_")),
            sum(v for k, v in c.items() if k.startswith("ds_")),
            sum(v for k, v in c.items() if k.startswith(("buffer_", "global_", "scratch_", "flat_")))))
        print("   ", c.most_common(30))
        if "--loops" in sys.argv:
            pos = {}
            idx = 0
            loops = []
            for k, t in lines:
                if k == "L":
                    pos[t] = idx
                else:
                    m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", t)
                    if m:
                        tgt = m.group(1) or m.group(2)
                        if tgt in pos:
                            loops.append((pos[tgt], idx + 1, tgt))
                    idx += 1
            for a, b, tgt in sorted(loops, key=lambda x: x[1] - x[0]):
                c, valu, slow, dpp = hist(ins[a:b])
                print("   loop %s: %d instructions, valu %d (slow class %d), salu %d, ds %d, vmem %d" % (
                    tgt, b - a, valu, slow, sum(v for k, v in c.items() if k.startswith("s_")),
                    sum(v for k, v in c.items() if k.startswith("ds_")),
                    sum(v for k, v in c.items() if k.startswith(("buffer_", "global_", "scratch_", "flat_")))))


if __name__ == "__main__":
    main()
