#!/usr/bin/env python3
"""tools/isa_loop_counts.py <file.s> <kernel-substring> -- static instruction counts per loop of one kernel
(VALU / SALU / LDS / VMEM / branch / waitcnt), from hipcc -S output.  Every basic block is attributed to the
outermost loop named in its '; in Loop: Header=' comment; blocks outside any loop go to 'straight'.  Static
counts: blocks under a rarely-taken branch (slow paths) count like any other."""
import re
import sys
from collections import defaultdict, OrderedDict


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    cur = "straight"
    counts = OrderedDict()
    depth_of = {}
    for l in lines[start:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            label, rest = m.group(1), m.group(2)
            if "Loop Header: Depth=1" in rest or "This Inner Loop Header: Depth=1" in rest or "=>This Loop Header: Depth=1" in rest:
                cur = label[1:]
            elif "in Loop: Header=" in rest:
                # nested blocks name their innermost header; map it to its depth-1 ancestor
                h = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", rest)
                parent = re.search(r"Parent Loop (BB\d+_\d+) Depth=1", rest)
                cur = parent.group(1) if parent else (h.group(1) if h.group(2) == "1" else depth_of.get(h.group(1), h.group(1)))
            else:
                cur = "straight"
            if "Depth=2" in rest or "Depth=3" in rest:
                p = re.search(r"Parent Loop (BB\d+_\d+) Depth=1", rest)
                if p:
                    depth_of[label[1:]] = p.group(1)
                    cur = p.group(1)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        c = counts.setdefault(cur, defaultdict(int))
        if op.startswith("v_"):
            c["VALU"] += 1
            if op.startswith("v_pk_"):
                c["pk"] += 1
            if op.endswith("_f64") or "_f64_" in op:
                c["f64"] += 1
            if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                c["lane"] += 1
            if op.startswith("v_cndmask"):
                c["cndmask"] += 1
            if op.startswith(("v_mov", "v_accvgpr")):
                c["mov"] += 1
        elif op.startswith("s_waitcnt"):
            c["wait"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            c["branch"] += 1
        elif op.startswith("s_barrier"):
            c["barrier"] += 1
        elif op.startswith("s_"):
            c["SALU"] += 1
        elif op.startswith("ds_"):
            c["LDS"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            c["VMEM"] += 1
        else:
            c["other"] += 1
    keys = ["VALU", "pk", "f64", "lane", "cndmask", "mov", "SALU", "LDS", "VMEM", "wait", "branch", "barrier", "other"]
    print(f"{'loop':14s}" + "".join(f"{k:>8s}" for k in keys))
    for name, c in counts.items():
        print(f"{name:14s}" + "".join(f"{c.get(k, 0):8d}" for k in keys))


if __name__ == "__main__":
    main()
