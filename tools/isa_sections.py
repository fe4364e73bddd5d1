"""Vector / scalar / LDS instructions of one kernel BY SOURCE SECTION, from a listing with line tables:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -S --cuda-device-only -I scanfold_amd/csrc -I include \
          -o /tmp/mfe_one.s tools/dev/mfe_one.hip
    python tools/isa_sections.py /tmp/mfe_one.s [--blocks N] [--json out.json]

Every instruction carries the `.loc` in force where it stands; lines of sf_mfe_fast.hip.h map to the sections of a cell
(SECTIONS below).  Instructions whose `.loc` points into a helper header (sf_pk16.h, sf_energy.h, sf_launch.h: packed min / add,
sfd_min, lane reads — inlined everywhere) inherit the section of the nearest preceding instruction of the same basic block that
lies in the main file.  Output: per basic block (largest first) and per section — vector instructions split by the measured
full-rate / half-rate classes (tools/isa_blocks.py), scalar, LDS, memory, idle issue slots (s_waitcnt, s_nop)."""
import collections
import json
import re
import sys

from isa_blocks import classify, valu_ns

MAIN = "sf_mfe_fast.hip.h"
# (first line, last line, name) in scanfold_amd/csrc/sf_mfe_fast.hip.h — kept in step with the file by tests/test_isa_sections.py
SECTIONS = []


def load_sections(path):
    """Sections are delimited IN THE SOURCE by comments of the form `// @section NAME` (a section runs to the next marker)."""
    out, cur, start = [], None, 0
    for n, ln in enumerate(open(path), 1):
        m = re.search(r"//\s*@section\s+(\S+)", ln)
        if m:
            if cur is not None:
                out.append((start, n - 1, cur))
            cur, start = m.group(1), n
    if cur is not None:
        out.append((start, 10 ** 9, cur))
    return out


def section_of(line):
    for a, b, name in SECTIONS:
        if a <= line <= b:
            return name
    return "other"


def parse(path):
    files, blocks = {}, []
    cur, on, loc = None, False, (None, 0)
    for ln in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', ln)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if not on:
            if ln.startswith("_Z") and ln.rstrip().split(";")[0].rstrip().endswith(":"):
                on, cur = True, ["entry", []]
            continue
        if ln.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = ln.strip()
        if t and not t.startswith((";", ".")):
            cur[1].append((t.split()[0], loc))
    blocks.append(cur)
    return blocks


def main():
    import os
    path = sys.argv[1]
    args = sys.argv[2:]
    nblocks = int(args[args.index("--blocks") + 1]) if "--blocks" in args else 10
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scanfold_amd", "csrc", MAIN)
    SECTIONS.extend(load_sections(src))
    blocks = parse(path)
    result = {}
    rows = sorted(blocks, key=lambda b: -len(b[1]))[:nblocks]
    for name, ins in rows:
        per = collections.OrderedDict()
        last = "other"
        for op, (f, line) in ins:
            if f.endswith(MAIN):
                last = section_of(line)
            sec = last
            c = per.setdefault(sec, collections.Counter())
            k = classify(op)
            c[k] += 1
            if k == "valu":
                c["valu_full" if valu_ns(op) < 1.5 else "valu_half"] += 1
                c["valu_ns"] += valu_ns(op)
        result[name] = {s: dict(c) for s, c in per.items()}
        tot = collections.Counter()
        for c in per.values():
            tot.update(c)
        print("\n%s: %d instructions, %d vector (%d full-rate + %d half-rate = %.0f ns), %d scalar, %d LDS, %d wait, %d nop"
              % (name, len(ins), tot["valu"], tot["valu_full"], tot["valu_half"], tot["valu_ns"], tot["salu"], tot["lds"], tot["wait"], tot["nop"]))
        print("  %-22s %5s %5s %5s %7s %5s %5s %5s %5s %5s" % ("section", "valu", "full", "half", "ns", "salu", "lds", "smem", "vmem", "idle"))
        for s, c in sorted(per.items(), key=lambda kv: -kv[1]["valu"]):
            print("  %-22s %5d %5d %5d %7.0f %5d %5d %5d %5d %5d" % (s, c["valu"], c["valu_full"], c["valu_half"], c["valu_ns"], c["salu"], c["lds"],
                                                                    c["smem"], c["vmem"], c["wait"] + c["nop"]))
    if "--json" in args:
        json.dump(result, open(args[args.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
