"""Per-basic-block instruction statistics of one kernel in a `hipcc -S --cuda-device-only` listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/sf.s scanfold_amd/csrc/scanfold_hip.hip
    python tools/isa_blocks.py /tmp/sf.s sf_mfe_fast_kernelILi128ELi120ELb0ELb0ELb0 [n_blocks] [--dump LABEL]

For the largest blocks: vector / scalar / LDS / scalar-memory / vector-memory instructions, and the issue slots that do no
work — s_waitcnt and s_nop (on gfx950 a packed VOP3P result read by the very next instruction costs one s_nop: chains of
v_pk_min_i16 / v_pk_add_i16 are full of them unless independent work sits in between)."""
import collections
import json
import re
import sys

# MI355X, four waves per SIMD (tools/micro/valu_classes.hip, profiles/r04/valu_classes.json): VOP1/VOP2 moves, 32-bit adds /
# subtracts / and / or, 16-bit add / min / max and f32 add / fma issue at the full rate (~1.05 ns per wave-instruction and
# SIMD); everything else measured — every VOP3-only opcode, every packed (VOP3P) one, 32-bit min / max, shifts, multiplies,
# SDWA / DPP forms — at half of it (~1.9 ns)
FULL_RATE = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_add_f32", "v_sub_f32",
             "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_min_i16", "v_max_i16", "v_min_u16", "v_max_u16", "v_add_u16", "v_sub_u16",
             "v_add_co_u32", "v_addc_co_u32", "v_not_b32")


def valu_ns(op, full=1.05, half=1.9):
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    if base.endswith(("_sdwa", "_dpp")) or op.endswith("_e64"):
        return half
    return full if base in FULL_RATE else half


def classify(op):
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "flat_", "buffer_")):
        return "vmem"
    return "other"


def blocks_of(path, symbol):
    out, cur, on = [], None, False
    for ln in open(path):
        if not on:
            if ln.startswith("_Z") and symbol in ln and ln.rstrip().split(";")[0].rstrip().endswith(":"):
                on, cur = True, ["entry", []]
            continue
        if ln.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            out.append(cur)
            cur = [m.group(1), []]
            continue
        t = ln.strip()
        if t and not t.startswith((";", ".")):
            cur[1].append(t)
    if cur:
        out.append(cur)
    return out


def main():
    path, symbol = sys.argv[1], sys.argv[2]
    args = sys.argv[3:]
    blocks = blocks_of(path, symbol)
    if "--dump" in args:
        lab = args[args.index("--dump") + 1]
        for name, ins in blocks:
            if name == lab:
                print("\n".join(ins))
        return
    n = int(args[0]) if args else 12
    tot = collections.Counter()
    rows = []
    for name, ins in blocks:
        c = collections.Counter(classify(i.split()[0]) for i in ins)
        tot.update(c)
        rows.append((len(ins), name, c))
    rows.sort(reverse=True)
    keys = ("valu", "salu", "lds", "smem", "vmem", "wait", "nop")
    ns_of = {name: sum(valu_ns(i.split()[0]) for i in ins if classify(i.split()[0]) == "valu") for name, ins in blocks}
    print("%-12s %6s  " % ("block", "insts") + " ".join("%5s" % k for k in keys) + "  valu_ns  ns/valu")
    for ln, name, c in rows[:n]:
        print("%-12s %6d  " % (name, ln) + " ".join("%5d" % c.get(k, 0) for k in keys)
              + "  %7.0f  %7.2f" % (ns_of[name], ns_of[name] / max(c.get("valu", 0), 1)))
    print("%-12s %6d  " % ("whole kernel", sum(r[0] for r in rows)) + " ".join("%5d" % tot.get(k, 0) for k in keys))
    hot = rows[:4]
    hv = sum(c.get("valu", 0) for _, _, c in hot)
    print(json.dumps({"hot_blocks": [r[1] for r in hot], "hot_valu": hv,
                      "hot_valu_ns_per_inst": sum(ns_of[r[1]] for r in hot) / max(hv, 1)}))


if __name__ == "__main__":
    main()
