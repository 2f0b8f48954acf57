#!/usr/bin/env python3
"""Static instruction counts of the traversal kernel's two inner loops, from the gfx950 assembly hipcc emits.

  python tools/isa_count.py [--out profiles/r2_isa_counts.json]

bench.py's VALU-issue roofline is "wave-level VALU instructions the algorithm needs / launch time" against the chip's
vector issue rate; the per-step instruction counts it multiplies the MEASURED loop trip counts with come from here, so
they follow the code instead of being typed in.  Method: compile csrc/prt_kernels.hip with -save-temps, take the
default instance of k_traverse8_persistent (and the other instances bench.py can run), split it into basic blocks,
find the loops (a branch to an earlier label closes one) and pick
  * the node loop  = the largest loop that holds the five 16-B loads of an 80-B node and the v_cvt_f32_ubyte
                     conversions of its quantized planes, and no IEEE division,
  * the triangle loop = the INNERMOST loop that holds IEEE divisions (v_div_fmas_f32: Triangle::Intersect) and the
                     three loads of a triangle record, and no plane conversion.
Counts are of ALL instructions between the loop's first label and its back edge (rarely taken side blocks such as the
overflow path included: an upper bound of what one trip issues, within a few per cent).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parallelraytracing_amd", "csrc")

# template arguments <STACK_L, WAVES, STATS, INST, LEAN, PRIM> as they appear in the mangled names
INSTANCES = {
    "lean8_5waves": "k_traverse8_persistentILi8ELi5ELb0ELb0ELb1ELb0ELb0EE",        # default: trees of <= 9 levels (C2, C3, C4)
    "lean8_5waves_primary": "k_traverse8_persistentILi8ELi5ELb0ELb0ELb1ELb1ELb0EE",  # the same reading compact primary rays (bounce 0)
    "deep15_4waves": "k_traverse8_persistentILi15ELi4ELb0ELb0ELb0ELb0ELb0EE",       # deeper device-built trees (no 4-wide fallback)
    "wide11_5waves": "k_traverse8_persistentILi11ELi5ELb0ELb0ELb0ELb0ELb0EE",       # A/B (stack_lds = 5)
    "inst12_4waves": "k_traverse8_persistentILi12ELi4ELb0ELb1ELb0ELb0ELb0EE",       # placed copies (C5I)
}


def compile_asm(tmp):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
           "--offload-arch=gfx950", "-c", os.path.join(CSRC, "prt_kernels.hip"), "-o", os.path.join(tmp, "k.o"),
           "-save-temps=obj"]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(tmp, "prt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")


def function_lines(asm, symbol_part):
    """Instruction / label lines of the first function whose mangled name contains symbol_part."""
    out, inside = [], False
    for line in asm:
        s = line.split(";")[0].strip()
        if not inside:
            if s.endswith(":") and symbol_part in s and not s.startswith("."):
                inside = True
            continue
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            out.append(s)
            break
        if not s or (s.startswith(".") and not s.startswith(".LBB")):
            continue
        out.append(s)
    return out


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def loops(lines):
    label_at = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    res = []
    for i, l in enumerate(lines):
        m = re.match(r"s_c?branch\S*\s+(\.LBB\S+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            res.append((label_at[m.group(1)], i))
    return res


def count(lines, lo, hi):
    c = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "other": 0, "total": 0}
    detail = {}
    for l in lines[lo:hi + 1]:
        if l.endswith(":"):
            continue
        op = l.split()[0]
        k = classify(op)
        c[k] += 1
        c["total"] += 1
        if k == "valu":
            base = re.sub(r"_e32|_e64|_dpp|_sdwa", "", op)
            detail[base] = detail.get(base, 0) + 1
    c["valu_by_opcode"] = dict(sorted(detail.items(), key=lambda kv: -kv[1]))
    return c


def pick(lines, lps, need, forbid, smallest=False):
    """The LARGEST loop that has the `need` signature and nothing of `forbid`: a source loop with `continue`s and
    several exits compiles to nested back edges to neighbouring headers; the outermost of them is the whole body.
    smallest=True picks the INNERMOST such loop instead (the triangle rounds sit inside the phase's pass loops, which
    also expand the queued groups: a round is the inner loop)."""
    best = None
    for lo, hi in lps:
        body = lines[lo:hi + 1]
        if all(sum(1 for l in body if l.startswith(p)) >= n for p, n in need.items()) and \
                not any(l.startswith(forbid) for l in body):
            if best is None or (hi - lo < best[1] - best[0] if smallest else hi - lo > best[1] - best[0]):
                best = (lo, hi)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--asm", default="", help="use this .s instead of compiling")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        path = args.asm or compile_asm(tmp)
        asm = open(path).read().splitlines()
    result = {"source": "parallelraytracing_amd/csrc/prt_kernels.hip, hipcc -O3 --offload-arch=gfx950 (tools/isa_count.py)",
              "instances": {}}
    for name, sym in INSTANCES.items():
        lines = function_lines(asm, sym)
        if not lines:
            print(f"{name}: instance {sym} not found", file=sys.stderr)
            continue
        lps = loops(lines)
        node = pick(lines, lps, {"global_load_dwordx4": 5, "v_cvt_f32_ubyte": 24}, "v_div_fmas_f32")
        tri = pick(lines, lps, {"v_div_fmas_f32": 3, "global_load_dword": 3}, "v_cvt_f32_ubyte", smallest=True)
        entry = {"function_instructions": sum(1 for l in lines if not l.endswith(":"))}
        if node:
            entry["node_step"] = count(lines, *node)
        if tri:
            entry["triangle_round"] = count(lines, *tri)
        result["instances"][name] = entry
        ns, tr = entry.get("node_step", {}), entry.get("triangle_round", {})
        print(f"{name}: node step {ns.get('valu')} VALU / {ns.get('total')} instructions, "
              f"triangle round {tr.get('valu')} VALU / {tr.get('total')}")
    if args.out:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from kernel_sha import kernel_sha
        result["kernel_sha16"] = kernel_sha()
        json.dump(result, open(args.out, "w"), indent=1)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
