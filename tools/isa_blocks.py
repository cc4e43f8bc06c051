"""Static instruction counts per basic block of one kernel (hipcc -S for gfx950; no GPU needed).
Usage: python tools/isa_blocks.py <source.hip> <kernel name substring> [min_valu] [-DFLAG ...]
Prints, per basic block with >= min_valu VALU instructions: VALU (of which quarter-rate transcendental / 32-bit multiply),
SALU, memory (global / LDS) instructions and the branch that ends it; then the kernel's totals."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUARTER = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos|mul_lo_u32|mul_hi_u32|mul_hi_i32|mul_lo_i32)")


def asm_of(src, flags):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/tensoir_amd/csrc",
           "-S", "--cuda-device-only", src, "-o", out] + flags
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read().splitlines()


def blocks(lines, kernel):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kernel in l and l.rstrip().endswith(tuple(": ;".split())) or (l.startswith("_Z") and kernel in l and ":" in l))
    cur, out = {"label": "entry", "line": start, "valu": 0, "q": 0, "salu": 0, "vmem": 0, "lds": 0, "mfma": 0, "end": "", "ops": {}}, []
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith(".Lfunc_end"):
            break
        if re.match(r"^\.LBB\d+_\d+:", l):
            out.append(cur)
            cur = {"label": l.split(":")[0], "line": i, "valu": 0, "q": 0, "salu": 0, "vmem": 0, "lds": 0, "mfma": 0, "end": "", "ops": {}}
            continue
        if not l or l.startswith((";", ".")):
            continue
        op = l.split()[0]
        cur["ops"][op] = cur["ops"].get(op, 0) + 1
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            cur["mfma"] += 1
        elif op.startswith("v_"):
            cur["valu"] += 1
            if QUARTER.match(op):
                cur["q"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            cur["end"] += (" | " if cur["end"] else "") + l.split(";")[0].strip()
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
    out.append(cur)
    return out


if __name__ == "__main__":
    src, kernel = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    min_valu = int(rest[0]) if rest and not rest[0].startswith("-") else 40
    flags = [a for a in rest if a.startswith("-")]
    bl = blocks(asm_of(src, flags), kernel)
    tot = {k: sum(b[k] for b in bl) for k in ("valu", "q", "salu", "vmem", "lds", "mfma")}
    print(f"{kernel}: {len(bl)} basic blocks, static totals {tot}")
    for b in bl:
        if b["valu"] >= min_valu or b["mfma"]:
            print(f"{b['label']:12s} line {b['line']:6d} valu {b['valu']:4d} (quarter-rate {b['q']:3d}) mfma {b['mfma']:3d} salu {b['salu']:4d} vmem {b['vmem']:3d} lds {b['lds']:3d}  {b['end'][:120]}")
    if os.environ.get("ISA_OPS"):
        want = os.environ["ISA_OPS"]
        for b in bl:
            if b["label"] == want:
                for op, n in sorted(b["ops"].items(), key=lambda kv: -kv[1]):
                    print(f"  {n:5d} {op}")
