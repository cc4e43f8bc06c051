"""Per-kernel register / scratch / LDS use of the gfx950 code objects inside libtensoir_hip.so (no GPU needed).
Usage: python tools/kernel_resources.py [library.so] [regex on the demangled-ish name]"""
import os
import re
import subprocess
import sys
import tempfile

import yaml

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(so):
    data = open(so, "rb").read()
    pos = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, p in enumerate(pos):
            end = pos[k + 1] if k + 1 < len(pos) else len(data)
            b = os.path.join(tmp, f"b{k}.bundle")
            open(b, "wb").write(data[p:end])
            co = b + ".co"
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            doc = notes[notes.index("---"):notes.rindex("...")]
            for kd in yaml.safe_load(doc)["amdhsa.kernels"]:
                name = subprocess.run(["c++filt", kd[".name"]], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"\(anonymous namespace\)::", "", name)
                out.append({"name": re.sub(r"\(.*", "", name).replace("void ", ""), "vgpr": kd[".vgpr_count"], "agpr": kd[".agpr_count"],
                            "sgpr": kd[".sgpr_count"], "vgpr_spill": kd[".vgpr_spill_count"], "sgpr_spill": kd[".sgpr_spill_count"],
                            "scratch": kd[".private_segment_fixed_size"], "lds": kd[".group_segment_fixed_size"],
                            "max_wg": kd[".max_flat_workgroup_size"]})
    return out


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tensoir_amd", "libtensoir_hip.so")
    filt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    for k in sorted(kernels(so), key=lambda k: k["name"]):
        if filt.search(k["name"]):
            print(f"{k['name'][:70]:70s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} spill v/s {k['vgpr_spill']}/{k['sgpr_spill']} "
                  f"scratch {k['scratch']} lds {k['lds']} wg {k['max_wg']}")
