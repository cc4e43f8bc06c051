"""Summarise rocprofv3 output dirs (kernel stats + PMC counters per kernel) into a small text table."""
import csv, glob, os, sys, collections

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    for p in ("void ", "(anonymous namespace)::"):
        name = name.replace(p, "")
    name = name.split("(")[0]
    return name[:60]


print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for f in find("*kernel_stats.csv"):
    if "/trace" not in f:
        continue
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    print(f"# {os.path.relpath(f, root)}")
    print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}")
    for r in rows[:25]:
        print(f"{short(r['Name']):60s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):7.2f}")

print("\n== PMC counters (per-dispatch mean by kernel)")
for f in find("*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"# {os.path.relpath(f, root)}")
    for k, cs in sorted(agg.items()):
        if not any(t in k for t in ("k_march", "k_mlp", "k_vm_app", "k_composite", "k_density", "k_shade", "k_indirect")):
            continue
        parts = [f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())]
        print(f"  {k:40s} " + "  ".join(parts))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CU_CYCLES" in cs and sum(cs["SQ_BUSY_CU_CYCLES"]) > 0:
            # MFMA busy cycles are summed over the 4 SIMDs of every CU
            util = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / (4.0 * sum(cs["SQ_BUSY_CU_CYCLES"]))
            print(f"  {'':40s} MFMA pipe utilisation = MFMA_BUSY / (4 SIMD x BUSY_CU_CYCLES) = {util:.3f}")


# ---- per-launch HBM-side traffic (bytes) per C-ABI entry point, for bench.py's roofline.traffic ----------------
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM
# section) -> doubled.  WRITE_SIZE is taken as reported (uncalibrated per the guide).  Collected in separate passes.
import json


def library_hash():
    """Source hash of the library these counters were collected with (csrc/build.sh writes it next to the .so)."""
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        return open(os.path.join(here, "..", "tensoir_amd", "libtensoir_hip.so.srchash")).read().strip()
    except OSError:
        return None


OP_OF = (("k_indirect_fused", "tir_indirect_fused_fwd"), ("k_mlp_f16_auxt", "tir_mlp_fwd_auxtab_f16"), ("k_vm_app_h16", "tir_vm_app_fwd_h16"), ("k_mlp_bf16_auxt", "tir_mlp_fwd_bf16x3"), ("k_mlp_bf16<3", "tir_mlp_fwd_bf16x3"), ("k_mlp_bf16_multi<3", "tir_mlp_fwd_bf16x3"), ("k_mlp_bf16<1", "tir_mlp_fwd_bf16"), ("k_mlp_mfma", "tir_mlp_fwd"),
         ("k_vm_app_mfma", "tir_vm_app_fwd"), ("k_vm_app_primary", "tir_vm_app_fwd"), ("k_march_secondary", "tir_march_secondary_fwd"),
         ("k_march_primary", "tir_march_primary_fwd"), ("k_composite_primary", "tir_composite_primary"),
         ("k_density_grad", "tir_density_grad_fwd"), ("k_shade_integrate", "tir_shade_integrate"))
tot = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for f in find("*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            c = r["Counter_Name"]
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = short(r["Kernel_Name"])
            for pat, op in OP_OF:
                if pat in k:
                    t = tot[op][c]
                    t[0] += float(r["Counter_Value"]); t[1] += 1
                    break
traffic = {}
for op, d in tot.items():
    fe = d["FETCH_SIZE"][0] / max(1, d["FETCH_SIZE"][1]) * 1024.0 * 2.0
    wr = d["WRITE_SIZE"][0] / max(1, d["WRITE_SIZE"][1]) * 1024.0
    traffic[op] = round(fe + wr)
    traffic[op + ":detail"] = {"fetch_bytes_x2_corrected": round(fe), "write_bytes": round(wr),
                               "launches_sampled": d["FETCH_SIZE"][1]}
if traffic:
    traffic["_library_source_hash"] = library_hash()
    with open(os.path.join(root, "pmc_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    print("\n== pmc_traffic.json (bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE)")
    for k, v in traffic.items():
        if not k.endswith(":detail") and not k.startswith("_"):
            print(f"  {k:32s} {v/1e6:10.2f} MB")

# ---- issue fractions per entry point (bench.py's bound labels): VALU share of the SIMD issue time and matrix-pipe busy share,
# from an SQ pass (SQ_ACTIVE_INST_VALU counts quad-cycles = one VALU instruction slot of a SIMD; a CU has 4 SIMDs):
#   valu_issue_frac = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES        mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)
iss = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in find("*counter_collection.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            c = r["Counter_Name"]
            if c not in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"):
                continue
            k = short(r["Kernel_Name"])
            for pat, op in OP_OF:
                if pat in k:
                    t = iss[op][c]
                    t[0] += float(r["Counter_Value"]); t[1] += 1
                    break
issue = {}
for op, d in iss.items():
    mean = {c: v[0] / max(1, v[1]) for c, v in d.items()}
    cu = mean.get("SQ_BUSY_CU_CYCLES", 0.0)
    if cu <= 0:
        continue
    e = {"launches_sampled": int(max(v[1] for v in d.values()))}
    if "SQ_ACTIVE_INST_VALU" in mean:
        e["valu_issue_frac"] = round(mean["SQ_ACTIVE_INST_VALU"] / cu, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
        e["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * cu), 4)
    if "SQ_INSTS_VALU" in mean:
        e["valu_instructions_per_launch"] = round(mean["SQ_INSTS_VALU"])
    if "SQ_WAIT_INST_ANY" in mean and "SQ_WAVE_CYCLES" in mean and mean["SQ_WAVE_CYCLES"] > 0:
        e["wait_frac_of_wave_cycles"] = round(mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"], 4)
    issue[op] = e
if issue:
    issue["_library_source_hash"] = library_hash()
    issue["_note"] = ("valu_issue_frac = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES (VALU instruction slots per CU-cycle over the 4 SIMDs' "
                      "one slot per 4 cycles each); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES); separate rocprofv3 --pmc pass")
    with open(os.path.join(root, "pmc_issue.json"), "w") as fh:
        json.dump(issue, fh, indent=1)
    print("\n== pmc_issue.json")
    for k, v in issue.items():
        if not k.startswith("_"):
            print(f"  {k:32s} {v}")
