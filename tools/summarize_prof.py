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
    if "/trace/" not in f:
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
        if not any(t in k for t in ("k_march", "k_mlp", "k_vm_app", "k_composite", "k_density")):
            continue
        parts = [f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())]
        print(f"  {k:40s} " + "  ".join(parts))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CU_CYCLES" in cs and sum(cs["SQ_BUSY_CU_CYCLES"]) > 0:
            # MFMA busy cycles are summed over the 4 SIMDs of every CU
            util = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / (4.0 * sum(cs["SQ_BUSY_CU_CYCLES"]))
            print(f"  {'':40s} MFMA pipe utilisation = MFMA_BUSY / (4 SIMD x BUSY_CU_CYCLES) = {util:.3f}")
