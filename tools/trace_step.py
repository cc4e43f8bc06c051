"""Per-step GPU timeline from a rocprofv3 kernel trace (csv): wall and busy time per step, idle gaps, per-kernel totals.
usage: python tools/trace_step.py <dir with *_kernel_trace.csv> [first-kernel-prefix] [--timeline]"""
import collections, csv, glob, sys
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "k_march_primary("
path = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in csv.DictReader(open(path)))
idx = [i for i, e in enumerate(ev) if e[2].startswith(first)]
n = min(10, len(idx) - 3)
a, b = idx[-n - 2], idx[-2]
sel = ev[a:b]
busy, cs, ce = 0, None, None
gaps = collections.Counter()
for s, e, name, _ in sel:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
            gaps[name.replace("(anonymous namespace)::", "").replace("void ", "")[:50]] += s - ce
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
wall = (ev[b][0] - ev[a][0]) / n / 1e6
print(f"steps {n}  wall {wall:.3f} ms/step  GPU busy {busy / n / 1e6:.3f}  idle {wall - busy / n / 1e6:.3f}  kernels/step {len(sel) / n:.0f}")
print("idle before (us/step):", [(k, round(v / n / 1e3, 1)) for k, v in gaps.most_common(8)])
agg = collections.defaultdict(lambda: [0, 0])
for s, e, name, _ in sel:
    k = name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    agg[k][0] += e - s; agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:24]:
    print(f"{k:60s} {c / n:6.1f} {t / n / 1e3:8.1f} us")
if "--timeline" in sys.argv:
    a, b = idx[-3], idx[-2]
    t0, pe = ev[a][0], ev[a][0]
    for s, e, name, st in ev[a:b]:
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap{(s - pe) / 1e3:7.1f} s{st} {name.replace('(anonymous namespace)::', '').replace('void ', '')[:48]}")
        pe = max(pe, e)
