import collections, csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]) for r in rows))
marks = [i for i, e in enumerate(ev) if e[2].startswith("triad_kernel")]
# markers: each wae_bench_triad launches 2 warm-up + 1 timed triads; take the last pass = between the last two marker groups
groups = []
for i in marks:
    if groups and i - groups[-1][-1] <= 3: groups[-1].append(i)
    else: groups.append([i])
a, b = groups[-2][-1] + 1, groups[-1][0]
seg = ev[a:b]
span = seg[-1][1] - seg[0][0]
busy = 0; end = seg[0][0]; gaps = collections.defaultdict(lambda: [0, 0]); kt = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
for s, e, n in seg:
    kt[n][0] += 1; kt[n][1] += e - s
prev = None
cur_end = seg[0][0]
for s, e, n in seg:
    if s > cur_end:
        g = s - cur_end
        key = (prev, n)
        gaps[key][0] += 1; gaps[key][1] += g
        hist["<10us" if g < 10e3 else "<50us" if g < 50e3 else "<200us" if g < 200e3 else "<1ms" if g < 1e6 else ">=1ms"] += g
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; prev = n
print("pass span %.3f s, GPU busy %.3f s (%.1f %%), idle %.3f s over %d dispatches" % (span / 1e9, busy / 1e9, 100 * busy / span, (span - busy) / 1e9, len(seg)))
print("idle time by gap length:", {k: round(v / 1e9, 4) for k, v in hist.items()})
print("largest idle totals by (previous kernel -> next kernel):")
for (p, n), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %8.1f ms  %5d x %8.1f us   %s -> %s" % (t / 1e6, c, t / c / 1e3, p, n))
print("kernel time:")
for n, (c, t) in sorted(kt.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %8.1f ms %6d x %8.1f us  %s" % (t / 1e6, c, t / c / 1e3, n))
