"""GPU box: per-workgroup finish times of dw_kernel by the jobs it touched, from a -DNTX_TRAIN_CLOCKS build (NERFTEX_LIB=build_dev/libntx_clocks.so NERFTEX_DW_CLOCKS=<file>)."""
import sys, re, collections
rows = []
for l in open(sys.argv[1]):
    p = l.split()
    g, t0, t1 = int(p[0]), int(p[1]), int(p[2])
    pieces = re.findall(r"j(\d+) cost (\d+) blocks (\d+) (\d+)-(\d+)", l)
    rows.append((g, t0, t1, [(int(j), int(c), int(b), int(s), int(e)) for j, c, b, s, e in pieces]))
end = [r[2] for r in rows]
print("workgroups", len(rows), "finish (10 ns ticks): min %d mean %.0f max %d  -> max/mean %.4f" % (min(end), sum(end) / len(end), max(end), max(end) / (sum(end) / len(end))))
print("start: min %d max %d" % (min(r[1] for r in rows), max(r[1] for r in rows)))
per = collections.defaultdict(list)
for g, t0, t1, pcs in rows:
    for j, c, b, s, e in pcs:
        if b > 0: per[(j, c)].append((e - s) / b)
for (j, c), v in sorted(per.items()):
    print("job %2d cost %3d: pieces %3d, ticks per block mean %.3f  -> per cost unit %.5f" % (j, c, len(v), sum(v) / len(v), sum(v) / len(v) / c))
late = sorted(rows, key=lambda r: -r[2])[:6]
for g, t0, t1, pcs in late: print("late wg", g, t1, [(j, b) for j, c, b, s, e in pcs])
early = sorted(rows, key=lambda r: r[2])[:6]
for g, t0, t1, pcs in early: print("early wg", g, t1, [(j, b) for j, c, b, s, e in pcs])
