import sys, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t0 = a[:, 0].min()
s, m, e = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0  # us (100 MHz counter)
print("waves", len(a))
for name, x in (("start", s), ("after setup", m), ("end", e), ("setup dur", m - s), ("work dur", e - m)):
    print("%-12s min %.2f p10 %.2f med %.2f p90 %.2f max %.2f" % (name, x.min(), np.percentile(x, 10), np.median(x), np.percentile(x, 90), x.max()))
