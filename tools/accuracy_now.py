import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import test_accuracy_gpu as T
gd = "/root/repo/tests/golden"
for tag, batch in (("A", 50), ("B", 20)):
    accs, chk, ref, _ = T._run(gd, tag, batch)
    d = np.abs(accs - ref)
    print("config %s: %d episodes: engine mean %.3f  reference mean %.3f  identical %.3f  p90 %.2f  p99 %.2f  max %.2f" % (
        tag, len(accs), accs.mean(), ref.mean(), np.mean(d < 1e-6), np.percentile(d, 90), np.percentile(d, 99), d.max()))
