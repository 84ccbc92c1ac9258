import time, torch, numpy as np, sys
sys.path.insert(0, "/root/repo")
import meta_fine_tuning_amd
from meta_fine_tuning_amd.methods.meta_template import linear_head_adapt
torch.manual_seed(0); np.random.seed(0)
zs = torch.randn(25, 512, device="cuda").abs(); zq = torch.randn(75, 512, device="cuda").abs()
y = np.repeat(np.arange(5), 5)
w = torch.randn(5, 512, device="cuda") * 0.04; b = torch.zeros(5, device="cuda")
for _ in range(2): s = linear_head_adapt(zs, y, zq, w, b, 5, 5)
torch.cuda.synchronize(); t = time.time()
for _ in range(10): s = linear_head_adapt(zs, y, zq, w, b, 5, 5)
torch.cuda.synchronize(); print("linear_head_adapt %.2f ms" % ((time.time() - t) * 100))
