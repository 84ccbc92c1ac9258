"""Per-launch time of the Wcompute layer + its statistics merge at the meta-training step's size (one 5-shot episode: 16 graphs of 30
nodes, 7,440 pair rows), register-K form (rk=1) against the 128-row tile kernel (rk=0); 50 dependent launches replayed from a hipGraph.
    gpurun -- python3 tools/pair_rk_micro.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import meta_fine_tuning_amd
from meta_fine_tuning_amd import functional as Fn, functional_bwd as FB, ops, synthetic, _lib as L
dev = "cuda"
sd = synthetic.gnn_head_state_dict(seed=5)
G = Fn.GnnHeadWeights(sd, dev, 5)
lib = L.lib()
N, B = 30, 16
P = N * (N + 1) // 2
rows = B * P
ij = Fn.pair_index_table(N, dev)
x = torch.randn(B * N, 256, device=dev)
layers, (w5, b5) = G.wc["layer_w0"]
def time_it(fn, n=50):
    """n dependent launches replayed from one hipGraph (what the meta-training step does): us per launch"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (10 * n) * 1000
for rk in (1, 0):
    tiles_m = int(lib.mft_pair_mlp_tiles_m_rk(B, N) if rk else lib.mft_pair_mlp_tiles_m(B, N))
    wsm = torch.empty(tiles_m * 192, device=dev); wsq = torch.empty_like(wsm); wsn = torch.empty(tiles_m, device=dev)
    sc = torch.rand(1, 192, device=dev) + 0.5; sh = torch.randn(1, 192, device=dev)
    z0 = torch.randn(rows, 192, device=dev)
    for li, (w, b, gam, beta, cout) in enumerate(layers):
        K = 133 if li == 0 else layers[li - 1][4]
        Kpad = 160 if li == 0 else K
        h_in = x if li == 0 else z0[:, :K].contiguous()
        z = torch.empty(rows, cout, device=dev)
        def run():
            if rk:
                L.check(lib.mft_pair_mlp_layer_rk(ops._p(h_in), h_in.shape[1], 0 if li == 0 else 1, ops._p(ij), ops._p(sc), ops._p(sh), ops._p(w), K, Kpad,
                        ops._p(b), ops._p(z), cout, 1, B, N, 0.01, ops._p(wsm), ops._p(wsq), ops._p(wsn), ops._stream()), "rk")
            else:
                L.check(lib.mft_pair_mlp_layer(ops._p(h_in), h_in.shape[1], 0 if li == 0 else 1, ops._p(ij), ops._p(sc), ops._p(sh), ops._p(w), K, Kpad,
                        ops._p(b), ops._p(z), cout, 1, B, N, 0.01, ops._p(wsm), ops._p(wsq), ops._p(wsn), 0, ops._stream()), "tile")
        t = time_it(run)
        m = torch.empty(1, cout, device=dev); s_ = torch.empty(1, cout, device=dev); so = torch.empty(1, cout, device=dev); sho = torch.empty(1, cout, device=dev)
        fin = lib.mft_pair_mlp_stats_finalize_rk if rk else lib.mft_pair_mlp_stats_finalize
        tf = time_it(lambda: L.check(fin(ops._p(wsm), ops._p(wsq), ops._p(wsn), 1, tiles_m, cout, ops._p(gam), ops._p(beta), 1e-5, ops._p(so), ops._p(sho), ops._p(m), ops._p(s_), ops._stream()), "fin"))
        print("rk=%d layer %d K=%d Cout=%d: layer %.2f us  finalize %.2f us" % (rk, li, Kpad, cout, t, tf), flush=True)
