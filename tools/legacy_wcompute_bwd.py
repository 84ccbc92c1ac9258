"""Round-2 form of the meta-training Wcompute forward / backward (functional_bwd.wcompute_taped / wcompute_backward): the
MATERIALISED pair tensor |x_i - x_j| [B*N*N, F] through generic GEMM / BatchNorm launches over all N*N rows.  Not product code:
kept only so that tools/metatrain_time.py can measure the fused upper-triangle form against it (install())."""
import torch  # noqa: F401

from meta_fine_tuning_amd import functional_bwd as FB
from meta_fine_tuning_amd import ops
from meta_fine_tuning_amd.functional_bwd import LRELU, L, _empty, _linear_bwd, _linear_fwd, _zeros, bn_bwd  # noqa: F401


def wcompute_taped(G, name, x, F, n_graphs, N):
    layers, (w5, b5) = G.wc[name]
    Kp = ops.round_up(F, 32)
    rows = n_graphs * N * N
    d = ops.pair_absdiff(x, N, F, Kp)
    t = {"name": name, "F": F, "Kp": Kp, "d": d, "raw": [], "act": [], "stats": []}
    h, K = d, Kp
    for (w, b, g, beta, cout) in layers:
        o = _linear_fwd(h, K, w, b, cout)
        m, s = ops.bn_stats(o, cout, rows, 1)
        oa = ops.bn_apply(o, cout, rows, 1, m, s, g, beta, act=LRELU, out=_empty(o.shape, o.device))
        t["raw"].append(o); t["act"].append(oa); t["stats"].append((m, s))
        h, K = oa, cout
    sc = _linear_fwd(h, K, w5, b5, 1)                     # [rows, 32], column 0 is the score
    A = ops.masked_softmax(sc, N)
    t["A"] = A
    return A, t


def wcompute_backward(G, t, dA, x, dX, n_graphs, N, grads, prefix):
    """Accumulates d(x) into dX[:, :F]; writes parameter gradients into ``grads`` under ``prefix``."""
    layers, (w5, b5) = G.wc[t["name"]]
    rows = n_graphs * N * N
    dev = x.device
    ds = _zeros((rows, 32), dev)
    L.check(L.lib().mft_masked_softmax_backward(ops._p(t["A"]), ops._p(dA), ops._p(ds), 32, n_graphs, N, ops._stream()),
            "mft_masked_softmax_backward")
    h4 = t["act"][3]
    dh, dW, db = _linear_bwd(h4, 96, w5, ds, 1)
    grads[prefix + ".conv2d_last.weight"] = dW[:, :96].reshape(1, 96, 1, 1).contiguous()
    grads[prefix + ".conv2d_last.bias"] = db
    for li in (3, 2, 1, 0):
        w, b, g, beta, cout = layers[li]
        m, s = t["stats"][li]
        do, dg, dbt = bn_bwd(t["raw"][li], dh, cout, rows, m, s, g, y_act=t["act"][li], act=LRELU)
        grads[prefix + ".bn_%d.weight" % (li + 1)], grads[prefix + ".bn_%d.bias" % (li + 1)] = dg, dbt
        hin = t["act"][li - 1] if li > 0 else t["d"]
        K = layers[li - 1][4] if li > 0 else t["Kp"]
        dh, dW, db = _linear_bwd(hin, K, w, do, cout)
        kin = K if li > 0 else t["F"]
        grads[prefix + ".conv2d_%d.weight" % (li + 1)] = dW[:, :kin].reshape(cout, kin, 1, 1).contiguous()
        grads[prefix + ".conv2d_%d.bias" % (li + 1)] = db
    L.check(L.lib().mft_pair_absdiff_backward(ops._p(x), x.shape[1], ops._p(dh), dh.shape[1], ops._p(dX), dX.shape[1],
                                              n_graphs, N, t["F"], ops._stream()), "mft_pair_absdiff_backward")



def install():
    FB.wcompute_taped, FB.wcompute_backward = wcompute_taped, wcompute_backward
