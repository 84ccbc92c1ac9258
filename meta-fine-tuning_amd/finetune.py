"""Test-time fine-tuning driver (mirror of finetune.py:182-328,424-682 for --method gnnnet).

``finetune(liz_x, y, model, state_in, save_it, ...)`` keeps the reference's signature and semantics
(module-global ``params`` supplies ``model`` and ``fine_tune_epoch``; permutations come from the global numpy
RNG) and runs on FinetuneEngine with a batch of one.  ``finetune_batched`` is the throughput entry point:
E independent episodes in lockstep.  Real datasets are out of scope: ``main`` evaluates
on the in-repo synthetic episodes with the reference's printed accuracy line.
"""
import argparse

import numpy as np
import torch

from . import engine as eng
from . import synthetic
from .io_utils import model_dict, parse_args  # noqa: F401  (re-exported like the reference)

params = None          # set by main(); finetune() reads params.model / params.fine_tune_epoch (finetune.py:185,261)

_ENGINES = {}


def _engine_for(state_in, model, n_way, n_support, n_query, size, n_views, epochs, E, fold50=False):
    head_key = tuple((p.data_ptr(), p._version) for p in list(model.fc.parameters()) + list(model.gnn.parameters()))
    key = (id(state_in), head_key, n_way, n_support, n_query, size, n_views, epochs, E, fold50)
    e = _ENGINES.get(key)
    if e is None:
        if len(_ENGINES) >= 4:
            _ENGINES.clear()
        head = {"fc." + k: v for k, v in model.fc.state_dict().items()}
        head.update({"gnn." + k: v for k, v in model.gnn.state_dict().items()})
        e = eng.FinetuneEngine(state_in, n_way, n_support, n_query, size, n_views=n_views, fine_tune_epoch=epochs,
                               episodes_per_batch=E, head_state=head, fold50=fold50)
        _ENGINES[key] = e
    return e


def finetune(liz_x, y, model, state_in, save_it, linear=False, flatten=True, n_query=15, ds=False,
             pretrained_dataset='miniImageNet', freeze_backbone=False, n_way=5, n_support=5):
    """One episode: liz_x = [x0, x0, aug_1, ...] each [n_way, n_support+n_query, 3, H, W]; returns softmax scores
    [n_way*n_query, n_way] (finetune.py:182-328)."""
    if linear or not flatten or ds:
        raise NotImplementedError("finetune(): only the GNN scoring branch with a flattened backbone is on the HIP hot path "
                                  "(the linear branch is finetune_linear(); ds = DampNet, out of scope)")
    if params is None or params.model != 'ResNet10':
        raise RuntimeError("finetune.params must be set (Namespace(model='ResNet10', fine_tune_epoch=...))")
    model = model.cuda()
    x0 = liz_x[0]
    n_query = x0.size(1) - n_support
    if freeze_backbone:
        # finetune.py:253-266: eval-mode backbone, no backbone optimiser; the classifier gets no gradient either, so the
        # loop at :270-299 changes nothing -- it only consumes one permutation per epoch.  Scores = GNN on eval features.
        for _ in range(params.fine_tune_epoch):
            np.random.permutation(n_way * n_support * (len(liz_x) + 1))
        feat = model_dict[params.model](flatten=True)
        feat.load_state_dict({k.replace("feature.", "", 1): v for k, v in state_in.items()
                              if k.startswith("feature.") and not k.startswith(("feature2.", "feature3."))})
        feat = feat.cuda().eval()
        with torch.no_grad():
            out_all = feat(x0.cuda().reshape(-1, *x0.shape[2:])).view(n_way, n_support + n_query, -1)
            model.n_query = n_query
            return torch.nn.functional.softmax(model.set_forward(out_all, is_feature=True), dim=1)
    e = _engine_for(state_in, model, n_way, n_support, n_query, x0.size(-1), len(liz_x), params.fine_tune_epoch, 1,
                    fold50=getattr(model, "FOLD50", False))
    model.n_query = n_query                                          # finetune.py:312
    return e.run_batch([liz_x])[0].clone()


_LIN_ENGINES = {}


def classifier_init(n_way, dim=512, n=1):
    """Initial weights of ``Classifier(dim, n_way)`` (finetune.py:33-42,65): torch's nn.Linear default draw from the
    global torch RNG, one fresh classifier per episode as in the reference."""
    ws, bs = [], []
    for _ in range(n):
        lin = torch.nn.Linear(dim, n_way)
        ws.append(lin.weight.detach())
        bs.append(lin.bias.detach())
    return torch.stack(ws), torch.stack(bs)


def _linear_engine(state_in, n_way, n_support, n_query, size, n_views, E):
    key = (id(state_in), n_way, n_support, n_query, size, n_views, E)
    e = _LIN_ENGINES.get(key)
    if e is None:
        if len(_LIN_ENGINES) >= 2:
            _LIN_ENGINES.clear()
        e = eng.FinetuneEngine(state_in, n_way, n_support, n_query, size, n_views=n_views, fine_tune_epoch=20,
                               episodes_per_batch=E, mode="linear")
        _LIN_ENGINES[key] = e
    return e


def _finetune_linear_frozen(x0, state_in, n_way, n_support, n_query, classifier):
    """finetune_linear(freeze_backbone=True) (finetune.py:45-174, frozen branch): eval-mode features are constants, only the
    Linear(512, n_way) classifier is trained -- 20 epochs x 5 mini-batches of 5 with Adam(lr .01, weight_decay .001), all 100
    steps in one launch (mft_linear_head_adam_run) -- and the scores are softmax(classifier(features of the queries))."""
    from . import ops
    feat = model_dict[params.model if params is not None else 'ResNet10'](flatten=True)
    feat.load_state_dict({k.replace("feature.", "", 1): v for k, v in state_in.items()
                          if k.startswith("feature.") and not k.startswith(("feature2.", "feature3."))})
    feat = feat.cuda().eval()
    support_size, batch_size, epochs = n_way * n_support, 5, 20
    with torch.no_grad():
        x = x0.cuda()
        za = feat(x[:, :n_support].reshape(support_size, *x.shape[2:])).float().contiguous()
        zb = feat(x[:, n_support:].reshape(n_way * n_query, *x.shape[2:])).float().contiguous()
    w0, b0 = classifier_init(n_way) if classifier is None else (torch.as_tensor(classifier[0]), torch.as_tensor(classifier[1]))
    W = w0.detach().clone().float().cuda().contiguous().view(1, n_way, -1)
    b = b0.detach().clone().float().cuda().contiguous().view(1, n_way)
    steps = []
    for _ in range(epochs):                                          # finetune.py:139-141: one permutation per epoch
        rand_id = np.random.permutation(support_size)
        for j in range(0, support_size, batch_size):
            ids = rand_id[j:min(j + batch_size, support_size)]
            steps.append(np.concatenate([ids, -np.ones(batch_size - len(ids), dtype=ids.dtype)]))
    table = torch.from_numpy(np.stack(steps).astype(np.int32)).cuda()
    y_dev = torch.from_numpy(np.repeat(np.arange(n_way), n_support).astype(np.int32)).cuda()
    D = za.shape[1]
    rc = ops._lib.lib().mft_linear_head_adam_run(ops._p(za), ops._p(y_dev), ops._p(table), 1, support_size, D, n_way, table.shape[0],
                                                 batch_size, ops._p(W), ops._p(b), 0.01, 0.9, 0.999, 1e-8, 0.001, ops._stream())
    ops._lib.check(rc, "mft_linear_head_adam_run")
    return torch.nn.functional.softmax(zb @ W[0].t() + b[0], dim=1)


def finetune_linear(liz_x, y, state_in, save_it, linear=False, flatten=True, n_query=15, ds=False,
                    pretrained_dataset='miniImageNet', freeze_backbone=False, n_way=5, n_support=5, classifier=None):
    """finetune.finetune_linear (finetune.py:45-174): the "baseline" branch of the README ensemble.  ``classifier`` =
    (w0 [n_way,512], b0 [n_way]) pins the initial Linear weights (default: torch's nn.Linear draw, as the reference)."""
    if not flatten:
        raise NotImplementedError("finetune_linear(): flatten=False is outside the HIP hot path")
    x0 = liz_x[0]
    n_query = x0.size(1) - n_support
    if freeze_backbone:
        return _finetune_linear_frozen(x0, state_in, n_way, n_support, n_query, classifier)
    e = _linear_engine(state_in, n_way, n_support, n_query, x0.size(-1), len(liz_x), 1)
    w0, b0 = classifier_init(n_way) if classifier is None else (torch.as_tensor(classifier[0]).view(1, n_way, -1),
                                                                 torch.as_tensor(classifier[1]).view(1, n_way))
    return e.run_batch([liz_x], classifier_init=(w0, b0))[0].clone()


def finetune_all(liz_x, y, model, state_baseline, state_gnn, n_way=5, n_support=5, classifier=None):
    """``--method all`` (finetune.py:634-649): scores_out = finetune_linear(baseline state) + finetune(gnnnet state);
    the numpy permutation stream is consumed in that order."""
    out = finetune_linear(liz_x, y, state_baseline, None, linear=True, n_way=n_way, n_support=n_support,
                          classifier=classifier)
    out = out + finetune(liz_x, y, model, state_gnn, 600, n_way=n_way, n_support=n_support)
    return out


def finetune_batched(episodes, model, state_in, fine_tune_epoch, n_way=5, n_support=5, episodes_per_batch=32,
                     perms=None):
    """Throughput path: ``episodes`` (list of liz_x) processed ``episodes_per_batch`` at a time in lockstep.
    Permutations are drawn episode by episode from the global numpy RNG (the reference's order) unless given."""
    model = model.cuda()
    x0 = episodes[0][0]
    n_query = x0.size(1) - n_support
    e = _engine_for(state_in, model, n_way, n_support, n_query, x0.size(-1), len(episodes[0]), fine_tune_epoch,
                    episodes_per_batch, fold50=getattr(model, "FOLD50", False))
    out = []
    for i in range(0, len(episodes), episodes_per_batch):
        chunk = episodes[i:i + episodes_per_batch]
        p = None if perms is None else perms[i:i + episodes_per_batch]
        out.append(e.run_batch(chunk, perms=p).clone())
    return torch.cat(out)


def evaluate(model, state, n_episodes, n_way, n_shot, n_query, size, gen_examples, fine_tune_epoch, seed0=0,
             episodes_per_batch=32, verbose=True):
    """600-episode loop of finetune.py:599-682 on synthetic episodes; returns per-episode accuracies."""
    y_query = np.repeat(range(n_way), n_query)
    accs = []
    for i in range(0, n_episodes, episodes_per_batch):
        eps = [[v.cuda() for v in synthetic.test_episode(seed0 + j, n_way, n_shot, n_query, size, gen_examples)]
               for j in range(i, min(i + episodes_per_batch, n_episodes))]
        for j, ep in enumerate(eps):
            assert torch.all(torch.eq(ep[0], ep[1]))                 # finetune.py:606
        sc = finetune_batched(eps, model, state, fine_tune_epoch, n_way, n_shot, episodes_per_batch)
        pred = sc.argmax(2).cpu().numpy()
        for p in pred:
            accs.append(float(np.mean(p == y_query)) * 100)
    accs = np.asarray(accs)
    if verbose:
        print('%d Test Acc = %4.2f%% +- %4.2f%%' % (len(accs), accs.mean(), 1.96 * accs.std() / np.sqrt(len(accs))))
    return accs


def main(argv=None):
    global params
    np.random.seed(10)                                               # finetune.py:425
    params = parse_args('train', argv)
    from .methods.gnnnet import GnnNet
    from .methods import gnnnet_copy
    if params.method not in ('gnnnet',):
        raise NotImplementedError("--method %s: only 'gnnnet' is on the HIP hot path" % params.method)
    size = int(__import__("os").environ.get("MFT_IMAGE_SIZE", "84"))
    cls = gnnnet_copy.GnnNet if params.n_shot == 50 else GnnNet
    model = cls(model_dict[params.model], n_way=params.test_n_way, n_support=params.n_shot).cuda()
    state = synthetic.gnnnet_state_dict(seed=0, n_way=params.test_n_way)   # no checkpoints offline
    model.load_state_dict(state)
    return evaluate(model, state, 600, params.test_n_way, params.n_shot, 15, size, params.gen_examples,
                    params.fine_tune_epoch)


if __name__ == '__main__':
    main()
