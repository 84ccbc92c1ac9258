"""50-shot test-time driver (mirror of finetune_50.py:48-748).

The reference's finetune_50.py is finetune.py with ``GnnNet`` imported from methods/gnnnet_copy (finetune_50.py:20) and one
dead ``classifier(output)`` call (:294); ``finetune`` / ``finetune_linear`` are otherwise the same functions, and the module
keeps its OWN ``params`` global (:185,261 read ``finetune_50.params``).  Here both modules share one implementation
(finetune._finetune / _finetune_linear); this module binds its own ``params`` and the gnnnet_copy model class.

The drivers pass the TRUE support count (``n_support=50``, finetune_50.py:548,619) while gnnnet_copy.GnnNet stores
round(50/2)=25 graph nodes per class (gnnnet_copy.py:34); the engine folds support k with support k+25 before the GNN
(gnnnet_copy.py:67-72 -> mft_build_graph_nodes(fold=1)).
"""
import numpy as np  # noqa: F401

from . import finetune as _ft
from .finetune import (LookaheadLoader, SyntheticNovelLoader, classifier_init, draw_episode_perms, evaluate,  # noqa: F401
                       finetune_batched, finetune_linear_batched, scores_batched)
from .io_utils import model_dict, parse_args  # noqa: F401
from .methods.gnnnet_copy import GnnNet  # noqa: F401  (finetune_50.py:20)

params = None


def finetune(liz_x, y, model, state_in, save_it, linear=False, flatten=True, n_query=15, ds=False,
             pretrained_dataset='miniImageNet', freeze_backbone=False, n_way=5, n_support=5):
    """finetune_50.py:182-330."""
    return _ft._finetune(params, liz_x, y, model, state_in, save_it, linear, flatten, n_query, ds, pretrained_dataset,
                         freeze_backbone, n_way, n_support)


def finetune_linear(liz_x, y, state_in, save_it, linear=False, flatten=True, n_query=15, ds=False,
                    pretrained_dataset='miniImageNet', freeze_backbone=False, n_way=5, n_support=5, classifier=None):
    """finetune_50.py:48-176."""
    return _ft._finetune_linear(params, liz_x, y, state_in, save_it, linear, flatten, n_query, ds, pretrained_dataset,
                                freeze_backbone, n_way, n_support, classifier)


def main(argv=None, n_episodes=600, episodes_per_batch=None):
    """finetune_50.py:424-748: the same driver with gnnnet_copy.GnnNet for every ``--n_shot``."""
    global params
    accs = _ft.main(argv, model_cls=GnnNet, n_episodes=n_episodes, episodes_per_batch=episodes_per_batch)
    params = _ft.params
    return accs


if __name__ == '__main__':
    main()
