"""50-shot meta-training driver (mirror of train_50.py): train.py with gnnnet_copy.GnnNet and the ``train_loop50`` /
``train_loop_finetune50`` loops when ``--n_shot 50`` (train_50.py:42-45,60-63,154-157) and a hard-coded checkpoint period of
10 epochs (train_50.py:53,66)."""
from . import train as _tr
from .train import SyntheticBatchLoader, SyntheticEpisodeLoader  # noqa: F401


def train(base_loader, model, optimization, start_epoch, stop_epoch, params):
    return _tr.train(base_loader, model, optimization, start_epoch, stop_epoch, params, variant50=True)


def main(argv=None, n_episode=100, size=84):
    return _tr.main(argv, n_episode=n_episode, size=size, variant50=True)


if __name__ == '__main__':
    main()
