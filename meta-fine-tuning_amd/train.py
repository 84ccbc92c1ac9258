"""Meta-training driver (mirror of train.py:26-207 for --method gnnnet): Adam over all parameters, 100
episodes per epoch, checkpoints ``{'epoch','state'}`` under <save_dir>/checkpoints/<dataset>/<model>_<method>
[_aug]_<n>way_<k>shot/<epoch>.tar, optional first-order-MAML meta-fine-tuning (--fine_tune).
The episode source is the in-repo synthetic miniImageNet-shaped sampler (real data is out of scope)."""
import os

import numpy as np
import torch

from . import configs, optim, synthetic
from .io_utils import get_assigned_file, model_dict, parse_args
from .methods.gnnnet import GnnNet


class SyntheticEpisodeLoader:
    """Stand-in for miniImageNet_few_shot.SetDataManager(...).get_data_loader(): ``n_episode`` episodes of
    [n_way, n_support+n_query, 3, size, size] (datasets/miniImageNet_few_shot.py:105-183)."""

    def __init__(self, n_way, n_support, n_query, size=84, n_episode=100, seed0=0):
        self.a = (n_way, n_support, n_query, size)
        self.n_episode, self.seed0, self.epoch = n_episode, seed0, 0

    def __len__(self):
        return self.n_episode

    def __iter__(self):
        base = self.seed0 + self.epoch * self.n_episode
        self.epoch += 1
        for i in range(self.n_episode):
            n_way, ns, nq, size = self.a
            yield synthetic.train_episode(base + i, n_way, ns, nq, size), None


class SyntheticBatchLoader:
    """Stand-in for SimpleDataManager(image_size, batch_size=16).get_data_loader(aug) (train.py:105-108;
    datasets/miniImageNet_few_shot.py:146-163): ``n_batch`` supervised mini-batches (x [B,3,size,size], y [B]) over
    ``num_classes`` class-structured synthetic classes."""

    def __init__(self, num_classes, size=84, batch_size=16, n_batch=100, seed0=0):
        self.num_classes, self.size, self.bs, self.n_batch, self.seed0, self.epoch = num_classes, size, batch_size, n_batch, seed0, 0
        rs = np.random.RandomState(seed0 + 12345)
        low = torch.from_numpy(rs.standard_normal((num_classes, 3, 7, 7)).astype(np.float32))
        self.templates = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)

    def __len__(self):
        return self.n_batch

    def __iter__(self):
        rs = np.random.RandomState(self.seed0 + self.epoch)
        self.epoch += 1
        for _ in range(self.n_batch):
            y = rs.randint(0, self.num_classes, size=self.bs)
            x = self.templates[torch.from_numpy(y)] + torch.from_numpy(rs.standard_normal((self.bs, 3, self.size, self.size)).astype(np.float32))
            yield x, torch.from_numpy(y)


def train(base_loader, model, optimization, start_epoch, stop_epoch, params):
    if optimization != 'Adam':
        raise ValueError('Unknown optimization, please define by yourself')
    optimizer = optim.Adam(model.parameters())          # torch.optim.Adam semantics, fused HIP update
    for epoch in range(start_epoch, stop_epoch):
        model.train()
        if params.method == 'baseline':
            model.train_loop(epoch, base_loader, optimizer)              # train.py:41-42: every other method -> train_loop
        elif not params.fine_tune:
            model.train_loop2(epoch, base_loader, optimizer)
        else:
            model.train_loop_finetune(epoch, base_loader, optimizer)
            if epoch == (stop_epoch - 1):
                model.MAML_update()
        if not os.path.isdir(params.checkpoint_dir):
            os.makedirs(params.checkpoint_dir)
        if (epoch % params.save_freq == 0) or (epoch == stop_epoch - 1):
            outfile = os.path.join(params.checkpoint_dir, '{:d}.tar'.format(epoch))
            torch.save({'epoch': epoch, 'state': model.state_dict()}, outfile)
    return model


def main(argv=None, n_episode=100, size=84):
    params = parse_args('train', argv)
    if not params.start_epoch > 0:
        np.random.seed(10)
    if params.method not in ('gnnnet', 'baseline'):
        raise NotImplementedError("--method %s: 'gnnnet' and 'baseline' are on the HIP path" % params.method)
    params.checkpoint_dir = '%s/checkpoints/%s/%s_%s' % (configs.save_dir, params.dataset, params.model, params.method)
    if params.train_aug:
        params.checkpoint_dir += '_aug'
    if params.method == 'baseline':                                      # train.py:101-108,176-180
        from .methods.baselinetrain import BaselineTrain
        base_loader = SyntheticBatchLoader(params.num_classes, size, 16, n_episode)
        model = BaselineTrain(model_dict[params.model], params.num_classes).cuda()
    else:
        n_query = max(1, int(16 * params.test_n_way / params.train_n_way))
        base_loader = SyntheticEpisodeLoader(params.train_n_way, params.n_shot, n_query, size, n_episode)
        model = GnnNet(model_dict[params.model], n_way=params.train_n_way, n_support=params.n_shot).cuda()
        params.checkpoint_dir += '_%dway_%dshot' % (params.train_n_way, params.n_shot)
    os.makedirs(params.checkpoint_dir, exist_ok=True)
    if params.start_epoch > 0:
        tmp = torch.load(get_assigned_file(params.checkpoint_dir, params.start_epoch - 1))
        state = {k: v for k, v in tmp['state'].items() if "feature2." not in k and "feature3." not in k}
        model.load_state_dict(state)
    return train(base_loader, model, 'Adam', params.start_epoch, params.stop_epoch, params)


if __name__ == '__main__':
    main()
