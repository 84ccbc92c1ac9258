"""CLI flags and checkpoint lookup -- same flag names, defaults and return values as the reference's
io_utils (io_utils.py:7-69), restated table-driven."""
import argparse
import glob
import os

import numpy as np

from . import backbone

model_dict = dict(ResNet10=backbone.ResNet10)      # ResNet10_FW / ResNet18 are off the hot path

# (flag, kwargs) for every script
_COMMON = [
    ("--dataset", dict(default="miniImagenet", help="training base model")),
    ("--test_dataset", dict(default="", help="test dataset")),
    ("--unsupervised", dict(default="", help="unsupervised dataset")),
    ("--model", dict(default="ResNet10", help="backbone architecture")),
    ("--method", dict(default="baseline", help="baseline/protonet/gnnnet/all")),
    ("--train_n_way", dict(default=5, type=int, help="class num to classify for training")),
    ("--test_n_way", dict(default=5, type=int, help="class num to classify for testing (validation)")),
    ("--n_shot", dict(default=5, type=int, help="number of labeled data in each class, same as n_support")),
    ("--train_aug", dict(action="store_true", help="perform data augmentation or not during training")),
    ("--both", dict(action="store_true", help="use both tuned and untuned model")),
    ("--freeze_backbone", dict(action="store_true", help="Freeze the backbone network for finetuning")),
    ("--save_iter", dict(default=-1, type=int, help="checkpoint epoch to load; best model if -1")),
    ("--fine_tune_all_models", dict(action="store_true", help="fine-tune each model before selection")),
    ("--fine_tune_epoch", dict(default=100, type=int, help="number of epochs to finetune")),
    ("--gen_examples", dict(default=10, type=int, help="number of examples to generate (data augmentation)")),
]
_PER_SCRIPT = {
    "train": [
        ("--fine_tune", dict(action="store_true", help="fine tuning during training")),
        ("--num_classes", dict(default=200, type=int, help="total number of classes in softmax (baseline only)")),
        ("--save_freq", dict(default=50, type=int, help="Save frequency")),
        ("--start_epoch", dict(default=0, type=int, help="Starting epoch")),
        ("--stop_epoch", dict(default=400, type=int, help="Stopping epoch")),
        # (not in the reference: k episodes per optimizer step in lockstep on one GPU = the update of a k-rank episode-parallel run)
        ("--episodes_per_rank", dict(default=1, type=int, help="gnnnet without --fine_tune: episodes per optimizer step (lockstep)")),
    ],
    "save_features": [("--split", dict(default="novel", help="base/val/novel"))],
    "test": [
        ("--split", dict(default="novel", help="base/val/novel")),
        ("--adaptation", dict(action="store_true", help="further adaptation in test time or not")),
        ("--unsup", dict(action="store_true", help="unsupervised learning or not")),
        ("--unsup_cluster", dict(action="store_true", help="unsupervised learning with clustering or not")),
    ],
}


def build_parser(script):
    if script not in _PER_SCRIPT:
        raise ValueError("Unknown script")
    parser = argparse.ArgumentParser(description="few-shot script %s" % script)
    for flag, kw in _COMMON:
        parser.add_argument(flag, **kw)
    parser.add_argument("--models_to_use", "--names-list", nargs="+",
                        default=["miniImageNet", "caltech256", "DTD", "cifar100", "CUB"], help="pretained model to use")
    for flag, kw in _PER_SCRIPT[script]:
        parser.add_argument(flag, **kw)
    return parser


def parse_args(script, argv=None):
    return build_parser(script).parse_args(argv)


def get_assigned_file(checkpoint_dir, num):
    return os.path.join(checkpoint_dir, "{:d}.tar".format(num))


def get_resume_file(checkpoint_dir):
    files = [f for f in glob.glob(os.path.join(checkpoint_dir, "*.tar")) if os.path.basename(f) != "best_model.tar"]
    if not files:
        return None if not glob.glob(os.path.join(checkpoint_dir, "*.tar")) else None
    epochs = np.array([int(os.path.splitext(os.path.basename(f))[0]) for f in files])
    return os.path.join(checkpoint_dir, "{:d}.tar".format(int(epochs.max())))


def get_best_file(checkpoint_dir):
    best = os.path.join(checkpoint_dir, "best_model.tar")
    return best if os.path.isfile(best) else get_resume_file(checkpoint_dir)
