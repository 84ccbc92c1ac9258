"""BaselineFinetune: frozen features + a freshly trained linear head (mirror of methods/baselinefinetune.py:9-61)."""
from .meta_template import MetaTemplate


class BaselineFinetune(MetaTemplate):
    def __init__(self, model_func, n_way, n_support, loss_type="softmax"):
        super().__init__(model_func, n_way, n_support)
        if loss_type != "softmax":
            raise NotImplementedError("loss_type='dist' (distLinear) is off the hot path")
        self.loss_type = loss_type

    def set_forward(self, x, is_feature=True):
        return self.set_forward_adaptation(x, is_feature)       # Baseline always adapts

    def set_forward_loss(self, x):
        raise ValueError('Baseline predict on pretrained feature and do not support finetune backbone')
