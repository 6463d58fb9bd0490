"""The three backbone families the reference's scripts switch on with ``--network`` (train_rpn_step1.py:56-79 and the
other step scripts): which builders, preprocessing, conv-size rule, stride, regularisers and step-3 freeze list go
with ``vgg16`` / ``resnet50`` / ``resnet101``."""
from . import resnet, vgg
from .weights import synthetic_resnet, synthetic_vgg16

NETWORKS = ("vgg16", "resnet50", "resnet101")


class Family:
    def __init__(self, network):
        assert network in NETWORKS, network
        self.name = network
        if network == "vgg16":
            self.mod, self.depth = vgg, None
            self.base, self.rpn, self.classifier = vgg.vgg16_base, vgg.vgg16_rpn, vgg.vgg16_classifier
            self.freeze_all = [1, 2, 3, 4, 5]                   # train_rpn_step3.py:60
        else:
            self.mod, self.depth = resnet, 50 if network == "resnet50" else 101
            r101 = self.depth == 101
            self.base = resnet.resnet101_base if r101 else resnet.resnet50_base
            self.rpn = resnet.resnet101_rpn if r101 and hasattr(resnet, "resnet101_rpn") else resnet.resnet50_rpn
            self.classifier = resnet.resnet101_classifier if r101 else resnet.resnet50_classifier
            self.freeze_all = [1, 2, 3, 4]                      # train_rpn_step3.py:68,75
        self.preprocess, self.conv_dims, self.stride = self.mod.preprocess, self.mod.get_conv_rows_cols, self.mod.STRIDE
        self.weight_regularizer, self.bias_regularizer = self.mod.WEIGHT_REGULARIZER, self.mod.BIAS_REGULARIZER

    def synthetic_weights(self, anchors_per_loc, num_classes=21, seed=1):
        """Seeded random weights standing in for the ImageNet download the reference does (no network here)."""
        if self.depth is None:
            return synthetic_vgg16(anchors_per_loc=anchors_per_loc, num_classes=num_classes, seed=seed)
        return synthetic_resnet(self.depth, anchors_per_loc=anchors_per_loc, num_classes=num_classes, seed=seed)

    def base_kwargs(self, bf16=False):
        """dtype is a ResNet-only knob (mixed precision); VGG runs f32."""
        return {"dtype": "bf16"} if (bf16 and self.depth is not None) else {}
