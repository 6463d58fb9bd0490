"""Entry point mirroring the reference's train_rpn_step1.py (train_rpn_step1.py:11-94): step 1 of the
4-step alternating training -- train the RPN.  Same flags; add `torchrun --nproc-per-node N` (or
`python -m torch.distributed.run`) in front for N-GPU data parallelism (1 image per GPU per step, one
RCCL all-reduce of the flat gradient buffer).

    python -m faster_rcnn_amd.train_rpn_step1 --voc_paths /data/VOC2007 --phases 60000:1e-3,20000:1e-4 \\
           --optimizer sgd --network resnet50 --anchor_scales 128,256,512
"""
import argparse

from . import dp
from ._networks import NETWORKS, Family
from .args_util import anchor_scales_from_str, base_paths_to_imgs, optimizer_from_str, phases_from_str, resize_dims_from_str
from .rpn_util import RpnTrainingManager
from .train_util import train_rpn
from .util import get_anchors, resize_imgs


def build_parser():
    p = argparse.ArgumentParser(description="Train the region proposal network (step 1)")
    p.add_argument("--voc_paths", dest="voc_paths", required=True, help="comma-separated VOC-style dataset roots")
    p.add_argument("--phases", dest="phases", default="60000:1e-3,20000:1e-4")
    p.add_argument("--optimizer", dest="optimizer", choices=("adam", "sgd"), default="sgd")
    p.add_argument("--img_set", dest="img_set", choices=("train", "val", "trainval", "test"), default="trainval")
    p.add_argument("--network", dest="network", choices=NETWORKS, default="vgg16")
    p.add_argument("--resize_dims", dest="resize_dims", default="600,1000")
    p.add_argument("--anchor_scales", dest="anchor_scales", default="128,256,512")
    p.add_argument("--save_weights_dest", dest="save_weights_dest", default=None)
    p.add_argument("--save_model_dest", dest="save_model_dest", default=None)
    p.add_argument("--init_weights", dest="init_weights", default=None, help=".npz keyed by Keras layer names (default: seeded synthetic)")
    p.add_argument("--bf16", action="store_true",
                   help="mixed precision (not in the reference): bf16 activations / gradients / packed filters, f32 master weights and optimiser")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    dp.init_from_env()
    train_imgs = base_paths_to_imgs(args.voc_paths, img_set=args.img_set)
    resize_min, resize_max = resize_dims_from_str(args.resize_dims)
    anchors = get_anchors(anchor_scales_from_str(args.anchor_scales))
    processed_imgs, _ = resize_imgs(train_imgs, min_size=resize_min, max_size=resize_max)
    net = Family(args.network)
    from .weights import load_npz
    weights = load_npz(args.init_weights) if args.init_weights else net.synthetic_weights(len(anchors))
    base_model = net.base(weight_regularizer=net.weight_regularizer, bias_regularizer=net.bias_regularizer, weights=weights,
                          **net.base_kwargs(args.bf16))
    rpn_model = net.rpn(base_model, weight_regularizer=net.weight_regularizer, bias_regularizer=net.bias_regularizer,
                        anchors_per_loc=len(anchors))
    save_weights_dest = args.save_weights_dest or "models/rpn_weights_{}_step1.h5".format(args.network)
    save_model_dest = args.save_model_dest or "models/rpn_model_{}_step1.h5".format(args.network)
    manager = RpnTrainingManager(net.conv_dims, net.stride, preprocess_func=net.preprocess, anchor_dims=anchors)
    train_rpn(rpn_model, processed_imgs, manager, optimizer_from_str(args.optimizer), phases=phases_from_str(args.phases),
              save_frequency=2000, save_weights_dest=save_weights_dest, save_model_dest=save_model_dest)
    if dp.rank() == 0:
        rpn_model.save_weights(save_weights_dest)
        print("Saved {} rpn weights to {}".format(args.network, save_weights_dest))
        rpn_model.save(save_model_dest)
        print("Saved {} rpn model to {}".format(args.network, save_model_dest))


if __name__ == "__main__":
    main()
