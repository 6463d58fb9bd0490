"""Entry point mirroring the reference's train_rpn_step3.py (train_rpn_step3.py:11-103): step 3 -- fine-tune
the RPN heads on top of the step-2 detector's base, every base block frozen and unregularised.

    python -m faster_rcnn_amd.train_rpn_step3 --step2_weights_path models/detector_weights_resnet50_step2.npz --voc_paths ...
"""
from . import dp
from ._networks import Family
from .args_util import anchor_scales_from_str, base_paths_to_imgs, optimizer_from_str, phases_from_str, resize_dims_from_str
from .rpn_util import RpnTrainingManager
from .train_rpn_step1 import build_parser as _step1_parser
from .train_util import train_rpn
from .util import get_anchors, resize_imgs


def build_parser():
    p = _step1_parser()
    p.description = "Fine-tune the region proposal network with a frozen base (step 3)"
    p.add_argument("--step2_weights_path", dest="step2_weights_path", default=None,
                   help="detector weights from step 2; layers are matched by name (load_weights(by_name=True))")
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
    rpn_base = net.base(freeze_blocks=net.freeze_all, weights=weights)                   # train_rpn_step3.py:60,68,75
    rpn_model = net.rpn(rpn_base, weight_regularizer=net.weight_regularizer, bias_regularizer=net.bias_regularizer,
                        anchors_per_loc=len(anchors))
    if args.step2_weights_path is not None:
        rpn_model.load_weights(args.step2_weights_path, by_name=True)
    save_weights_dest = args.save_weights_dest or "models/rpn_weights_{}_step3.h5".format(args.network)
    manager = RpnTrainingManager(net.conv_dims, net.stride, preprocess_func=net.preprocess, anchor_dims=anchors)
    train_rpn(rpn_model, processed_imgs, manager, optimizer_from_str(args.optimizer), phases=phases_from_str(args.phases),
              save_frequency=2000, save_weights_dest=save_weights_dest, save_model_dest=args.save_model_dest)
    if dp.rank() == 0:
        rpn_model.save_weights(save_weights_dest)
        print("Saved {} rpn weights to {}".format(args.network, save_weights_dest))


if __name__ == "__main__":
    main()
