"""Entry point mirroring the reference's train_det_step4.py (train_det_step4.py:13-121): step 4 -- train the
detector head on conv features of the (frozen) step-3 RPN network; the RPN model is saved with the conv
map as third output for inference (voc_dets needs it, train_det_step4.py:80, 121).

    python -m faster_rcnn_amd.train_det_step4 models/rpn_weights_resnet50_step3.npz --voc_paths ...
"""
from . import dp
from ._networks import Family
from .args_util import anchor_scales_from_str, base_paths_to_imgs, optimizer_from_str, phases_from_str, resize_dims_from_str
from .data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
from .det_util import DetTrainingManager
from .shared_constants import NUM_ROIS
from .train_det_step2 import build_parser as _step2_parser
from .train_util import train_detector_step4
from .util import get_anchors, resize_imgs


def build_parser():
    p = _step2_parser()
    p.description = "Train the Fast R-CNN detector head on shared conv features (step 4)"
    p.add_argument("--save_rpn_model_dest", dest="save_rpn_model_dest", default=None)
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    dp.init_from_env()
    train_imgs = base_paths_to_imgs(args.voc_paths, img_set=args.img_set)
    resize_min, resize_max = resize_dims_from_str(args.resize_dims)
    anchors = get_anchors(anchor_scales_from_str(args.anchor_scales))
    processed_imgs, _ = resize_imgs(train_imgs, min_size=resize_min, max_size=resize_max)
    class_mapping = KITTI_CLASS_MAPPING if args.kitti else VOC_CLASS_MAPPING
    num_classes = len(class_mapping)
    net = Family(args.network)
    from .weights import load_npz
    rpn_model = net.rpn(net.base(weights=load_npz(args.step1_weights_path)), include_conv=True, anchors_per_loc=len(anchors))
    det_weights = load_npz(args.init_weights) if args.init_weights else net.synthetic_weights(len(anchors), num_classes)
    detector_model = net.classifier(NUM_ROIS, num_classes, weight_regularizer=net.weight_regularizer, bias_regularizer=net.bias_regularizer,
                                    weights=det_weights)
    save_weights_dest = args.save_weights_dest or "models/detector_weights_{}_step4.h5".format(args.network)
    manager = DetTrainingManager(rpn_model=rpn_model, class_mapping=class_mapping, preprocess_func=net.preprocess,
                                 stride=net.stride, anchor_dims=anchors)
    train_detector_step4(detector_model, processed_imgs, manager, optimizer_from_str(args.optimizer), phases=phases_from_str(args.phases),
                         save_frequency=2000, save_weights_dest=save_weights_dest, save_model_dest=args.save_model_dest)
    if dp.rank() == 0:
        detector_model.save_weights(save_weights_dest)
        rpn_model.save(args.save_rpn_model_dest or "models/rpn_model_{}_step3.h5".format(args.network))
        print("Saved {} detector weights to {}".format(args.network, save_weights_dest))


if __name__ == "__main__":
    main()
