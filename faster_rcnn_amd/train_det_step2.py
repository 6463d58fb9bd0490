"""Entry point mirroring the reference's train_det_step2.py (train_det_step2.py:13-119): step 2 --
train the detector (own base + stage-5 head) on proposals from the frozen step-1 RPN.  Same flags;
prefix with torchrun for N-GPU data parallelism.

    python -m faster_rcnn_amd.train_det_step2 models/rpn_weights_resnet50_step1.npz --voc_paths /data/VOC2007
"""
import argparse

from . import dp
from ._networks import NETWORKS, Family
from .args_util import anchor_scales_from_str, base_paths_to_imgs, optimizer_from_str, phases_from_str, resize_dims_from_str
from .data.voc_data_helpers import KITTI_CLASS_MAPPING, VOC_CLASS_MAPPING
from .det_util import DetTrainingManager
from .shared_constants import NUM_ROIS
from .train_util import train_detector_step2
from .util import get_anchors, resize_imgs


def build_parser():
    p = argparse.ArgumentParser(description="Train the Fast R-CNN detector (step 2)")
    p.add_argument("step1_weights_path", type=str, help="RPN weights saved by step 1")
    p.add_argument("--voc_paths", dest="voc_paths", required=True)
    p.add_argument("--phases", dest="phases", default="60000:1e-3,20000:1e-4")
    p.add_argument("--optimizer", dest="optimizer", choices=("adam", "sgd"), default="sgd")
    p.add_argument("--kitti", dest="kitti", action="store_true")
    p.add_argument("--img_set", dest="img_set", choices=("train", "val", "trainval", "test"), default="trainval")
    p.add_argument("--network", dest="network", choices=NETWORKS, default="vgg16")
    p.add_argument("--resize_dims", dest="resize_dims", default="600,1000")
    p.add_argument("--anchor_scales", dest="anchor_scales", default="128,256,512")
    p.add_argument("--save_weights_dest", dest="save_weights_dest", default=None)
    p.add_argument("--save_model_dest", dest="save_model_dest", default=None)
    p.add_argument("--init_weights", dest="init_weights", default=None)
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
    class_mapping = KITTI_CLASS_MAPPING if args.kitti else VOC_CLASS_MAPPING
    num_classes = len(class_mapping)
    net = Family(args.network)
    from .weights import load_npz
    rpn_weights = load_npz(args.step1_weights_path)
    rpn_model = net.rpn(net.base(weights=rpn_weights), anchors_per_loc=len(anchors))        # frozen, not regularised
    det_weights = load_npz(args.init_weights) if args.init_weights else net.synthetic_weights(len(anchors), num_classes)
    detector_base = net.base(weight_regularizer=net.weight_regularizer, bias_regularizer=net.bias_regularizer, weights=det_weights,
                             **net.base_kwargs(args.bf16))
    detector_model = net.classifier(NUM_ROIS, num_classes, detector_base, weight_regularizer=net.weight_regularizer,
                                    bias_regularizer=net.bias_regularizer)
    save_weights_dest = args.save_weights_dest or "models/detector_weights_{}_step2.h5".format(args.network)
    save_model_dest = args.save_model_dest or "models/detector_model_{}_step2.h5".format(args.network)
    manager = DetTrainingManager(rpn_model=rpn_model, class_mapping=class_mapping, preprocess_func=net.preprocess,
                                 stride=net.stride, anchor_dims=anchors)
    train_detector_step2(detector_model, processed_imgs, manager, optimizer_from_str(args.optimizer), phases=phases_from_str(args.phases),
                         save_frequency=2000, save_weights_dest=save_weights_dest, save_model_dest=save_model_dest)
    if dp.rank() == 0:
        detector_model.save_weights(save_weights_dest)
        print("Saved {} detector weights to {}".format(args.network, save_weights_dest))


if __name__ == "__main__":
    main()
