"""Mirror of the reference's args_util.py flag grammar (args_util.py:7-77)."""
from .data.voc_data_helpers import extract_img_data, get_img_names_from_set
from .train import optimizer_from_str  # noqa: F401  ('sgd' -> SGD(momentum 0.9), else Adam; args_util.py:48-59)


def base_paths_to_imgs(base_path_str, img_set="trainval", do_flip=True):
    """args_util.py:7-27: comma-separated VOC roots -> Image list (+ horizontally flipped copies)."""
    imgs = []
    for path in base_path_str.split(","):
        imgs.extend(extract_img_data(path, name) for name in get_img_names_from_set(path, img_set))
    if do_flip:
        imgs += [img.horizontal_flip() for img in imgs]
    return imgs


def phases_from_str(phases_str):
    """'60000:1e-3,20000:1e-4' -> [[60000, 1e-3], [20000, 1e-4]] (args_util.py:30-45)."""
    return [[int(p.split(":")[0]), float(p.split(":")[1])] for p in phases_str.split(",")]


def resize_dims_from_str(resize_dims_str):
    return [int(d) for d in resize_dims_str.split(",")]


def anchor_scales_from_str(anchor_scales_str):
    return [int(d) for d in anchor_scales_str.split(",")]
