"""Class maps and the VOC annotation loader (data/voc_data_helpers.py:10-138), OpenCV-free."""
import os
from xml.etree import ElementTree

from ..shapes import Box, GroundTruthBox, Image, Metadata

VOC_CLASS_MAPPING = {name: i for i, name in enumerate(
    ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
     "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor", "bg"])}
KITTI_CLASS_MAPPING = {name: i for i, name in enumerate(
    ["car", "person", "Cyclist", "DontCare", "Misc", "Person_sitting", "Tram", "Truck", "Van", "bg"])}

IMAGES_DIR, ANNOTATIONS_DIR, IMAGESETS_DIR = "JPEGImages", "Annotations", os.path.join("ImageSets", "Main")


def extract_img_metadata(base_path, img_num):
    """voc_data_helpers.py:68-122 (annotation coords are shifted -1 on load, :111-114)."""
    xml = ElementTree.parse(os.path.join(base_path, ANNOTATIONS_DIR, img_num + ".xml")).getroot()
    image_path = os.path.join(base_path, IMAGES_DIR, xml.find("filename").text)
    size = xml.find("size")
    width, height = int(size.find("width").text), int(size.find("height").text)
    gt_boxes = []
    for obj in xml.findall("object"):
        bb = obj.find("bndbox")
        xmin, xmax, ymin, ymax = (int(float(bb.find(k).text)) - 1 for k in ("xmin", "xmax", "ymin", "ymax"))
        gt_boxes.append(GroundTruthBox(obj_cls=obj.find("name").text, difficult=int(obj.find("difficult").text) == 1,
                                       box=Box(xmin, ymin, xmax, ymax)))
    return Metadata(img_num, width=width, height=height, gt_boxes=gt_boxes, image_path=image_path)


def extract_img_data(base_path, img_num):
    return Image(metadata=extract_img_metadata(base_path, img_num))


def get_img_names_from_set(base_path, set_name):
    with open(os.path.join(base_path, IMAGESETS_DIR, set_name + ".txt")) as f:
        return [line.rstrip("\n") for line in f]
