"""Class-name table for example_img_classify.py.  The 1000 ImageNet names are data that is not redistributed here:
put them, one per line, in `utils/imagenet_labels.txt` (or point PIO_IMAGENET_LABELS at such a file)."""
import os

_path = os.environ.get("PIO_IMAGENET_LABELS", os.path.join(os.path.dirname(__file__), "imagenet_labels.txt"))
if os.path.exists(_path):
    IMAGENET_LABELS = {i: line.strip() for i, line in enumerate(open(_path))}
else:
    IMAGENET_LABELS = {i: f"class_{i}" for i in range(1000)}
