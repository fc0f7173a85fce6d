"""Class-name table for example_multimodal.py; see utils/imagenet_labels.py for the convention."""
import os

_path = os.environ.get("PIO_KINETICS_LABELS", os.path.join(os.path.dirname(__file__), "kinetics_700_classes.txt"))
if os.path.exists(_path):
    KINETICS_CLASSES = {i: line.strip() for i, line in enumerate(open(_path))}
else:
    KINETICS_CLASSES = {i: f"class_{i}" for i in range(700)}
