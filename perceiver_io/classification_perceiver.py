from perceiverio_pytorch_amd.models import ClassificationPerceiver, PrepType  # noqa: F401
