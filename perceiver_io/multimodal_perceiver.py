from perceiverio_pytorch_amd.models import MultiModalPerceiver  # noqa: F401
