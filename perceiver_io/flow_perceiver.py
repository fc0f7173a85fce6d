from perceiverio_pytorch_amd.models import FlowPerceiver  # noqa: F401
