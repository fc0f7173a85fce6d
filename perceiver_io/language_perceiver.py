from perceiverio_pytorch_amd.models import LanguagePerceiver  # noqa: F401
