from perceiverio_pytorch_amd.position_encoding import *  # noqa: F401,F403
