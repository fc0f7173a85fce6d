from perceiverio_pytorch_amd.perceiver import *  # noqa: F401,F403
