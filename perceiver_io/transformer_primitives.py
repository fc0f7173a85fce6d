from perceiverio_pytorch_amd.transformer_primitives import *  # noqa: F401,F403
