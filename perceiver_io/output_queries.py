from perceiverio_pytorch_amd.output_queries import *  # noqa: F401,F403
