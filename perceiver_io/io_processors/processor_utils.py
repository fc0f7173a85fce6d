from perceiverio_pytorch_amd.io_processors import (Conv2DDownsample, extract_patches, patches_for_flow,  # noqa: F401
                                                   reverse_space_to_depth, space_to_depth)
