from perceiverio_pytorch_amd.io_processors import (AudioPostprocessor, ClassificationPostprocessor,  # noqa: F401
                                                   EmbeddingPostprocessor, FlowPostprocessor, IdentityPostprocessor,
                                                   ImagePostprocessor, ProjectionPostprocessor)
