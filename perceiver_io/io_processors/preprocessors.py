from perceiverio_pytorch_amd.io_processors import (AudioPreprocessor, EmbeddingPreprocessor, ImagePreprocessor,  # noqa: F401
                                                   OneHotPreprocessor)
