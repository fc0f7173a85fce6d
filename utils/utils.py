"""Small host-side helpers the reference's examples import from utils.utils."""
import pickle

import numpy as np
import torch

from perceiverio_pytorch_amd.io_processors import same_padding, unravel_index  # noqa: F401


def dump_pickle(obj, file_path):
    with open(file_path, "wb") as f:
        pickle.dump(obj, f)


def load_pickle(file_path):
    with open(file_path, "rb") as f:
        return pickle.load(f)


def load_image(imfile, device):
    """Image file -> float tensor [1, 3, H, W] (0..255) on `device`."""
    from PIL import Image
    img = torch.from_numpy(np.array(Image.open(imfile)).astype(np.uint8)).permute(2, 0, 1).float()
    return img[None].to(device)


def show_animation(images: np.ndarray, fps: int = 25, title: str = "animation"):
    import matplotlib.pyplot as plt
    from matplotlib.animation import ArtistAnimation
    fig = plt.figure(title)
    frames = [[plt.imshow(im, animated=True)] for im in images]
    _anim = ArtistAnimation(fig, frames, interval=1000 / fps, blit=True, repeat_delay=1000)  # noqa: F841
    plt.show()


def conv_output_shape(input_size, kernel_size, stride=1, padding=0, dilation=1, dims: int = 2):
    as_list = lambda v: [v] * dims if isinstance(v, int) else list(v)  # noqa: E731
    k, s, p, d = as_list(kernel_size), as_list(stride), as_list(padding), as_list(dilation)
    lead = len(input_size) - dims
    return list(input_size[:lead]) + [(input_size[lead + i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1
                                      for i in range(dims)]
