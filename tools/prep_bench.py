"""Dev tool (GPU): the Conv2DDownsample tail -- one HIP pass (pio_bn_relu_maxpool_tokens) against the torch ops."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT]
import torch  # noqa: E402
from perceiverio_pytorch_amd.io_processors import Conv2DDownsample  # noqa: E402

dev = torch.device("cuda:0")
net = Conv2DDownsample(num_layers=1, num_channels=64, use_batchnorm=True).to(dev).eval()
x = torch.randn(32, 3, 224, 224, device=dev)


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    def torch_path():
        y = net(x)
        return y.movedim(1, -1).reshape(y.shape[0], -1, y.shape[1]).contiguous()
    print(f"conv + torch ops (BatchNorm, ReLU, pad + max-pool, channels-last copy): {timed(torch_path):8.1f} us")
    print(f"conv + pio_bn_relu_maxpool_tokens:                                       {timed(lambda: net.forward_tokens(x)):8.1f} us")
    import torch.nn.functional as F
    from perceiverio_pytorch_amd.io_processors import same_padding
    conv = net.convs[0]
    print(f"conv alone:                                                              {timed(lambda: conv(F.pad(x, same_padding(x.shape[1:], conv.kernel_size, conv.stride)))):8.1f} us")
