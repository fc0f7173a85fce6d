"""Where does the non-hot-path time of the full classifier go? (dev tool, GPU)"""
import os, sys, time
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")]
import torch
import perceiverio_pytorch_amd as P
from perceiverio_pytorch_amd.models import ClassificationPerceiver
P.set_precision_policy("fp16")
dev = torch.device("cuda:0")
m = ClassificationPerceiver(precision_policy="fp16").to(dev).eval()
x = torch.randn(32, 3, 224, 224, device=dev)
pio = m.perceiver
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.inference_mode():
    prep = pio._multi_preprocessor._preprocessors["__default"]
    print("convnet           %.2f ms" % t(lambda: prep.convnet(x)))
    print("preprocessor      %.2f ms" % t(lambda: prep(x)))
    print("multi_preproc     %.2f ms" % t(lambda: pio._multi_preprocessor({"__default": x})))
    xin, sizes, wo = pio._multi_preprocessor({"__default": x})
    print("decoder_query     %.2f ms" % t(lambda: pio.decoder_query(xin, sizes, wo)))
    lat = pio._encoder.latents(xin)
    print("encoder           %.2f ms" % t(lambda: pio._encoder(xin, lat)))
    z = pio._encoder(xin, lat); q, _ = pio.decoder_query(xin, sizes, wo)
    print("decoder           %.2f ms" % t(lambda: pio._decoder(q, z)))
    print("full model        %.2f ms" % t(lambda: m(x)))
