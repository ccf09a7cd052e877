"""Dev aid: checkerboard split / merge kernels against the HBM roofline (algorithmic bytes = read + write of the tensor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd.ops import ckbd_embed, ckbd_unembed
for shape, dt in [((1, 192, 32, 48), torch.float32), ((1, 2304, 32, 48), torch.float32), ((24, 2304, 32, 48), torch.float32),
                  ((1, 320, 136, 240), torch.float32), ((1, 3840, 136, 240), torch.float16), ((8, 3840, 136, 240), torch.float16)]:
    y = torch.randn(shape, device="cuda").to(dt)
    for name, fn, arg in (("unembed", ckbd_unembed, y), ("embed", ckbd_embed, ckbd_unembed(y))):
        for _ in range(3): fn(arg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn(arg)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        b = 2 * y.numel() * y.element_size()
        print(f"{name:8s} {str(tuple(shape)):22s} {str(dt)[6:]:8s} {b/1e6:9.1f} MB  {ms*1e3:8.1f} us  {b/ms/1e6:7.1f} GB/s  ({b/ms/1e6/8000:.3f} of 8 TB/s; incl. torch.empty + launch)")
