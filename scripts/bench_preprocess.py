#!/usr/bin/env python3
"""Timing of the batched preprocessing kernel (ihmr_preprocess_images): 64 random hand crops -> (64,3,224,224).
HBM-bound byte work: algorithmic bytes = source bytes + 224*224*3*4 output bytes per image."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ihmr_amd.preprocess import DataProcessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.RandomState(0)
shapes = [(int(rng.randint(200, 600)), int(rng.randint(200, 600))) for _ in range(B)]
images = [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in shapes]
proc = DataProcessor(final_size=224)
buf, off, sz = proc.pack(images)
buf, off, sz = buf.cuda(), off.cuda(), sz.cuda()
for _ in range(5):
    proc.preprocess_packed(buf, off, sz)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    proc.preprocess_packed(buf, off, sz)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
src = sum(im.size for im in images)
# bytes actually touched: a down-scaled image reads at most 4 source pixels per output pixel
out_bytes = B * 224 * 224 * 3 * 4
print(f"B={B}: {ms * 1e3:.1f} us per batch ({B / ms * 1e3:.0f} images/s); source {src / 1e6:.1f} MB + output {out_bytes / 1e6:.1f} MB "
      f"-> {(src + out_bytes) / ms / 1e6:.0f} GB/s algorithmic")
