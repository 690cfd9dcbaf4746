"""A few 64-image passes of the IHMR-Baseline encoder and nothing else (the program `scripts/prof_encoder_layers.sh` traces)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ihmr_amd.networks import InterHandEncoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
enc = InterHandEncoder(types.SimpleNamespace(total_params_dim=122), torch.zeros(B, 122)).cuda()
img = torch.rand(B, 3, 224, 224, device="cuda") * 2 - 1
for _ in range(6):
    enc(img)
torch.cuda.synchronize()
