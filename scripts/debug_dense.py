import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ihmr_amd import hip
from ihmr_amd.assets import synthetic_mano
right, left = synthetic_mano(True), synthetic_mano(False)
v = torch.tensor(right['v_template']); l = v.clone(); l[:,0] = -l[:,0]; l = l + torch.tensor([0.15, 0.0, 0.02])
hv = torch.stack([v, l])[None].contiguous().cuda()
B = 1
fr = torch.tensor(right['faces'].astype(np.int32)).cuda(); fl = torch.tensor(left['faces'].astype(np.int32)).cuda()
phi = torch.full((B,2,32,32,32), -7.0, device='cuda')
nbytes = hip.lib().ihmr_sdf_workspace_bytes(B)
ws = torch.zeros(nbytes, dtype=torch.uint8, device='cuda')
rc = hip.lib().ihmr_sdf_dense_grid(hip.ptr(fr), hip.ptr(fl), hip.ptr(hv), B, hip.ptr(phi), hip.ptr(ws), hip.stream_ptr())
torch.cuda.synchronize()
print('rc', rc, 'phi min/max', phi.min().item(), phi.max().item(), 'n>0', (phi>0).sum().item(), 'n==-7', (phi==-7).sum().item())
H=2
off_box=0; off_tri=H*16; off_phi=off_tri+H*20*1600*4; off_needed=off_phi+H*32768*4; off_cnt=off_needed+H*1024*4
w = ws.cpu().numpy()
print('box', np.frombuffer(w[off_box:off_box+32].tobytes(), np.float32))
print('col_count', np.frombuffer(w[off_cnt:off_cnt+8].tobytes(), np.int32))
nd = np.frombuffer(w[off_needed:off_needed+H*4096].tobytes(), np.uint32)
print('needed nonzero', (nd!=0).sum(), 'first', nd[:4])
p2 = np.frombuffer(w[off_phi:off_phi+H*32768*4].tobytes(), np.float32)
print('ws.phi n>0', (p2>0).sum())
