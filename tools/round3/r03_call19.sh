#!/bin/bash
set -u
bash tools/round3/ab.sh base
bash tools/round3/ab.sh bwd_no_atomics KAMD_BWD_MODE=1
bash tools/round3/ab.sh bwd_no_gathers KAMD_BWD_MODE=2
bash tools/round3/ab.sh bwd_neither KAMD_BWD_MODE=3
bash tools/round3/ab.sh bwd_per_cu8 KAMD_SOFT_BWD_PER_CU=8
bash tools/round3/ab.sh bwd_per_cu32 KAMD_SOFT_BWD_PER_CU=32
python - <<'P'
import torch, kaolin_amd as kal
from kaolin_amd.utils import testing as T
V,H,W=8,1024,1024
fz,fimg,feats,nz=T.sphere_scene(level=50,num_views=V,device='cuda')
feat=torch.cat(feats,-1).contiguous()
a=fimg.clone().requires_grad_()
out,soft,idx=kal.render.mesh.dibr_rasterization(H,W,fz,a,feat,nz)
sv=out.grad_fn.saved_tensors
work=sv[-1]
M=kal._C.render.mesh
items=M.work_items(work,V,H,W)
counts = work[M.WORK_FLAT_WORD:M.WORK_FLAT_WORD + M.FLAT_SHARDS*M.COUNTER_STRIDE:M.COUNTER_STRIDE]
print('work items', items.numel(), 'flat records', int(counts.sum()), 'per shard max', int(counts.max()), 'covered tiles', M.covered_tiles(work,V,H,W).numel())
hc = None
P
