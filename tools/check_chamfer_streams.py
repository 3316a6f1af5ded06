"""Three streams issuing the fused chamfer operator (persistent grid-build kernel with grid barriers) concurrently: results
must equal the single-stream ones and nothing may hang.  Run under `timeout`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kaolin_amd as kal
pc = kal.metrics.pointcloud
g = torch.Generator().manual_seed(0)
clouds = [(torch.rand(1, 60000, 3, generator=g).cuda(), torch.rand(1, 50000, 3, generator=g).cuda()) for _ in range(3)]
ref = [pc.chamfer_distance(a, b) for a, b in clouds]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in clouds]
for it in range(30):
    outs = []
    for s, (a, b) in zip(streams, clouds):
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            outs.append(pc.chamfer_distance(a, b))
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert all(torch.equal(o, r) for o, r in zip(outs, ref)), it
print('STREAMS OK')
