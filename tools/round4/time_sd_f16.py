"""sided_distance on at::Half clouds at 100k x 100k: the exact grid search (round 4) vs the all-pairs kernel (KAMD_SIDED_DISTANCE=brute)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import kaolin_amd as kal
g = torch.Generator().manual_seed(0)
p1, p2 = torch.rand(1, 100000, 3, generator=g).half().cuda(), torch.rand(1, 100000, 3, generator=g).half().cuda()


def ms(n=10):
    for _ in range(3):
        kal.metrics.pointcloud.sided_distance(p1, p2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        kal.metrics.pointcloud.sided_distance(p1, p2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


grid = ms()
os.environ['KAMD_SIDED_DISTANCE'] = 'brute'
brute = ms(3)
print(f'sided_distance fp16 100k x 100k: grid search {grid:.3f} ms, all-pairs {brute:.3f} ms')
