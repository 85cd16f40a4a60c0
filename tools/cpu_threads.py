import sys, time, os, torch
sys.path.insert(0, '/root/repo')
from oracle import perceiver as operc, weights as ow
print('cpus', os.cpu_count())
shapes = operc.param_shapes(6, 100, 4, num_latents=2048, voxel_patch_size=5, voxel_patch_stride=5)
P = ow.hashed_state_dict(shapes, 0)
ins = torch.randn(1, 10, 100, 100, 100); pr = torch.randn(1, 4); lt = torch.randn(1, 77, 512)
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    with torch.no_grad():
        operc.forward(P, ins, pr, lt, depth=6)
    print(nt, 'threads: fwd %.1f s' % (time.perf_counter() - t0), flush=True)
