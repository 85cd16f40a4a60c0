import torch, json
dev='cuda:0'
small = torch.zeros(1 << 16, device=dev)
torch.cuda.synchronize()
def pair(pre, body, n=50):
    tot=0.0
    for _ in range(n):
        torch.cuda.synchronize()
        if pre:
            for _ in range(40): small.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(body): small.add_(1.0)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot/n*1e3
print(json.dumps({'empty_pair_idle_us': pair(False,0), 'empty_pair_busy_us': pair(True,0), 'one_kernel_idle': pair(False,1), 'one_kernel_busy': pair(True,1), 'six_kernels_idle': pair(False,6), 'six_kernels_busy': pair(True,6)}))
