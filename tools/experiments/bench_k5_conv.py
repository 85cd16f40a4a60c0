import os, sys, torch
sys.path.insert(0, os.getcwd())
from voxactb_amd import ops
from tools.bench_halo import timeit
dev='cuda:0'
B=16
for (Cin,N,S_in,S_out,k,off,rep,name) in ((128,64,20,20,5,-2,True,'fwd 128->64 S20'), (64,128,20,24,5,-4,False,'dgrad 64->128 S24')):
    x=torch.randn(B,S_in,S_in,S_in,Cin,device=dev)
    W=torch.randn(N,k**3*Cin,device=dev)*0.02
    wb=ops.split_bf16(W,True)
    out=None
    for dl in (True, False):
        ops.DL_GEMM=dl
        ops.new_step()
        t=timeit(lambda: ops.conv3d_bf16w(x,wb,N,B,S_in,S_out,k,off,replicate=rep), n=5)
        print(name,'DL' if dl else 'staged','%.3f ms %.1f TF/s'%(t, 2.0*B*S_out**3*N*k**3*Cin/t*1e-9))
