"""Repro of the packed-fp32 observation recorded in tools/experiments/README.md (round 3, "wgin" fold).

With every voxel input = 1 the ten columns of dW_in must equal db_in bit for bit (same terms, same order).  Build the library with
VXB_EXTRA_FLAGS=-DWGIN_NO_OPAQUE (the compiler then emits v_pk_fma_f32 ... op_sel:[0,1,0] for the odd columns) and run this on the GPU:
the odd columns differ from db in a few workgroups' worth of terms, and differently from run to run.  The default build prints zeros.

    VOXACTB_WGIN_FOLD=1 python tools/experiments/wgin_pk_fma_repro.py
"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from voxactb_amd import ops
from tests.test_ops_gpu import rnd, cl, DEV
C = 64
ops.PRECISION, ops.WGRAD_PRECISION, ops.WGIN_FOLD = 'bf16x3', 'fp16', True
B, S = 9, 22
du = cl(rnd(B, C, S, S, S, seed=1)).to(DEV)
Wf = (rnd(C, 2 * C, 3, 3, 3, seed=2) * 0.05).to(DEV)
d0 = cl(rnd(B, C, S, S, S, seed=3)).to(DEV); u0 = cl(rnd(B, C, S, S, S, seed=4)).to(DEV)
wt = ops.conv_weight_dgrad(Wf)
vox = torch.ones(B, S, S, S, 10, device=DEV)
res = []
for rep in range(3):
    dW, db = torch.zeros(C, 10, device=DEV), torch.zeros(C, device=DEV)
    ops.conv3_dgrad_fold(du, wt, B, S, 2 * C, [(None, False, None), (torch.empty_like(d0), False, u0)], leaf_blocks=(0,),
                         wgin={0: (d0, vox, dW, db)})
    torch.cuda.synchronize()
    res.append((dW.clone(), db.clone()))
print('dW repeatable:', all(torch.equal(res[0][0], r[0]) for r in res), ' db repeatable:', all(torch.equal(res[0][1], r[1]) for r in res))
dW, db = res[0]
diff = (dW - db[:, None]) / db.abs().max()
print('max |dW[:, j] - db| / max|db| per column j:', ['%.1e' % v for v in diff.abs().max(0).values.tolist()])
print('channels with a difference:', (diff.abs().max(1).values > 0).nonzero().flatten().tolist())
