"""GPU: deterministic column sums (bias gradients) -- the flat float4 path (row length dividing 1024) and the general
path, against float64 sums; ragged row counts, accumulate, strided input."""
import pytest
import torch

from voxactb_amd import ops
from .test_ops_gpu import rnd, close, DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,N', [(100003, 64), (4097, 512), (70001, 128), (999, 1024), (5000, 72), (3001, 4096), (40, 64)])
def test_colsum_matches_fp64(rows, N):
    x = rnd(rows, N)
    out = torch.full((N,), 0.5, device=DEV)
    ops.colsum(x.to(DEV), out, accumulate=True)
    close(out, (x.double().sum(0) + 0.5).float(), 2e-5, 'colsum accumulate')
    out2 = torch.empty(N, device=DEV)
    ops.colsum(x.to(DEV), out2)
    out3 = torch.empty(N, device=DEV)
    ops.colsum(x.to(DEV), out3)
    assert torch.equal(out2, out3)                 # fixed summation order
    close(out2, x.double().sum(0).float(), 2e-5, 'colsum')


def test_colsum_strided_rows_take_the_general_path():
    x = rnd(5000, 192)
    out = torch.empty(64, device=DEV)
    ops.colsum(x.to(DEV)[:, 64:128], out)
    close(out, x[:, 64:128].double().sum(0).float(), 2e-5, 'colsum strided')


@pytest.mark.parametrize('rows,cols', [(1, 1000000), (3, 70001), (2, 65536)])
def test_softmax_long_rows_matches_torch(rows, cols):
    """the multi-workgroup softmax of a few very long rows (act(): B x V^3) against torch's softmax in float64."""
    import torch
    from voxactb_amd import ops
    ld = (cols + 3) & ~3
    x = torch.randn(rows, ld, device='cuda:0') * 3.0
    want = torch.softmax(x[:, :cols].double(), dim=1)
    got = ops.softmax_rows(x.clone(), rows, cols, ld)
    assert float((got[:, :cols].double() - want).abs().max()) < 1e-9 + 2e-6 * float(want.max())
    assert abs(float(got[:, :cols].double().sum(1).max()) - 1.0) < 1e-5
    if ld > cols:
        assert float(got[:, cols:].abs().max()) == 0.0
