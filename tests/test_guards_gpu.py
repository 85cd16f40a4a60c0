"""GPU: failure behaviour of the C-ABI layer -- invalid labels never become out-of-bounds reads, launches follow the
tensors' device, the package refuses to run without its HIP library."""
import math

import pytest
import torch

from voxactb_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_cross_entropy_rejects_out_of_range_labels_with_nan():
    """the reference raises an indexing error for a label outside the class range (agent :519-545); the fused kernels
    return NaN for that sample's loss and gradient row instead of reading out of bounds (VXB has no device-side raise)."""
    torch.manual_seed(0)
    x = torch.randn(3, 4096, device=DEV)
    dx = torch.empty_like(x)
    lab = torch.tensor([5, 4096, -1], dtype=torch.int32, device=DEV)
    loss, lse, arg = ops.ce_big(x, lab, dx, 1.0)
    ref = torch.nn.functional.cross_entropy(x[:1], lab[:1].long(), reduction='none')
    assert abs(float(loss[0]) - float(ref[0])) < 1e-5 and math.isnan(float(loss[1])) and math.isnan(float(loss[2]))
    assert torch.isfinite(dx[0]).all() and torch.isnan(dx[1]).all() and torch.isnan(dx[2]).all()
    assert torch.equal(arg.cpu().long(), x.argmax(1).cpu())
    o = torch.randn(2, 220, device=DEV)
    d_o = torch.empty_like(o)
    labs = torch.tensor([[3, 71, 0, 1, 1], [72, 5, -3, 2, 0]], dtype=torch.int32, device=DEV)
    l, pred = ops.ce_rows(o, [(0, 72), (72, 72), (144, 72), (216, 2), (218, 2)], labs, d_o, 0.5)
    assert torch.isfinite(l[0]).all() and torch.isfinite(d_o[0]).all()
    assert [math.isnan(float(v)) for v in l[1]] == [True, False, True, True, False]
    assert torch.isnan(d_o[1, :72]).all() and torch.isfinite(d_o[1, 72:144]).all() and torch.isnan(d_o[1, 216:218]).all()


def test_launch_follows_the_tensor_device_not_the_current_device():
    if torch.cuda.device_count() < 2:
        # single-GPU box: the guard is exercised through its context manager and the agent's set_device
        with _lib.on_device(torch.zeros(1, device=DEV)):
            assert torch.cuda.current_device() == 0
        return
    torch.cuda.set_device(0)
    x = torch.randn(4, 64, device='cuda:1')
    w = torch.ones(64, device='cuda:1')
    b = torch.zeros(64, device='cuda:1')
    y, mean, rstd = ops.layernorm_fwd(x, w, b)            # launched while cuda:0 is current
    ref = torch.nn.functional.layer_norm(x, (64,), w, b)
    assert float((y - ref).abs().max()) < 1e-5 and torch.cuda.current_device() == 0


def test_cpu_tensors_are_refused():
    with pytest.raises(_lib.VoxactbHipError):
        ops.layernorm_fwd(torch.randn(4, 64), torch.ones(64), torch.zeros(64))
