"""CPU: the learning-rate schedule against the function the reference calls
(transformers.get_cosine_with_hard_restarts_schedule_with_warmup, agent :273-279)."""
import torch
import transformers

from voxactb_amd.helpers.optim.schedule import CosineWithHardRestarts


def test_matches_transformers_schedule():
    for warm, total, cycles in ((3000, 40000, 4), (10, 100, 1), (0, 50, 3), (5, 5, 2)):
        pa, pb = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
        oa, ob = torch.optim.SGD([pa], lr=5e-4), torch.optim.SGD([pb], lr=5e-4)
        ref = transformers.get_cosine_with_hard_restarts_schedule_with_warmup(oa, num_warmup_steps=warm, num_training_steps=total,
                                                                              num_cycles=cycles)
        mine = CosineWithHardRestarts(ob, warm, total, cycles)
        for step in range(min(total + 20, 400)):
            assert abs(oa.param_groups[0]['lr'] - ob.param_groups[0]['lr']) < 1e-18, (warm, total, cycles, step)
            assert abs(ref.get_last_lr()[0] - mine.get_last_lr()[0]) < 1e-18
            oa.step(); ref.step(); mine.step()
