"""bench.py takes `roofline.traffic` (HBM bytes per launch of the dominant kernel) and the voxelizer's real traffic from the NEWEST
committed rocprofv3 PMC passes under profiles/ instead of constants typed into the script (round-4 review: they would go stale silently
the day a kernel changes): the parsing works on the committed files, and every kernel a timer label maps to is present in them."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_dominant_kernel_traffic_comes_from_the_newest_profile():
    b = _bench()
    prof = b.newest_pmc_profiles()
    assert prof is not None
    fetch = b.pmc_mean_per_dispatch(prof[0])
    assert len(fetch) > 10
    for label, (kernels, _, factor) in b.TRAFFIC_KERNELS.items():
        assert factor in (1.0, 2.0)
        for k in kernels:
            assert any(k in name for name in fetch), (k, os.path.basename(prof[0]))
        nbytes, note = b.profile_traffic(label + ']')
        assert 1e9 < nbytes < 1e11 and os.path.basename(prof[0]).split('_pmc_')[0] in note
    # the final conv's forward: 12.3 GB compulsory; 14 - 17 GB moved (raw FETCH_SIZE: this kernel's 64-byte segment loads are tallied at
    # face value, profiles/r05_final_conv_tile_order.log) -- a change of the kernel must show up here
    nbytes, note = b.profile_traffic('conv3d_bf16[k3 s1 128->64 S100]')
    assert 1.23e10 < nbytes < 2.0e10 and 'r06_fetch_size_calibration' in note


def test_voxelizer_traffic_comes_from_the_newest_profile():
    b = _bench()
    inc, src = b.voxel_profile_traffic(True)
    full, _ = b.voxel_profile_traffic(False)
    assert src is not None
    assert 5e7 < inc < 6e8                  # the incremental call touches ~2 x 40 B per occupied cell + ~100 B per point
    assert 6.4e8 < full < 1.2e9             # the stateless call writes all 640 MB of the grid
