"""GPU (-m gpu): more launch geometries than the context keeps tile schedules for (16), queued on two streams without
a host wait in between: an evicted schedule's device buffer is reused for the next one, which must not happen while a
kernel queued on the OTHER stream still reads it (prepare_schedule, dst_api.cpp)."""
import numpy as np
import pytest

import distance_amd as da
from helpers import random_alignment

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("path", ["dense", "consensus", "hybrid"])
def test_forty_row_ranges_on_two_streams(path):
    n, L = 2_600, 2_000
    codes = random_alignment(n, L, 11, divergence=0.004)
    codes[: n // 3, ::50] = 72            # clade-like columns: the hybrid path has hot sites to hand to the dense kernels
    with da.Engine(0) as eng:
        eng.set_prep_threshold(0)
        eng.set_path("dense")
        eng.upload(0, codes)
        want = eng.run_square("raw")
        eng.set_path(path)
        eng.upload(0, codes)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        bounds = [int(x) for x in np.linspace(0, n - 1, 41)]
        outs = []
        for k in range(40):
            r0, r1 = bounds[k], bounds[k + 1]
            pairs = da.square_row_start(n, r1) - da.square_row_start(n, r0)
            out = torch.full((max(pairs, 1),), -1.0, dtype=torch.float64, device="cuda")
            eng.run_square_device("raw", r0, r1, out.data_ptr(), out.numel() * 8, stream=streams[k % 2].cuda_stream)
            outs.append((r0, r1, pairs, out))
        torch.cuda.synchronize()
        assert eng.last_path() == path
        for r0, r1, pairs, out in outs:
            lo = da.square_row_start(n, r0)
            assert np.array_equal(out[:pairs].cpu().numpy(), want[lo:lo + pairs], equal_nan=True), (path, r0, r1)
