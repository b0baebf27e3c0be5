"""Prediction session under load: batches of different sizes in rotation with several in flight -- every tail form
(short-chain, staged split, row-resident), both force-feature forms, the overflow-free steady state of the neighbour
search -- each result compared bit for bit with the synchronous call on the same batch (tools/session_stress.py runs
the same loop for tens of thousands of calls)."""
from collections import deque

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mixed_batches_in_flight_reproduce_the_synchronous_results():
    import bench
    from gpr_calculator_amd.parallel import Comm
    from gpr_calculator_amd.synthetic import neb_images
    gp = bench.build_model("Dot", 400, Comm())
    gp.fit(show=False, opt=False)
    sizes = [1, 2, 8, 1, 12, 16, 40, 64, 5]
    batches = [neb_images(n, seed=700 + i) for i, n in enumerate(sizes)]
    want = []
    for b in batches:
        for _ in range(2):
            r = gp.predict_structures(b, return_std=True)
        want.append([np.array(x, copy=True) for x in r])
    for depth in (2, 4):
        gp.session_depth = depth
        q, bad = deque(), 0
        for c in range(1200):
            if len(q) == depth:
                i, tk = q.popleft()
                bad += any(not np.array_equal(np.asarray(a), b) for a, b in zip(gp.predict_structures_collect(tk), want[i]))
            i = (c * 7 + c // 13) % len(batches)
            q.append((i, gp.predict_structures_submit(batches[i], return_std=True)))
        while q:
            i, tk = q.popleft()
            bad += any(not np.array_equal(np.asarray(a), b) for a, b in zip(gp.predict_structures_collect(tk), want[i]))
        assert bad == 0, f"{bad} of 1200 results differ from the synchronous call at depth {depth}"
