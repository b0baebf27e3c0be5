"""GPU: the one-launch neighbour search (nbr_fused_kernel behind gpr_nbr_build_device_rows: geometry upload, count,
offsets across structures by look-back, fill, force-point regrouping by table look-up) against the five-launch path it
replaces -- bit for bit, array by array -- on batches that reach its corners: more structures than lanes of a wavefront,
clusters without periodic images next to small cells with 125 of them, 108 atoms (two passes of 64 lanes per centre), two
species with the sign-flipping weights; and the answer to capacities that are too small."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _arrays(strucs):
    from gpr_calculator_amd.neighbors import atoms_arrays
    arrs = [atoms_arrays(s) for s in strucs]
    atom0 = np.concatenate(([0], np.cumsum([len(a[0]) for a in arrs]))).astype(np.int32)
    num = np.ascontiguousarray(np.concatenate([a[0] for a in arrs]).astype(np.int32))
    pos = np.ascontiguousarray(np.concatenate([a[1] for a in arrs]).astype(np.float64))
    cells = np.ascontiguousarray(np.stack([np.asarray(a[2], dtype=np.float64).reshape(9) for a in arrs]))
    pbc = np.ascontiguousarray(np.stack([np.asarray(a[3], dtype=np.int32).reshape(3) for a in arrs]).astype(np.int32))
    return atom0, pos, cells, pbc, num


def _build(strucs, capP, capR, fused, weight_on=0):
    import torch
    from gpr_calculator_amd import _lib, device as dev
    L = _lib.lib()
    vp = C.c_void_p
    atom0, pos, cells, pbc, num = _arrays(strucs)
    natoms = int(atom0[-1])
    ri = torch.full((2 * capR,), -7, dtype=torch.int32, device=dev.device())
    h, done = vp(), C.c_int(0)
    L.gpr_nbr_set_fused(int(fused))
    try:
        dev.check(L.gpr_nbr_build_device_rows(C.byref(h), dev.stream_ptr(), len(strucs), atom0.ctypes.data, pos.ctypes.data, cells.ctypes.data,
                                              pbc.ctypes.data, num.ctypes.data, 5.0, weight_on, capP, capR, ri.data_ptr(),
                                              ri.data_ptr() + 4 * capR, C.byref(done)), "gpr_nbr_build_device_rows")
    finally:
        L.gpr_nbr_set_fused(1)
    assert done.value == int(fused)
    if not done.value:
        dev.check(L.gpr_nbr_force_rows(h, natoms, None, None, capR, ri.data_ptr(), ri.data_ptr() + 4 * capR), "gpr_nbr_force_rows")
    info = torch.zeros(4, dtype=torch.float64, device=dev.device())
    dev.check(L.gpr_nbr_info_to(h, info.data_ptr()), "gpr_nbr_info_to")
    p = [vp() for _ in range(9)]
    dev.check(L.gpr_nbr_device_arrays(h, *[C.byref(x) for x in p]), "gpr_nbr_device_arrays")
    info = info.cpu().numpy()
    over, P, S = int(info[1]), int(info[2]), int(info[3])

    def grab(ptr, n, dtype):
        t = torch.empty(n, dtype=dtype, device=dev.device())
        if n:
            dev.check(L.gpr_memcpy_dtod(dev.stream_ptr(), t.data_ptr(), ptr, n * t.element_size()), "gpr_memcpy_dtod")
        return t.cpu().numpy()
    out = {"info": info, "cr0": grab(p[3], natoms + 1, torch.int32), "cp0": grab(p[2], natoms + 1, torch.int32)}
    if not over:
        out.update(vec=grab(p[0], 3 * P, torch.float64), w=grab(p[1], P, torch.float64), rp0=grab(p[4], S + 1, torch.int32),
                   self_row=grab(p[5], natoms, torch.int32), seq=grab(p[6], 2 * S, torch.int32), numbers=grab(p[7], natoms, torch.int32),
                   atom0=grab(p[8], len(strucs) + 1, torch.int32), row_index=ri[:S].cpu().numpy(), x_index=ri[capR:capR + S].cpu().numpy())
    L.gpr_nbr_destroy(h)
    return out


def _batches():
    from gpr_calculator_amd.neighbors import SimpleAtoms
    from gpr_calculator_amd.synthetic import fcc_cell, neb_images
    rng = np.random.default_rng(5)
    many = neb_images(70, seed=300)                                             # more structures than lanes
    small = [fcc_cell(seed=k, reps=1, species=(46, 78), sigma=0.1) for k in range(6)]      # 4 atoms, 125 images
    big = [fcc_cell(seed=9, reps=3, species=(46,), sigma=0.05)]                 # 108 atoms
    clusters = [SimpleAtoms(rng.choice([8, 12, 46], n), rng.normal(0.0, 2.2, (n, 3)), np.eye(3) * 30.0, [False] * 3) for n in (5, 13, 1, 9)]
    slab = [SimpleAtoms(s.numbers, s.positions, s.cell, [True, True, False]) for s in small[:2]]
    return {"many": many, "mixed": small + big + clusters + slab + many[:3]}


@pytest.mark.parametrize("name", ["many", "mixed"])
@pytest.mark.parametrize("weight_on", [0, 1])
def test_one_launch_build_is_the_five_launch_build(name, weight_on):
    strucs = _batches()[name]
    natoms = sum(len(s) for s in strucs)
    capP, capR = natoms * 140, natoms * 40
    a, b = _build(strucs, capP, capR, True, weight_on), _build(strucs, capP, capR, False, weight_on)
    assert a["info"][1] == 0 and b["info"][1] == 0 and a["info"][0] == 0
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert a["info"][3] == a["cr0"][-1] and a["info"][2] == a["cp0"][-1]


def test_overflow_leaves_a_prefix_of_the_batch_and_empty_ranges_behind_it():
    """capacities that are too small: the flag goes up and the true sizes are reported; the tables describe the structures
    that fitted (the same entries as the full build) and close every later centre at their end -- nothing queued behind
    the build can leave the arrays.  (The five-launch builder empties the whole table instead.)"""
    strucs = _batches()["mixed"]
    natoms = sum(len(s) for s in strucs)
    a0 = np.concatenate([[0], np.cumsum([len(s) for s in strucs])])
    full = _build(strucs, natoms * 140, natoms * 40, True)
    P, S = int(full["info"][2]), int(full["info"][3])
    for capP, capR in ((P - 1, S), (P, S - 1), (64, 64), (P // 2, S // 2)):
        o = _build(strucs, capP, capR, True)
        assert o["info"][1] == 1 and int(o["info"][2]) == P and int(o["info"][3]) == S          # flag up, true sizes reported
        cr0, cp0 = o["cr0"], o["cp0"]
        assert (np.diff(cr0) >= 0).all() and (np.diff(cp0) >= 0).all() and cr0[-1] <= capR and cp0[-1] <= capP
        # the structures that fit: the longest prefix whose totals stay inside both capacities
        k = 0
        while k < len(strucs) and full["cr0"][a0[k + 1]] <= capR and full["cp0"][a0[k + 1]] <= capP:
            k += 1
        assert k < len(strucs)
        assert np.array_equal(cr0[:a0[k] + 1], full["cr0"][:a0[k] + 1]) and np.array_equal(cp0[:a0[k] + 1], full["cp0"][:a0[k] + 1])
        assert (cr0[a0[k]:] == cr0[a0[k]]).all() and (cp0[a0[k]:] == cp0[a0[k]]).all()       # every later centre empty
        o5 = _build(strucs, capP, capR, False)                                                  # the five launches: all or nothing
        assert o5["info"][1] == 1 and not o5["cr0"].any() and not o5["cp0"].any()
    exact = _build(strucs, P, S, True)                                                         # capacities that just fit
    assert exact["info"][1] == 0 and np.array_equal(exact["seq"], full["seq"])


def _rows_done(strucs, capP, capR):
    """does a capped build with regrouping take the one-launch kernel right now?"""
    import torch
    from gpr_calculator_amd import _lib, device as dev
    L = _lib.lib()
    atom0, pos, cells, pbc, num = _arrays(strucs)
    ri = torch.full((2 * capR,), -7, dtype=torch.int32, device=dev.device())
    h, done = C.c_void_p(), C.c_int(0)
    dev.check(L.gpr_nbr_build_device_rows(C.byref(h), dev.stream_ptr(), len(strucs), atom0.ctypes.data, pos.ctypes.data, cells.ctypes.data,
                                          pbc.ctypes.data, num.ctypes.data, 5.0, 0, capP, capR, ri.data_ptr(), ri.data_ptr() + 4 * capR,
                                          C.byref(done)), "gpr_nbr_build_device_rows")
    torch.cuda.synchronize()
    L.gpr_nbr_destroy(h)
    return done.value


def test_the_admission_rule_of_the_one_launch_build_when_a_capacity_is_pretended():
    """The kernel's workgroups wait for their predecessors only (ticket order), so a build needs no particular residency and
    is always admitted.  The rule the first form of the kernel needed -- all such workgroups in flight within half of what
    the device keeps resident -- still exists behind gpr_nbr_debug_fused_capacity: a build that does not fit the pretended
    capacity takes the five-launch path, with the same tables."""
    import torch
    from gpr_calculator_amd import _lib
    from gpr_calculator_amd.synthetic import neb_images
    L = _lib.lib()
    strucs = neb_images(40, seed=310)
    natoms = sum(len(s) for s in strucs)
    capP, capR = natoms * 140, natoms * 40
    assert _rows_done(strucs, capP, capR) == 1
    cap, fly = C.c_int(-1), C.c_int(-1)
    L.gpr_nbr_fused_state(C.byref(cap), C.byref(fly))
    assert cap.value >= 256 and fly.value == 0, (cap.value, fly.value)        # >= one 1024-thread workgroup per CU; nothing left flying
    want = _build(strucs, capP, capR, False)
    try:
        L.gpr_nbr_debug_fused_capacity(2 * len(strucs) - 1)                   # one workgroup short of "half the device"
        assert _rows_done(strucs, capP, capR) == 0
        L.gpr_nbr_debug_fused_capacity(2 * len(strucs))
        assert _rows_done(strucs, capP, capR) == 1
        # two streams: the first launch is still flying (the stream is held back by a long kernel in front of it) when the
        # second asks -- together they exceed the pretended capacity, so the second build is declined
        L.gpr_nbr_debug_fused_capacity(3 * len(strucs))
        s1 = torch.cuda.Stream()
        big = torch.empty((8192, 8192), dtype=torch.float64, device="cuda").normal_()
        with torch.cuda.stream(s1):
            for _ in range(4):
                big = big @ big * 1e-4                                          # tens of milliseconds ahead of the build
            first = _rows_done_async(strucs, capP, capR)
        second = _rows_done(strucs, capP, capR)
        s1.synchronize()
        first[1]()
        assert first[0] == 1 and second == 0
    finally:
        L.gpr_nbr_debug_fused_capacity(-1)
    got = _build(strucs, capP, capR, True)                                    # admitted again, and the same tables either way
    for k in want:
        assert np.array_equal(got[k], want[k]), k


def _rows_done_async(strucs, capP, capR):
    """(rows_done, release) of a capped build queued on the current stream without waiting for it"""
    import torch
    from gpr_calculator_amd import _lib, device as dev
    L = _lib.lib()
    atom0, pos, cells, pbc, num = _arrays(strucs)
    ri = torch.full((2 * capR,), -7, dtype=torch.int32, device=dev.device())
    h, done = C.c_void_p(), C.c_int(0)
    dev.check(L.gpr_nbr_build_device_rows(C.byref(h), dev.stream_ptr(), len(strucs), atom0.ctypes.data, pos.ctypes.data, cells.ctypes.data,
                                          pbc.ctypes.data, num.ctypes.data, 5.0, 0, capP, capR, ri.data_ptr(), ri.data_ptr() + 4 * capR,
                                          C.byref(done)), "gpr_nbr_build_device_rows")
    keep = (atom0, pos, cells, pbc, num, ri)
    return done.value, (lambda: (L.gpr_nbr_destroy(h), keep)[0])


def test_a_build_that_needs_more_than_64_kb_of_lds_still_runs():
    """128 atoms in a cell small enough for hundreds of periodic images: the tables of the one-launch kernel pass the
    64 KB default of dynamic LDS.  The limit is raised (or the build declined) -- never a failed launch."""
    from gpr_calculator_amd.neighbors import SimpleAtoms
    rng = np.random.default_rng(11)
    n = 128
    cell = np.diag([2.6, 2.6, 40.0])
    pos = rng.uniform(0.0, 1.0, (n, 3)) @ cell
    s = SimpleAtoms(np.full(n, 46), pos, cell, [True, True, False])            # 5 x 5 x 1 images and more
    natoms = n
    a, b = _build_any([s], natoms * 600, natoms * 130), _build([s], natoms * 600, natoms * 130, False)
    assert a["info"][1] == 0 and b["info"][1] == 0
    for k in b:
        assert np.array_equal(a[k], b[k]), k


def _build_any(strucs, capP, capR):
    """_build without insisting on which path served it"""
    import gpr_calculator_amd._lib as _lib
    L = _lib.lib()
    try:
        return _build(strucs, capP, capR, True)
    except AssertionError:
        return _build(strucs, capP, capR, False)


def test_first_build_on_streams_that_have_never_seen_one():
    """The look-back slots of the one-launch build belong to the stream and are cleared when the stream first needs them --
    ON that stream: cleared on the null stream (which a non-blocking stream does not wait for) the first launch could read
    them before they were cleared, or lose what it had published.  Thirty fresh non-blocking streams, first build on each
    with other work keeping the device busy, every table compared with the five-launch build."""
    import torch
    from gpr_calculator_amd.synthetic import neb_images
    strucs = neb_images(48, seed=77)
    natoms = sum(len(s) for s in strucs)
    capP, capR = natoms * 140, natoms * 40
    want = _build(strucs, capP, capR, False)
    busy = torch.empty((4096, 4096), dtype=torch.float64, device="cuda").normal_()
    for k in range(30):
        st = torch.cuda.Stream()
        busy = busy @ busy * 1e-3                                            # the null stream has work queued: a clear queued there comes late
        with torch.cuda.stream(st):
            got = _build(strucs, capP, capR, True)
        assert got["info"][0] == 0 and got["info"][1] == 0, (k, got["info"])
        for key in want:
            assert np.array_equal(got[key], want[key]), (k, key)
    torch.cuda.synchronize()
