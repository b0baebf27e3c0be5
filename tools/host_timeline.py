"""Host-side inclusive wall time per function of one predict step (no device synchronisation inside
the step, so device time only shows up where the host actually waits).  usage: python tools/host_timeline.py [Dot|RBF]"""
import sys, os, time, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from gpr_calculator_amd import device as dev, SO3 as so3mod, gaussianprocess as gpmod
from gpr_calculator_amd.synthetic import neb_images
from gpr_calculator_amd.parallel import Comm

acc, cnt = collections.defaultdict(float), collections.Counter()

def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj)}.{name}"
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0; cnt[label] += 1
    setattr(obj, name, w)

kind = sys.argv[1] if len(sys.argv) > 1 else "Dot"
gp = bench.build_model(kind, 2000, Comm())
gp.fit(show=False, opt=False)
batch = neb_images(64, seed=100)
for _ in range(3): gp.predict_structures(batch, return_std=True)
S = so3mod.SO3
for n in ("neighbor_index_batch", "calculate_batch_dev"): wrap(S, n, "SO3." + n)
for n in ("_batch_data_dev", "_predict_dev", "get_train_x", "_get_fixed_atoms"): wrap(gpmod.GP, n, "GP." + n)
wrap(dev.PackedSet, "__init__", "PackedSet.__init__"); wrap(dev.PackedSet, "close", "PackedSet.close")
for n in ("to_dev", "dot2_features", "gemm", "rowdot", "diag_ff", "diag_ee", "gemv_t", "trmm_lower", "colsumsq", "block_ff", "block_ef", "block_ee"):
    wrap(dev, n, "dev." + n)
wrap(type(gp.kernel), "diag_dev", "kernel.diag_dev"); wrap(type(gp.kernel), "_psi", "kernel._psi") if hasattr(gp.kernel, "_psi") else None
wrap(torch.Tensor, "cpu", "Tensor.cpu"); wrap(torch.Tensor, "index_select", "Tensor.index_select")
wrap(np, "argsort", "np.argsort")
N = 20
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): gp.predict_structures(batch, return_std=True)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N * 1e3
print(f"step {tot:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:32s} {v / N * 1e3:8.3f} ms  x{cnt[k] / N:.0f}")

# ---- finer split of the descriptor front end (host view)
import ctypes as C
from gpr_calculator_amd import _lib
from gpr_calculator_amd.neighbors import atoms_arrays
def split(n=20):
    t = collections.defaultdict(float)
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); arrs = [atoms_arrays(s) for s in batch]; t1 = time.perf_counter()
        atom0 = np.zeros(len(arrs) + 1, dtype=np.int64); np.cumsum([len(a[0]) for a in arrs], out=atom0[1:])
        a0 = atom0.astype(np.int32)
        numbers = np.ascontiguousarray(np.concatenate([a[0] for a in arrs]), dtype=np.int32)
        pos = np.ascontiguousarray(np.concatenate([a[1] for a in arrs], axis=0), dtype=np.float64)
        cells = np.ascontiguousarray(np.stack([a[2] for a in arrs]).reshape(-1, 9), dtype=np.float64)
        pbc = np.ascontiguousarray(np.stack([a[3] for a in arrs]), dtype=np.int32)
        t2 = time.perf_counter()
        L = _lib.lib(); h = C.c_void_p()
        ip = lambda a: a.ctypes.data_as(_lib.c_int_p); dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        L.gpr_nbr_build_device(C.byref(h), dev.stream_ptr(), len(arrs), ip(a0), dp(pos), dp(cells), ip(pbc), ip(numbers), 5.0, 0)
        t3 = time.perf_counter()
        L.gpr_nbr_destroy(h)
        t4 = time.perf_counter()
        nb = gp.descriptor.neighbor_batch_dev(batch); t5 = time.perf_counter()
        fr = nb.force_rows(np.arange(nb.natoms)); t6 = time.perf_counter()
        desc = gp.descriptor.calculate_batch_dev(batch, nbr=nb); t7 = time.perf_counter()
        t["atoms_arrays x64"] += t1 - t0; t["concat"] += t2 - t1; t["gpr_nbr_build_device"] += t3 - t2
        t["destroy"] += t4 - t3; t["NeighborBatch total"] += t5 - t4; t["force_rows"] += t6 - t5; t["so3 launch"] += t7 - t6
    for k, v in t.items(): print(f"  {k:28s} {v / n * 1e3:7.3f} ms")
split()

def split2(n=50):
    """statement-level timing of NeighborBatch.__init__ (copied body)"""
    rcut, weight_on = 5.0, False
    t = collections.defaultdict(float)
    L = _lib.lib()
    for _ in range(n):
        torch.cuda.synchronize()
        T = [time.perf_counter()]
        def tick(): T.append(time.perf_counter())
        arrs = [atoms_arrays(s) for s in batch]; tick()
        nstruct = len(arrs)
        atom0_ = np.zeros(nstruct + 1, dtype=np.int64)
        np.cumsum([len(a[0]) for a in arrs], out=atom0_[1:])
        atom0 = atom0_.astype(np.int32); tick()
        numbers_ = np.concatenate([a[0] for a in arrs]).astype(np.int64)
        numbers = np.ascontiguousarray(numbers_, dtype=np.int32); tick()
        pos = np.ascontiguousarray(np.concatenate([a[1] for a in arrs], axis=0), dtype=np.float64)
        cells = np.ascontiguousarray(np.stack([a[2] for a in arrs]).reshape(-1, 9), dtype=np.float64)
        pbc = np.ascontiguousarray(np.stack([a[3] for a in arrs]), dtype=np.int32); tick()
        ip = lambda a: a.ctypes.data_as(_lib.c_int_p)
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        h = C.c_void_p()
        sp = dev.stream_ptr(); tick()
        dev.check(L.gpr_nbr_build_device(C.byref(h), sp, nstruct, ip(atom0), dp(pos), dp(cells), ip(pbc), ip(numbers), rcut, int(weight_on)), "x"); tick()
        P_, S_ = C.c_int(), C.c_int()
        dev.check(L.gpr_nbr_sizes(h, C.byref(P_), C.byref(S_)), "x")
        cr0 = np.empty(int(atom0[-1]) + 1, np.int32); row0 = np.empty(nstruct + 1, np.int32)
        dev.check(L.gpr_nbr_export(h, None, None, None, ip(cr0), None, None, None, ip(row0)), "x"); tick()
        names = ("vec", "w", "centre_pair0", "centre_row0", "row_pair0", "self_row", "seq", "numbers", "atom0")
        raw = [C.c_void_p() for _ in names]
        dev.check(L.gpr_nbr_device_arrays(h, *[C.byref(r) for r in raw]), "x"); tick()
        L.gpr_nbr_destroy(h); tick()
        for k, lab in enumerate(("atoms_arrays", "atom0", "numbers", "pos/cells/pbc", "stream_ptr", "build_device", "sizes+export", "device_arrays", "destroy")):
            t[lab] += T[k + 1] - T[k]
    for k, v in t.items(): print(f"    {k:20s} {v / n * 1e3:7.3f} ms")
split2()

def split3(n=50):
    """statement-level timing of GP._batch_data_dev + predict tail (copied body)"""
    t = collections.defaultdict(float)
    self = gp
    strucs = batch
    for _ in range(n):
        torch.cuda.synchronize()
        T = [time.perf_counter()]
        def tick(): T.append(time.perf_counter())
        nbr = self.descriptor.neighbor_batch_dev(strucs); tick()
        atom0 = nbr.atom0
        natoms = int(atom0[-1]); nper = np.diff(atom0)
        keep, frees = None, []
        for k, s in enumerate(strucs):
            fix = self._get_fixed_atoms(s)
            frees.append(None if not len(fix) else list(fix))
        tick()
        counts_f, row_index, x_index, f_row0 = nbr.force_rows(None); tick()
        e_row0, ele = nbr.atom0_dev(), nbr.numbers_dev(); tick()
        desc = self.descriptor.calculate_batch_dev(strucs, nbr=nbr); tick()
        E = dev.SetPair(desc["x"], None, ele, nper); E._row0_dev = e_row0
        F = dev.SetPair(desc["x"], desc["dxdr"], ele, counts_f, x_index=x_index, dxdr_index=row_index); F._row0_dev = f_row0; F._nbr = nbr
        X = {"energy": E, "force": F}; tick()
        mean, var, flat = self._predict_dev(X, self.get_train_x(), 1e-8, True, want_flat=True); tick()
        nbr.status_into(flat[-1:]); tick()
        both = flat.cpu().numpy(); tick()
        for k, lab in enumerate(("neighbor_batch_dev", "keep/frees loop", "force_rows", "meta upload", "so3 launch", "SetPairs", "_predict_dev", "status_into", "download (waits GPU)")):
            t[lab] += T[k + 1] - T[k]
    for k, v in t.items(): print(f"      {k:22s} {v / n * 1e3:7.3f} ms")
split3()
