"""SO3: SO(3) power-spectrum descriptor with the reference's class interface
(gpr_calc/SO3.py:7-323), evaluated by csrc/so3.hip on the GPU.

Host side (this file): gathering the structures' arrays and the 3 x 80 table of radial quadrature weights, which
depends only on (nmax, lmax, rcut, alpha) and is built once.  Device side: the periodic neighbour search and the
(centre, neighbour) row bookkeeping (csrc/nbr.hip; a host builder with identical output serves the single-structure
`calculate` API), the radial Chebyshev table built from that quadrature (gpr_so3_tables), and everything per
neighbour pair and per centre -- spherical harmonics, expansion coefficients, power spectrum, its derivatives and,
with stress=True, the stress rows."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import device as dev
from .neighbors import atoms_arrays, batch_arrays, neighbor_pairs, symbol_of


def radial_table(nmax, lmax, rcut, alpha):
    """Gauss-Chebyshev nodes r_q and G[n, q] = g_n(r_q) w_q r_q^2 exp(-alpha r_q^2) sqrt(1 - t_q^2)
    (gpr_calc/SO3.py:417-453, 623-655).  A (nmax x 10(nmax+lmax+1)) constant table."""
    NQ = (nmax + lmax + 1) * 10
    t = np.cos((2 * np.arange(1, NQ + 1) - 1) * np.pi / 2 / NQ)
    rq = rcut / 2 * (t + 1)
    S = np.zeros((nmax, nmax))
    for a in range(1, nmax + 1):
        for b in range(1, a + 1):
            S[a - 1, b - 1] = S[b - 1, a - 1] = np.sqrt((2 * a + 5) * (2 * a + 6) * (2 * a + 7) * (2 * b + 5) * (2 * b + 6) * (2 * b + 7)) \
                / (5 + a + b) / (6 + a + b) / (7 + a + b)
    evals, V = np.linalg.eig(np.linalg.inv(S))
    w = V @ np.diag(np.sqrt(evals)) @ np.linalg.inv(V)         # S^(-1/2), non-symmetric eig as in W()
    G = np.zeros((nmax, NQ))
    for n in range(nmax):
        for a in range(1, nmax + 1):
            G[n] += w[n, a - 1] * (rcut - rq) ** (a + 2) / np.sqrt(2 * rcut ** (2 * a + 7) / (2 * a + 5) / (2 * a + 6) / (2 * a + 7))
    G *= (np.pi / NQ * rcut / 2) * rq ** 2 * np.exp(-alpha * rq ** 2) * np.sqrt(1 - t ** 2)
    return rq, np.ascontiguousarray(G.real)


class SO3:
    '''
    SO3 power spectrum components of the Gaussian atomic neighbour density.

    args (as gpr_calc.SO3.SO3): nmax, lmax, rcut, alpha, derivative, stress, cutoff_function
    (only 'cosine' exists upstream), weight_on, primitive.
    '''

    def __init__(self, nmax=3, lmax=3, rcut=3.5, alpha=2.0, derivative=True, stress=False,
                 cutoff_function='cosine', weight_on=False, primitive=False):
        self.nmax = nmax
        self.lmax = lmax
        self.rcut = rcut
        self.alpha = alpha
        self.derivative = derivative
        self.stress = stress
        self._type = "SO3"
        self.cutoff_function = cutoff_function
        self.weight_on = weight_on
        self.primitive = primitive
        self._tables = None

    def __str__(self):
        s = "SO3 descriptor with Cutoff: {:6.3f}".format(self.rcut)
        s += " lmax: {:d}, nmax: {:d}, alpha: {:.3f}\n".format(self.lmax, self.nmax, self.alpha)
        return s

    __repr__ = __str__

    def load_from_dict(self, dict0):
        self.nmax = dict0["nmax"]
        self.lmax = dict0["lmax"]
        self.rcut = dict0["rcut"]
        self.alpha = dict0["alpha"]
        self.derivative = dict0["derivative"]
        self.stress = dict0["stress"]
        self._tables = None

    def save_dict(self):
        return {"nmax": self.nmax, "lmax": self.lmax, "rcut": self.rcut, "alpha": self.alpha,
                "derivative": self.derivative, "stress": self.stress, "_type": "SO3"}

    @staticmethod
    def _check_device_limits(nmax, lmax):
        """The device kernels cover nmax <= 4 with lmax <= 6, nmax = 5 with lmax <= 4 and nmax = 6 with lmax <= 3 -- up to
        85 power-spectrum components (one thread per column of a dxdr row in the descriptor kernel; the covariance kernels
        take up to 128 columns; csrc/so3.hip, csrc/pack.hip): say so when the object is configured, not deep inside a
        launch (the reference allows nmax <= 11, SO3.py:76-77).  Stress rows (stress=True) additionally need nmax <= 4,
        nmax (lmax + 1) <= 16 and ncoef <= 32, checked at the launch."""
        if nmax is None or lmax is None:
            return
        ncoef = nmax * (nmax + 1) // 2 * (lmax + 1)
        if nmax > 6 or lmax > 6 or (nmax == 5 and lmax > 4) or (nmax > 5 and lmax > 3) or ncoef > 85:
            raise NotImplementedError(
                f"SO3(nmax={nmax}, lmax={lmax}) gives ncoef = {ncoef}; this build's device kernels cover nmax <= 4 with "
                f"lmax <= 6, nmax = 5 with lmax <= 4, nmax = 6 with lmax <= 3 (ncoef <= 85; the reference's default is "
                f"nmax=3, lmax=4: ncoef = 30)")

    # -- validated attributes (gpr_calc/SO3.py:67-182)
    @property
    def nmax(self):
        return self._nmax

    @nmax.setter
    def nmax(self, nmax):
        if not isinstance(nmax, int):
            raise ValueError('nmax must be an integer')
        if nmax < 1:
            raise ValueError('nmax must be greater than or equal to 1')
        if nmax > 11:
            raise ValueError('nmax > 11 yields complex eigenvalues which will mess up the calculation')
        self._check_device_limits(nmax, getattr(self, "_lmax", None))
        self._nmax = nmax
        self._tables = None

    @property
    def lmax(self):
        return self._lmax

    @lmax.setter
    def lmax(self, lmax):
        if not isinstance(lmax, int):
            raise ValueError('lmax must be an integer')
        if lmax < 0:
            raise ValueError('lmax must be greater than or equal to zero')
        if lmax > 32:
            raise NotImplementedError('lmax up to 32 only')
        self._check_device_limits(getattr(self, "_nmax", None), lmax)
        self._lmax = lmax
        self._tables = None

    @property
    def rcut(self):
        return self._rcut

    @rcut.setter
    def rcut(self, rcut):
        if not isinstance(rcut, (float, int)):
            raise ValueError('rcut must be a float')
        if rcut <= 0:
            raise ValueError('rcut must be greater than zero')
        self._rcut = rcut
        self._tables = None

    @property
    def alpha(self):
        return self._alpha

    @alpha.setter
    def alpha(self, alpha):
        if not isinstance(alpha, (float, int)):
            raise ValueError('alpha must be a float')
        if alpha <= 0:
            raise ValueError('alpha must be greater than zero')
        self._alpha = alpha
        self._tables = None

    @property
    def derivative(self):
        return self._derivative

    @derivative.setter
    def derivative(self, derivative):
        if not isinstance(derivative, bool):
            raise ValueError('derivative must be a boolean value')
        self._derivative = derivative

    @property
    def stress(self):
        return self._stress

    @stress.setter
    def stress(self, stress):
        if not isinstance(stress, bool):
            raise ValueError('stress must be a boolean value')
        self._stress = stress

    @property
    def cutoff_function(self):
        return self._cutoff_function

    @cutoff_function.setter
    def cutoff_function(self, cutoff_function):
        if not isinstance(cutoff_function, str):
            raise ValueError('You must specify the cutoff function as a string')
        if cutoff_function != 'cosine':
            # the reference accepts tanh / poly1-4 / exp / unity here but defines none of them (SO3.py:155-170)
            raise NotImplementedError('The requested cutoff function has not been implemented')
        self._cutoff_function = cutoff_function

    @property
    def ncoefs(self):
        return self.nmax * (self.nmax + 1) // 2 * (self.lmax + 1)

    # ------------------------------------------------------------------ host bookkeeping
    NCHEB = 16      # Chebyshev coefficients per segment of the radial table

    def _device_tables(self):
        """(table, nseg, ncheb): radial Chebyshev table + angular constants on the device (gpr_so3_tables), built
        once per (nmax, lmax, rcut, alpha) from the reference's quadrature table G, rq"""
        if self._tables is None:
            rq, G = radial_table(self.nmax, self.lmax, float(self.rcut), float(self.alpha))
            nseg = max(8, int(np.ceil(1.2 * float(self.rcut) * np.sqrt(float(self.alpha)))))
            L = _lib.lib()
            size = int(L.gpr_so3_table_size(self.nmax, self.lmax, nseg, self.NCHEB))
            table = torch.empty(size, dtype=dev.F64, device=dev.device())
            d_rq, d_G = dev.to_dev(rq), dev.to_dev(G)
            dev.check(L.gpr_so3_tables(dev.stream_ptr(), self.nmax, self.lmax, float(self.rcut), float(self.alpha),
                                       len(rq), dev.ptr(d_G), dev.ptr(d_rq), nseg, self.NCHEB, dev.ptr(table)),
                      "gpr_so3_tables")
            self._tables = (table, nseg, self.NCHEB, d_G, d_rq)
        return self._tables

    def radial_functions(self, r, mode="table"):
        """F0_nl(r), F1_nl(r) [len(r), nmax, lmax+1, 2] from the device table or the direct quadrature"""
        table, nseg, ncheb, d_G, d_rq = self._device_tables()
        rs = dev.to_dev(np.asarray(r, dtype=np.float64))
        out = torch.empty((len(rs), self.nmax, self.lmax + 1, 2), dtype=dev.F64, device=dev.device())
        dev.check(_lib.lib().gpr_so3_radial(dev.stream_ptr(), 1 if mode == "table" else 0, self.nmax, self.lmax,
                                            float(self.rcut), float(self.alpha), int(d_rq.shape[0]), dev.ptr(d_G),
                                            dev.ptr(d_rq), nseg, ncheb, dev.ptr(table), len(rs), dev.ptr(rs),
                                            dev.ptr(out)), "gpr_so3_radial")
        return out.cpu().numpy()

    def neighbor_index(self, atoms, atom_ids=None):
        """pairs, `seq` and the pair/row maps of one structure (the bookkeeping of
        SO3.build_neighbor_list, gpr_calc/SO3.py:348-407), as numpy arrays."""
        numbers, pos, cell, pbc = atoms_arrays(atoms)
        n = len(numbers)
        ci, cj, vec = neighbor_pairs(atoms, float(self.rcut), atom_ids)
        centres = np.arange(n) if atom_ids is None else np.asarray(list(atom_ids), dtype=np.int64)
        key = ci * n + cj
        keys = np.union1d(key, centres * n + centres)            # sorted unique (i, j) incl. (i, i)
        seq = np.stack((keys // n, keys % n), axis=1).astype(np.int64)
        row_of_pair = np.searchsorted(keys, key)
        S = len(keys)
        # pairs are ordered by (centre, neighbour, image) => by row
        row_pair0 = np.concatenate(([0], np.cumsum(np.bincount(row_of_pair, minlength=S)))).astype(np.int32)
        cidx = np.searchsorted(centres, ci)
        centre_pair0 = np.concatenate(([0], np.cumsum(np.bincount(cidx, minlength=len(centres))))).astype(np.int32)
        ridx = np.searchsorted(centres, seq[:, 0])
        centre_row0 = np.concatenate(([0], np.cumsum(np.bincount(ridx, minlength=len(centres))))).astype(np.int32)
        self_row = np.searchsorted(keys, centres * n + centres).astype(np.int32)
        w = numbers[cj].astype(np.float64)
        if self.weight_on:                                       # SO3.py:381-385
            w = np.where(numbers[cj] != numbers[ci], -w, w)
        return dict(n=n, centres=centres, numbers=numbers, vec=vec, w=w, seq=seq, centre_pair0=centre_pair0,
                    centre_row0=centre_row0, row_pair0=row_pair0, self_row=self_row)

    def neighbor_index_batch(self, strucs):
        """neighbour pairs, `seq` and the pair / row maps of a batch of structures through the native
        host routine gpr_nbr_build (csrc/nbr.hip); same content as neighbor_index, concatenated"""
        arrs = [atoms_arrays(s) for s in strucs]
        atom0 = np.concatenate(([0], np.cumsum([len(a[0]) for a in arrs]))).astype(np.int32)
        numbers = np.ascontiguousarray(np.concatenate([a[0] for a in arrs]), dtype=np.int32)
        pos = np.ascontiguousarray(np.concatenate([a[1] for a in arrs], axis=0), dtype=np.float64)
        cells = np.ascontiguousarray(np.stack([a[2] for a in arrs]).reshape(-1, 9), dtype=np.float64)
        pbc = np.ascontiguousarray(np.stack([a[3] for a in arrs]), dtype=np.int32)
        L = _lib.lib()
        h = C.c_void_p()
        ip = lambda a: a.ctypes.data_as(_lib.c_int_p)
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        dev.check(L.gpr_nbr_build(C.byref(h), len(strucs), ip(atom0), dp(pos), dp(cells), ip(pbc), ip(numbers),
                                  float(self.rcut), int(bool(self.weight_on))), "gpr_nbr_build")
        try:
            P, S = C.c_int(), C.c_int()
            dev.check(L.gpr_nbr_sizes(h, C.byref(P), C.byref(S)), "gpr_nbr_sizes")
            P, S, n = P.value, S.value, int(atom0[-1])
            out = dict(vec=np.empty((P, 3)), w=np.empty(P), centre_pair0=np.empty(n + 1, np.int32),
                       centre_row0=np.empty(n + 1, np.int32), row_pair0=np.empty(S + 1, np.int32),
                       self_row=np.empty(n, np.int32), seq=np.empty((S, 2), np.int32),
                       row0=np.empty(len(strucs) + 1, np.int32))
            dev.check(L.gpr_nbr_export(h, dp(out["vec"]), dp(out["w"]), ip(out["centre_pair0"]), ip(out["centre_row0"]),
                                       ip(out["row_pair0"]), ip(out["self_row"]), ip(out["seq"]), ip(out["row0"])),
                      "gpr_nbr_export")
        finally:
            L.gpr_nbr_destroy(h)
        out["seq"] = out["seq"].astype(np.int64)
        out["atom0"], out["numbers"] = atom0.astype(np.int64), numbers.astype(np.int64)
        out["row0"] = out["row0"].astype(np.int64)
        return out

    def neighbor_batch_dev(self, strucs):
        """device-resident neighbour pairs / seq rows of a batch (gpr_nbr_build_device); NeighborBatch"""
        return NeighborBatch(strucs, float(self.rcut), bool(self.weight_on))

    def calculate_batch_dev(self, strucs, host_index=False, nbr=None):
        """Descriptors of several structures in one launch.  Returns a dict of device tensors
        x [sum n, d], dxdr [sum S, d, 3] plus `nbr` (the NeighborBatch the rows refer to) and the host
        offsets atom0 / row0 / numbers_all.  With host_index=True the neighbour search runs on the host
        (gpr_nbr_build) and the per-structure `seq` lists are returned as numpy arrays as well."""
        table, nseg, ncheb = self._device_tables()[:3]
        D = dev.device()
        d = self.ncoefs
        if host_index:
            nb = self.neighbor_index_batch(strucs)
            atom0, row0, numbers = nb["atom0"], nb["row0"], nb["numbers"]
            nc, P, S = int(atom0[-1]), len(nb["w"]), int(row0[-1])
            t = lambda a, dt: dev.to_dev(a, dt)
            keep = [t(nb["vec"], dev.F64), t(nb["w"], dev.F64)] + \
                   [t(nb[k], torch.int32) for k in ("centre_pair0", "centre_row0", "row_pair0", "self_row")]
            ptrs = [dev.ptr(a) for a in keep]
            nbr = None
        else:
            if nbr is None:
                nbr = self.neighbor_batch_dev(strucs)
            atom0, row0, numbers = nbr.atom0, nbr.row0, nbr.numbers
            nc, P, S = nbr.natoms, nbr.P, nbr.S
            ptrs = [nbr.ptrs[k] for k in ("vec", "w", "centre_pair0", "centre_row0", "row_pair0", "self_row")]
        x = torch.empty((nc, d), dtype=dev.F64, device=D)
        dxdr = torch.empty((S, d, 3), dtype=dev.F64, device=D)
        rdxdr = None
        if self._stress:
            # stress rows (SO3.py:240-306): centre positions as given (not wrapped) and 1 / cell volume per centre
            arrs = [atoms_arrays(s) for s in strucs]
            pos = np.concatenate([a[1] for a in arrs], axis=0)
            ivol = np.concatenate([np.full(len(a[0]), 1.0 / abs(np.linalg.det(a[2]))) for a in arrs])
            d_pos, d_ivol = dev.to_dev(pos), dev.to_dev(ivol)
            rdxdr = torch.empty((S, d, 3, 3), dtype=dev.F64, device=D)
            dev.check(_lib.lib().gpr_so3_power_spectrum_stress(
                dev.stream_ptr(), self.nmax, self.lmax, float(self.rcut), float(self.alpha), nseg, ncheb, dev.ptr(table),
                nc, P, S, *ptrs, dev.ptr(x), dev.ptr(dxdr), dev.ptr(d_pos), dev.ptr(d_ivol), dev.ptr(rdxdr)),
                "gpr_so3_power_spectrum_stress")
        else:
            dev.check(_lib.lib().gpr_so3_power_spectrum(
                dev.stream_ptr(), self.nmax, self.lmax, float(self.rcut), float(self.alpha), nseg, ncheb, dev.ptr(table),
                nc, P, S, *ptrs, dev.ptr(x), dev.ptr(dxdr)), "gpr_so3_power_spectrum")
        out = dict(x=x, dxdr=dxdr, rdxdr=rdxdr, atom0=atom0, row0=row0, numbers_all=numbers, nbr=nbr,
                   numbers=[numbers[atom0[k]:atom0[k + 1]] for k in range(len(strucs))])
        if host_index:
            seq = nb["seq"]
            out["seq_all"] = seq
            out["seq"] = [seq[row0[k]:row0[k + 1]] for k in range(len(strucs))]
        return out

    def calculate(self, atoms, atom_ids=None, use_mpi=False):
        '''
        Power spectrum x [n, ncoef], its derivatives dxdr [S, ncoef, 3] per (centre, neighbour) row of
        seq [S, 2], the chemical symbols, and rdxdr (None unless stress) -- the dictionary of
        gpr_calc.SO3.SO3.calculate (SO3.py:186-323), as numpy arrays.  `use_mpi` is accepted for
        signature compatibility; the work is already one GPU launch.
        '''
        out = self.calculate_batch_dev([atoms], host_index=True)
        numbers = out["numbers"][0]
        symbols = [str(s) for s in (atoms.symbols if hasattr(atoms, "symbols") else [symbol_of(z) for z in numbers])]
        x = out["x"].cpu().numpy()
        if atom_ids is not None:
            # only the listed centres (SO3.py:353-399): x keeps its [n, ncoef] shape with the other rows zero; seq / dxdr
            # hold the rows of the listed centres, centre by centre in the order given
            ids = [int(i) for i in atom_ids]
            full, x = x, np.zeros_like(x)
            x[ids] = full[ids]
        if not self.derivative:                       # SO3.py:309-320: no seq, no derivative rows
            return {'x': x, 'dxdr': None, 'rdxdr': None, 'elements': symbols}
        seq, dxdr = out["seq"][0], out["dxdr"].cpu().numpy()
        rdxdr = None if out["rdxdr"] is None else out["rdxdr"].cpu().numpy()
        if atom_ids is not None:
            rows = np.concatenate([np.flatnonzero(seq[:, 0] == i) for i in ids] or [np.zeros(0, dtype=np.int64)])
            seq, dxdr = seq[rows], dxdr[rows]
            rdxdr = None if rdxdr is None else rdxdr[rows]
        return {'x': x, 'dxdr': dxdr, 'rdxdr': rdxdr, 'elements': symbols, 'seq': seq}


class NeighborBatch:
    """Neighbour pairs and `seq` rows of a batch of structures, built and kept on the device
    (csrc/nbr.hip: gpr_nbr_build_device).  Host side: atom0 / row0 offsets, numbers, centre_row0."""

    def __init__(self, strucs, rcut, weight_on):
        self.atom0, self.numbers, pos, cells, pbc = batch_arrays(strucs)
        self.nstruct = len(strucs)
        atom0 = self.atom0.astype(np.int32)
        numbers = self.numbers.astype(np.int32)
        L = _lib.lib()
        ip = lambda a: a.ctypes.data_as(_lib.c_int_p)
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        self._h = C.c_void_p()
        self.natoms = int(atom0[-1])
        dev.check(L.gpr_nbr_build_device(C.byref(self._h), dev.stream_ptr(), self.nstruct, ip(atom0), dp(pos), dp(cells),
                                         ip(pbc), ip(numbers), rcut, int(weight_on)), "gpr_nbr_build_device")
        P, S = C.c_int(), C.c_int()
        dev.check(L.gpr_nbr_sizes(self._h, C.byref(P), C.byref(S)), "gpr_nbr_sizes")
        self.P, self.S = P.value, S.value
        self.centre_row0 = np.empty(self.natoms + 1, np.int32)
        row0 = np.empty(self.nstruct + 1, np.int32)
        dev.check(L.gpr_nbr_export(self._h, None, None, None, ip(self.centre_row0), None, None, None, ip(row0)),
                  "gpr_nbr_export")
        self.row0 = row0.astype(np.int64)
        names = ("vec", "w", "centre_pair0", "centre_row0", "row_pair0", "self_row", "seq", "numbers", "atom0")
        raw = [C.c_void_p() for _ in names]
        dev.check(L.gpr_nbr_device_arrays(self._h, *[C.byref(r) for r in raw]), "gpr_nbr_device_arrays")
        self.ptrs = dict(zip(names, raw))

    def force_rows(self, atoms=None):
        """(counts, row_index, x_index, pt_row0) of the force points of the moving atoms `atoms` (global ids; None =
        every atom): device int32 index arrays into dxdr rows / x rows and the point row offsets (gpr_nbr_force_rows)"""
        if atoms is None:                                          # nothing to upload: point a is atom a
            counts = self.centre_row0[1:] - self.centre_row0[:-1]
            nrows, npts = self.S, self.natoms
            p_atoms, p_row0 = C.c_void_p(None), C.c_void_p(None)
            row0_dev = dev.DevArray(self.ptrs["centre_row0"].value, self.natoms + 1, torch.int32, self)
        else:
            atoms = np.ascontiguousarray(atoms, dtype=np.int32)
            counts = (self.centre_row0[1:] - self.centre_row0[:-1])[atoms]
            meta = np.empty(2 * len(atoms) + 1, dtype=np.int32)    # [atoms | pt_row0] in one upload
            meta[:len(atoms)] = atoms
            meta[len(atoms)] = 0
            np.cumsum(counts, out=meta[len(atoms) + 1:])
            nrows, npts = int(meta[-1]), len(atoms)
            d_meta = dev.to_dev(meta, torch.int32)
            p_atoms, p_row0 = dev.ptr(d_meta), C.c_void_p(d_meta.data_ptr() + 4 * len(atoms))
            row0_dev = d_meta[len(atoms):]
        idx = torch.empty((2, max(nrows, 1)), dtype=torch.int32, device=dev.device())
        dev.check(_lib.lib().gpr_nbr_force_rows(self._h, npts, p_atoms, p_row0, nrows, dev.ptr(idx[0]), dev.ptr(idx[1])),
                  "gpr_nbr_force_rows")
        return counts, idx[0, :nrows], idx[1, :nrows], row0_dev

    def numbers_dev(self):
        """atomic numbers of all atoms, int32, on the device (uploaded with the geometry)"""
        return dev.DevArray(self.ptrs["numbers"].value, self.natoms, torch.int32, self)

    def atom0_dev(self):
        """atom offsets of the structures, int32 [nstruct + 1], on the device"""
        return dev.DevArray(self.ptrs["atom0"].value, self.nstruct + 1, torch.int32, self)

    def status_into(self, slot):
        """queue the missing-row count (as a double) into the one-element device tensor `slot`"""
        dev.check(_lib.lib().gpr_nbr_status_to(self._h, dev.ptr(slot)), "gpr_nbr_status_to")

    def check(self):
        n = C.c_int()
        dev.check(_lib.lib().gpr_nbr_check(self._h, C.byref(n)), "gpr_nbr_check")
        if n.value:
            raise RuntimeError(f"neighbour table is not symmetric: {n.value} force rows missing")

    def export(self):
        """host copies in the layout of SO3.neighbor_index_batch (tests, debugging)"""
        out = dict(vec=np.empty((self.P, 3)), w=np.empty(self.P), centre_pair0=np.empty(self.natoms + 1, np.int32),
                   centre_row0=np.empty(self.natoms + 1, np.int32), row_pair0=np.empty(self.S + 1, np.int32),
                   self_row=np.empty(self.natoms, np.int32), seq=np.empty((self.S, 2), np.int32),
                   row0=np.empty(self.nstruct + 1, np.int32))
        ip = lambda a: a.ctypes.data_as(_lib.c_int_p)
        dp = lambda a: a.ctypes.data_as(_lib.c_double_p)
        dev.check(_lib.lib().gpr_nbr_export(self._h, dp(out["vec"]), dp(out["w"]), ip(out["centre_pair0"]),
                                            ip(out["centre_row0"]), ip(out["row_pair0"]), ip(out["self_row"]),
                                            ip(out["seq"]), ip(out["row0"])), "gpr_nbr_export")
        return out

    def close(self):
        if self._h:
            _lib.lib().gpr_nbr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
