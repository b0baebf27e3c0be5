"""Periodic neighbour search with the semantics the reference obtains from
ase.neighborlist.NeighborList(cutoffs=[rcut/2]*n, self_interaction=False, bothways=True, skin=0.0)
(gpr_calc/SO3.py:348-407): every (i, j, image) with |r_j + image.cell - r_i| < rcut, periodic
self-images (i == j, image != 0) included, non-periodic axes not replicated.

Host-side index arithmetic only (integers and a distance predicate); the descriptor arithmetic on
these pairs runs on the GPU (csrc/so3.hip)."""
import numpy as np


class SimpleAtoms:
    """Minimal stand-in for ase.Atoms (positions, cell, pbc, numbers) used when ASE is not
    installed; SO3 / GP accept either."""

    def __init__(self, numbers, positions, cell=None, pbc=False, fixed=()):
        self.numbers = np.asarray(numbers, dtype=np.int64)
        self.positions = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        self.cell = np.zeros((3, 3)) if cell is None else np.asarray(cell, dtype=np.float64).reshape(3, 3)
        self.pbc = np.array([pbc] * 3 if isinstance(pbc, (bool, np.bool_)) else pbc, dtype=bool)
        self.fixed = np.asarray(list(fixed), dtype=np.int64)

    def __len__(self):
        return len(self.numbers)

    def __getitem__(self, i):
        return _AtomView(int(self.numbers[i]), self.positions[i])

    def get_cell(self):
        return self.cell

    def get_atomic_numbers(self):
        return self.numbers

    def get_chemical_symbols(self):
        return [symbol_of(z) for z in self.numbers]

    @property
    def symbols(self):
        return self.get_chemical_symbols()

    def get_volume(self):
        return abs(np.linalg.det(self.cell))

    def copy(self):
        return SimpleAtoms(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(), self.fixed.copy())


class _AtomView:
    def __init__(self, number, position):
        self.number, self.position = number, position


_SYMBOLS = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
            "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu "
            "Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr").split()
_Z_OF = {s: z for z, s in enumerate(_SYMBOLS)}


def symbol_of(z):
    return _SYMBOLS[int(z)]


def atomic_number(symbol):
    """Element(symbol).z of pyxtal (gaussianprocess.py:788, 847): a periodic-table lookup."""
    return _Z_OF[symbol]


def atoms_arrays(atoms):
    """(numbers[n], positions[n,3], cell[3,3], pbc[3]) of an ase.Atoms or SimpleAtoms"""
    numbers = np.asarray(atoms.get_atomic_numbers() if hasattr(atoms, "get_atomic_numbers") else atoms.numbers)
    pos = np.asarray(atoms.positions, dtype=np.float64)
    cell = np.asarray(atoms.get_cell() if hasattr(atoms, "get_cell") else atoms.cell, dtype=np.float64).reshape(3, 3)
    pbc = np.asarray(atoms.pbc, dtype=bool).reshape(-1)
    if pbc.size == 1:
        pbc = np.repeat(pbc, 3)
    return numbers.astype(np.int64), pos, cell, pbc


def batch_arrays(strucs):
    """(atom0[ns+1] int64, numbers[N] int64, positions[N,3], cells[ns,9], pbc[ns,3] int32) of a list of ase.Atoms /
    SimpleAtoms: the concatenated form the batched neighbour search takes.  One pass of attribute reads and four
    numpy calls (the per-structure atoms_arrays conversions cost more than the search itself for small cells)."""
    ns = len(strucs)
    atom0 = np.zeros(ns + 1, dtype=np.int64)
    if ns == 0:
        return atom0, np.zeros(0, np.int64), np.zeros((0, 3)), np.zeros((0, 9)), np.zeros((0, 3), np.int32)
    try:
        nums = [s.numbers for s in strucs]                  # ase.Atoms and SimpleAtoms both expose these as arrays
        poss = [s.positions for s in strucs]
        cells = np.array([s.cell for s in strucs], dtype=np.float64).reshape(ns, 9)
        pbc = np.array([s.pbc for s in strucs]).astype(np.int32)
        if pbc.shape != (ns, 3):
            raise ValueError
        np.cumsum([len(a) for a in nums], out=atom0[1:])
        numbers = np.concatenate(nums).astype(np.int64)
        pos = np.ascontiguousarray(np.concatenate(poss, axis=0), dtype=np.float64)
        if pos.shape != (int(atom0[-1]), 3):
            raise ValueError
        return atom0, numbers, pos, np.ascontiguousarray(cells), np.ascontiguousarray(pbc)
    except (AttributeError, ValueError, TypeError):
        arrs = [atoms_arrays(s) for s in strucs]
        np.cumsum([len(a[0]) for a in arrs], out=atom0[1:])
        numbers = np.concatenate([a[0] for a in arrs]).astype(np.int64)
        pos = np.ascontiguousarray(np.concatenate([a[1] for a in arrs], axis=0), dtype=np.float64)
        cells = np.ascontiguousarray(np.stack([a[2] for a in arrs]).reshape(-1, 9), dtype=np.float64)
        pbc = np.ascontiguousarray(np.stack([a[3] for a in arrs]), dtype=np.int32)
        return atom0, numbers, pos, cells, pbc


def has_fixed_atoms(atoms):
    """cheap test before fixed_indices: does the structure carry any constraint at all"""
    fx = getattr(atoms, "fixed", None)
    if fx is not None:
        return len(fx) > 0
    return bool(getattr(atoms, "constraints", None))


def fixed_indices(atoms):
    """indices held by an ASE FixAtoms constraint (gaussianprocess.py:823-832) or SimpleAtoms.fixed"""
    if hasattr(atoms, "fixed"):
        return np.asarray(atoms.fixed, dtype=np.int64)
    out = []
    for c in getattr(atoms, "constraints", []) or []:
        if type(c).__name__ == "FixAtoms":
            out = c.get_indices()
            break
    return np.asarray(out, dtype=np.int64)


def image_ranges(cell, pbc, rcut):
    """number of periodic images needed along each axis to cover a sphere of radius rcut"""
    nimg = np.zeros(3, dtype=np.int64)
    if not pbc.any():
        return nimg
    vol = abs(np.linalg.det(cell))
    if vol < 1e-12:
        raise ValueError("periodic axis with a singular cell")
    for k in range(3):
        if pbc[k]:
            a, b = cell[(k + 1) % 3], cell[(k + 2) % 3]
            height = vol / np.linalg.norm(np.cross(a, b))
            nimg[k] = int(np.ceil(rcut / height))
    return nimg


def neighbor_pairs(atoms, rcut, atom_ids=None):
    """All (centre i, neighbour j, displacement r_ij) with |r_ij| < rcut.

    Returns (centres[P], neighbours[P], vectors[P,3]) ordered by centre, then by neighbour index,
    then by image -- the per-centre grouping the reference builds in SO3.build_neighbor_list."""
    numbers, pos, cell, pbc = atoms_arrays(atoms)
    n = len(numbers)
    ids = np.arange(n) if atom_ids is None else np.asarray(list(atom_ids), dtype=np.int64)
    wrapped = pos.copy()
    if pbc.any():
        frac = np.linalg.solve(cell.T, pos.T).T
        for k in range(3):
            if pbc[k]:
                frac[:, k] -= np.floor(frac[:, k])
        wrapped = frac @ cell
    nimg = image_ranges(cell, pbc, rcut)
    rng = [np.arange(-m, m + 1) for m in nimg]
    imgs = np.array(np.meshgrid(*rng, indexing="ij")).reshape(3, -1).T      # [I,3]
    shifts = imgs @ cell                                                     # [I,3]
    zero_img = np.all(imgs == 0, axis=1)
    # displacement tensor [C, n, I, 3] (sizes here are a few hundred atoms x <= 125 images)
    disp = wrapped[None, :, None, :] + shifts[None, None, :, :] - wrapped[ids][:, None, None, :]
    dist2 = np.einsum("cjik,cjik->cji", disp, disp)
    mask = dist2 < rcut * rcut
    self_mask = (ids[:, None] == np.arange(n)[None, :])[:, :, None] & zero_img[None, None, :]
    mask &= ~self_mask
    c, j, im = np.nonzero(mask)
    return ids[c], j, disp[c, j, im]


class NeighborListCompat:
    """Drop-in for the three calls SO3.build_neighbor_list makes on ase's NeighborList
    (constructor, update(atoms), get_neighbors(i) -> (indices, offsets)).  Used only by
    tests/golden/make_golden_python.py to feed the reference's SO3.calculate a neighbour list in
    a container without ASE.  Offsets are integer image vectors relative to the WRAPPED positions;
    update() therefore also wraps atoms.positions' periodic components in place of ASE's own
    handling so that positions[j] + offset.cell - positions[i] is the true displacement."""

    def __init__(self, cutoffs, self_interaction=False, bothways=True, skin=0.0):
        assert bothways and not self_interaction and skin == 0.0
        self.rcut = 2.0 * float(cutoffs[0])

    def update(self, atoms):
        numbers, pos, cell, pbc = atoms_arrays(atoms)
        frac_shift = np.zeros((len(numbers), 3), dtype=np.int64)
        if pbc.any():
            frac = np.linalg.solve(cell.T, pos.T).T
            for k in range(3):
                if pbc[k]:
                    frac_shift[:, k] = np.floor(frac[:, k]).astype(np.int64)
        c, j, vec = neighbor_pairs(atoms, self.rcut)
        self._by_centre = {}
        # offset o such that pos[j] + o.cell - pos[i] = vec
        if len(c):
            raw = vec - (pos[j] - pos[c])
            off = np.rint(np.linalg.solve(cell.T, raw.T).T).astype(np.int64) if pbc.any() else np.zeros((len(c), 3), np.int64)
        else:
            off = np.zeros((0, 3), np.int64)
        for i in range(len(numbers)):
            sel = c == i
            self._by_centre[i] = (j[sel], off[sel])

    def get_neighbors(self, i):
        return self._by_centre[i]
