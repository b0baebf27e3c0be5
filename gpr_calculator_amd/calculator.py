"""GPR: the ASE calculator of the reference (gpr_calc/calculator.py:10-181) on top of the GPU-resident GP.

`GPR(base=..., ff=gp, save=..., tag=..., freq=..., stress=..., f_tol=..., return_std=...)` predicts with
the surrogate, compares the predictive standard deviations with the tolerances, falls back to the base
calculator when they are exceeded, queues the new data and triggers a refit -- the same decision logic,
thresholds and counters as the reference.  ASE is imported lazily: the class derives from
ase.calculators.calculator.Calculator when ASE is installed and from a minimal stand-in otherwise (so
that the decision logic stays testable in containers without ASE).
"""
import numpy as np

from .neighbors import fixed_indices

try:                                            # pragma: no cover - depends on the environment
    from ase.calculators.calculator import Calculator, all_changes
    HAVE_ASE = True
except Exception:                               # ASE absent: minimal protocol stand-in
    HAVE_ASE = False
    all_changes = ['positions', 'numbers', 'cell', 'pbc', 'initial_charges', 'initial_magmoms']

    class _Params(dict):
        __getattr__ = dict.__getitem__

        def __setattr__(self, k, v):
            self[k] = v

    class Calculator:
        """the part of ase.calculators.calculator.Calculator that GPR relies on"""
        implemented_properties = []
        nolabel = False

        def __init__(self, **kwargs):
            self.parameters = _Params(kwargs)
            self.results = {}
            self.atoms = None
            self.name = self.__class__.__name__.lower()

        def calculate(self, atoms=None, properties=('energy',), system_changes=all_changes):
            if atoms is not None:
                self.atoms = atoms.copy()

        def get_potential_energy(self, atoms=None):
            self.calculate(atoms)
            return self.results['energy']

        def get_forces(self, atoms=None):
            self.calculate(atoms)
            return self.results['forces']


class GPR(Calculator):
    implemented_properties = ['energy', 'forces', 'stress', 'var_e', 'var_f']
    nolabel = True

    def __init__(self, **kwargs):
        Calculator.__init__(self, **kwargs)
        self.results = {}
        self.force_base = False
        self.allow_base = True
        self.update_gpr = True
        self.verbose = True
        self.ignore_E_std = True
        self.tag = self.parameters.tag if 'tag' in self.parameters else 'GPR'
        self.freq = self.parameters.freq if 'freq' in self.parameters else 10
        self.save = self.parameters.save if 'save' in self.parameters else True

    def freeze(self):
        self.allow_base = False
        self.update = False

    def unfreeze(self):
        self.update = True
        self.allow_base = True

    def calculate(self, atoms=None, properties=['energy', 'forces'], system_changes=all_changes):
        """One force call of an optimiser (gpr_calc/calculator.py:48-117)."""
        gp_model = self.parameters.ff
        comm, rank = gp_model.comm, gp_model.comm.rank
        fix_ids = fixed_indices(atoms)

        atoms.positions = comm.bcast(atoms.positions, root=0)
        if hasattr(atoms.cell, "array"):
            atoms.cell.array = comm.bcast(atoms.cell.array, root=0)

        self._calculate(atoms, properties, system_changes)
        if self.ignore_E_std:
            e_tol = 100
        else:
            e_tol = 1.2 * len(atoms) * gp_model.noise_e
        f_tol = 1.2 * gp_model.noise_f
        E_std, F_std = self.results['var_e'] * len(atoms), self.results['var_f'].max()
        E = self.results['energy']
        Fmax = np.abs(self.results['forces']).max()
        E_fail = E_std > e_tol
        f_ref = max(f_tol, Fmax / 2.5)
        force_fail = not (F_std < f_ref).all()
        if self.force_base or (self.allow_base and (E_fail or force_fail)):
            gp_model.use_base += 1
            if rank == 0:
                atoms.calc = self.parameters.base
                eng = atoms.get_potential_energy()
                forces = atoms.get_forces()
                forces[fix_ids] = 0.0
                atoms.calc = None
                data = (atoms.copy(), eng, forces)
                f_max = np.abs(forces).max()
                if self.verbose:
                    print(f"From Base model E: {E_std:.3f}/{E:.3f}/{eng:.3f}, F: {F_std:.3f}/{Fmax:.3f}/{f_max:.3f}")
            else:
                data, eng, forces = None, None, None
            data, eng, forces = comm.bcast((data, eng, forces), root=0)
            comm.barrier()
            gp_model.add_structure(data)
            self.results["energy"] = eng
            self.results["forces"] = forces
            atoms.calc = self
        else:
            gp_model.use_surrogate += 1
            if rank == 0 and self.verbose:
                print(f"From Surrogate  E: {E_std:.3f}/{e_tol:.3f}/{E:.3f}, F: {F_std:.3f}/{f_tol:.3f}/{Fmax:.3f}")

        # refit when enough points are queued (gpr_calc/calculator.py:101-117)
        freq = max([2, self.freq // 2]) if gp_model.N_forces > 100 else self.freq
        if self.update_gpr and (gp_model.N_queue > freq or gp_model.N_energy_queue >= 2):
            gp_model.fit(opt=True, show=False, maxiter=10)
            if rank == 0 and self.save:
                gp_model.save(f'{self.tag}-gpr.json', f'{self.tag}-gpr.db', verbose=False)
                print(gp_model)
            gp_model.validate_data(show=True)
            if gp_model.error['energy_mae'] > 0.1 or gp_model.error['forces_mae'] > 0.3:
                print("ERROR: The error is too large, check the data.")
                print(gp_model.error)
                print("The program stops here!\n")
                import sys
                sys.exit()

    def _calculate(self, atoms, properties, system_changes):
        """E / F / sigma from the GP (gpr_calc/calculator.py:119-155)"""
        Calculator.calculate(self, atoms, properties, system_changes)
        stress = self.parameters.stress if 'stress' in self.parameters else False
        f_tol = self.parameters.f_tol if 'f_tol' in self.parameters else 1e-12
        return_std = self.parameters.return_std if 'return_std' in self.parameters else True

        res = self.parameters.ff.predict_structure(atoms, stress, return_std, f_tol=f_tol)
        if return_std:
            self.results['var_e'] = res[3]
            self.results['var_f'] = res[4]
        self.results['energy'] = res[0]
        self.results['free_energy'] = res[0]
        self.results['forces'] = res[1]
        self.results['stress'] = res[2].sum(axis=0) if stress else None
        self.forces = res[1]

    def get_var_e(self, total=False):
        if total:
            return self.results["var_e"] * len(self.results["forces"])
        return self.results["var_e"]

    def get_var_f(self):
        return self.results["var_f"]

    def get_e(self, peratom=True):
        if peratom:
            return self.results["energy"] / len(self.results["forces"])
        return self.results["energy"]


class LJ:
    """Pairwise Lennard-Jones base potential with the reference's interface (gpr_calc/calculator.py:183-283):
    `LJ(parameters={'rc', 'sigma', 'epsilon'})`, `calculate(atoms) -> (energy, forces[n, 3], stress)`, `save_dict` /
    `load_from_dict`, usable as `GP(base_potential=LJ(...))` (gaussianprocess.py:579-583).  Truncated at rc and
    shifted by its value there; stress = per-atom virial in Voigt order (xx, yy, zz, yz, xz, xy) divided by the cell
    volume for fully periodic cells, None otherwise.  Host arithmetic (numpy over the periodic pair list): the base
    potential is not on the GPR hot path."""

    def __init__(self, parameters=None):
        _parameters = {'name': 'LJ', 'rc': 5.0, 'sigma': 1.0, 'epsilon': 1.0}
        if parameters is not None:
            _parameters.update(parameters)
        self.load_from_dict(_parameters)

    def __str__(self):
        return "LJ(eps: {:.3f}, sigma: {:.3f}, cutoff: {:.3f})".format(self.epsilon, self.sigma, self.rc)

    def load_from_dict(self, dict0):
        self._parameters = dict0
        self.name = self._parameters["name"]
        self.epsilon = self._parameters["epsilon"]
        self.sigma = self._parameters["sigma"]
        self.rc = self._parameters["rc"]

    def save_dict(self):
        return self._parameters

    def calculate(self, atoms):
        from .neighbors import atoms_arrays, neighbor_pairs
        sigma, epsilon, rc = self.sigma, self.epsilon, self.rc
        numbers, pos, cell, pbc = atoms_arrays(atoms)
        n = len(numbers)
        ci, cj, vec = neighbor_pairs(atoms, rc)                 # both directions: every atom sees all its neighbours
        r2 = (vec ** 2).sum(axis=1)
        c6 = (sigma ** 2 / r2) ** 3
        c12 = c6 ** 2
        e0 = 4 * epsilon * ((sigma / rc) ** 12 - (sigma / rc) ** 6)
        pair_e = 4 * epsilon * (c12 - c6) - e0
        pair_f = (-24 * epsilon * (2 * c12 - c6) / r2)[:, None] * vec        # (du/dr / r) d, d pointing to the neighbour
        energy = 0.5 * pair_e.sum()
        forces = np.zeros((n, 3))
        np.add.at(forces, ci, pair_f)
        stress = None
        if bool(np.all(pbc)):
            s33 = np.zeros((n, 3, 3))
            np.add.at(s33, ci, 0.5 * pair_f[:, :, None] * vec[:, None, :])
            voigt = np.stack((s33[:, 0, 0], s33[:, 1, 1], s33[:, 2, 2], s33[:, 1, 2], s33[:, 0, 2], s33[:, 0, 1]), axis=1)
            stress = voigt / abs(np.linalg.det(cell))
        return energy, forces, stress
