"""GPU: (1) the multi-GPU sharding of the covariance assembly is exact -- the outputs of all shards sum,
bit for bit, to the unsharded matrix (world simulated on one device); (2) the GPR calculator's
surrogate-or-base decision, queueing and refit trigger (gpr_calc/calculator.py:48-117)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_assembly_sums_exactly_to_the_full_matrix():
    import torch
    from gpr_calculator_amd import device as dev
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from tests.test_gpu_blocks import make_data, tupled
    train = tupled(make_data(5, 9, 61))
    new = make_data(6, 3, 17)
    k = RBF_mb(para=[2.5, 1.3], zeta=2.0)
    full = k.k_total_dev(train)
    full_t = k.k_total_dev(new, train, 1e-8, transpose=True)
    for world in (2, 3, 8):
        acc, acc_t = torch.zeros_like(full), torch.zeros_like(full_t)
        try:
            for r in range(world):
                dev.set_shard(world, r)
                acc += k.k_total_dev(train)
                acc_t += k.k_total_dev(new, train, 1e-8, transpose=True)
        finally:
            dev.set_shard(1, 0)
        # (C_fe is each rank's copy of its own partial C_ef^T, so it sums to the full block as well)
        assert torch.equal(acc, full) and torch.equal(acc_t, full_t)


class _Base:
    """a fake 'ab initio' base calculator: Lennard-Jones labels"""
    name = "lj"

    def __init__(self):
        self.calls = 0


def test_gpr_calculator_decision_and_refit():
    from gpr_calculator_amd.calculator import GPR
    from gpr_calculator_amd.gaussianprocess import GP
    from gpr_calculator_amd.SO3 import SO3
    from gpr_calculator_amd.kernels.RBF_mb import RBF_mb
    from gpr_calculator_amd.synthetic import fcc_cell, lj_energy_forces

    class Atoms(type(fcc_cell(0, reps=1))):
        calc = None

        def get_potential_energy(self):
            self.calc_obj.calls += 1
            return lj_energy_forces(self)[0]

        def get_forces(self):
            return lj_energy_forces(self)[1]

        def copy(self):
            c = Atoms(self.numbers.copy(), self.positions.copy(), self.cell.copy(), self.pbc.copy(), self.fixed.copy())
            c.calc_obj = self.calc_obj
            return c

    base = _Base()

    def make(seed):
        s = fcc_cell(seed, reps=1, sigma=0.05)
        a = Atoms(s.numbers, s.positions, s.cell, s.pbc)
        a.calc_obj = base
        return a

    gp = GP(kernel=RBF_mb(para=[1.0, 0.5], zeta=2.0), descriptor=SO3(nmax=3, lmax=4, rcut=5.0), noise_e=0.002,
            noise_f=0.1, log_file="/tmp/gpr_test.log")
    for s in range(3):
        a = make(s)
        E, F = lj_energy_forces(a)
        gp.add_structure((a, E, F))
    gp.fit(show=False, opt=False)
    calc = GPR(base=base, ff=gp, save=False)
    calc.verbose = False
    calc.update_gpr = False
    # a training structure: the surrogate is confident -> no base call
    a0 = make(0)
    calc.calculate(a0)
    assert gp.use_surrogate == 1 and gp.use_base == 0 and base.calls == 0
    assert calc.results["forces"].shape == (4, 3) and calc.results["stress"] is None
    assert calc.get_var_f().shape == (4, 3) and calc.get_e(peratom=False) == calc.results["energy"]
    # an unseen, distorted structure with the energy criterion switched on: E_std * n exceeds
    # 1.2 * n * noise_e -> base model is called and the structure is queued
    far = make(50)
    far.positions = far.positions + np.random.default_rng(1).normal(0, 0.25, far.positions.shape)
    n_before = gp.N_queue
    calc.ignore_E_std = False
    calc.calculate(far)
    calc.ignore_E_std = True
    assert gp.use_base == 1 and base.calls == 1 and gp.N_queue > n_before
    assert abs(calc.results["energy"] - lj_energy_forces(far)[0]) < 1e-12
    # forcing the base model always takes the fallback branch
    calc.force_base = True
    calc.calculate(make(7))
    assert gp.use_base == 2
