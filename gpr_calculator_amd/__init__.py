"""gpr_calculator_amd -- MI355X-native GPR hot path (descriptor -> covariance -> factor/solve/variance)
behind the GP / GPR(ase.Calculator) / kernel-object API of MaterSim/GPR_calculator."""
