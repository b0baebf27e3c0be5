"""Host-side packing and bookkeeping helpers with the reference's names and semantics
(gpr_calc/utilities.py): pure index/array shuffling, no kernel arithmetic."""
import numpy as np

from .neighbors import atomic_number


def list_to_tuple(data, stress=False, include_value=False, mode="force"):
    """Flatten per-point tuples into contiguous (X, [dXdR,] ELE, counts[, values])
    (gpr_calc/utilities.py:340-390)."""
    rows = 0
    for fd in data:
        rows += fd[0].shape[0]
    ncol = fd[0].shape[1]
    X = np.zeros([rows, ncol])
    ELE, indices, values = [], [], []
    if mode == "force":
        dXdR = np.zeros([rows, ncol, 9 if stress else 3])
    c = 0
    for fd in data:
        if mode == "force":
            if include_value:
                (x, dxdr, f, ele) = fd
                values.append(f)
            else:
                (x, dxdr, ele) = fd
            n = x.shape[0]
            dXdR[c:c + n, :ncol, :] = dxdr
        else:
            if include_value:
                (x, e, ele) = fd
                values.append(e)
            else:
                (x, ele) = fd
            n = x.shape[0]
        indices.append(n)
        X[c:c + n, :ncol] = x
        ELE.extend(ele)
        c += n
    ELE = np.ravel(ELE)
    if mode == "force":
        return (X, dXdR, ELE, indices, values) if include_value else (X, dXdR, ELE, indices)
    return (X, ELE, indices, values) if include_value else (X, ELE, indices)


def tuple_to_list(data, mode="force"):
    """Inverse of list_to_tuple (gpr_calc/utilities.py:393-406)."""
    out, c = [], 0
    if mode == "force":
        X, dXdR, ELE, indices = data
        for n in indices:
            out.append((X[c:c + n], dXdR[c:c + n], ELE[c:c + n]))
            c += n
    else:
        X, ELE, indices = data
        for n in indices:
            out.append((X[c:c + n], ELE[c:c + n]))
            c += n
    return out


def new_pt(data, Refs, d_tol=1e-1, eps=1e-8):
    """True unless an existing reference of the same species has 1 - cos^2 < d_tol
    (gpr_calc/utilities.py:32-42, including its `norm(X1 + eps)` normalisation)."""
    (X, ele) = data
    X = X / (np.linalg.norm(X) + eps)
    for (X1, ele1) in Refs:
        if ele1 == ele:
            X1 = X1 / np.linalg.norm(X1 + eps)
            d = X @ X1.T
            if 1 - d ** 2 < d_tol:
                return False
    return True


def species_numbers(symbols):
    """[Element(s).z for s in symbols] (gaussianprocess.py:788, 847; utilities.py:109)"""
    return np.array([atomic_number(s) for s in symbols])


def force_point_rows(seq, atom):
    """rows of `seq` that feel atom `atom`: np.argwhere(seq[:,1]==atom) (gaussianprocess.py:795)"""
    return np.argwhere(seq[:, 1] == atom).flatten()


def force_point_rows_all(seq, natoms):
    """force_point_rows(seq, i) for every atom i at once: one stable sort of the second column instead of natoms scans of
    it (220 scans of 5 000 rows per labelled Pd4 / MgO structure were 7 ms of the 13 an add_structure took).  Same row
    indices in the same (ascending) order as the scans give."""
    col = np.asarray(seq[:, 1])
    order = np.argsort(col, kind="stable")
    bounds = np.searchsorted(col[order], np.arange(natoms + 1))
    return [order[bounds[i]:bounds[i + 1]] for i in range(natoms)]


def convert_train_data(data, des, N_force=100000):
    """structures -> {"energy": [(x, E/n, ele)], "force": [(x_rows, dxdr_rows, F_i, ele_rows)], "db": [...]}
    (gpr_calc/utilities.py:97-129; its dedup branch is unreachable, so every atom becomes a force
    point -- SURVEY.md Appendix A.20)."""
    energy_data, force_data, db_data = [], [], []
    for (struc, energy, forces) in data:
        d = des.calculate(struc)
        ele = species_numbers(d["elements"])
        f_ids = []
        rows_of = force_point_rows_all(d["seq"], len(struc))
        for i in range(len(struc)):
            if len(force_data) < N_force:
                ids = rows_of[i]
                _i = d["seq"][ids, 0]
                force_data.append((d["x"][_i, :], d["dxdr"][ids], forces[i], ele[_i]))
                f_ids.append(i)
        energy_data.append((d["x"], energy / len(struc), ele))
        db_data.append((struc, energy, forces, True, f_ids))
    return {"energy": energy_data, "force": force_data, "db": db_data}


def rmse(true, predicted):
    true, predicted = np.array(true), np.array(predicted)
    return float(np.sqrt(np.sum((true - predicted) ** 2 / len(true))))


def mae(true, predicted):
    true, predicted = np.array(true), np.array(predicted)
    return float(np.sum(np.abs(true - predicted) / len(true)))


def r2(true, predicted):
    """1 - SSE / (variance + 1e-8) (gpr_calc/utilities.py:54-63)"""
    if len(true) == 0:
        return 1
    true, predicted = np.array(true), np.array(predicted)
    t_bar = np.sum(true) / len(true)
    return float(1 - np.sum((true - predicted) ** 2) / (np.sum((true - t_bar) ** 2) + 1e-8))


def metric_values(y, y_pred):
    """(r2, mae, rmse) (gpr_calc/utilities.py:81-85)"""
    return r2(y, y_pred), mae(y, y_pred), rmse(y, y_pred)
