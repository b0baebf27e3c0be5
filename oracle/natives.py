"""oracle/natives.py -- TEST INFRASTRUCTURE ONLY (checker; never imported by gpr_calculator_amd).

ctypes bindings for the 13 CPU natives in three interchangeable flavours:
  * NativeSet("ref")   -> oracle/_ref/lib{rbf,dot}_ref.so, the UNMODIFIED reference C++
                          (gpr_calc/kernels/rbf_kernel.cpp, dot_kernel.cpp) built by oracle/Makefile
  * NativeSet("port")  -> oracle/libgpr_oracle.so, our plain-C restatement (kernels_port.c)
Signatures follow gpr_calc/kernels/rbf_kernel.h:4-38 and dot_kernel.h:4-27.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_D, _I = C.POINTER(C.c_double), C.POINTER(C.c_int)
i, f = C.c_int, C.c_double

SIGS = {
    "rbf_kee_many":           [i, i, i, i, f, f, f, _D, _I, _I, _D, _I, _I, _D],
    "rbf_kee_many_with_grad": [i, i, i, i, f, f, f, _D, _I, _I, _D, _I, _I, _D, _D],
    "rbf_kef_many":           [i, i, i, i, f, f, f, _D, _I, _I, _D, _D, _I, _I, _D],
    "rbf_kef_many_with_grad": [i, i, i, i, f, f, f, _D, _I, _I, _D, _D, _I, _I, _D],
    "rbf_kef_many_stress":    [i, i, i, i, f, f, f, _D, _I, _I, _D, _D, _I, _I, _D],
    "rbf_kff_many":           [i, i, i, i, i, i, f, f, f, f, _D, _D, _I, _I, _D, _D, _I, _I, _D],
    "rbf_kff_many_with_grad": [i, i, i, i, i, i, f, f, f, _D, _D, _I, _I, _D, _D, _I, _I, _D, _D],
    "rbf_kff_many_stress":    [i, i, i, i, i, i, f, f, f, f, _D, _D, _I, _I, _D, _D, _I, _I, _D],
    "dot_kee_many":           [i, i, i, i, f, f, f, _D, _I, _I, _D, _I, _I, _D],
    "dot_kef_many":           [i, i, i, i, f, _D, _I, _I, _D, _D, _I, _I, _D],
    "dot_kef_many_stress":    [i, i, i, i, f, _D, _I, _I, _D, _D, _I, _I, _D],
    "dot_kff_many":           [i, i, i, i, i, i, f, _D, _D, _I, _I, _D, _D, _I, _I, _D],
    "dot_kff_many_stress":    [i, i, i, i, i, i, f, _D, _D, _I, _I, _D, _D, _I, _I, _D],
}


def _as(arr, ctype):
    return arr.ctypes.data_as(ctype)


class NativeSet:
    def __init__(self, flavour="port", libs=None, prefix=None):
        self.flavour = flavour
        if libs is None:
            if flavour == "ref":
                libs = [os.path.join(HERE, "_ref", "librbf_ref.so"),
                        os.path.join(HERE, "_ref", "libdot_ref.so")]
                prefix = ""
            elif flavour == "port":
                libs = [os.path.join(HERE, "libgpr_oracle.so")]
                prefix = "port_"
            else:
                raise ValueError(flavour)
        self._libs = [C.CDLL(p) for p in libs]
        self.fn = {}
        for name, sig in SIGS.items():
            for lib in self._libs:
                try:
                    fn = getattr(lib, (prefix or "") + name)
                except AttributeError:
                    continue
                fn.argtypes, fn.restype = sig, None
                self.fn[name] = fn
                break
            else:
                raise OSError(f"{name} not exported by {libs}")

    @staticmethod
    def available(flavour):
        if flavour == "ref":
            return os.path.exists(os.path.join(HERE, "_ref", "librbf_ref.so"))
        return os.path.exists(os.path.join(HERE, "libgpr_oracle.so"))

    def call(self, name, *args):
        """Scalars pass through; numpy arrays are passed by pointer (must be C-contiguous
        float64 / int32). Outputs are accumulated into the arrays the caller passes."""
        sig = SIGS[name]
        conv = []
        keep = []
        for a, t in zip(args, sig):
            if t is _D:
                assert a.dtype == np.float64 and a.flags.c_contiguous
                conv.append(_as(a, _D)); keep.append(a)
            elif t is _I:
                assert a.dtype == np.int32 and a.flags.c_contiguous
                conv.append(_as(a, _I)); keep.append(a)
            else:
                conv.append(a)
        assert len(conv) == len(sig), (name, len(conv), len(sig))
        self.fn[name](*conv)


def expand_counts(counts):
    """per-row point id from per-point row counts (rbf_kernel.py:32-36)."""
    return np.repeat(np.arange(len(counts), dtype=np.int32), counts).astype(np.int32)


def make_case(seed, d, counts1, counts2, species=(46,), zeta=2.0, sigma=2.5, l=1.3, sigma0=2.0,
              tol=1e-12, n2_range=None, zero_rows=()):
    """Random-row inputs in the style of SURVEY.md section 8(d) micro-bench inputs."""
    rng = np.random.default_rng(seed)
    n1, n2 = int(sum(counts1)), int(sum(counts2))
    c = dict(d=d, zeta=float(zeta), sigma=float(sigma), l=float(l), sigma0=float(sigma0),
             tol=float(tol), counts1=np.asarray(counts1, np.int32),
             counts2=np.asarray(counts2, np.int32))
    for tag, n in (("1", n1), ("2", n2)):
        c["x" + tag] = rng.random((n, d))
        c["dx%sdr" % tag] = rng.standard_normal((n, d, 3))
        c["dx%sdr9" % tag] = np.concatenate(
            [c["dx%sdr" % tag], rng.standard_normal((n, d, 6))], axis=2).copy()
        c["ele" + tag] = rng.choice(np.asarray(species, np.int32), size=n).astype(np.int32)
    for (tag, row) in zero_rows:
        c["x" + tag][row] = 0.0
    c["n2_range"] = np.asarray(n2_range if n2_range is not None else (0, n2), np.int32)
    return c


def run_all_natives(ns, c):
    """Call the 13 natives on case c with zeroed outputs; returns {name: array}."""
    d, zeta = int(c["d"]), float(c["zeta"])
    s2, l, l2, s02, tol = c["sigma"] ** 2, float(c["l"]), c["l"] ** 2, c["sigma0"] ** 2, float(c["tol"])
    x1, x2 = np.ascontiguousarray(c["x1"]), np.ascontiguousarray(c["x2"])
    A3, B3 = np.ascontiguousarray(c["dx1dr"]), np.ascontiguousarray(c["dx2dr"])
    A9, B9 = np.ascontiguousarray(c["dx1dr9"]), np.ascontiguousarray(c["dx2dr9"])
    e1, e2 = c["ele1"].astype(np.int32), c["ele2"].astype(np.int32)
    i1, i2 = expand_counts(c["counts1"]), expand_counts(c["counts2"])
    n1, n2, m1, m2 = len(x1), len(x2), len(c["counts1"]), len(c["counts2"])
    s, e = int(c["n2_range"][0]), int(c["n2_range"][1])
    z = lambda *shape: np.zeros(shape)
    o = {}
    o["rbf_kee_many"] = z(m1, m2)
    ns.call("rbf_kee_many", n1, n2, d, m2, zeta, s2, l2, x1, e1, i1, x2, e2, i2, o["rbf_kee_many"])
    o["rbf_kee_many_with_grad"], o["rbf_kee_many_with_grad.dl"] = z(m1, m2), z(m1, m2)
    ns.call("rbf_kee_many_with_grad", n1, n2, d, m2, zeta, s2, l2, x1, e1, i1, x2, e2, i2,
            o["rbf_kee_many_with_grad"], o["rbf_kee_many_with_grad.dl"])
    o["rbf_kef_many"] = z(m1, m2, 3)
    ns.call("rbf_kef_many", n1, n2, d, m2, zeta, s2, l2, x1, e1, i1, x2, B3, e2, i2, o["rbf_kef_many"])
    o["rbf_kef_many_with_grad"] = z(m1, m2, 6)
    ns.call("rbf_kef_many_with_grad", n1, n2, d, m2, zeta, s2, l, x1, e1, i1, x2, B3, e2, i2,
            o["rbf_kef_many_with_grad"])
    o["rbf_kef_many_stress"] = z(m1, m2, 9)
    ns.call("rbf_kef_many_stress", n1, n2, d, m2, zeta, s2, l2, x1, e1, i1, x2, B9, e2, i2,
            o["rbf_kef_many_stress"])
    o["rbf_kff_many"] = z(m1, 3, m2, 3)
    ns.call("rbf_kff_many", n1, n2, s, e, d, m2, zeta, s2, l2, tol, x1, A3, e1, i1, x2, B3, e2, i2,
            o["rbf_kff_many"])
    o["rbf_kff_many_with_grad"], o["rbf_kff_many_with_grad.dl"] = z(m1, 3, m2, 3), z(m1, 3, m2, 3)
    ns.call("rbf_kff_many_with_grad", n1, n2, s, e, d, m2, zeta, s2, l, x1, A3, e1, i1, x2, B3, e2,
            i2, o["rbf_kff_many_with_grad"], o["rbf_kff_many_with_grad.dl"])
    o["rbf_kff_many_stress"] = z(m1, 9, m2, 3)
    ns.call("rbf_kff_many_stress", n1, n2, s, e, d, m2, zeta, s2, l2, tol, x1, A9, e1, i1, x2, B3,
            e2, i2, o["rbf_kff_many_stress"])
    o["dot_kee_many"] = z(m1, m2)
    ns.call("dot_kee_many", n1, n2, d, m2, zeta, s2, s02, x1, e1, i1, x2, e2, i2, o["dot_kee_many"])
    o["dot_kef_many"] = z(m1, m2, 3)
    ns.call("dot_kef_many", n1, n2, d, m2, zeta, x1, e1, i1, x2, B3, e2, i2, o["dot_kef_many"])
    o["dot_kef_many_stress"] = z(m1, m2, 9)
    ns.call("dot_kef_many_stress", n1, n2, d, m2, zeta, x1, e1, i1, x2, B9, e2, i2,
            o["dot_kef_many_stress"])
    o["dot_kff_many"] = z(m1, 3, m2, 3)
    ns.call("dot_kff_many", n1, n2, s, e, d, m2, zeta, x1, A3, e1, i1, x2, B3, e2, i2, o["dot_kff_many"])
    o["dot_kff_many_stress"] = z(m1, 9, m2, 3)
    ns.call("dot_kff_many_stress", n1, n2, s, e, d, m2, zeta, x1, A9, e1, i1, x2, B3, e2, i2,
            o["dot_kff_many_stress"])
    return o
