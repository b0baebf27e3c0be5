/* _pygather -- marshalling helper of the batched prediction (host side only, no model arithmetic): walks a list of
 * ase.Atoms-like objects once and concatenates numbers / positions / cell / pbc into flat buffers, the job
 * neighbors.batch_arrays does with a dozen numpy calls.  gather(strucs) returns
 *     (atom0 int32[ns+1], numbers int32[N], positions f64[3N], cells f64[9 ns], pbc int32[3 ns], any_fixed, mask)
 * mask: uint8[N], 0 for the atoms listed in a structure's `fixed` array, or None when nothing is fixed or the fixed
 * atoms come as ase constraints (the caller resolves those)
 * as bytearrays (np.frombuffer views them without a copy), or None when some structure does not expose plain
 * C-contiguous array buffers (ase.Cell objects, lists, ...): the caller then takes the numpy path. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

static PyObject *s_numbers, *s_positions, *s_cell, *s_pbc, *s_fixed, *s_constraints;

static char fmt_char(const Py_buffer *v) {
    const char *f = v->format ? v->format : "B";
    while (*f == '<' || *f == '>' || *f == '=' || *f == '@' || *f == '!') f++;
    return *f;
}

/* copies an integer / bool buffer of `count` items into int32; 0 on success */
static int copy_ints(const Py_buffer *v, Py_ssize_t count, int32_t *dst) {
    const char c = fmt_char(v);
    if (v->len != count * v->itemsize) return -1;
    if (v->itemsize == 8 && (c == 'l' || c == 'q' || c == 'n')) { const int64_t *s = (const int64_t *)v->buf; for (Py_ssize_t i = 0; i < count; i++) dst[i] = (int32_t)s[i]; return 0; }
    if (v->itemsize == 4 && (c == 'i' || c == 'l')) { memcpy(dst, v->buf, (size_t)count * 4); return 0; }
    if (v->itemsize == 1 && (c == '?' || c == 'b' || c == 'B')) { const unsigned char *s = (const unsigned char *)v->buf; for (Py_ssize_t i = 0; i < count; i++) dst[i] = s[i] ? 1 : 0; return 0; }
    return -1;
}

static PyObject *gather(PyObject *self, PyObject *arg) {
    (void)self;
    PyObject *seq = PySequence_Fast(arg, "gather expects a sequence of structures");
    if (!seq) return NULL;
    const Py_ssize_t ns = PySequence_Fast_GET_SIZE(seq);
    PyObject **items = PySequence_Fast_ITEMS(seq);
    PyObject *b_atom0 = PyByteArray_FromStringAndSize(NULL, (ns + 1) * 4), *b_cells = PyByteArray_FromStringAndSize(NULL, ns * 72),
             *b_pbc = PyByteArray_FromStringAndSize(NULL, ns * 12), *b_num = NULL, *b_pos = NULL, *b_mask = NULL, *result = NULL;
    int any_fixed = 0, plain = 1, unresolved = 0;
    if (!b_atom0 || !b_cells || !b_pbc) goto done;
    int32_t *atom0 = (int32_t *)PyByteArray_AS_STRING(b_atom0);
    /* pass 1: atom counts */
    atom0[0] = 0;
    for (Py_ssize_t s = 0; s < ns && plain; s++) {
        PyObject *num = PyObject_GetAttr(items[s], s_numbers);
        if (!num) { PyErr_Clear(); plain = 0; break; }
        const Py_ssize_t n = PyObject_Length(num);
        Py_DECREF(num);
        if (n < 0) { PyErr_Clear(); plain = 0; break; }
        atom0[s + 1] = atom0[s] + (int32_t)n;
    }
    if (!plain) goto done;
    const Py_ssize_t N = atom0[ns];
    b_num = PyByteArray_FromStringAndSize(NULL, N * 4);
    b_pos = PyByteArray_FromStringAndSize(NULL, N * 24);
    if (!b_num || !b_pos) goto done;
    int32_t *numbers = (int32_t *)PyByteArray_AS_STRING(b_num), *pbc = (int32_t *)PyByteArray_AS_STRING(b_pbc);
    double *pos = (double *)PyByteArray_AS_STRING(b_pos), *cells = (double *)PyByteArray_AS_STRING(b_cells);
    /* pass 2: contents */
    for (Py_ssize_t s = 0; s < ns && plain; s++) {
        const Py_ssize_t n = atom0[s + 1] - atom0[s];
        PyObject *attrs[4] = {PyObject_GetAttr(items[s], s_numbers), PyObject_GetAttr(items[s], s_positions),
                              PyObject_GetAttr(items[s], s_cell), PyObject_GetAttr(items[s], s_pbc)};
        Py_buffer v;
        for (int a = 0; a < 4 && plain; a++) {
            if (!attrs[a] || PyObject_GetBuffer(attrs[a], &v, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); plain = 0; break; }
            const char c = fmt_char(&v);
            if (a == 0) { if (copy_ints(&v, n, numbers + atom0[s])) plain = 0; }
            else if (a == 1) { if (c == 'd' && v.itemsize == 8 && v.len == n * 24) memcpy(pos + 3 * (size_t)atom0[s], v.buf, (size_t)n * 24); else plain = 0; }
            else if (a == 2) { if (c == 'd' && v.itemsize == 8 && v.len == 72) memcpy(cells + 9 * s, v.buf, 72); else plain = 0; }
            else { if (copy_ints(&v, 3, pbc + 3 * s)) plain = 0; }
            PyBuffer_Release(&v);
        }
        for (int a = 0; a < 4; a++) Py_XDECREF(attrs[a]);
        if (!plain) break;
        /* fixed atoms: SimpleAtoms.fixed (array) or a non-empty ase constraints list */
        PyObject *fx = PyObject_GetAttr(items[s], s_fixed);
        if (fx) {
            const Py_ssize_t nfx = fx != Py_None ? PyObject_Length(fx) : 0;
            if (PyErr_Occurred()) PyErr_Clear();
            if (nfx > 0) {
                any_fixed = 1;
                Py_buffer fv;
                int ok = 0;
                if (PyObject_GetBuffer(fx, &fv, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) == 0) {
                    const char c = fmt_char(&fv);
                    if (!b_mask) {
                        b_mask = PyByteArray_FromStringAndSize(NULL, N);
                        if (b_mask) memset(PyByteArray_AS_STRING(b_mask), 1, (size_t)N);
                    }
                    if (b_mask && fv.len == nfx * fv.itemsize) {
                        unsigned char *mk = (unsigned char *)PyByteArray_AS_STRING(b_mask) + atom0[s];
                        ok = 1;
                        for (Py_ssize_t i = 0; i < nfx; i++) {
                            long long idx;
                            if (fv.itemsize == 8 && (c == 'l' || c == 'q' || c == 'n')) idx = ((const int64_t *)fv.buf)[i];
                            else if (fv.itemsize == 4 && (c == 'i' || c == 'l')) idx = ((const int32_t *)fv.buf)[i];
                            else { ok = 0; break; }
                            if (idx < 0 || idx >= n) { ok = 0; break; }
                            mk[idx] = 0;
                        }
                    }
                    PyBuffer_Release(&fv);
                } else PyErr_Clear();
                if (!ok) unresolved = 1;
            }
            Py_DECREF(fx);
        } else {
            PyErr_Clear();
            PyObject *cs = PyObject_GetAttr(items[s], s_constraints);
            if (cs) { const int t = PyObject_IsTrue(cs); if (t > 0) { any_fixed = 1; unresolved = 1; } if (t < 0) PyErr_Clear(); Py_DECREF(cs); }
            else PyErr_Clear();
        }
    }
    /* element 7: (addresses of atom0, numbers, positions, cells, pbc, number of atoms) -- the caller hands them to the C-ABI as they
       are (an array view and its .ctypes.data per buffer cost more host time than this whole pass) */
    if (plain) result = Py_BuildValue("(OOOOOOO(nnnnnn))", b_atom0, b_num, b_pos, b_cells, b_pbc, any_fixed ? Py_True : Py_False,
                                      (b_mask && !unresolved) ? b_mask : Py_None,
                                      (Py_ssize_t)(size_t)atom0, (Py_ssize_t)(size_t)numbers, (Py_ssize_t)(size_t)pos, (Py_ssize_t)(size_t)cells,
                                      (Py_ssize_t)(size_t)pbc, (Py_ssize_t)N);
done:
    Py_XDECREF(b_atom0); Py_XDECREF(b_cells); Py_XDECREF(b_pbc); Py_XDECREF(b_num); Py_XDECREF(b_pos); Py_XDECREF(b_mask);
    Py_DECREF(seq);
    if (!result && !PyErr_Occurred()) Py_RETURN_NONE;
    return result;
}

static PyMethodDef methods[] = {{"gather", gather, METH_O, "concatenate numbers / positions / cell / pbc of a list of structures"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pygather", NULL, -1, methods, NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__pygather(void) {
    s_numbers = PyUnicode_InternFromString("numbers"); s_positions = PyUnicode_InternFromString("positions");
    s_cell = PyUnicode_InternFromString("cell"); s_pbc = PyUnicode_InternFromString("pbc");
    s_fixed = PyUnicode_InternFromString("fixed"); s_constraints = PyUnicode_InternFromString("constraints");
    return PyModule_Create(&moddef);
}
