/* _fastmat -- the per-region loop of region_batch._materialize_groups in C (round 6).
 *
 * Solution.materialize() touches every field of every region of a level: six array views, six lists and one dictionary per region.  In
 * Python that loop is ~1.5 us per region (a zip over fifteen iterables, ten slice / view constructions); here the same objects are built
 * through the C API in ~0.9 us.  Host-side convenience only: no arithmetic, nothing of the solve runs here; region_batch.py keeps the
 * Python loop and uses it when this module has not been built (same objects either way: tests/test_host_logic.py compares them).
 *
 *   fill(regs, hd, hi, er, js, (n_x, n_t, k, oA, ob, oC, od, iact, iom, ila, iri, irc)) -> None
 *     regs  list of BatchCriticalRegion (any object with an instance dictionary), one per entry of js
 *     hd    float64 (n_slots, fd), hi int32 (n_slots, fi): the level's head arrays (rows may be strided, elements contiguous)
 *     er    float64 (rows, n_t + 1): the pool of region rows [f | E]
 *     js    int64 (len(regs),): slot row of each region
 *   Fields set: A, b, C, d (views of hd), E, f (views of er), active_set, omega_set, lambda_set, regular_set -- the index conventions of
 *   ppopt_amd/region_batch.py (BatchCriticalRegion); the fields are those of the reference's dataclass (critical_region.py:10-60).
 *   A field that is already in the region's dictionary is left as it is.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

static PyObject *s_A, *s_b, *s_C, *s_d, *s_E, *s_f, *s_act, *s_om, *s_la, *s_re;

static PyObject *view2(PyArrayObject *base, char *data, npy_intp d0, npy_intp d1, npy_intp s0, npy_intp s1, int flags) {
    npy_intp dims[2] = {d0, d1}, strides[2] = {s0, s1};
    PyArray_Descr *descr = PyArray_DescrFromType(NPY_DOUBLE);      /* new reference, stolen below */
    PyObject *v = PyArray_NewFromDescr(&PyArray_Type, descr, 2, dims, strides, data, flags, NULL);
    if (!v) return NULL;
    Py_INCREF(base);
    if (PyArray_SetBaseObject((PyArrayObject *)v, (PyObject *)base) < 0) { Py_DECREF(v); return NULL; }
    return v;
}

static PyObject *int_list(const npy_int32 *p, npy_intp n) {
    PyObject *l = PyList_New(n);
    if (!l) return NULL;
    for (npy_intp i = 0; i < n; ++i) {
        PyObject *v = PyLong_FromLong((long)p[i]);
        if (!v) { Py_DECREF(l); return NULL; }
        PyList_SET_ITEM(l, i, v);
    }
    return l;
}

static int put(PyObject *dict, PyObject *existing, PyObject *key, PyObject *val) {      /* steals val */
    int rc = 0;
    if (!val) return -1;
    if (!(existing && PyDict_Contains(existing, key) == 1)) rc = PyDict_SetItem(dict, key, val);
    Py_DECREF(val);
    return rc;
}

static PyObject *fill(PyObject *self, PyObject *args) {
    PyObject *regs, *o_hd, *o_hi, *o_er, *o_js;
    long n_x, n_t, k, oA, ob, oC, od, iact, iom, ila, iri, irc;
    if (!PyArg_ParseTuple(args, "O!OOOO(llllllllllll)", &PyList_Type, &regs, &o_hd, &o_hi, &o_er, &o_js,
                          &n_x, &n_t, &k, &oA, &ob, &oC, &od, &iact, &iom, &ila, &iri, &irc)) return NULL;
    if (!PyArray_Check(o_hd) || !PyArray_Check(o_hi) || !PyArray_Check(o_er) || !PyArray_Check(o_js)) { PyErr_SetString(PyExc_TypeError, "fill: arrays expected"); return NULL; }
    PyArrayObject *hd = (PyArrayObject *)o_hd, *hi = (PyArrayObject *)o_hi, *er = (PyArrayObject *)o_er, *js = (PyArrayObject *)o_js;
    if (PyArray_TYPE(hd) != NPY_DOUBLE || PyArray_NDIM(hd) != 2 || PyArray_STRIDE(hd, 1) != 8 ||
        PyArray_TYPE(hi) != NPY_INT32 || PyArray_NDIM(hi) != 2 || PyArray_STRIDE(hi, 1) != 4 ||
        PyArray_TYPE(er) != NPY_DOUBLE || PyArray_NDIM(er) != 2 || (PyArray_DIM(er, 0) > 0 && PyArray_STRIDE(er, 1) != 8) || PyArray_DIM(er, 1) != n_t + 1 ||
        PyArray_TYPE(js) != NPY_INT64 || PyArray_NDIM(js) != 1 || (PyArray_DIM(js, 0) > 1 && PyArray_STRIDE(js, 0) != 8)) {
        PyErr_SetString(PyExc_ValueError, "fill: unexpected array layout");
        return NULL;
    }
    const npy_intp n = PyList_GET_SIZE(regs), n_slots = PyArray_DIM(hd, 0), fd = PyArray_DIM(hd, 1), fi = PyArray_DIM(hi, 1), n_rows = PyArray_DIM(er, 0);
    if (PyArray_DIM(js, 0) != n || PyArray_DIM(hi, 0) != n_slots || od + k > fd || ob + n_x > fd || irc > fi || iact + k > fi) { PyErr_SetString(PyExc_ValueError, "fill: sizes do not fit"); return NULL; }
    const int fl_hd = (PyArray_FLAGS(hd) & NPY_ARRAY_WRITEABLE) | NPY_ARRAY_ALIGNED, fl_er = (PyArray_FLAGS(er) & NPY_ARRAY_WRITEABLE) | NPY_ARRAY_ALIGNED;
    const npy_intp s_hd = PyArray_STRIDE(hd, 0), s_hi = PyArray_STRIDE(hi, 0), s_er = n_rows > 0 ? PyArray_STRIDE(er, 0) : (n_t + 1) * 8;
    const npy_int64 *jp = (const npy_int64 *)PyArray_DATA(js);
    const npy_intp w_re = (fi - irc);      /* columns of each of the two regular-set blocks */
    for (npy_intp r = 0; r < n; ++r) {
        const npy_int64 j = jp[r];
        if (j < 0 || j >= n_slots) { PyErr_SetString(PyExc_IndexError, "fill: slot out of range"); return NULL; }
        PyObject *reg = PyList_GET_ITEM(regs, r);
        PyObject **dp = _PyObject_GetDictPtr(reg);
        if (!dp) { PyErr_SetString(PyExc_TypeError, "fill: region without an instance dictionary"); return NULL; }
        PyObject *existing = (*dp && PyDict_GET_SIZE(*dp) > 0) ? *dp : NULL;
        PyObject *fields = existing ? existing : _PyDict_NewPresized(10);      /* (ten fields: no resize on the way) */
        if (!fields) return NULL;
        char *row_d = (char *)PyArray_DATA(hd) + j * s_hd;
        const npy_int32 *h = (const npy_int32 *)((char *)PyArray_DATA(hi) + j * s_hi);
        const npy_intp m = h[2], n_om = h[3], n_la = h[4], n_re = h[5], off = h[6];
        int bad = 0;
        if (m < 0 || off < 0 || off + m > n_rows || n_om < 0 || iom + n_om > fi || n_la < 0 || ila + n_la > fi || n_re < 0 || n_re > w_re || iri + n_re > fi) {
            PyErr_SetString(PyExc_ValueError, "fill: a region header does not fit its arrays");
            bad = 1;
        }
        char *row_e = (char *)PyArray_DATA(er) + off * s_er;
        if (!bad) bad = put(fields, existing, s_A, view2(hd, row_d + oA * 8, n_x, n_t, n_t * 8, 8, fl_hd)) < 0;
        if (!bad) bad = put(fields, existing, s_b, view2(hd, row_d + ob * 8, n_x, 1, 8, 8, fl_hd)) < 0;
        if (!bad) bad = put(fields, existing, s_C, view2(hd, row_d + oC * 8, k, n_t, n_t * 8, 8, fl_hd)) < 0;
        if (!bad) bad = put(fields, existing, s_d, view2(hd, row_d + od * 8, k, 1, 8, 8, fl_hd)) < 0;
        if (!bad) bad = put(fields, existing, s_E, view2(er, row_e + 8, m, n_t, s_er, 8, fl_er)) < 0;
        if (!bad) bad = put(fields, existing, s_f, view2(er, row_e, m, 1, s_er, 8, fl_er)) < 0;
        if (!bad) bad = put(fields, existing, s_act, int_list(h + iact, k)) < 0;
        if (!bad) bad = put(fields, existing, s_om, int_list(h + iom, n_om)) < 0;
        if (!bad) bad = put(fields, existing, s_la, int_list(h + ila, n_la)) < 0;
        if (!bad) {
            PyObject *pair = PyList_New(2), *l0 = int_list(h + iri, n_re), *l1 = int_list(h + irc, n_re);
            if (!pair || !l0 || !l1) { Py_XDECREF(pair); Py_XDECREF(l0); Py_XDECREF(l1); bad = 1; }
            else { PyList_SET_ITEM(pair, 0, l0); PyList_SET_ITEM(pair, 1, l1); bad = put(fields, existing, s_re, pair) < 0; }
        }
        if (bad) { if (!existing) Py_DECREF(fields); return NULL; }
        if (!existing) {
            PyObject *old = *dp;
            *dp = fields;      /* (the region's dictionary was absent or empty) */
            Py_XDECREF(old);
        }
    }
    Py_RETURN_NONE;
}

/* make(cls, batch, slots) -> [cls instances]: RegionBatch.regions_of without a Python-level __init__ per region.  cls has the two __slots__
 * members `_batch` and `_j` (BatchCriticalRegion); every instance gets (batch, slots[i]).  ~40 ns per region instead of ~100. */
#include <structmember.h>
static PyObject *make(PyObject *self, PyObject *args) {
    PyObject *cls, *batch, *slots;
    if (!PyArg_ParseTuple(args, "OOO!", &cls, &batch, &PyList_Type, &slots)) return NULL;
    if (!PyType_Check(cls)) { PyErr_SetString(PyExc_TypeError, "make: a class expected"); return NULL; }
    Py_ssize_t off[2];
    const char *names[2] = {"_batch", "_j"};
    for (int i = 0; i < 2; ++i) {
        PyObject *desc = PyObject_GetAttrString(cls, names[i]);
        if (!desc) return NULL;
        if (Py_TYPE(desc) != &PyMemberDescr_Type || ((PyMemberDescrObject *)desc)->d_member->type != T_OBJECT_EX) {
            Py_DECREF(desc);
            PyErr_SetString(PyExc_TypeError, "make: the class has no such slot member");
            return NULL;
        }
        off[i] = ((PyMemberDescrObject *)desc)->d_member->offset;
        Py_DECREF(desc);
    }
    PyTypeObject *tp = (PyTypeObject *)cls;
    const Py_ssize_t n = PyList_GET_SIZE(slots);
    PyObject *out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *obj = tp->tp_alloc(tp, 0);
        if (!obj) { Py_DECREF(out); return NULL; }
        PyObject *j = PyList_GET_ITEM(slots, i);
        Py_INCREF(batch); Py_INCREF(j);
        *(PyObject **)((char *)obj + off[0]) = batch;
        *(PyObject **)((char *)obj + off[1]) = j;
        PyList_SET_ITEM(out, i, obj);
    }
    return out;
}

static PyMethodDef methods[] = {{"make", make, METH_VARARGS, "make(cls, batch, slots): one instance of cls per slot, its members _batch and _j set"},
                                {"fill", fill, METH_VARARGS, "fill(regs, hd, hi, er, js, layout): the fields of every region of a level"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastmat", "per-region loop of Solution.materialize (host side)", -1, methods};

PyMODINIT_FUNC PyInit__fastmat(void) {
    import_array();
    s_A = PyUnicode_InternFromString("A"); s_b = PyUnicode_InternFromString("b"); s_C = PyUnicode_InternFromString("C"); s_d = PyUnicode_InternFromString("d");
    s_E = PyUnicode_InternFromString("E"); s_f = PyUnicode_InternFromString("f"); s_act = PyUnicode_InternFromString("active_set");
    s_om = PyUnicode_InternFromString("omega_set"); s_la = PyUnicode_InternFromString("lambda_set"); s_re = PyUnicode_InternFromString("regular_set");
    return PyModule_Create(&moddef);
}
