/* obsparse.c -- host-side traversal of a list of observation tuples into padded arrays (CPython extension, no numpy C API: the
 * outputs are writable buffers the caller allocated).  What molgym/agents/covariant/agent.py:165-197 does per sample with
 * Python loops and what molgym_amd/observations.py::ParsedObservations.from_list did with three list comprehensions + np.array
 * (1.7 us per sample: the largest host cost of ppo.train's prepare_rollout and of the per-mini-batch parse of the autograd path).
 * An observation is ((label, (x, y, z)) x canvas_size, (bag counts x num_labels)); sequences may be tuples or lists.
 * parse(observations, labels int64[T][N], xyz float64[T][N][3], bags int64[T][Z]) -> None; raises ValueError on any shape
 * mismatch (the caller then takes the numpy path, whose error messages name the offending shape). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static int as_double(PyObject* o, double* out) {
  if (PyFloat_CheckExact(o)) { *out = PyFloat_AS_DOUBLE(o); return 0; }
  double v = PyFloat_AsDouble(o);
  if (v == -1.0 && PyErr_Occurred()) return -1;
  *out = v;
  return 0;
}
static int as_int64(PyObject* o, int64_t* out) {
  long long v = PyLong_AsLongLong(o);
  if (v == -1 && PyErr_Occurred()) {  /* numpy integer scalars etc.: through __index__ */
    PyErr_Clear();
    PyObject* idx = PyNumber_Index(o);
    if (!idx) return -1;
    v = PyLong_AsLongLong(idx);
    Py_DECREF(idx);
    if (v == -1 && PyErr_Occurred()) return -1;
  }
  *out = (int64_t)v;
  return 0;
}

static PyObject* parse(PyObject* self, PyObject* args) {
  PyObject* obs;
  Py_buffer lb, xb, bb;
  if (!PyArg_ParseTuple(args, "Ow*w*w*", &obs, &lb, &xb, &bb)) return NULL;
  PyObject* result = NULL;
  PyObject* seq = PySequence_Fast(obs, "observations must be a sequence");
  if (!seq) goto done;
  {
    const Py_ssize_t T = PySequence_Fast_GET_SIZE(seq);
    if (T == 0 || lb.len % (T * 8) || bb.len % (T * 8)) { PyErr_SetString(PyExc_ValueError, "buffer sizes do not match the observations"); goto done_seq; }
    const Py_ssize_t N = lb.len / (T * 8), Z = bb.len / (T * 8);
    if (xb.len != T * N * 3 * 8) { PyErr_SetString(PyExc_ValueError, "position buffer size"); goto done_seq; }
    int64_t* labels = (int64_t*)lb.buf;
    double* xyz = (double*)xb.buf;
    int64_t* bags = (int64_t*)bb.buf;
    for (Py_ssize_t t = 0; t < T; ++t) {
      PyObject* o = PySequence_Fast(PySequence_Fast_GET_ITEM(seq, t), "observation must be a (canvas, bag) pair");
      if (!o) goto done_seq;
      if (PySequence_Fast_GET_SIZE(o) != 2) { Py_DECREF(o); PyErr_SetString(PyExc_ValueError, "observation must be a (canvas, bag) pair"); goto done_seq; }
      PyObject* canvas = PySequence_Fast(PySequence_Fast_GET_ITEM(o, 0), "canvas must be a sequence");
      PyObject* bag = canvas ? PySequence_Fast(PySequence_Fast_GET_ITEM(o, 1), "bag must be a sequence") : NULL;
      int bad = !canvas || !bag;
      if (!bad && (PySequence_Fast_GET_SIZE(canvas) != N || PySequence_Fast_GET_SIZE(bag) != Z)) {
        PyErr_SetString(PyExc_ValueError, "ragged observations");
        bad = 1;
      }
      for (Py_ssize_t i = 0; !bad && i < N; ++i) {
        PyObject* item = PySequence_Fast(PySequence_Fast_GET_ITEM(canvas, i), "canvas item must be (label, position)");
        if (!item) { bad = 1; break; }
        if (PySequence_Fast_GET_SIZE(item) != 2) { PyErr_SetString(PyExc_ValueError, "canvas item must be (label, position)"); bad = 1; }
        PyObject* p = bad ? NULL : PySequence_Fast(PySequence_Fast_GET_ITEM(item, 1), "position must be a sequence");
        if (!bad && !p) bad = 1;
        if (!bad && PySequence_Fast_GET_SIZE(p) != 3) { PyErr_SetString(PyExc_ValueError, "position must have three components"); bad = 1; }
        if (!bad && as_int64(PySequence_Fast_GET_ITEM(item, 0), labels + t * N + i)) bad = 1;
        for (int k = 0; !bad && k < 3; ++k)
          if (as_double(PySequence_Fast_GET_ITEM(p, k), xyz + (t * N + i) * 3 + k)) bad = 1;
        Py_XDECREF(p);
        Py_DECREF(item);
      }
      for (Py_ssize_t z = 0; !bad && z < Z; ++z)
        if (as_int64(PySequence_Fast_GET_ITEM(bag, z), bags + t * Z + z)) bad = 1;
      Py_XDECREF(bag);
      Py_XDECREF(canvas);
      Py_DECREF(o);
      if (bad) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "malformed observation");
        goto done_seq;
      }
    }
    result = Py_None;
    Py_INCREF(result);
  }
done_seq:
  Py_DECREF(seq);
done:
  PyBuffer_Release(&lb);
  PyBuffer_Release(&xb);
  PyBuffer_Release(&bb);
  return result;
}

static PyMethodDef methods[] = {{"parse", parse, METH_VARARGS, "observations -> (labels, xyz, bags) buffers"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_obsparse", NULL, -1, methods};
PyMODINIT_FUNC PyInit__obsparse(void) { return PyModule_Create(&moddef); }
