/* Host glue, not part of the C ABI: hands the UTF-8 bytes of a Python list of str to libamdretrieval WITHOUT copying
 * them.  amdr_tokenizer_encode_ptrs (include/amdretrieval.h) takes plain `const char*` / length arrays; for a caller that
 * holds Python strings the cheapest way to those is CPython's own cached UTF-8 view of each object
 * (PyUnicode_AsUTF8AndSize: for an ASCII string the pointer INTO the object, no allocation) — "\0".join(qs).encode()
 * touches and copies every string and was 70 % of a 9 344-query tokeniser call.  The list must stay alive while the
 * pointers are in use (the caller holds it across the native call).
 *
 *   utf8_views(seq, ptrs_addr, lens_addr) -> total bytes
 *     seq: list / tuple of str (None counts as ""), ptrs_addr / lens_addr: addresses of int64 arrays of len(seq) entries
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject* utf8_views(PyObject* self, PyObject* args) {
  PyObject* seq;
  unsigned long long pa, la;
  if (!PyArg_ParseTuple(args, "OKK", &seq, &pa, &la)) return NULL;
  PyObject* fast = PySequence_Fast(seq, "utf8_views: a sequence of str is required");
  if (!fast) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  PyObject** items = PySequence_Fast_ITEMS(fast);
  int64_t* ptrs = (int64_t*)(uintptr_t)pa;
  int64_t* lens = (int64_t*)(uintptr_t)la;
  int64_t total = 0;
  static const char kEmpty[1] = {0};
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* o = items[i];
    if (o == Py_None) {
      ptrs[i] = (int64_t)(uintptr_t)kEmpty;
      lens[i] = 0;
      continue;
    }
    if (!PyUnicode_Check(o)) {
      Py_DECREF(fast);
      PyErr_Format(PyExc_TypeError, "utf8_views: item %zd is not a str", i);
      return NULL;
    }
    Py_ssize_t len = 0;
    const char* p = PyUnicode_AsUTF8AndSize(o, &len);
    if (!p) {
      Py_DECREF(fast);
      return NULL;
    }
    ptrs[i] = (int64_t)(uintptr_t)p;
    lens[i] = (int64_t)len;
    total += (int64_t)len;
  }
  Py_DECREF(fast);
  return PyLong_FromLongLong((long long)total);
}

static PyMethodDef kMethods[] = {{"utf8_views", utf8_views, METH_VARARGS, "UTF-8 pointers and sizes of a list of str"},
                                 {NULL, NULL, 0, NULL}};
static struct PyModuleDef kModule = {PyModuleDef_HEAD_INIT, "_amdr_pystrings", NULL, -1, kMethods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__amdr_pystrings(void) { return PyModule_Create(&kModule); }
