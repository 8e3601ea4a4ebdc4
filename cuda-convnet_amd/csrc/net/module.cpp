// CPython 3 extension module `_ConvNet`: the drop-in boundary of the reference
// (src/pyconvnet.cu:47-204, seven methods with the same names, argument meaning and
// return values) plus six additions the reference has no need for:
//   prefetchBatch(data) / pinnedArray    asynchronous upload of the next batch (copy stream, own thread) / page-locked arrays
//   loadBatch(data)                      upload a data batch and keep it resident in HBM (blocking);
//                                        a following startBatch([], test) trains/tests on it
//   commUniqueId() -> bytes              RCCL unique id (rank 0 creates, the launcher distributes it)
//   initDataParallel(rank, world, id)    join the replica group (one process per GPU)
//   destroyModel()                       stop the worker thread and release the device
//   startSteps(n, test=0, profile=0)     n minibatch steps cycling over the resident batch (finishBatch collects)
//   getProfile()                         per-operation device time of the last profiled startSteps
//   getCommLog() / getGradSegments()     the data-parallel exchange as it was issued, and the arena layout (tests)
//   commInfo()                           what the replica group's communicator reports about itself
//   failNextBackwardAt(layerName)        TEST BUILD ONLY (_ConvNetHooks): one injected failure in the next backward pass
// Differences from the reference that a caller can observe: errors raise RuntimeError
// instead of exit(1)/assert; checkGradients returns its per-matrix results instead of
// printing and calling exit(0); the GIL is released while a call blocks.
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <numpy/arrayobject.h>

#include <deque>
#include <map>

#include "mi_convnet.h"
#include "net.h"

using namespace mcn;

static ConvNet* model = nullptr;         // pyconvnet.cu:45: one model per process
static PyObject* modelLayers = nullptr;  // keeps the weight arrays alive (they are written in place by syncWithHost)
static std::deque<std::vector<PyObject*>> inflight;  // arrays lent to queued work items

// page-locked allocations handed out by pinnedArray: [begin, end) address ranges
static std::map<uintptr_t, uintptr_t> pinnedRanges;
static bool isPinned(const void* p, size_t bytes) {
    const uintptr_t a = (uintptr_t)p;
    auto it = pinnedRanges.upper_bound(a);
    if (it == pinnedRanges.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->second;
}
static void pinnedCapsuleFree(PyObject* cap) {
    void* p = PyCapsule_GetPointer(cap, "mcn_pinned");
    if (p) {
        pinnedRanges.erase((uintptr_t)p);
        mcn_host_free(p);
    }
}

static bool toHostMat(PyObject* o, HostMat& m) {
    if (!PyArray_Check(o)) return false;
    PyArrayObject* a = (PyArrayObject*)o;
    if (PyArray_TYPE(a) != NPY_FLOAT32 || PyArray_NDIM(a) != 2) return false;
    const bool c = PyArray_IS_C_CONTIGUOUS(a), f = PyArray_IS_F_CONTIGUOUS(a);
    if (!c && !f) return false;
    m.data = (float*)PyArray_DATA(a);
    m.rows = (long)PyArray_DIM(a, 0);
    m.cols = (long)PyArray_DIM(a, 1);
    m.fortran = !c;  // matrix.cpp:71-90: C-contiguous -> not transposed, F-contiguous -> transposed view
    m.pinned = isPinned(m.data, m.size() * sizeof(float));
    return true;
}

static Value toValue(PyObject* o, int depth = 0) {
    Value v;
    if (o == Py_None || depth > 8) return v;
    if (PyBool_Check(o)) { v.kind = Value::INT; v.i = o == Py_True; return v; }
    if (PyLong_Check(o)) { v.kind = Value::INT; v.i = PyLong_AsLong(o); return v; }
    if (PyFloat_Check(o)) { v.kind = Value::FLOAT; v.f = PyFloat_AsDouble(o); return v; }
    if (PyUnicode_Check(o)) { v.kind = Value::STR; const char* s = PyUnicode_AsUTF8(o); v.s = s ? s : ""; return v; }
    if (PyArray_IsScalar(o, Integer)) { v.kind = Value::INT; PyObject* l = PyNumber_Long(o); v.i = PyLong_AsLong(l); Py_XDECREF(l); return v; }
    if (PyArray_IsScalar(o, Floating)) { v.kind = Value::FLOAT; PyObject* f = PyNumber_Float(o); v.f = PyFloat_AsDouble(f); Py_XDECREF(f); return v; }
    if (PyArray_Check(o)) {
        if (toHostMat(o, v.mat)) v.kind = Value::MAT;
        return v;  // other arrays (e.g. integer filterConns) are not read natively
    }
    if (PyList_Check(o) || PyTuple_Check(o)) {
        v.kind = Value::LIST;
        const Py_ssize_t n = PySequence_Size(o);
        for (Py_ssize_t i = 0; i < n; i++) {
            PyObject* e = PySequence_GetItem(o, i);
            v.list.push_back(toValue(e, depth + 1));
            Py_XDECREF(e);
        }
        return v;
    }
    if (PyDict_Check(o)) {
        v.kind = Value::DICT;
        v.dict = std::make_shared<Params>();
        PyObject *k, *val;
        Py_ssize_t pos = 0;
        while (PyDict_Next(o, &pos, &k, &val)) {
            if (!PyUnicode_Check(k)) continue;
            const char* ks = PyUnicode_AsUTF8(k);
            if (ks) (*v.dict)[ks] = toValue(val, depth + 1);
        }
        return v;
    }
    return v;
}

// data list -> CPUData; arrays that are not float32 C-contiguous are converted (copied) first
static bool toCPUData(PyObject* list, CPUData& out, std::vector<PyObject*>& refs, Py_ssize_t count) {
    for (Py_ssize_t i = 0; i < count; i++) {
        PyObject* a = PyArray_FROM_OTF(PyList_GET_ITEM(list, i), NPY_FLOAT32, NPY_ARRAY_C_CONTIGUOUS | NPY_ARRAY_ALIGNED);
        if (!a) return false;
        refs.push_back(a);
        HostMat m;
        if (PyArray_NDIM((PyArrayObject*)a) != 2 || !toHostMat(a, m)) {
            PyErr_SetString(PyExc_ValueError, "data matrices must be 2-D (dims, numCases) float32 arrays");
            return false;
        }
        out.mats.push_back(m);
    }
    return true;
}

static void releaseRefs(std::vector<PyObject*>& refs) {
    for (PyObject* r : refs) Py_XDECREF(r);
    refs.clear();
}

static bool needModel() {
    if (!model) PyErr_SetString(PyExc_RuntimeError, "_ConvNet: initModel has not been called");
    return model != nullptr;
}

static bool waitResult(WorkResult& r) {
    Py_BEGIN_ALLOW_THREADS
    r = model->resultQueue().dequeue();
    Py_END_ALLOW_THREADS
    if (!inflight.empty()) {
        releaseRefs(inflight.front());
        inflight.pop_front();
    }
    if (r.type == WorkResult::ERROR) {
        PyErr_SetString(PyExc_RuntimeError, ("_ConvNet: " + r.error).c_str());
        return false;
    }
    return true;
}

static PyObject* initModel(PyObject*, PyObject* args) {  // pyconvnet.cu:69-88
    PyObject* layers;
    int minibatch, device;
    if (!PyArg_ParseTuple(args, "O!ii", &PyList_Type, &layers, &minibatch, &device)) return nullptr;
    if (model) { PyErr_SetString(PyExc_RuntimeError, "_ConvNet: a model already exists in this process (destroyModel first)"); return nullptr; }
    try {
        std::vector<Params> ps;
        for (Py_ssize_t i = 0; i < PyList_GET_SIZE(layers); i++) {
            Value v = toValue(PyList_GET_ITEM(layers, i));
            if (v.kind != Value::DICT) throw ParamError("layers[" + std::to_string(i) + "] is not a dict");
            ps.push_back(*v.dict);
        }
        model = new ConvNet(ps, minibatch, device);
    } catch (std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, (std::string("Error creating ConvNet: ") + e.what()).c_str());
        return nullptr;
    }
    Py_INCREF(layers);
    modelLayers = layers;
    model->start();
    WorkResult r;
    inflight.emplace_back();
    if (!waitResult(r)) {
        delete model;
        model = nullptr;
        Py_CLEAR(modelLayers);
        return nullptr;
    }
    return Py_BuildValue("i", 0);
}

static std::vector<PyObject*> prefetchRefs;
static PyObject* destroyModel(PyObject*, PyObject*) {
    if (model) {
        Py_BEGIN_ALLOW_THREADS
        try { model->waitPrefetchUploaded(); } catch (...) {}
        delete model;
        Py_END_ALLOW_THREADS
        model = nullptr;
    }
    releaseRefs(prefetchRefs);
    for (auto& v : inflight) releaseRefs(v);
    inflight.clear();
    Py_CLEAR(modelLayers);
    return Py_BuildValue("i", 0);
}

static PyObject* startBatch(PyObject*, PyObject* args) {  // pyconvnet.cu:93-107 (asynchronous)
    PyObject* data;
    int test = 0;
    if (!PyArg_ParseTuple(args, "O!|i", &PyList_Type, &data, &test)) return nullptr;
    if (!needModel()) return nullptr;
    CPUData cd;
    std::vector<PyObject*> refs;
    const bool resident = PyList_GET_SIZE(data) == 0;
    if (!toCPUData(data, cd, refs, PyList_GET_SIZE(data))) { releaseRefs(refs); return nullptr; }
    inflight.push_back(refs);
    if (resident) model->dp().claimPrefetch();
    model->workQueue().enqueue([cd, test, resident](ConvNet& n) { n.runTraining(cd, test != 0, resident); });
    return Py_BuildValue("i", 0);
}

static PyObject* loadBatch(PyObject*, PyObject* args) {
    PyObject* data;
    if (!PyArg_ParseTuple(args, "O!", &PyList_Type, &data)) return nullptr;
    if (!needModel()) return nullptr;
    CPUData cd;
    std::vector<PyObject*> refs;
    if (!toCPUData(data, cd, refs, PyList_GET_SIZE(data))) { releaseRefs(refs); return nullptr; }
    inflight.push_back(refs);
    model->workQueue().enqueue([cd](ConvNet& n) { n.runLoad(cd); });
    WorkResult r;
    if (!waitResult(r)) return nullptr;
    return Py_BuildValue("i", 0);
}

// pinnedArray(rows, cols) -> C-contiguous float32 array in page-locked host memory (freed with the array).  Data providers
// that fill such arrays skip the staging copy: the batch goes to the device by DMA straight from them.
static PyObject* pinnedArray(PyObject*, PyObject* args) {
    long rows, cols;
    if (!PyArg_ParseTuple(args, "ll", &rows, &cols)) return nullptr;
    if (rows <= 0 || cols <= 0) { PyErr_SetString(PyExc_ValueError, "pinnedArray: positive dimensions expected"); return nullptr; }
    void* p = nullptr;
    const size_t bytes = (size_t)rows * cols * sizeof(float);
    if (mcn_host_alloc(&p, bytes) != MCN_OK) { PyErr_SetString(PyExc_RuntimeError, mcn_last_error()); return nullptr; }
    // the capsule owns the allocation from here on: every later failure only drops references
    PyObject* cap = PyCapsule_New(p, "mcn_pinned", pinnedCapsuleFree);
    if (!cap) {
        mcn_host_free(p);
        return nullptr;
    }
    npy_intp dims[2] = {rows, cols};
    PyObject* arr = PyArray_SimpleNewFromData(2, dims, NPY_FLOAT32, p);
    if (!arr) {
        Py_DECREF(cap);
        return nullptr;
    }
    if (PyArray_SetBaseObject((PyArrayObject*)arr, cap) != 0) {  // (steals the capsule's reference, also when it fails)
        Py_DECREF(arr);
        return nullptr;
    }
    pinnedRanges[(uintptr_t)p] = (uintptr_t)p + bytes;
    return arr;
}

// prefetchBatch(data): asynchronous upload of the NEXT batch into the spare device slot (own thread, own stream) while the
// current one trains; the following startBatch([], test) trains / tests on it.  The arrays are kept alive until the transfer
// has completed (checked at the next prefetchBatch / destroyModel).
static PyObject* prefetchBatch(PyObject*, PyObject* args) {
    PyObject* data;
    if (!PyArg_ParseTuple(args, "O!", &PyList_Type, &data)) return nullptr;
    if (!needModel()) return nullptr;
    CPUData cd;
    std::vector<PyObject*> refs;
    if (!toCPUData(data, cd, refs, PyList_GET_SIZE(data))) { releaseRefs(refs); return nullptr; }
    std::string err;
    Py_BEGIN_ALLOW_THREADS
    try {
        model->waitPrefetchUploaded();  // the previous prefetch no longer reads its host arrays
        model->prefetchFromHost(cd);
    } catch (std::exception& e) {
        err = e.what();
    }
    Py_END_ALLOW_THREADS
    releaseRefs(prefetchRefs);
    if (!err.empty()) {
        releaseRefs(refs);
        PyErr_SetString(PyExc_RuntimeError, ("_ConvNet: " + err).c_str());
        return nullptr;
    }
    prefetchRefs = refs;
    return Py_BuildValue("i", 0);
}

static std::map<std::string, ProfEntry> lastProfile;

static PyObject* startSteps(PyObject*, PyObject* args) {
    int numSteps, test = 0, profile = 0;
    if (!PyArg_ParseTuple(args, "i|ii", &numSteps, &test, &profile)) return nullptr;
    if (!needModel()) return nullptr;
    inflight.emplace_back();
    model->workQueue().enqueue([numSteps, test, profile](ConvNet& n) { n.runSteps(numSteps, test != 0, profile); });
    return Py_BuildValue("i", 0);
}

static PyObject* getProfile(PyObject*, PyObject*) {
    PyObject* dict = PyDict_New();
    for (auto& kv : lastProfile) {
        PyObject* t = Py_BuildValue("(ld)", kv.second.count, kv.second.ms);
        PyDict_SetItemString(dict, kv.first.c_str(), t);
        Py_DECREF(t);
    }
    return dict;
}

static PyObject* startMultiviewTest(PyObject*, PyObject* args) {  // pyconvnet.cu:112-127
    PyObject* data;
    int numViews, logregIdx;
    if (!PyArg_ParseTuple(args, "O!ii", &PyList_Type, &data, &numViews, &logregIdx)) return nullptr;
    if (!needModel()) return nullptr;
    CPUData cd;
    std::vector<PyObject*> refs;
    if (!toCPUData(data, cd, refs, PyList_GET_SIZE(data))) { releaseRefs(refs); return nullptr; }
    inflight.push_back(refs);
    model->workQueue().enqueue([cd, numViews, logregIdx](ConvNet& n) { n.runMultiview(cd, numViews, logregIdx); });
    return Py_BuildValue("i", 0);
}

static PyObject* startFeatureWriter(PyObject*, PyObject* args) {  // pyconvnet.cu:129-145: last list element is the output
    PyObject* data;
    int layerIdx;
    if (!PyArg_ParseTuple(args, "O!i", &PyList_Type, &data, &layerIdx)) return nullptr;
    if (!needModel()) return nullptr;
    const Py_ssize_t n = PyList_GET_SIZE(data);
    if (n < 2) { PyErr_SetString(PyExc_ValueError, "startFeatureWriter: data must end with the output matrix"); return nullptr; }
    PyObject* fo = PyList_GET_ITEM(data, n - 1);
    HostMat ftrs;
    if (!toHostMat(fo, ftrs) || ftrs.fortran || !PyArray_ISWRITEABLE((PyArrayObject*)fo)) {
        PyErr_SetString(PyExc_ValueError, "startFeatureWriter: the output must be a writeable C-contiguous float32 (numCases, numFeatures) array");
        return nullptr;
    }
    CPUData cd;
    std::vector<PyObject*> refs;
    if (!toCPUData(data, cd, refs, n - 1)) { releaseRefs(refs); return nullptr; }
    Py_INCREF(fo);
    refs.push_back(fo);
    inflight.push_back(refs);
    model->workQueue().enqueue([cd, ftrs, layerIdx](ConvNet& net) { net.runFeatures(cd, ftrs, layerIdx); });
    return Py_BuildValue("i", 0);
}

static PyObject* finishBatch(PyObject*, PyObject*) {  // pyconvnet.cu:150-171
    if (!needModel()) return nullptr;
    WorkResult r;
    if (!waitResult(r)) return nullptr;
    if (r.type != WorkResult::BATCH_DONE) { PyErr_SetString(PyExc_RuntimeError, "_ConvNet: unexpected result type"); return nullptr; }
    lastProfile = r.profile;
    PyObject* dict = PyDict_New();
    for (const std::string& nm : r.cost.names) {
        const std::vector<double>& vals = r.cost.values[nm];
        PyObject* lst = PyList_New(0);
        for (double d : vals) {
            PyObject* f = PyFloat_FromDouble(d);
            PyList_Append(lst, f);
            Py_DECREF(f);
        }
        PyDict_SetItemString(dict, nm.c_str(), lst);
        Py_DECREF(lst);
    }
    return Py_BuildValue("Nl", dict, r.cost.numCases);
}

static PyObject* checkGradients(PyObject*, PyObject* args) {  // pyconvnet.cu:173-189
    PyObject* data;
    long maxElems = 0;
    int central = 0;
    if (!PyArg_ParseTuple(args, "O!|li", &PyList_Type, &data, &maxElems, &central)) return nullptr;
    if (!needModel()) return nullptr;
    CPUData cd;
    std::vector<PyObject*> refs;
    if (!toCPUData(data, cd, refs, PyList_GET_SIZE(data))) { releaseRefs(refs); return nullptr; }
    inflight.push_back(refs);
    model->workQueue().enqueue([cd, maxElems, central](ConvNet& n) { n.runGradCheck(cd, maxElems, central != 0); });
    WorkResult r;
    if (!waitResult(r)) return nullptr;
    PyObject* out = PyList_New(0);
    for (const GradCheckResult& g : r.gradChecks) {
        PyObject* t = Py_BuildValue("(sdddi)", g.name.c_str(), g.analNorm, g.numNorm, g.relErr, g.fail ? 1 : 0);
        PyList_Append(out, t);
        Py_DECREF(t);
    }
    return out;
}

static PyObject* syncWithHost(PyObject*, PyObject*) {  // pyconvnet.cu:194-204
    if (!needModel()) return nullptr;
    inflight.emplace_back();
    model->workQueue().enqueue([](ConvNet& n) {
        n.copyToCPU();
        WorkResult r;
        r.type = WorkResult::SYNC_DONE;
        n.resultQueue().enqueue(std::move(r));
    });
    WorkResult r;
    if (!waitResult(r)) return nullptr;
    return Py_BuildValue("i", 0);
}

static PyObject* commUniqueId(PyObject*, PyObject*) {
    char id[128];
    if (mcn_comm_unique_id(id) != MCN_OK) { PyErr_SetString(PyExc_RuntimeError, mcn_last_error()); return nullptr; }
    return PyBytes_FromStringAndSize(id, 128);
}

static PyObject* initDataParallel(PyObject*, PyObject* args) {
    int rank, world, force = 0;
    const char* id;
    Py_ssize_t idLen;
    if (!PyArg_ParseTuple(args, "iiy#|i", &rank, &world, &id, &idLen, &force)) return nullptr;
    if (!needModel()) return nullptr;
    if (idLen != 128) { PyErr_SetString(PyExc_ValueError, "initDataParallel: the unique id must be 128 bytes"); return nullptr; }
    std::string ids(id, 128);
    inflight.emplace_back();
    model->workQueue().enqueue([rank, world, ids, force](ConvNet& n) {
        n.initComm(rank, world, ids.data(), force != 0);
        WorkResult r;
        r.type = WorkResult::SYNC_DONE;
        n.resultQueue().enqueue(std::move(r));
    });
    WorkResult r;
    if (!waitResult(r)) return nullptr;
    return Py_BuildValue("i", 0);
}

// the data-parallel plan as it ran: [(kind, arena offset, count)] since the last call, and the arena layout it refers to
static PyObject* getCommLog(PyObject*, PyObject*) {
    if (!needModel()) return nullptr;
    PyObject* out = PyList_New(0);
    for (const ConvNet::CommRec& r : model->takeCommLog()) {
        PyObject* t = Py_BuildValue("(snn)", r.kind == 0 ? "allreduce" : "allgather", (Py_ssize_t)r.offset, (Py_ssize_t)r.count);
        PyList_Append(out, t);
        Py_DECREF(t);
    }
    return out;
}

static PyObject* getGradSegments(PyObject*, PyObject*) {
    if (!needModel()) return nullptr;
    PyObject* out = PyList_New(0);
    for (const ConvNet::SegmentInfo& s : model->gradSegments()) {
        PyObject* t = Py_BuildValue("(snn)", s.layer.c_str(), (Py_ssize_t)s.begin, (Py_ssize_t)s.end);
        PyList_Append(out, t);
        Py_DECREF(t);
    }
    return Py_BuildValue("Nn", out, (Py_ssize_t)model->bucketFloats());
}

#ifdef MCN_TEST_HOOKS
// TEST BUILD ONLY (_ConvNetHooks): failNextBackwardAt(layerName): the next backward pass throws when it reaches that weight
// layer -- after the layers above it have queued their work on both streams (tests of the recovery after a failed step)
static PyObject* failNextBackwardAt(PyObject*, PyObject* args) {
    const char* name = nullptr;
    if (!PyArg_ParseTuple(args, "s", &name)) return nullptr;
    if (!needModel()) return nullptr;
    model->failNextBackwardAt(name);
    Py_RETURN_NONE;
}
#endif

static PyObject* commInfo(PyObject*, PyObject*) {
    if (!needModel()) return nullptr;
    int n = 0, v = 0, r = 0;
    try {
        model->commInfo(n, v, r);
    } catch (std::exception& e) {
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
    return Py_BuildValue("(iii)", n, v, r);
}

static PyMethodDef methods[] = {
    {"initModel", initModel, METH_VARARGS, "initModel(layers, minibatchSize, deviceID)"},
    {"startBatch", startBatch, METH_VARARGS, "startBatch(data, test=0): asynchronous; data == [] uses the batch left by loadBatch"},
    {"finishBatch", finishBatch, METH_VARARGS, "finishBatch() -> ({cost name: [values]}, numCases)"},
    {"checkGradients", checkGradients, METH_VARARGS, "checkGradients(data, maxElems=0, central=0) -> [(name, analNorm, numNorm, relErr, failed)]"},
    {"startMultiviewTest", startMultiviewTest, METH_VARARGS, "startMultiviewTest(data, numViews, logregIdx)"},
    {"startFeatureWriter", startFeatureWriter, METH_VARARGS, "startFeatureWriter(data + [ftrs], layerIdx)"},
    {"syncWithHost", syncWithHost, METH_VARARGS, "syncWithHost(): fills the weight arrays of `layers` in place"},
    {"loadBatch", loadBatch, METH_VARARGS, "loadBatch(data): upload and keep resident (blocking)"},
    {"prefetchBatch", prefetchBatch, METH_VARARGS, "prefetchBatch(data): upload the NEXT batch on the copy stream while the current one trains; startBatch([], test) then uses it"},
    {"pinnedArray", pinnedArray, METH_VARARGS, "pinnedArray(rows, cols) -> float32 C array in page-locked host memory"},
    {"startSteps", startSteps, METH_VARARGS, "startSteps(numSteps, test=0, profile=0): minibatch steps cycling over the resident batch (asynchronous)"},
    {"getProfile", getProfile, METH_NOARGS, "getProfile() -> {operation: (count, total device ms)} of the last profiled startSteps"},
    {"commUniqueId", commUniqueId, METH_NOARGS, "commUniqueId() -> 128 bytes"},
    {"initDataParallel", initDataParallel, METH_VARARGS, "initDataParallel(rank, worldSize, uniqueId, force=0)"},
    {"getCommLog", getCommLog, METH_NOARGS, "getCommLog() -> [(kind, arenaOffset, count)] issued since the last call (data parallel)"},
    {"getGradSegments", getGradSegments, METH_NOARGS, "getGradSegments() -> ([(layer, begin, end)] in arena order, bucketFloats)"},
#ifdef MCN_TEST_HOOKS
    {"failNextBackwardAt", failNextBackwardAt, METH_VARARGS, "failNextBackwardAt(layerName): inject one failure into the next backward pass (tests)"},
#endif
    {"commInfo", commInfo, METH_NOARGS, "commInfo() -> (ranks the communicator counts, RCCL version code, 1 if RCCL)"},
    {"destroyModel", destroyModel, METH_NOARGS, "destroyModel()"},
    {nullptr, nullptr, 0, nullptr}};

#ifdef MCN_TEST_HOOKS  // the test build is a module of its own: a process cannot mistake one for the other
static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_ConvNetHooks", "MI355X-native cuda-convnet compute core (test build with hooks)", -1, methods};
PyMODINIT_FUNC PyInit__ConvNetHooks(void) {
#else
static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_ConvNet", "MI355X-native cuda-convnet compute core", -1, methods};
PyMODINIT_FUNC PyInit__ConvNet(void) {
#endif
    import_array();
    return PyModule_Create(&moduledef);
}
