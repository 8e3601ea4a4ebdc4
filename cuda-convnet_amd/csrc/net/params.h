// Language-neutral image of the `layers` list-of-dicts that the Python side hands to
// initModel (schema: SURVEY.md section 8b; reference readers: util.cu:74-104 pyDictGet*).
// module.cpp converts PyObject -> Params once, so the layer graph itself never touches
// the CPython API (and can be driven from plain C++).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mcn {

// A host matrix borrowed from the caller (numpy buffer): reference Matrix(PyArrayObject*)
// matrix.cpp:71-90.  `fortran` = column-major storage of the (rows, cols) matrix.
struct HostMat {
    float* data = nullptr;
    long rows = 0, cols = 0;
    bool fortran = false;
    bool pinned = false;  // page-locked host memory (pinnedArray): DMA reads it directly, no staging copy
    size_t size() const { return (size_t)rows * cols; }
};

struct Value;
using Params = std::map<std::string, Value>;

struct Value {
    enum Kind { NONE, INT, FLOAT, STR, LIST, DICT, MAT } kind = NONE;
    long i = 0;
    double f = 0;
    std::string s;
    std::vector<Value> list;
    std::shared_ptr<Params> dict;
    HostMat mat;
};

struct ParamError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline const Value& pget(const Params& p, const std::string& key) {
    auto it = p.find(key);
    if (it == p.end()) throw ParamError("layer dict: missing key '" + key + "'");
    return it->second;
}
inline bool phas(const Params& p, const std::string& key) { return p.count(key) != 0; }
inline long pint(const Params& p, const std::string& key) {
    const Value& v = pget(p, key);
    if (v.kind == Value::INT) return v.i;
    if (v.kind == Value::FLOAT) return (long)v.f;
    throw ParamError("layer dict: key '" + key + "' is not a number");
}
inline long pint(const Params& p, const std::string& key, long dflt) { return phas(p, key) ? pint(p, key) : dflt; }
inline double pfloat(const Params& p, const std::string& key) {
    const Value& v = pget(p, key);
    if (v.kind == Value::INT) return (double)v.i;
    if (v.kind == Value::FLOAT) return v.f;
    throw ParamError("layer dict: key '" + key + "' is not a number");
}
inline const std::string& pstr(const Params& p, const std::string& key) {
    const Value& v = pget(p, key);
    if (v.kind != Value::STR) throw ParamError("layer dict: key '" + key + "' is not a string");
    return v.s;
}
inline std::vector<long> pintv(const Params& p, const std::string& key) {
    const Value& v = pget(p, key);
    std::vector<long> out;
    if (v.kind == Value::INT) { out.push_back(v.i); return out; }
    if (v.kind != Value::LIST) throw ParamError("layer dict: key '" + key + "' is not a list");
    for (const Value& e : v.list) out.push_back(e.kind == Value::FLOAT ? (long)e.f : e.i);
    return out;
}
inline std::vector<double> pfloatv(const Params& p, const std::string& key) {
    const Value& v = pget(p, key);
    std::vector<double> out;
    if (v.kind == Value::INT) { out.push_back((double)v.i); return out; }
    if (v.kind == Value::FLOAT) { out.push_back(v.f); return out; }
    if (v.kind != Value::LIST) throw ParamError("layer dict: key '" + key + "' is not a list");
    for (const Value& e : v.list) out.push_back(e.kind == Value::INT ? (double)e.i : e.f);
    return out;
}
inline HostMat pmat(const Params& p, const std::string& key) {
    const Value& v = pget(p, key);
    if (v.kind != Value::MAT) throw ParamError("layer dict: key '" + key + "' is not a float32 matrix");
    return v.mat;
}
inline std::vector<HostMat> pmatv(const Params& p, const std::string& key) {
    const Value& v = pget(p, key);
    if (v.kind != Value::LIST) throw ParamError("layer dict: key '" + key + "' is not a list of matrices");
    std::vector<HostMat> out;
    for (const Value& e : v.list) {
        if (e.kind != Value::MAT) throw ParamError("layer dict: key '" + key + "' holds a non-matrix");
        out.push_back(e.mat);
    }
    return out;
}

}  // namespace mcn
