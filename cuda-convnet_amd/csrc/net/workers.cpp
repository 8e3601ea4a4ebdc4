// Work items executed on the device-owning thread (reference: worker.cu) and the
// finite-difference gradient check (convnet.cu:280-330).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <random>

#include "mi_convnet.h"
#include "net.h"

namespace mcn {

// name: weight matrix name; eps: finite-difference step.  Unlike the reference, which perturbs
// every element, at most `maxElems` (seeded choice; <= 0 means all) are checked so that the test
// is usable on real layer sizes; norms are taken over the checked subset.
// central: (err(+eps) - err(-eps)) / (2 N eps) instead of the reference's forward difference, which carries an
// O(eps) curvature term that is large at the FC step of 0.1.
bool ConvNet::checkGradient(const std::string& name, float eps, Weights& w, std::vector<GradCheckResult>& out, long maxElems, bool central) {
    std::vector<float> W;
    w.hostImage(W, _stream);
    const long rows = w.rows(), cols = w.cols(), total = rows * cols;
    std::vector<long> idx(total);
    for (long i = 0; i < total; i++) idx[i] = i;
    if (maxElems > 0 && maxElems < total) {
        std::mt19937_64 rng(12345 + total);
        std::shuffle(idx.begin(), idx.end(), rng);
        idx.resize(maxElems);
        std::sort(idx.begin(), idx.end());
    }
    std::vector<double> num(idx.size());
    for (size_t k = 0; k < idx.size(); k++) {
        const long r = idx[k] / cols, c = idx[k] % cols;
        const float v = W[idx[k]];
        w.setElement(r, c, v + eps, _stream);
        zeroCosts();
        fpropCurrent(PASS_GC);
        const double err = costValue();
        if (central) {
            w.setElement(r, c, v - eps, _stream);
            zeroCosts();
            fpropCurrent(PASS_GC);
            num[k] = (err - costValue()) / (2.0 * _numCases * (double)eps);
        } else {
            num[k] = (err - _baseErr) / (_numCases * (double)eps);
        }
        if (std::isnan(num[k]) || std::isinf(num[k]))
            throw NetError("Numerical computation produced nan or inf when checking '" + name +
                           "'. Consider reducing the sizes of the weights or finite difference steps.");
        w.setElement(r, c, v, _stream);
    }
    // analytic gradient, scaled by -1/numCases (convnet.cu:307-309); w.getGrad is inc for FC/local layers
    std::vector<float> g(w.size());
    chk(mcn_d2h(g.data(), w.getGrad(false), w.size() * 4, _stream), "d2h grad");
    sync();
    double an2 = 0, nn2 = 0, dn2 = 0;
    for (size_t k = 0; k < idx.size(); k++) {
        const long r = idx[k] / cols, c = idx[k] % cols;
        const double a = (double)g[w.memIndex(r, c)] * (-1.0 / _numCases);
        an2 += a * a;
        nn2 += num[k] * num[k];
        dn2 += (num[k] - a) * (num[k] - a);
    }
    GradCheckResult res;
    res.name = name;
    res.analNorm = std::sqrt(an2);
    res.numNorm = std::sqrt(nn2);
    res.relErr = std::sqrt(dn2) / res.analNorm;
    res.fail = !(res.relErr < 0.02);  // GC_REL_ERR_THRESH util.cuh:47
    out.push_back(res);
    return res.fail;
}

// TrainingWorker::run (worker.cu:88-102).  useResident: run on the batch a previous runLoad left on the device.
void ConvNet::runTraining(const CPUData& data, bool test, bool useResident) {
    if (!useResident) _dp->setData(data);
    else if (_dp->prefetchPending()) _dp->adoptPrefetched();  // the batch prefetchBatch uploaded while the last one trained
    else if (!_dp->resident()) throw NetError("no resident data batch: call loadBatch or prefetchBatch first");
    WorkResult res;
    zeroCosts();
    if (_dp->streamed()) {
        // batch kept on the host: slices arrive over the copy stream two steps ahead (the reference uploads each slice on the
        // compute thread and waits for it, data.cu:72-98)
        _dp->streamBegin();
        for (int i = 0; i < _dp->numMinibatches(); i++) {
            const long start = (long)i * _minibatchSize, end = std::min(start + _minibatchSize, _dp->numCases());
            _numCases = (int)(end - start);
            joinGradWork();
            _dp->streamSlice(i, _curData);
            fpropCurrent(test ? PASS_TEST : PASS_TRAIN);
            if (!test) {
                bprop(PASS_TRAIN);
                updateWeights();
                _stepsDone++;
            }
            joinGradWork();  // (cost kernels on the side stream read the labels of this slice)
            _dp->streamSliceDone(i);
        }
        readCosts(res.cost, _dp->numCases());
        _dp->clearHost();
        res.type = WorkResult::BATCH_DONE;
        _results.enqueue(std::move(res));
        return;
    }
    for (int i = 0; i < _dp->numMinibatches(); i++) {
        if (test) fprop(i, PASS_TEST);
        else trainStep(i);
    }
    readCosts(res.cost, _dp->numCases());
    _dp->clearHost();
    res.type = WorkResult::BATCH_DONE;
    _results.enqueue(std::move(res));
}

void ConvNet::runSteps(int numSteps, bool test, int profile) {
    if (!_dp->resident()) throw NetError("no resident data batch: call loadBatch first");
    if (numSteps < 0) throw NetError("negative step count");
    WorkResult res;
    zeroCosts();
    _profiling = profile == 1;
    _phases = profile == 2;
    _commTiming = profile == 3 && dataParallel();  // steps as in a normal run, every collective bracketed by events on the exchange stream
    const int nm = _dp->numMinibatches();
    long cases = 0;
    joinGradWork();  // (nothing of an earlier work item on the side thread: its accounting starts at zero)
    const auto t0 = std::chrono::steady_clock::now();
    g_eventCallMs = 0;
    g_blockedMs = 0;
    const uint64_t launchNs0 = mcn_launch_ns();
    const size_t launches0 = mcn_launch_count();
    _sideLaunchNs = _sideLaunches = _sidePollNs = _sideWorkNs = 0;
    try {
        for (int s = 0; s < numSteps; s++) {
            if (!test && (!profile || profile == 3)) {
                trainStep(s % nm);
                cases += _numCases;
                continue;
            }
            if (_phases) profBegin("phase.forward");
            fprop(s % nm, test ? PASS_TEST : PASS_TRAIN);
            if (_phases) profEnd();
            cases += _numCases;
            if (!test) {
                bprop(PASS_TRAIN);
                if (_phases) profBegin("phase.update");
                updateWeights();
                if (_phases) profEnd();
                _stepsDone++;
            }
        }
        const double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (profile == 1 || profile == 2) profCollect(res.profile);
        if (_commTiming) {
            sync();
            _commSub.wait(_commSub.pushed());
            commCollect(res.profile);
        }
        // host-side submission time of the whole loop (launch-bound when it approaches the wall time)
        res.profile["host.enqueue"].count = numSteps;
        res.profile["host.enqueue"].ms = enq;
        res.profile["host.events"].count = numSteps;
        res.profile["host.events"].ms = g_eventCallMs;
        // where the submitting threads' time went (bench.py host_breakdown): the main thread inside the runtime's launch call,
        // blocked until the side thread had issued what it was given, in event / tick calls; the side thread polling for
        // fork ticks and running its closures
        auto put = [&](const char* k, double ms, long count) {
            res.profile[k].count = count;
            res.profile[k].ms = ms;
        };
        const double launchMs = (double)(mcn_launch_ns() - launchNs0) * 1e-6;
        put("host.main.launch_calls", launchMs, (long)(mcn_launch_count() - launches0));
        put("host.main.blocked_on_side_thread", g_blockedMs, numSteps);
        put("host.main.event_and_tick_calls", g_eventCallMs, numSteps);
        put("host.main.graph_walk_and_other", enq - launchMs - g_blockedMs - g_eventCallMs, numSteps);
        put("host.side.launch_calls", _sideLaunchNs.load() * 1e-6, _sideLaunches.load());
        put("host.side.polling_fork_ticks", _sidePollNs.load() * 1e-6, numSteps);
        put("host.side.closures_total", _sideWorkNs.load() * 1e-6, numSteps);
    } catch (...) {
        _profiling = _phases = _commTiming = false;
        throw;
    }
    _profiling = _phases = _commTiming = false;
    readCosts(res.cost, cases);
    res.type = WorkResult::BATCH_DONE;
    _results.enqueue(std::move(res));
}

void ConvNet::runLoad(const CPUData& data) {
    _dp->setData(data);
    if (!_dp->resident()) throw NetError("data batch exceeds MCN_MAX_DATA_ON_GPU_MB; cannot keep it resident");
    _dp->clearHost();
    WorkResult res;
    res.type = WorkResult::SYNC_DONE;
    _results.enqueue(std::move(res));
}

// FeatureWorker::run (worker.cu:191-214): ftrs is (numCases, numFeatures) row-major on the host
void ConvNet::runFeatures(const CPUData& data, HostMat ftrs, int layerIdx) {
    if (layerIdx < 0 || layerIdx >= numLayers()) throw NetError("feature writer: bad layer index");
    _dp->setData(data);
    Layer& fl = layer(layerIdx);
    const long F = fl.outputs();
    if (ftrs.rows != _dp->numCases() || ftrs.cols != F || ftrs.fortran)
        throw NetError("feature writer: output must be a C-ordered (numCases, " + std::to_string(F) + ") float32 matrix");
    DBuf tmp;
    tmp.ensure((size_t)F * _minibatchSize);
    WorkResult res;
    zeroCosts();
    for (int i = 0; i < _dp->numMinibatches(); i++) {
        fprop(i, PASS_TEST);
        chk(mcn_transpose(fl.getActs(), tmp.p, (int)F, _numCases, _stream), "transpose features");
        chk(mcn_d2h(ftrs.data + (size_t)i * _minibatchSize * F, tmp.p, (size_t)_numCases * F * 4, _stream), "d2h features");
        sync();
    }
    readCosts(res.cost, _dp->numCases());
    _dp->clearHost();
    res.type = WorkResult::BATCH_DONE;
    _results.enqueue(std::move(res));
}

// MultiviewTestWorker::run (worker.cu:142-174): average the softmax output over numViews views of each case
void ConvNet::runMultiview(const CPUData& data, int numViews, int logregIdx) {
    if (numViews <= 0 || data.numCases() % numViews != 0) throw NetError("multiview test: numCases not divisible by numViews");
    if (logregIdx < 0 || logregIdx >= numLayers() || layer(logregIdx).type() != "cost.logreg")
        throw NetError("multiview test: logregIdx does not name a cost.logreg layer");
    _dp->setData(data);
    Layer& logreg = layer(logregIdx);
    const long real = _dp->numCases() / numViews;
    const int numMini = (int)((real + _minibatchSize - 1) / _minibatchSize);
    const long K = logreg.prev()[1]->outputs();
    DBuf probs;
    probs.ensure((size_t)K * _minibatchSize);
    WorkResult res;
    zeroCosts();
    for (int i = 0; i < numMini; i++) {
        for (int v = 0; v < numViews; v++) {
            if (v == numViews - 1) zeroCosts();  // the other cost layers report their LAST view, as in the reference
            fpropSlice(v * real + (long)i * _minibatchSize, std::min((v + 1) * real, v * real + (long)(i + 1) * _minibatchSize), PASS_TEST);
            const size_t n = (size_t)K * _numCases;
            if (v == 0) chk(mcn_d2d(probs.p, logreg.prev()[1]->getActs(), n * 4, _stream), "copy probs");
            else chk(mcn_axpby(probs.p, logreg.prev()[1]->getActs(), n, 1.f, 1.f, _stream), "add probs");
        }
        chk(mcn_scale(probs.p, probs.p, (size_t)K * _numCases, 1.0f / numViews, _stream), "scale probs");
        Cost last, avg;
        readCosts(last, 0);
        zeroCosts();
        logreg.fprop({logreg.prev()[0]->getActs(), probs.p}, PASS_TEST);  // cost of the averaged prediction
        readCosts(avg, 0);
        zeroCosts();
        for (const std::string& nm : last.names) {
            const std::vector<double>& src = nm == logreg.name() ? avg.values[nm] : last.values[nm];
            auto& acc = res.cost.values[nm];
            if (acc.empty()) res.cost.names.push_back(nm);
            acc.resize(src.size(), 0.0);
            for (size_t j = 0; j < src.size(); j++) acc[j] += src[j];
            res.cost.coeffs[nm] = last.coeffs[nm];
        }
        res.cost.numCases += _numCases;
    }
    _dp->clearHost();
    res.type = WorkResult::BATCH_DONE;
    _results.enqueue(std::move(res));
}

// GradCheckWorker::run (worker.cu:126-130).  The reference prints and calls exit(0); here the
// per-matrix results go back to the caller, which decides what to do.
void ConvNet::runGradCheck(const CPUData& data, long maxElems, bool central) {
    _dp->setData(data);
    WorkResult res;
    checkGradients(res.gradChecks, maxElems, central);
    _dp->clearHost();
    res.type = WorkResult::GRADCHECK_DONE;
    _results.enqueue(std::move(res));
}

}  // namespace mcn
