// See net.h.  Reference citations are file:line into /root/reference/src unless noted.
#include "net.h"
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "mi_convnet.h"

namespace mcn {

void chk(int rc, const char* what) {
    if (rc != MCN_OK) throw NetError(std::string(what) + ": " + mcn_last_error());
}

// host time spent in cross-stream synchronisation calls (reported as host.events by runSteps)
thread_local double g_eventCallMs = 0;
thread_local double g_blockedMs = 0;
// MCN_DEBUG_WAITS=1: host time per named wait site (printed when the model is destroyed): where a submitting thread stands still
static const bool g_debugWaits = getenv("MCN_DEBUG_WAITS") != nullptr;
static std::mutex g_waitSiteMutex;
static std::map<std::string, std::pair<double, long>> g_waitSites;
struct SiteTimer {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit SiteTimer(const char* n) : name(g_debugWaits ? n : nullptr) { if (name) t0 = std::chrono::steady_clock::now(); }
    ~SiteTimer() {
        if (!name) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(g_waitSiteMutex);
        auto& e = g_waitSites[name];
        e.first += ms;
        e.second++;
    }
};
static void printWaitSites() {
    if (!g_debugWaits) return;
    std::lock_guard<std::mutex> lk(g_waitSiteMutex);
    for (auto& kv : g_waitSites) fprintf(stderr, "[mcn] wait site %-40s %10.3f ms in %ld calls\n", kv.first.c_str(), kv.second.first, kv.second.second);
    g_waitSites.clear();
}
int timedEventRecord(void* e, void* s) {
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = mcn_event_record(e, s);
    g_eventCallMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
static thread_local int t_stream = 0;  // which stream this submitting thread feeds: 0 main, 1 side, 2 exchange, 3 copy
// Which submitting threads poll an event themselves (hipEventQuery) instead of queueing a wait on their stream: bit 0 the
// main thread, bit 1 the side thread.  A wait queued on a stream for an event that has not fired yet costs the waiting
// queue ~10 us after it fires (measured on MI355X / ROCm 7.2: DeepID2 step 0.315 ms with queued waits, 0.280 ms when the
// side thread polls -- it has nothing else to do, and launches the moment the main stream's kernel has finished).
// Default: the side thread polls.  MCN_HOST_WAIT overrides (0: every wait queued).
static const int g_hostWaitMask = getenv("MCN_HOST_WAIT") ? atoi(getenv("MCN_HOST_WAIT")) : 2;
static std::atomic<bool> g_noHostWait{false};  // (stream capture in progress: events cannot be queried; read by every submitting thread)
int timedStreamWait(void* s, void* e) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc = MCN_OK;
    if (!g_noHostWait && t_stream <= 2 && ((g_hostWaitMask >> t_stream) & 1)) {
        int q;
        unsigned spin = 0;
        while ((q = mcn_event_query(e)) == 0) {  // (a stream that hangs without faulting: give up like awaitTick does)
            if ((++spin & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                throw NetError("an event a submitting thread polls for was not reached within 60 s");
        }
        if (q < 0) rc = q;
    } else {
        rc = mcn_stream_wait_event(s, e);
    }
    g_eventCallMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

DBuf::~DBuf() {
    if (p) mcn_free(p);
}

float* DBuf::ensure(size_t n) {
    if (n > cap) {
        if (p) chk(mcn_free(p), "free");
        p = nullptr;
        void* q = nullptr;
        chk(mcn_alloc(&q, n * sizeof(float)), "alloc");
        p = (float*)q;
        cap = n;
    }
    return p;
}

static int neuronTypeId(const std::string& t) {
    static const char* names[] = {"ident", "relu", "logistic", "brelu", "abs", "tanh", "softrelu", "square", "sqrt", "linear"};
    for (int i = 0; i < 10; i++)
        if (t == names[i]) return i;
    throw NetError("unknown neuron type '" + t + "'");  // neuron.cu:32-75
}

bool NeuronSpec::needsInputs() const { return type == MCN_NEURON_ABS || type == MCN_NEURON_SQUARE; }
bool NeuronSpec::fusableGrad() const { return !needsInputs() && type != MCN_NEURON_SOFTRELU; }

// ============================================================================ Weights
Weights::Weights(const HostMat& hW, const HostMat& hInc, float eps, float wc, float mom, bool useGrad, bool devFortran)
    : _hW(hW), _hInc(hInc), _eps(eps), _wc(wc), _mom(mom), _useGrad(useGrad), _devFortran(devFortran) {
    if (hW.rows != hInc.rows || hW.cols != hInc.cols) throw NetError("weights and weightsInc differ in shape");
}

Weights::Weights(Weights& src, float eps)
    : _hW(src._hW), _hInc(src._hInc), _eps(eps), _wc(0), _mom(src._mom), _useGrad(src._useGrad),
      _devFortran(src._devFortran), _src(src._src ? src._src : &src) {}

void Weights::upload(const HostMat& h, float* dst, void* stream) {
    const bool vec = h.rows == 1 || h.cols == 1;
    if (vec || h.fortran == _devFortran) {
        chk(mcn_h2d(dst, h.data, h.size() * 4, stream), "h2d weights");
        chk(mcn_stream_sync(stream), "sync");
        return;
    }
    std::vector<float> tmp(h.size());
    for (long r = 0; r < h.rows; r++)
        for (long c = 0; c < h.cols; c++)
            tmp[memIndex(r, c)] = h.data[h.fortran ? c * h.rows + r : r * h.cols + c];
    chk(mcn_h2d(dst, tmp.data(), tmp.size() * 4, stream), "h2d weights");
    chk(mcn_stream_sync(stream), "sync");
}

void Weights::download(const HostMat& h, const float* src, void* stream) {
    const bool vec = h.rows == 1 || h.cols == 1;
    if (vec || h.fortran == _devFortran) {
        chk(mcn_d2h(h.data, src, h.size() * 4, stream), "d2h weights");
        chk(mcn_stream_sync(stream), "sync");
        return;
    }
    std::vector<float> tmp(h.size());
    chk(mcn_d2h(tmp.data(), src, tmp.size() * 4, stream), "d2h weights");
    chk(mcn_stream_sync(stream), "sync");
    for (long r = 0; r < h.rows; r++)
        for (long c = 0; c < h.cols; c++)
            h.data[h.fortran ? c * h.rows + r : r * h.cols + c] = tmp[memIndex(r, c)];
}

void Weights::copyToGPU(void* stream) {  // weights.cuh:134-147
    if (_src) return;
    _W.ensure(size());
    _inc.ensure(size());
    upload(_hW, _W.p, stream);
    upload(_hInc, _inc.p, stream);
}

void Weights::copyToCPU(void* stream) {  // weights.cuh:124-130: fills the caller's arrays in place
    if (_src) return;
    download(_hW, _W.p, stream);
    download(_hInc, _inc.p, stream);
}

void Weights::update(bool dp, void* stream) {
    if (_src == nullptr && _eps > 0) {
        chk(mcn_sgd_update(_W.p, _inc.p, _grad, size(), _mom, _wc, _eps, (_useGrad || dp) ? 1 : 0, stream), "sgd update");
        _numUpdates = 0;
    }
}

bool Weights::planItem(bool dp, ::mcn_sgd_item& it) {
    if (_src || !(_eps > 0)) return false;
    it.W = _W.p;
    it.inc = _inc.p;
    it.useGrad = (_useGrad || dp) ? 1 : 0;
    it.grad = it.useGrad ? _grad : nullptr;
    it.n = size();
    it.mom = _mom;
    it.wc = _wc;
    it.eps = _eps;
    return true;
}

void Weights::hostImage(std::vector<float>& out, void* stream) {
    std::vector<float> tmp(size());
    chk(mcn_d2h(tmp.data(), W(), size() * 4, stream), "d2h");
    chk(mcn_stream_sync(stream), "sync");
    out.resize(size());
    for (long r = 0; r < rows(); r++)
        for (long c = 0; c < cols(); c++) out[(size_t)r * cols() + c] = tmp[memIndex(r, c)];
}

void Weights::setElement(long r, long c, float v, void* stream) {
    chk(mcn_h2d(W() + memIndex(r, c), &v, 4, stream), "h2d element");
    chk(mcn_stream_sync(stream), "sync");
}

// ============================================================================ Layer
Layer::Layer(ConvNet* net, const Params& p) : _net(net) {  // layer.cu:44-57
    _name = pstr(p, "name");
    _type = pstr(p, "type");
    _outputs = (int)pint(p, "outputs", 0);
    _gradConsumer = pint(p, "gradConsumer") != 0;
    _actsTarget = (int)pint(p, "actsTarget");
    _actsGradTarget = (int)pint(p, "actsGradTarget");
}

float* Layer::getActs() {
    if (!_acts) throw NetError("layer '" + _name + "' has no activities yet");
    return _acts;
}

float* Layer::getActsGrad() {
    if (_actsGradTarget >= 0) return _prev[_actsGradTarget]->getActsGrad();
    return _ownActsGrad.p;
}

bool Layer::isGradConsumer() {  // layer.cu:202-210
    if (!_foundGradConsumers) {
        for (Layer* l : _prev) _gradConsumer |= l->isGradConsumer();
        _foundGradConsumers = true;
    }
    return _gradConsumer;
}

void Layer::allocate(int maxCases) {
    if (_actsTarget < 0) _ownActs.ensure(std::max<size_t>(actsElems(maxCases), 4));
    if (_actsGradTarget < 0 && isGradConsumer() && _outputs > 0) _ownActsGrad.ensure((size_t)_outputs * maxCases);
}

void Layer::fpropFromPrev(PassType pt) {  // layer.cu:75-84
    _rcvdF += 1;
    if (_rcvdF == (int)_prev.size()) {
        std::vector<float*> v;
        for (Layer* l : _prev) v.push_back(l->getActs());
        fprop(v, pt);
    }
}

void Layer::fprop(const std::vector<float*>& inputs, PassType pt) {  // layer.cu:92-114
    if (inputs.size() != _prev.size()) throw NetError("layer '" + _name + "': wrong number of inputs");
    _net->beforeLayerFprop(this);
    _inputs = inputs;
    _acts = _actsTarget < 0 ? _ownActs.p : _inputs[_actsTarget];
    _rcvdF = (int)_prev.size();
    if (_actsTarget >= 0) {
        ProfScope ps(_net, _name, "fprop", _actsTarget);
        fpropActs(_actsTarget, 0, pt);
    }
    for (int i = 0; i < (int)_prev.size(); i++)
        if (i != _actsTarget) {
            ProfScope ps(_net, _name, "fprop", i);
            fpropActs(i, (_actsTarget >= 0 || i > 0) ? 1.f : 0.f, pt);
        }
    // cost layers first: their work is forked to the side stream, and a fork covers everything the main stream holds at
    // that point -- after fc6's product had been queued, the L2-SN cost behind fc5 started a whole fc6 forward later
    static const bool costFirst = getenv("MCN_NO_COST_FIRST") == nullptr;
    if (costFirst && _next.size() > 1) {
        for (Layer* l : _next)
            if (l->type().compare(0, 5, "cost.") == 0) l->fpropFromPrev(pt);
        for (Layer* l : _next)
            if (l->type().compare(0, 5, "cost.") != 0) l->fpropFromPrev(pt);
    } else {
        for (Layer* l : _next) l->fpropFromPrev(pt);
    }
}

void Layer::bprop(PassType pt) {  // layer.cu:116-121
    if (_rcvdB == _numGradProducersNext) {
        _rcvdB++;
        bpropWith(getActsGrad(), pt);
    }
}

void Layer::awaitSeed() {
    if (!_seeded) return;
    _seeded = false;
    _net->waitSideSubmitted(_seedTicket);  // the side thread has recorded the event
    chk(timedStreamWait(_net->stream(), _seedEvent), "wait event");
}

void Layer::bpropWith(float* v, PassType pt) {  // layer.cu:123-158
    awaitSeed();
    bpropCommon(v, pt);
    if (isGradProducer()) {
        for (int i = 0; i < (int)_prev.size(); i++) {
            if (_prev[i]->isGradConsumer() && _actsGradTarget != i) {
                _prev[i]->awaitSeed();
                ProfScope ps(_net, _name, "bpropActs", i);
                bpropActs(v, i, _prev[i]->rcvdB() > 0 ? 1.f : 0.f, pt);
                _prev[i]->incRcvdB();
            }
        }
        if (_actsGradTarget >= 0 && _prev[_actsGradTarget]->isGradConsumer()) {
            _prev[_actsGradTarget]->awaitSeed();
            ProfScope ps(_net, _name, "bpropActs", _actsGradTarget);
            bpropActs(v, _actsGradTarget, _prev[_actsGradTarget]->rcvdB() > 0 ? 1.f : 0.f, pt);
            _prev[_actsGradTarget]->incRcvdB();
        }
        for (Layer* l : _prev)
            if (l->isGradConsumer()) l->bprop(pt);
    }
}

// ============================================================================ NeuronLayer (layer.cu:240-251)
NeuronLayer::NeuronLayer(ConvNet* net, const Params& p) : Layer(net, p) {
    const Value& nv = pget(p, "neuron");
    if (nv.kind != Value::DICT) throw NetError("layer '" + _name + "': 'neuron' must be a dict");
    spec.type = neuronTypeId(pstr(*nv.dict, "type"));
    if (phas(*nv.dict, "params")) {
        const Value& pv = pget(*nv.dict, "params");
        if (pv.kind == Value::DICT) {
            if (phas(*pv.dict, "a")) spec.a = (float)pfloat(*pv.dict, "a");
            if (phas(*pv.dict, "b")) spec.b = (float)pfloat(*pv.dict, "b");
        }
    }
}

void NeuronLayer::fpropActs(int, float, PassType) {
    if (fwdFused) return;  // the producer's epilogue already applied the activation in place
    if (spec.type == MCN_NEURON_IDENT && _acts == _inputs[0]) return;
    chk(mcn_neuron_fwd(_inputs[0], _acts, (size_t)_outputs * _net->numCases(), spec.type, spec.a, spec.b, _net->stream()), "neuron fwd");
}

void NeuronLayer::bpropActs(float* v, int, float scaleTargets, PassType) {
    if (bwdFused) return;  // the max-pool undo above already wrote neuronGrad(undo) into the shared gradient
    float* tgt = _prev[0]->getActsGrad();
    chk(mcn_neuron_bwd(v, _acts, _inputs[0], tgt, (size_t)_outputs * _net->numCases(), spec.type, spec.a, spec.b,
                       scaleTargets > 0 ? 1 : 0, _net->stream()), "neuron bwd");
}

// ============================================================================ WeightLayer (layer.cu:258-343)
WeightLayer::WeightLayer(ConvNet* net, const Params& p, bool useGrad, bool fortran) : Layer(net, p) {
    std::vector<HostMat> hW = pmatv(p, "weights"), hInc = pmatv(p, "weightsInc");
    HostMat hB = pmat(p, "biases"), hBInc = pmat(p, "biasesInc");
    std::vector<double> momW = pfloatv(p, "momW"), epsW = pfloatv(p, "epsW"), wc = pfloatv(p, "wc");
    std::vector<long> srcL = pintv(p, "weightSourceLayerIndices"), srcM = pintv(p, "weightSourceMatrixIndices");
    if (hW.size() != hInc.size() || hW.size() != epsW.size() || hW.size() != momW.size() || hW.size() != wc.size() ||
        hW.size() != srcL.size() || hW.size() != srcM.size())
        throw NetError("layer '" + _name + "': weight parameter lists differ in length");
    // Weight sharing (`weightSource`, weights.cuh:59-68, layer.cu:277-289): a sharer uses the owner's device matrices with its own
    // epsW; the update counter lives in the owner.  Every gradient piece of a net that shares weights is issued by ONE
    // submitting thread on ONE stream (ConvNet::finishSideWork / joinGradWork), in the order of the backward pass, so the
    // accumulation order -- and for FC / local matrices the reference's own "last contribution wins" rule, layer.cu:378,
    // :554 -- is the reference's.
    for (size_t i = 0; i < hW.size(); i++) {
        if (srcL[i] == net->numLayers()) {  // current layer (layer.cu:280-281)
            if (srcM[i] < 0 || srcM[i] >= (long)_weights.size()) throw NetError("layer '" + _name + "': weight source matrix not defined yet");
            _weights.emplace_back(new Weights(*_weights[srcM[i]], (float)epsW[i]));
            net->noteWeightSharing();
        } else if (srcL[i] >= 0) {
            WeightLayer* src = srcL[i] < net->numLayers() ? dynamic_cast<WeightLayer*>(&net->layer((int)srcL[i])) : nullptr;
            if (!src) throw NetError("layer '" + _name + "': weight source is not a weight layer");
            if (srcM[i] < 0 || srcM[i] >= src->numWeights()) throw NetError("layer '" + _name + "': weight source matrix index out of range");
            Weights& sw = src->weights((int)srcM[i]);
            if (sw.rows() != hW[i].rows || sw.cols() != hW[i].cols || sw.useGrad() != useGrad)
                throw NetError("layer '" + _name + "': weight source matrix has another shape or layer type");
            _weights.emplace_back(new Weights(sw, (float)epsW[i]));
            net->noteWeightSharing();
        } else {
            _weights.emplace_back(new Weights(hW[i], hInc[i], (float)epsW[i], (float)wc[i], (float)momW[i], useGrad, fortran));
        }
    }
    _biases.reset(new Weights(hB, hBInc, (float)pfloat(p, "epsB"), 0.f, (float)pfloat(p, "momB"), true, false));
}

float WeightLayer::gradScale(float eps, PassType pt) const {
    return pt == PASS_GC ? 1.f : eps / (float)(_net->numCases() * _net->worldSize());
}

void WeightLayer::bpropCommon(float* v, PassType pt) {  // layer.cu:306-317
    // two pieces of side-stream work: they start at once next to this layer's backward-data kernels, except for the
    // last weight layer of the pass, which has nothing left to hide behind and is split over both streams
    const bool last = _net->lastWeightLayerOfPass(this);
    // the bias gradient: a row sum queued for the pass's one batched launch -- unless the weight-gradient launch below
    // delivers it (conv, strip form).  The batch goes out with the LAST layer of the pass that adds to it, so that a last
    // layer without a bias piece of its own (DeepID's conv1) leaves nothing for the side stream to do at the very end.
    const bool biasHere = _biases->eps() > 0 && !(_biasInWgrad && pt == PASS_TRAIN && _net->fusedTailNow());
    const bool flushHere = _net->biasBatching() && _net->flushesBiasBatch(this, last);
    // (queued ahead of this layer's own pieces; not when this layer still adds to -- and flushes -- the pass's bias batch: the
    // early update reads the other layers' bias sums)
    if (last && pt == PASS_TRAIN && !flushHere && !(biasHere && _net->biasBatching())) _net->earlyUpdate();
    if (biasHere || flushHere) {
        _net->deferSideWork([this, v, pt, biasHere, flushHere] {
            if (biasHere) {
                if (_net->biasBatching()) {  // a row sum like every other layer's: all of them in one launch
                    int rows, cols;
                    biasSumShape(_net->numCases(), rows, cols);
                    _net->queueBiasSum(_biases->gradBuf(), v, rows, cols, gradScale(_biases->eps(), pt), _biasParts.p);
                } else {
                    ProfScope ps(_net, _name, "bpropBiases");
                    bpropBiases(v, pt);
                }
            }
            if (flushHere) _net->flushBiasSums();
        }, 1, true);
    }
    _net->deferSideWork([this, v, pt] {
        for (int i = 0; i < (int)_weights.size(); i++) {
            if (_weights[i]->eps() > 0) {
                ProfScope ps(_net, _name, "bpropWeights", i);
                bpropWeights(v, i, pt);
                _weights[i]->incNumUpdates();
            }
        }
        if (_net->dataParallel() && pt == PASS_TRAIN) _net->gradReady(this);
    }, (biasHere || flushHere || last) ? 1 : 2, last);  // (a layer without a bias piece still goes to the side stream at once)
}

void WeightLayer::updateWeights() {
    ProfScope ps(_net, _name, "update");
    for (auto& w : _weights) w->update(_net->dataParallel(), _net->stream());
    _biases->update(_net->dataParallel(), _net->stream());
}
void WeightLayer::weightsUpdated() {
    for (auto& w : _weights) w->updated();
    _biases->updated();
}
void WeightLayer::resetUpdateCounters() {
    for (auto& w : _weights) w->resetUpdates();
    _biases->resetUpdates();
}
bool WeightLayer::sharesWeights() const {
    for (auto& w : _weights)
        if (!w->owner()) return true;
    return false;
}
void WeightLayer::copyToCPU() {
    for (auto& w : _weights) w->copyToCPU(_net->stream());
    _biases->copyToCPU(_net->stream());
}
void WeightLayer::copyToGPU() {
    for (auto& w : _weights) w->copyToGPU(_net->stream());
    _biases->copyToGPU(_net->stream());
}
bool WeightLayer::planFusedTail(int numCases) {
    // bias gradient = row sums of the layer's gradient matrix; rows longer than one chunk (conv layers with shared biases)
    // take two stages, and the second one can ride in the update
    _biases->partials = Weights::Partials();
    if (!(_biases->eps() > 0)) return true;
    int rows, cols;
    biasSumShape(numCases, rows, cols);
    const size_t bytes = mcn_add_sum_workspace_bytes(rows, cols, 1);
    if (bytes == 0) return true;
    const int chunks = (int)(bytes / sizeof(float) / rows);
    if (chunks > 256) return false;  // (rows of more than 2 M columns: their second stage is a block reduction, not a wave's)
    _biasParts.ensure(bytes / sizeof(float));
    _biases->partials.data = _biasParts.p;
    _biases->partials.count = chunks;
    _biases->partials.kind = 2;
    _biases->partials.scale = _biases->eps() / (float)(numCases * _net->worldSize());
    return true;
}
void WeightLayer::collectWeights(std::vector<Weights*>& out) {
    for (auto& w : _weights) out.push_back(w.get());
    out.push_back(_biases.get());
}
void WeightLayer::checkGradients(std::vector<GradCheckResult>& out, long maxElems, bool central) {  // layer.cu:334-339
    for (int i = 0; i < (int)_weights.size(); i++)
        _net->checkGradient(_name + " weights[" + std::to_string(i) + "]", _wStep, *_weights[i], out, maxElems, central);
    _net->checkGradient(_name + " biases", _bStep, *_biases, out, maxElems, central);
}

// ============================================================================ FCLayer (layer.cu:350-384)
// Memory image of W_i is [outputs][inputs_i] (Fortran-order (inputs, outputs) host matrix,
// layer.py:788); activities are (features, cases).  fwd: Y = W X; bwd-data: dX = W^T dY;
// bwd-weights: inc = mom*inc + (eps/N) dY X^T.
FCLayer::FCLayer(ConvNet* net, const Params& p) : WeightLayer(net, p, false, true) {
    _wStep = 0.1f;  // layer.cu:351-352
    _bStep = 0.01f;
}

size_t FCLayer::workspaceBytes(int maxCases) {
    size_t b = mcn_add_sum_workspace_bytes(_outputs, maxCases, 1);
    for (Layer* l : _prev) {
        const int I = l->outputs();
        b = std::max(b, mcn_gemm_workspace_bytes(_outputs, maxCases, I, 0, 0));
        b = std::max(b, mcn_gemm_workspace_bytes(I, maxCases, _outputs, 1, 0));
        b = std::max(b, mcn_gemm_workspace_bytes(_outputs, I, maxCases, 0, 1));
    }
    return b;
}

void FCLayer::planMulti() {
    multiShapes = _prev.size() >= 2 && _prev.size() <= 4 && getenv("MCN_NO_FC_MULTI") == nullptr;  // (A/B switch)
    for (size_t a = 0; a < _prev.size(); a++) {
        multiShapes = multiShapes && _prev[a]->outputs() % 16 == 0;
        for (size_t b = a + 1; b < _prev.size(); b++) multiShapes = multiShapes && _prev[a] != _prev[b];
    }
    neuronAbove.assign(_prev.size(), nullptr);
    _bwdDone.assign(_prev.size(), 0);
}

void FCLayer::fpropActs(int i, float scaleTargets, PassType) {
    const int n = _net->numCases(), I = _prev[i]->outputs();
    if (multiShapes && _net->fusion() && _actsTarget < 0) {
        // every input in one launch (the reference: one sgemm per input, layer.cu:355-360)
        if (i == 0) {
            const float* W[4];
            const float* X[4];
            int K[4];
            const int cnt = (int)_prev.size();
            for (int k = 0; k < cnt; k++) { W[k] = _weights[k]->W(); X[k] = _inputs[k]; K[k] = _prev[k]->outputs(); }
            _fwdAllDone = mcn_fc_fprop_multi_supported(W, X, K, cnt, _acts, _outputs, n) != 0;
            if (_fwdAllDone) {
                chk(mcn_fc_fprop_multi(W, X, K, cnt, _biases->W(), _acts, _outputs, n, fusedNeuron ? fusedNeuron->spec.type : MCN_NEURON_IDENT,
                                       fusedNeuron ? fusedNeuron->spec.a : 0.f, fusedNeuron ? fusedNeuron->spec.b : 0.f, _net->stream()),
                    "fc fprop (all inputs)");
                return;
            }
        } else if (_fwdAllDone) {
            return;
        }
    }
    if (scaleTargets == 0 && softmaxBelow && _net->fusion() && softmaxBelow->part.p &&
        mcn_gemm_softmax_part_supported(_weights[i]->W(), _inputs[i], _acts, _outputs, n, I)) {
        // classifier layer: the product also leaves stage 1 of the softmax that follows (no pass of its own over the acts)
        chk(mcn_gemm_bias_neuron_softmax_part(_weights[i]->W(), _inputs[i], _biases->W(), _acts, _outputs, n, I,
                                              fusedNeuron ? fusedNeuron->spec.type : MCN_NEURON_IDENT, fusedNeuron ? fusedNeuron->spec.a : 0.f,
                                              fusedNeuron ? fusedNeuron->spec.b : 0.f, softmaxBelow->part.p, softmaxBelow->partBytes,
                                              _net->stream()), "fc fprop + softmax stage 1");
        softmaxBelow->partReady = true;
    } else if (scaleTargets == 0) {
        const bool fuse = fusedNeuron && _prev.size() == 1;  // (several inputs: the neuron follows the last product below)
        chk(mcn_gemm_bias_neuron(_weights[i]->W(), _inputs[i], _biases->W(), _acts, _outputs, n, I, 0.f,
                                 fuse ? fusedNeuron->spec.type : MCN_NEURON_IDENT, fuse ? fusedNeuron->spec.a : 0.f,
                                 fuse ? fusedNeuron->spec.b : 0.f, _net->workspace(), _net->workspaceBytes(), _net->stream()),
            "fc fprop");
    } else {
        chk(mcn_gemm(_weights[i]->W(), _inputs[i], _acts, _outputs, n, I, 0, 0, 1.f, scaleTargets, _net->workspace(),
                     _net->workspaceBytes(), _net->stream()), "fc fprop");
    }
    if (fusedNeuron && _prev.size() > 1 && i == (int)_prev.size() - 1)  // fused by plan, but this shape took the per-input path
        chk(mcn_neuron_fwd(_acts, _acts, (size_t)_outputs * n, fusedNeuron->spec.type, fusedNeuron->spec.a, fusedNeuron->spec.b, _net->stream()),
            "fc neuron");
}

void FCLayer::bpropActs(float* v, int i, float scaleTargets, PassType) {
    const int n = _net->numCases(), I = _prev[i]->outputs();
    if (_bwdDone[i]) {  // written by the all-inputs launch of an earlier input of this pass
        _bwdDone[i] = 0;
        return;
    }
    if (multiShapes && _net->fusion() && _actsGradTarget < 0) {
        // every gradient-consuming input in one launch (the reference: one sgemm per input, layer.cu:362-366)
        const float* W[4];
        float* T[4];
        const float* NA[4];
        int K[4], NT[4], idx[4];
        float sT[4], na[4], nb[4];
        int cnt = 0;
        for (int k = i; k < (int)_prev.size(); k++) {
            if (!_prev[k]->isGradConsumer()) continue;
            W[cnt] = _weights[k]->W();
            T[cnt] = _prev[k]->getActsGrad();
            K[cnt] = _prev[k]->outputs();
            sT[cnt] = k == i ? scaleTargets : (_prev[k]->rcvdB() > 0 ? 1.f : 0.f);
            NeuronLayer* r = neuronAbove[k];
            NA[cnt] = r ? r->getActs() : nullptr;
            NT[cnt] = r ? r->spec.type : MCN_NEURON_IDENT;
            na[cnt] = r ? r->spec.a : 0.f;
            nb[cnt] = r ? r->spec.b : 0.f;
            idx[cnt++] = k;
        }
        if (cnt >= 2 && mcn_fc_bprop_acts_multi_supported(W, T, K, cnt, v, _outputs, n)) {
            for (int c = 1; c < cnt; c++) _prev[idx[c]]->awaitSeed();
            chk(mcn_fc_bprop_acts_multi(W, T, K, sT, NA, NT, na, nb, cnt, v, _outputs, n, _net->stream()), "fc bprop acts (all inputs)");
            for (int c = 1; c < cnt; c++) _bwdDone[idx[c]] = 1;
            return;
        }
    }
    float* tgt = _prev[i]->getActsGrad();
    // the classifier layer above the feature layer: the feature layer's second gradient, from the L2-SN cost, is one elementwise
    // pass over the same matrix and waits for a stream gate -- both go into this product's reduce launch (two launches fewer on
    // the backward chain; MCN_NO_L2SN_FOLD=1 keeps them)
    const bool noFold = getenv("MCN_NO_L2SN_FOLD") != nullptr;  // (read per call: tests toggle it in one process)
    if (scaleTargets == 0 && !neuronAbove[i] && t_stream == 0 && _net->fusion() && !noFold && _prev[i]->numGradProducersNext() == 2)
        for (Layer* l : _prev[i]->next()) {
            auto* c = dynamic_cast<L2SNCostLayer*>(l);
            L2SNCostLayer::Fold f;
            if (c && c->foldable(_prev[i], f) && mcn_fc_bprop_acts_l2sn_supported(_weights[i]->W(), v, tgt, I, n, _outputs) == 1) {
                chk(mcn_fc_bprop_acts_l2sn(_weights[i]->W(), v, tgt, I, n, _outputs, _net->workspace(), _net->workspaceBytes(), f.data,
                                           f.stats, f.info, f.coeff, _net->gateWord(), f.gateSeq, _net->gateTimeoutMs(), _net->stream()),
                    "fc bprop acts + l2sn grad");
                c->gradFolded();
                return;
            }
        }
    chk(mcn_gemm(_weights[i]->W(), v, tgt, I, n, _outputs, 1, 0, 1.f, scaleTargets, _net->workspace(), _net->workspaceBytes(),
                 _net->stream()), "fc bprop acts");
    if (NeuronLayer* r = neuronAbove[i])  // its gradient is this layer's job (buildFusions); scaleTargets is 0 by construction
        chk(mcn_neuron_bwd(tgt, r->getActs(), r->getActs(), tgt, (size_t)I * n, r->spec.type, r->spec.a, r->spec.b, 0, _net->stream()),
            "fc bprop acts: neuron grad");
}

void FCLayer::bpropBiases(float* v, PassType pt) {
    chk(mcn_add_sum(_biases->gradBuf(), v, _outputs, _net->numCases(), 1, 0.f, gradScale(_biases->eps(), pt), _net->workspace(),
                    _net->workspaceBytes(), _net->stream()), "fc bprop biases");
}

void FCLayer::bpropWeights(float* v, int i, PassType pt) {
    const int n = _net->numCases(), I = _prev[i]->outputs();
    const bool dp = _net->dataParallel();
    // several inputs (fc5): the gradients of all weight matrices in one launch, issued with the first input
    static const bool noMulti = getenv("MCN_NO_FC_MULTI") != nullptr || getenv("MCN_NO_FC_WGRAD_MULTI") != nullptr;
    // (big layers -- DeepID2+ fc5, 2.7 M weights -- do better with one general NT product per input: 13 + 17 us on the side stream
    // against 28 in one launch of 32 x 32 tiles, and the two-stream step is 6 us shorter)
    static const size_t multiMax = getenv("MCN_FC_WGRAD_MULTI_MAX_ELEMS") ? (size_t)atol(getenv("MCN_FC_WGRAD_MULTI_MAX_ELEMS")) : (size_t)1 << 20;
    size_t elems = 0;
    for (auto& wk : _weights) elems += (size_t)wk->rows() * wk->cols();
    if (_weights.size() > 1 && _net->fusion() && !noMulti && elems <= multiMax) {
        if (i > 0 && _wgradAllDone) {
            if (i + 1 == (int)_weights.size()) _wgradAllDone = false;
            return;
        }
        if (i == 0) {
            const int cnt = (int)_weights.size();
            std::vector<const float*> xs(cnt);
            std::vector<float*> ts(cnt);
            std::vector<int> is(cnt);
            std::vector<float> so(cnt), st(cnt);
            bool all = cnt <= 4;
            for (int k = 0; k < cnt && all; k++) {
                Weights& wk = *_weights[k];
                all = wk.eps() > 0;
                xs[k] = _prev[k]->getActs();
                ts[k] = wk.getGrad(dp);
                is[k] = _prev[k]->outputs();
                so[k] = gradScale(wk.eps(), pt);
                st[k] = dp ? (wk.numUpdates() > 0 ? 1.f : 0.f) : ((wk.numUpdates() == 0 && pt != PASS_GC) ? wk.mom() : 0.f);
                for (int j = 0; j < k; j++) all = all && ts[j] != ts[k];  // (two inputs sharing one matrix: one after the other)
            }
            if (all && mcn_fc_bprop_weights_multi_supported(v, xs.data(), ts.data(), is.data(), cnt, _outputs, n) == 1) {
                chk(mcn_fc_bprop_weights_multi(v, xs.data(), ts.data(), is.data(), so.data(), st.data(), cnt, _outputs, n, _net->stream()),
                    "fc bprop weights (all inputs)");
                _wgradAllDone = true;
                return;
            }
        }
    }
    Weights& w = *_weights[i];
    // single GPU: momentum folded into the GEMM epilogue (layer.cu:378-381); data-parallel: raw
    // scaled gradient, momentum applied after the all-reduce (algebraically identical)
    const float scaleTgt = dp ? (w.numUpdates() > 0 ? 1.f : 0.f) : ((w.numUpdates() == 0 && pt != PASS_GC) ? w.mom() : 0.f);
    chk(mcn_gemm(v, _prev[i]->getActs(), w.getGrad(dp), _outputs, I, n, 0, 1, gradScale(w.eps(), pt), scaleTgt, _net->workspace(),
                 _net->workspaceBytes(), _net->stream()), "fc bprop weights");
}

// ============================================================================ LocalLayer / ConvLayer / LocalUnsharedLayer
LocalLayer::LocalLayer(ConvNet* net, const Params& p, bool useGrad, bool conv) : WeightLayer(net, p, useGrad, false), _conv(conv) {
    _padding = pintv(p, "padding");  // layer.cu:393-407
    _stride = pintv(p, "stride");
    _filterSize = pintv(p, "filterSize");
    _channels = pintv(p, "channels");
    _groups = pintv(p, "groups");
    _filterChannels = pintv(p, "filterChannels");
    std::vector<long> imgSize = pintv(p, "imgSize");
    _imgSizeX = phas(p, "imgSizeX") ? pintv(p, "imgSizeX") : imgSize;  // rectangular extension
    _imgSizeY = phas(p, "imgSizeY") ? pintv(p, "imgSizeY") : imgSize;
    _filters = (int)pint(p, "filters");
    _modulesX = (int)pint(p, "modulesX");
    _modulesY = (int)pint(p, "modulesY", _modulesX);
    _modules = (int)pint(p, "modules");
    if (_modules != _modulesX * _modulesY) throw NetError("layer '" + _name + "': modules != modulesY*modulesX");
    if (phas(p, "randSparse")) {  // layer.cu:410-418
        std::vector<long> rs = pintv(p, "randSparse");
        _sparse.resize(rs.size());
        for (size_t i = 0; i < rs.size(); i++) {
            if (!rs[i]) continue;
            const Value& fc = pget(p, "filterConns");
            if (fc.kind != Value::LIST || fc.list.size() <= i || fc.list[i].kind != Value::LIST)
                throw NetError("layer '" + _name + "': 'filterConns' must hold one list of channel indices per input");
            _sparse[i].reset(new Sparse);
            for (const Value& e : fc.list[i].list) _sparse[i]->conns.push_back((int)(e.kind == Value::FLOAT ? (long)e.f : e.i));
            const long wide = _groups[i] * _filterChannels[i];
            if (_groups[i] < 2 || _filterChannels[i] % 4 || _channels[i] % _filterChannels[i] || wide % _channels[i] ||
                (long)_sparse[i]->conns.size() != wide)
                throw NetError("layer '" + _name + "': inconsistent random-sparse connectivity (layer.py:908-914)");
            for (long s = 0; s < wide / _channels[i]; s++) {  // every run of `channels` entries is a permutation (layer.py:841-843)
                std::vector<char> seen(_channels[i], 0);
                for (long j = 0; j < _channels[i]; j++) {
                    const int c = _sparse[i]->conns[s * _channels[i] + j];
                    if (c < 0 || c >= _channels[i] || seen[c]) throw NetError("layer '" + _name + "': filterConns is not a list of channel permutations");
                    seen[c] = 1;
                }
            }
        }
    }
    if (conv) {
        _partialSum = (int)pint(p, "partialSum");  // layer.cu:439-440
        _sharedBiases = (int)pint(p, "sharedBiases");
    }
}

size_t LocalLayer::workspaceBytes(int maxCases) {
    size_t b = (_conv && _sharedBiases) ? mcn_add_sum_workspace_bytes(_filters, _modules * maxCases, 1)
                                        : mcn_add_sum_workspace_bytes(_filters * _modules, maxCases, 1);
    if (_conv)
        for (size_t i = 0; i < _prev.size(); i++)
            b = std::max(b, mcn_conv_weight_acts_reduced_workspace_bytes(sparse((int)i) ? wideChannels((int)i) : (int)_channels[i], _filters,
                                                                        (int)_filterSize[i], _modulesY, _modulesX, (int)_groups[i]));
    return b;
}

bool LocalLayer::planFusedTail(int numCases) {
    _biasInWgrad = false;
    if (!WeightLayer::planFusedTail(numCases)) return false;
    _wgradParts.resize(_weights.size());
    _wgradChunks.assign(_weights.size(), 0);
    for (size_t i = 0; i < _weights.size(); i++) {
        Weights& w = *_weights[i];
        w.partials = Weights::Partials();
        if (!_conv || !w.owner() || !(w.eps() > 0)) continue;
        const int channels = sparse((int)i) ? wideChannels((int)i) : (int)_channels[i];
        const size_t bytes = mcn_conv_weight_acts_reduced_workspace_bytes(channels, _filters, (int)_filterSize[i], _modulesY, _modulesX, (int)_groups[i]);
        if (!_wgradParts[i]) _wgradParts[i].reset(new DBuf());
        _wgradParts[i]->ensure(bytes / sizeof(float) + 4);
        w.partials.data = _wgradParts[i]->p;
        w.partials.kind = 1;
        int biasCapable = 0;
        w.partials.count = mcn_conv_weight_acts_partials_count(numCases, numCases, (int)_imgSizeY[i], (int)_imgSizeX[i], channels, _filters,
                                                               (int)_filterSize[i], _modulesY, _modulesX, (int)_stride[i], (int)_groups[i], &biasCapable);
        if (w.partials.count <= 0) w.partials = Weights::Partials();
        const bool noBiasInWgrad = getenv("MCN_NO_BIAS_IN_WGRAD") != nullptr;  // (A/B switch; read per model: tests toggle it in one process)
        if (i == 0 && w.partials.count > 0 && biasCapable && _sharedBiases && _biases->eps() > 0 && !noBiasInWgrad) {
            // the strip kernel stages every gradient row of the layer: it sums them for the shared biases on the way
            _biasParts.ensure((size_t)w.partials.count * _filters);
            _biases->partials.data = _biasParts.p;
            _biases->partials.count = w.partials.count;
            _biases->partials.kind = 1;
            _biases->partials.scale = 1.f;
            _biasInWgrad = true;
        }
    }
    return true;
}

void LocalLayer::allocate(int maxCases) {
    WeightLayer::allocate(maxCases);
    for (size_t i = 0; i < _sparse.size(); i++)
        if (_sparse[i]) {
            const size_t n = (size_t)wideChannels((int)i) * _imgSizeY[i] * _imgSizeX[i] * maxCases;
            _sparse[i]->gathered.ensure(n);
            if (_prev[i]->isGradConsumer()) _sparse[i]->gradTmp.ensure(n);
        }
}

void LocalLayer::copyToGPU() {  // layer.cu:421-431
    WeightLayer::copyToGPU();
    for (size_t i = 0; i < _sparse.size(); i++)
        if (_sparse[i]) {
            Sparse& sp = *_sparse[i];
            sp.dConns.ensure(sp.conns.size());
            sp.dInverse.ensure(sp.conns.size());
            chk(mcn_h2d(sp.dConns.p, sp.conns.data(), sp.conns.size() * sizeof(int), _net->stream()), "h2d filterConns");
            chk(mcn_invert_channel_indices((const int*)sp.dConns.p, (int*)sp.dInverse.p, (int)_channels[i],
                                           (int)(sp.conns.size() / _channels[i]), _net->stream()), "invert filterConns");
            chk(mcn_stream_sync(_net->stream()), "sync");
        }
}

void LocalLayer::fpropActs(int i, float scaleTargets, PassType) {  // layer.cu:443-462, :530-543
    const int n = _net->numCases();
    const float* in = _inputs[i];
    int channels = (int)_channels[i];
    if (sparse(i)) {  // the channel table applied once; the product is the dense grouped one over the re-ordered input (sparse_conv.hip)
        Sparse& sp = *_sparse[i];
        channels = wideChannels(i);
        chk(mcn_gather_channels(_inputs[i], n, (const int*)sp.dConns.p, sp.gathered.p, n, (int)_channels[i], channels,
                                (int)(_imgSizeY[i] * _imgSizeX[i]), _net->stream()), "gather channels");
        in = sp.gathered.p;
    }
    if (scaleTargets == 0) {
        const bool fuse = fusedNeuron && _prev.size() == 1;
        const int biasPerModule = (_conv && _sharedBiases) ? 0 : 1, ntype = fuse ? fusedNeuron->spec.type : MCN_NEURON_IDENT;
        float* pooled = fusedPool && _prev.size() == 1 ? fusedPool->ownActs() : nullptr;
        if (pooled && mcn_filter_acts_pool_supported(in, _weights[i]->W(), _biases->W(), _acts, n, n, (int)_imgSizeY[i], (int)_imgSizeX[i],
                                                     channels, _filters, (int)_filterSize[i], _modulesY, _modulesX, (int)_padding[i],
                                                     (int)_stride[i], (int)_groups[i], _conv ? 1 : 0, biasPerModule, ntype) == 1) {
            chk(mcn_filter_acts_bias_neuron_pool(in, _weights[i]->W(), _biases->W(), _acts, pooled, n, n, (int)_imgSizeY[i],
                                                 (int)_imgSizeX[i], channels, _filters, (int)_filterSize[i], _modulesY, _modulesX,
                                                 (int)_padding[i], (int)_stride[i], (int)_groups[i], _conv ? 1 : 0, biasPerModule, ntype,
                                                 fuse ? fusedNeuron->spec.a : 0.f, fuse ? fusedNeuron->spec.b : 0.f, _net->stream()),
                "filterActs + pool");
            fusedPool->fwdDoneBelow = true;
            return;
        }
        chk(mcn_filter_acts_bias_neuron(in, _weights[i]->W(), _biases->W(), _acts, n, n, (int)_imgSizeY[i], (int)_imgSizeX[i],
                                        channels, _filters, (int)_filterSize[i], _modulesY, _modulesX, (int)_padding[i],
                                        (int)_stride[i], (int)_groups[i], _conv ? 1 : 0, (_conv && _sharedBiases) ? 0 : 1,
                                        fuse ? fusedNeuron->spec.type : MCN_NEURON_IDENT, fuse ? fusedNeuron->spec.a : 0.f,
                                        fuse ? fusedNeuron->spec.b : 0.f, _net->stream()), "filterActs");
    } else {
        auto fn = _conv ? mcn_conv_filter_acts : mcn_local_filter_acts;
        chk(fn(in, _weights[i]->W(), _acts, n, n, (int)_imgSizeY[i], (int)_imgSizeX[i], channels, _filters,
               (int)_filterSize[i], _modulesY, _modulesX, (int)_padding[i], (int)_stride[i], (int)_groups[i], scaleTargets, 1.f,
               _net->stream()), "filterActs");
    }
}

void LocalLayer::bpropActs(float* v, int i, float scaleTargets, PassType) {  // layer.cu:498-513, :566-575
    // input = a max pool with non-overlapping windows, and every other consumer of that pool has already added its share:
    // the gradient goes through the pool (and the neuron under it) in the epilogue, never to the pool's own gradient matrix
    auto fn = _conv ? mcn_conv_img_acts : mcn_local_img_acts;
    if (sparse(i)) {  // layer.cu:499-508, :567-570: the dense grouped product, then every image channel collects its overSample copies
        Sparse& sp = *_sparse[i];
        chk(fn(v, _weights[i]->W(), sp.gradTmp.p, _net->numCases(), (int)_imgSizeY[i], (int)_imgSizeX[i], wideChannels(i), _filters,
               (int)_filterSize[i], _modulesY, _modulesX, (int)_padding[i], (int)_stride[i], (int)_groups[i], 0.f, 1.f, _net->stream()),
            "imgActs (sparse)");
        chk(mcn_scatter_channels_sum(sp.gradTmp.p, (const int*)sp.dInverse.p, _prev[i]->getActsGrad(), _net->numCases(), (int)_channels[i],
                                     wideChannels(i) / (int)_channels[i], (int)(_imgSizeY[i] * _imgSizeX[i]), scaleTargets, 1.f,
                                     _net->stream()), "scatter channels");
        return;
    }
    if (auto* pl = dynamic_cast<PoolLayer*>(_prev[i])) {
        if (pl->rcvdB() + 1 == pl->numGradProducersNext() && pl->prev()[0]->isGradConsumer() &&
            pl->undoFusable(_net->numCases(), (int)_channels[i], (int)_imgSizeY[i], (int)_imgSizeX[i])) {
            NeuronLayer* r = pl->fusedNeuronGrad;
            chk(mcn_img_acts_max_undo(v, _weights[i]->W(), scaleTargets != 0 ? pl->getActsGrad() : nullptr, _net->numCases(), (int)_imgSizeY[i],
                                      (int)_imgSizeX[i], (int)_channels[i], _filters, (int)_filterSize[i], _modulesY, _modulesX,
                                      (int)_padding[i], (int)_stride[i], (int)_groups[i], _conv ? 1 : 0, scaleTargets, 1.f,
                                      pl->prev()[0]->getActs(), pl->getActs(), pl->prev()[0]->getActsGrad(), r ? r->spec.type : -1,
                                      r ? r->spec.a : 0.f, r ? r->spec.b : 0.f, _net->stream()), "imgActs + max undo");
            pl->undoDoneAbove = true;
            return;
        }
    }
    chk(fn(v, _weights[i]->W(), _prev[i]->getActsGrad(), _net->numCases(), (int)_imgSizeY[i], (int)_imgSizeX[i], (int)_channels[i],
           _filters, (int)_filterSize[i], _modulesY, _modulesX, (int)_padding[i], (int)_stride[i], (int)_groups[i], scaleTargets, 1.f,
           _net->stream()), "imgActs");
}

void LocalLayer::bpropBiases(float* v, PassType pt) {  // layer.cu:464-474, :545-549
    const int n = _net->numCases();
    const float sc = gradScale(_biases->eps(), pt);
    if (_conv && _sharedBiases)
        chk(mcn_add_sum(_biases->gradBuf(), v, _filters, _modules * n, 1, 0.f, sc, _net->workspace(), _net->workspaceBytes(), _net->stream()),
            "conv bprop biases");
    else
        chk(mcn_add_sum(_biases->gradBuf(), v, _filters * _modules, n, 1, 0.f, sc, _net->workspace(), _net->workspaceBytes(), _net->stream()),
            "local bprop biases");
}

void LocalLayer::bpropWeights(float* v, int i, PassType pt) {
    const int n = _net->numCases();
    Weights& w = *_weights[i];
    const bool dp = _net->dataParallel();
    // random-sparse input: the re-ordered tensor forward left behind, as numGroups*filterChannels dense grouped channels
    const float* in = sparse(i) ? _sparse[i]->gathered.p : _prev[i]->getActs();
    const int channels = sparse(i) ? wideChannels(i) : (int)_channels[i];
    if (_conv && pt == PASS_TRAIN && _net->fusedTailNow() && w.partials.kind == 1 && w.partials.count > 0) {
        // first stage only: the update launch sums the module-chunk partials itself (ConvNet::fusedTailNow)
        int chunks = 0, biasDone = 0;
        const bool wantBias = i == 0 && _biasInWgrad;
        chk(mcn_conv_weight_acts_partials(in, v, _wgradParts[i]->p, _wgradParts[i]->cap * sizeof(float), n, n, (int)_imgSizeY[i],
                                          (int)_imgSizeX[i], channels, _filters, (int)_filterSize[i], _modulesY, _modulesX, (int)_padding[i],
                                          (int)_stride[i], (int)_groups[i], gradScale(w.eps(), pt), &chunks, wantBias ? _biasParts.p : nullptr,
                                          gradScale(_biases->eps(), pt), &biasDone, _net->stream()), "convWeightActs (partials)");
        if (chunks != w.partials.count || (wantBias && !biasDone))
            throw NetError("layer '" + _name + "': the weight gradient came in another form than planned");
        return;
    }
    if (_conv) {
        // layer.cu:476-496: weightActs into a (modules/partialSum) x weights temporary, then addSum over
        // its rows.  Here: one call with a private module split + fixed-order reduction; the value is
        // the sum over ALL modules either way, so `partialSum` only affects rounding order.
        chk(mcn_conv_weight_acts_reduced(in, v, w.getGrad(dp), _net->workspace(), _net->workspaceBytes(), n, n,
                                         (int)_imgSizeY[i], (int)_imgSizeX[i], channels, _filters, (int)_filterSize[i],
                                         _modulesY, _modulesX, (int)_padding[i], (int)_stride[i], (int)_groups[i],
                                         w.numUpdates() > 0 ? 1.f : 0.f, gradScale(w.eps(), pt), _net->stream()), "convWeightActs");
    } else {
        // layer.cu:551-564: straight into inc with momentum (single GPU) / raw gradient (data-parallel)
        const float scaleTgt = dp ? (w.numUpdates() > 0 ? 1.f : 0.f) : ((pt != PASS_GC && w.numUpdates() == 0) ? w.mom() : 0.f);
        chk(mcn_local_weight_acts(in, v, w.getGrad(dp), n, n, (int)_imgSizeY[i], (int)_imgSizeX[i], channels,
                                  _filters, (int)_filterSize[i], _modulesY, _modulesX, (int)_padding[i], (int)_stride[i],
                                  (int)_groups[i], scaleTgt, gradScale(w.eps(), pt), _net->stream()), "localWeightActs");
    }
}

// ============================================================================ SoftmaxLayer (layer.cu:582-607)
size_t SoftmaxLayer::workspaceBytes(int maxCases) { return mcn_softmax_workspace_bytes(_outputs, maxCases); }

void SoftmaxLayer::allocate(int maxCases) {
    Layer::allocate(maxCases);
    partBytes = mcn_softmax_workspace_bytes(_outputs, maxCases);
    part.ensure((partBytes + 3) / 4);
    costStateBytes = mcn_softmax_logreg_cost_state_bytes(maxCases);
    costState.ensure((costStateBytes + 3) / 4);
    chk(mcn_memset(costState.p, 0, costStateBytes, _net->stream()), "softmax cost state");
}

void SoftmaxLayer::fpropActs(int, float, PassType pt) {
    // a logreg cost on top (the DeepID nets): its gradient through this softmax needs only the probabilities and the
    // labels, so a training pass writes it while the probabilities are still in registers (one launch and one read of
    // the (classes, cases) matrix less on the backward chain).  Safe: fc's gradient buffer has no other writer.
    _gradFused = false;
    const bool havePart = partReady;  // the producer's forward product of THIS pass left stage 1
    partReady = false;
    CostLayer* cost = _next.size() == 1 && _next[0]->type() == "cost.logreg" ? static_cast<CostLayer*>(_next[0]) : nullptr;
    if (cost && pt != PASS_TEST && _net->fusion() && cost->coeff() != 0 && _prev[0]->isGradConsumer() && _prev[0]->next().size() == 1) {
        float* labels = cost->prev()[0]->getActs();
        static const bool noCostFusion = getenv("MCN_NO_LOGREG_FUSION") != nullptr;
        LogregCostLayer* lc = static_cast<LogregCostLayer*>(cost);
        if (!noCostFusion && mcn_softmax_logreg_cost_supported(_outputs, _net->numCases()) && lc->dCost) {
            // ... and the cost itself: the last block of the same launch finishes it (no cost kernels at the end of the step)
            chk(mcn_softmax_logreg_cost(_inputs[0], _acts, labels, _prev[0]->getActsGrad(), cost->coeff(), _outputs, _net->numCases(),
                                        havePart ? part.p : nullptr, partBytes, _net->workspace(), _net->workspaceBytes(), lc->logProbsBuf(),
                                        lc->correctBuf(), lc->dCost, costState.p, costStateBytes, 0, _net->stream()), "softmax + logreg grad + cost");
            lc->costDoneBySoftmax = true;
            lc->softmaxState = costState.p;
            lc->softmaxStateBytes = costStateBytes;
            _gradFused = true;
            return;
        }
        if (havePart)
            chk(mcn_softmax_from_part(_inputs[0], _acts, labels, _prev[0]->getActsGrad(), cost->coeff(), _outputs, _net->numCases(), part.p,
                                      partBytes, _net->stream()), "softmax stage 2 + logreg grad");
        else
            chk(mcn_softmax_logreg_grad(_inputs[0], _acts, labels, _prev[0]->getActsGrad(), cost->coeff(), _outputs, _net->numCases(),
                                        _net->workspace(), _net->workspaceBytes(), _net->stream()), "softmax + logreg grad");
        _gradFused = true;
        return;
    }
    if (havePart)
        chk(mcn_softmax_from_part(_inputs[0], _acts, nullptr, nullptr, 0.f, _outputs, _net->numCases(), part.p, partBytes, _net->stream()),
            "softmax stage 2");
    else
        chk(mcn_softmax(_inputs[0], _acts, _outputs, _net->numCases(), _net->workspace(), _net->workspaceBytes(), _net->stream()), "softmax");
}

void SoftmaxLayer::bpropActs(float* v, int, float scaleTargets, PassType) {
    const bool doLogregGrad = _next.size() == 1 && _next[0]->type() == "cost.logreg";
    if (doLogregGrad) {
        if (_gradFused && scaleTargets == 0) return;  // written during forward
        float* labels = _next[0]->prev()[0]->getActs();
        const float coeff = static_cast<CostLayer*>(_next[0])->coeff();
        chk(mcn_logreg_softmax_grad(labels, _acts, _prev[0]->getActsGrad(), _outputs, _net->numCases(), scaleTargets == 1 ? 1 : 0, coeff,
                                    _net->stream()), "logreg-softmax grad");
    } else {
        chk(mcn_softmax_grad(_acts, v, _prev[0]->getActsGrad(), _outputs, _net->numCases(), scaleTargets == 1 ? 1 : 0, _net->stream()),
            "softmax grad");
    }
}

// ============================================================================ Eltwise layers (layer.cu:614-653)
EltwiseSumLayer::EltwiseSumLayer(ConvNet* net, const Params& p) : Layer(net, p) { _coeffs = pfloatv(p, "coeffs"); }

void EltwiseSumLayer::fpropActs(int i, float scaleTargets, PassType) {
    const size_t n = (size_t)_outputs * _net->numCases();
    if (scaleTargets == 0) chk(mcn_scale(_inputs[i], _acts, n, (float)_coeffs[i], _net->stream()), "eltsum");
    else chk(mcn_axpby(_acts, _inputs[i], n, 1.f, (float)_coeffs[i], _net->stream()), "eltsum");
}

void EltwiseSumLayer::bpropActs(float* v, int i, float scaleTargets, PassType) {
    const size_t n = (size_t)_outputs * _net->numCases();
    float* tgt = _prev[i]->getActsGrad();
    if (scaleTargets == 0) chk(mcn_scale(v, tgt, n, (float)_coeffs[i], _net->stream()), "eltsum grad");
    else chk(mcn_axpby(tgt, v, n, scaleTargets, (float)_coeffs[i], _net->stream()), "eltsum grad");
}

void EltwiseMaxLayer::fpropActs(int i, float, PassType) {
    const size_t n = (size_t)_outputs * _net->numCases();
    if (i == 1) chk(mcn_eltwise_max(_inputs[1], _inputs[0], _acts, n, _net->stream()), "eltmax");
    else if (i > 1) chk(mcn_eltwise_max(_acts, _inputs[i], _acts, n, _net->stream()), "eltmax");
}

void EltwiseMaxLayer::bpropActs(float* v, int i, float scaleTargets, PassType) {
    chk(mcn_eltwise_max_grad(v, _inputs[i], _acts, _prev[i]->getActsGrad(), (size_t)_outputs * _net->numCases(), scaleTargets != 0 ? 1 : 0,
                             _net->stream()), "eltmax grad");
}

// ============================================================================ DataLayer (layer.cu:660-679)
DataLayer::DataLayer(ConvNet* net, const Params& p) : Layer(net, p) { dataIdx = (int)pint(p, "dataIdx"); }

void DataLayer::fpropData(float* slice, PassType pt) {
    _acts = slice;
    for (Layer* l : _next) l->fpropFromPrev(pt);
}

// ============================================================================ PoolLayer (layer.cu:685-736)
PoolLayer::PoolLayer(ConvNet* net, const Params& p) : Layer(net, p) {
    _channels = (int)pint(p, "channels");
    _sizeX = (int)pint(p, "sizeX");
    _start = (int)pint(p, "start");
    _stride = (int)pint(p, "stride");
    _outputsX = (int)pint(p, "outputsX");
    _outputsY = (int)pint(p, "outputsY", _outputsX);
    const int imgSize = (int)pint(p, "imgSize");
    _imgSizeX = (int)pint(p, "imgSizeX", imgSize);
    _imgSizeY = (int)pint(p, "imgSizeY", imgSize);
    const std::string& pool = pstr(p, "pool");
    if (pool != "max" && pool != "avg") throw NetError("Unknown pooling layer type " + pool);  // layer.cu:706
    _max = pool == "max";
}

bool PoolLayer::quadFusable(int filters, int modulesY, int modulesX) const {
    return _max && _actsTarget < 0 && _channels == filters && _imgSizeY == modulesY && _imgSizeX == modulesX && _sizeX == 2 && _stride == 2 &&
           _start == 0 && modulesY % 2 == 0 && modulesX % 2 == 0 && _outputsY == modulesY / 2 && _outputsX == modulesX / 2;
}

void PoolLayer::fpropActs(int, float, PassType) {
    if (fwdDoneBelow) {  // the conv layer's forward launch wrote this pool's output
        fwdDoneBelow = false;
        return;
    }
    chk(mcn_local_pool(_inputs[0], _acts, _net->numCases(), _channels, _imgSizeY, _imgSizeX, _sizeX, _start, _stride, _outputsY, _outputsX,
                       _max ? 0 : 1, _net->stream()), "pool");
}

bool PoolLayer::undoFusable(int numCases, int channels, int outY, int outX) const {
    static const bool off = getenv("MCN_NO_UNDO_FUSION") != nullptr;
    return !off && _net->fusion() && _max && channels == _channels && outY == _outputsY && outX == _outputsX &&
           (fusedNeuronGrad || _prev[0]->numGradProducersNext() == 1) &&
           mcn_img_acts_max_undo_supported(numCases, _sizeX, _start, _stride, _outputsY, _outputsX, _imgSizeY, _imgSizeX,
                                           fusedNeuronGrad ? fusedNeuronGrad->spec.type : -1) == 1;
}

void PoolLayer::bpropActs(float* v, int, float scaleTargets, PassType) {
    if (undoDoneAbove) {  // the last consumer's backward-data product wrote the gradient of this pool's input itself
        undoDoneAbove = false;
        return;
    }
    float* tgt = _prev[0]->getActsGrad();
    const int n = _net->numCases();
    if (!_max) {
        chk(mcn_local_avg_undo(v, tgt, n, _channels, _imgSizeY, _imgSizeX, _sizeX, _start, _stride, _outputsY, _outputsX, scaleTargets, 1.f,
                               _net->stream()), "avg undo");
    } else if (fusedNeuronGrad && scaleTargets == 0) {
        chk(mcn_local_max_undo_neuron(_prev[0]->getActs(), v, _acts, tgt, n, _channels, _imgSizeY, _imgSizeX, _sizeX, _start, _stride,
                                      _outputsY, _outputsX, fusedNeuronGrad->spec.type, fusedNeuronGrad->spec.a, fusedNeuronGrad->spec.b,
                                      _net->stream()), "max undo + neuron grad");
    } else {
        if (fusedNeuronGrad) throw NetError("layer '" + _name + "': fused neuron gradient with accumulation");
        chk(mcn_local_max_undo(_prev[0]->getActs(), v, _acts, tgt, n, _channels, _imgSizeY, _imgSizeX, _sizeX, _start, _stride, _outputsY,
                               _outputsX, scaleTargets, 1.f, _net->stream()), "max undo");
    }
}

// ============================================================================ image-op layers (layer.cu:743-840, :1124-1143)
ImageOpLayer::ImageOpLayer(ConvNet* net, const Params& p) : Layer(net, p) {
    _channels = (int)pint(p, "channels");
    const int imgSize = (int)pint(p, "imgSize");
    _imgSizeX = (int)pint(p, "imgSizeX", imgSize);
    _imgSizeY = (int)pint(p, "imgSizeY", imgSize);
}

NailbedLayer::NailbedLayer(ConvNet* net, const Params& p) : ImageOpLayer(net, p) {
    _start = (int)pint(p, "start");
    _stride = (int)pint(p, "stride");
    _outputsX = (int)pint(p, "outputsX");
    _outputsY = (int)pint(p, "outputsY", _outputsX);
}
void NailbedLayer::fpropActs(int, float, PassType) {  // layer.cu:751-753
    chk(mcn_bed_of_nails(_inputs[0], _acts, _net->numCases(), _channels, _imgSizeY, _imgSizeX, _start, _stride, _outputsY, _outputsX, 0.f, 1.f,
                         _net->stream()), "bedOfNails");
}
void NailbedLayer::bpropActs(float* v, int, float scaleTargets, PassType) {  // layer.cu:755-757
    chk(mcn_bed_of_nails_undo(v, _prev[0]->getActsGrad(), _net->numCases(), _channels, _imgSizeY, _imgSizeX, _start, _stride, _outputsY,
                              _outputsX, scaleTargets, 1.f, _net->stream()), "bedOfNailsUndo");
}

GaussianBlurLayer::GaussianBlurLayer(ConvNet* net, const Params& p) : ImageOpLayer(net, p) {
    _hFilter = pmat(p, "filter");
    if (_hFilter.size() < 1 || _hFilter.size() % 2 == 0) throw NetError("layer '" + _name + "': the blur filter needs an odd number of taps");
}
void GaussianBlurLayer::allocate(int maxCases) {
    Layer::allocate(maxCases);
    _tmp.ensure((size_t)_outputs * maxCases);
}
void GaussianBlurLayer::copyToGPU() {  // layer.cu:782-784
    _filter.ensure(_hFilter.size());
    chk(mcn_h2d(_filter.p, _hFilter.data, _hFilter.size() * 4, _net->stream()), "h2d blur filter");
    chk(mcn_stream_sync(_net->stream()), "sync");
}
void GaussianBlurLayer::fpropActs(int, float, PassType) {  // layer.cu:769-772: rows, then columns
    const int n = _net->numCases(), fs = (int)_hFilter.size();
    chk(mcn_gaussian_blur(_inputs[0], _filter.p, _tmp.p, n, _channels, _imgSizeY, _imgSizeX, fs, 1, 0.f, 1.f, _net->stream()), "blur rows");
    chk(mcn_gaussian_blur(_tmp.p, _filter.p, _acts, n, _channels, _imgSizeY, _imgSizeX, fs, 0, 0.f, 1.f, _net->stream()), "blur columns");
}
void GaussianBlurLayer::bpropActs(float* v, int, float scaleTargets, PassType) {  // layer.cu:776-780: the filter is symmetric
    const int n = _net->numCases(), fs = (int)_hFilter.size();
    chk(mcn_gaussian_blur(v, _filter.p, _tmp.p, n, _channels, _imgSizeY, _imgSizeX, fs, 1, 0.f, 1.f, _net->stream()), "blur rows");
    chk(mcn_gaussian_blur(_tmp.p, _filter.p, _prev[0]->getActsGrad(), n, _channels, _imgSizeY, _imgSizeX, fs, 0, scaleTargets, 1.f,
                          _net->stream()), "blur columns");
}

ResizeLayer::ResizeLayer(ConvNet* net, const Params& p) : ImageOpLayer(net, p) {
    _tgtSize = (int)pint(p, "tgtSize");
    _scale = (float)pfloat(p, "scale");
    if (_imgSizeX != _imgSizeY) throw NetError("layer '" + _name + "': resize needs square images");
}
void ResizeLayer::fpropActs(int, float, PassType) {  // layer.cu:798-800
    chk(mcn_resize_bilinear(_inputs[0], _acts, _net->numCases(), _channels, _imgSizeX, _tgtSize, _scale, _net->stream()), "resizeBilinear");
}
void ResizeLayer::bpropActs(float*, int, float, PassType) {  // layer.cu:803-805 (assert(false))
    throw NetError("layer '" + _name + "': a resize layer cannot propagate gradients");
}

ColorLayer::ColorLayer(ConvNet* net, const Params& p, bool lab) : ImageOpLayer(net, p), _lab(lab) {
    if (lab) _center = (int)pint(p, "center", 0);
    if (_channels != 3) throw NetError("layer '" + _name + "': colour transforms need 3 channels");
}
void ColorLayer::fpropActs(int, float, PassType) {  // layer.cu:815-817, :833-835
    const int n = _net->numCases(), px = _imgSizeY * _imgSizeX;
    if (_lab) chk(mcn_rgb_to_lab(_inputs[0], _acts, n, px, _center, _net->stream()), "rgbToLAB");
    else chk(mcn_rgb_to_yuv(_inputs[0], _acts, n, px, _net->stream()), "rgbToYUV");
}
void ColorLayer::bpropActs(float*, int, float, PassType) {  // layer.cu:820-822, :838-840 (assert(false))
    throw NetError("layer '" + _name + "': a colour transform cannot propagate gradients");
}

ShiftLayer::ShiftLayer(ConvNet* net, const Params& p) : ImageOpLayer(net, p) {
    _hFilter = pmat(p, "filter");
    if (_hFilter.fortran) throw NetError("layer '" + _name + "': the shift table must be C-contiguous");
    const long px = (long)_imgSizeY * _imgSizeX;
    if (_hFilter.rows * _hFilter.cols * _channels != _outputs) throw NetError("layer '" + _name + "': shift table does not match 'outputs'");
    for (size_t k = 0; k < _hFilter.size(); k++)
        if (!(_hFilter.data[k] >= 0 && _hFilter.data[k] < px)) throw NetError("layer '" + _name + "': shift table points outside the image");
}
void ShiftLayer::allocate(int maxCases) { Layer::allocate(maxCases); }
void ShiftLayer::copyToGPU() {  // layer.cu:1141-1143
    _filter.ensure(_hFilter.size());
    chk(mcn_h2d(_filter.p, _hFilter.data, _hFilter.size() * 4, _net->stream()), "h2d shift table");
    chk(mcn_stream_sync(_net->stream()), "sync");
}
void ShiftLayer::fpropActs(int, float, PassType) {  // layer.cu:1129-1131 (on the host in the reference, conv_util.cu:2902-2990)
    chk(mcn_shift(_inputs[0], _filter.p, _acts, _net->numCases(), _channels, _imgSizeY * _imgSizeX, (int)_hFilter.rows, (int)_hFilter.cols,
                  _net->stream()), "shift");
}
void ShiftLayer::bpropActs(float*, int, float, PassType) {  // layer.cu:1135-1139 is empty: the layer below would see an unwritten matrix
    throw NetError("layer '" + _name + "': a shift layer cannot propagate gradients");
}

// ============================================================================ ResponseNormLayer (layer.cu:847-885)
ResponseNormLayer::ResponseNormLayer(ConvNet* net, const Params& p, bool cross) : Layer(net, p), _cross(cross) {
    _channels = (int)pint(p, "channels");
    _size = (int)pint(p, "size");
    _scale = (float)pfloat(p, "scale");
    _pow = (float)pfloat(p, "pow");
    _blocked = (int)pint(p, "blocked", 0);
    const int imgSize = (int)pint(p, "imgSize", 0);
    _imgSizeX = (int)pint(p, "imgSizeX", imgSize);
    _imgSizeY = (int)pint(p, "imgSizeY", imgSize);
    if ((long)_channels * _imgSizeX * _imgSizeY != _outputs) throw NetError("layer '" + _name + "': geometry does not match outputs");
}

void ResponseNormLayer::allocate(int maxCases) {
    Layer::allocate(maxCases);
    _denoms.ensure((size_t)_outputs * maxCases);
}

void ResponseNormLayer::fpropActs(int, float, PassType) {
    const int n = _net->numCases();
    if (_cross)
        chk(mcn_response_norm_cross_map(_inputs[0], _denoms.p, _acts, n, _channels, _imgSizeY * _imgSizeX, _size, _scale, _pow, _blocked,
                                        _net->stream()), "cmrnorm");
    else
        chk(mcn_response_norm(_inputs[0], _denoms.p, _acts, n, _channels, _imgSizeY, _imgSizeX, _size, _scale, _pow, _net->stream()), "rnorm");
}

void ResponseNormLayer::bpropActs(float* v, int, float scaleTargets, PassType) {  // destroys own acts like the reference
    _net->joinGradWork();  // a (deferred) weight-gradient kernel of the layer above still has to read those acts
    const int n = _net->numCases();
    if (_cross)
        chk(mcn_response_norm_cross_map_undo(v, _denoms.p, _prev[0]->getActs(), _acts, _prev[0]->getActsGrad(), n, _channels,
                                             _imgSizeY * _imgSizeX, _size, _scale, _pow, _blocked, scaleTargets, 1.f, _net->stream()),
            "cmrnorm undo");
    else
        chk(mcn_response_norm_undo(v, _denoms.p, _prev[0]->getActs(), _acts, _prev[0]->getActsGrad(), n, _channels, _imgSizeY, _imgSizeX,
                                   _size, _scale, _pow, scaleTargets, 1.f, _net->stream()), "rnorm undo");
}

// ============================================================================ ContrastNormLayer (layer.cu:893-913)
void ContrastNormLayer::allocate(int maxCases) {
    ResponseNormLayer::allocate(maxCases);
    _meanDiffs.ensure((size_t)_outputs * maxCases);
}

void ContrastNormLayer::fpropActs(int, float, PassType) {  // layer.cu:897-902
    const int n = _net->numCases();
    // local mean over the (clipped) size x size window around every pixel, then images - mean
    chk(mcn_local_pool(_inputs[0], _meanDiffs.p, n, _channels, _imgSizeY, _imgSizeX, _size, -_size / 2, 1, _imgSizeY, _imgSizeX, 1,
                       _net->stream()), "cnorm mean");
    chk(mcn_axpby(_meanDiffs.p, _inputs[0], (size_t)_outputs * n, -1.f, 1.f, _net->stream()), "cnorm mean diffs");
    chk(mcn_contrast_norm(_inputs[0], _meanDiffs.p, _denoms.p, _acts, n, _channels, _imgSizeY, _imgSizeX, _size, _scale, _pow,
                          _net->stream()), "cnorm");
}

void ContrastNormLayer::bpropActs(float* v, int, float scaleTargets, PassType) {  // layer.cu:904-906; destroys own acts
    _net->joinGradWork();
    chk(mcn_response_norm_undo(v, _denoms.p, _meanDiffs.p, _acts, _prev[0]->getActsGrad(), _net->numCases(), _channels, _imgSizeY,
                               _imgSizeX, _size, _scale, _pow, scaleTargets, 1.f, _net->stream()), "cnorm undo");
}

// ============================================================================ Cost layers (layer.cu:920-1116)
CostLayer::CostLayer(ConvNet* net, const Params& p) : Layer(net, p) { _coeff = (float)pfloat(p, "coeff"); }

void CostLayer::bprop(PassType pt) {
    if (_coeff != 0) Layer::bprop(pt);
}

void CostLayer::allocate(int maxCases) {
    _ownActs.ensure(std::max<size_t>(actsElems(maxCases), 4));
    _aux.ensure(std::max<size_t>(maxCases, 4));
}

size_t LogregCostLayer::workspaceBytes(int maxCases) { return mcn_logreg_cost_workspace_bytes(_prev[1]->outputs(), maxCases); }

void LogregCostLayer::fpropActs(int i, float, PassType) {  // layer.cu:969-981
    if (i != 0) return;
    if (costDoneBySoftmax) {  // the softmax below left log p[label], hit flags and tie counts (this pass): one small block finishes
        costDoneBySoftmax = false;
        const int nc = _net->numCases();
        _net->deferLateWork([this, nc] {
            chk(mcn_softmax_logreg_cost_finish(_acts, _aux.p, dCost, softmaxState, softmaxStateBytes, nc, _net->stream()), "logreg cost");
        });
        return;
    }
    // the cost value is off the critical path (nothing downstream reads it): side stream
    float *labels = _inputs[0], *probs = _inputs[1];
    const int n = _net->numCases();  // (captured now: in a test pass the closure runs while the NEXT minibatch is being set up)
    _net->deferLateWork([this, labels, probs, n] {
        chk(mcn_logreg_cost(labels, probs, _acts, _aux.p, _prev[1]->outputs(), n, dCost, _net->workspace(),
                            _net->workspaceBytes(), _net->stream()), "logreg cost");
    });
}

void LogregCostLayer::bpropActs(float*, int i, float scaleTargets, PassType) {  // layer.cu:983-994
    if (i != 1) throw NetError("logreg cost: gradient requested for the labels");
    const bool doWork = _prev[1]->next().size() > 1 || _prev[1]->type() != "softmax";
    if (doWork)
        chk(mcn_logreg_grad(_prev[0]->getActs(), _prev[1]->getActs(), _prev[1]->getActsGrad(), _prev[1]->outputs(), _net->numCases(),
                            scaleTargets == 1 ? 1 : 0, _coeff, _net->stream()), "logreg grad");
}

void SumOfSquaresCostLayer::fpropActs(int, float, PassType) {  // layer.cu:1004-1008
    const size_t n = (size_t)_prev[0]->outputs() * _net->numCases();
    chk(mcn_neuron_fwd(_inputs[0], _acts, n, MCN_NEURON_SQUARE, 0.f, 0.f, _net->stream()), "sum2 cost");
    chk(mcn_sum_all(_acts, n, dCost, 1, _net->stream()), "sum2 cost");
}

void SumOfSquaresCostLayer::bpropActs(float*, int i, float scaleTargets, PassType) {  // layer.cu:1010-1012
    const size_t n = (size_t)_prev[0]->outputs() * _net->numCases();
    float* tgt = _prev[i]->getActsGrad();
    if (scaleTargets == 0) chk(mcn_scale(_inputs[0], tgt, n, -2 * _coeff, _net->stream()), "sum2 grad");
    else chk(mcn_axpby(tgt, _inputs[0], n, scaleTargets, -2 * _coeff, _net->stream()), "sum2 grad");
}

L2SNCostLayer::L2SNCostLayer(ConvNet* net, const Params& p) : CostLayer(net, p) { _m = (float)pfloat(p, "m"); }  // layer.cu:1055-1058

L2SNCostLayer::~L2SNCostLayer() {
    if (_evReady) mcn_event_destroy(_evReady);
}

void L2SNCostLayer::allocate(int maxCases) {
    CostLayer::allocate(maxCases);
    _pairStats.ensure(std::max(maxCases, 4));
    _info.ensure(4);
    _gathered.ensure((size_t)std::max(maxCases, 4) * std::max(1, _net->worldSize()));
    if (!_evReady) chk(mcn_event_create_sync(&_evReady), "event");
}

// The pair statistics, the margin search and the cost run on the side stream as soon as the features exist, next to
// the rest of the forward pass; the backward pass waits for them and is then one elementwise launch.
void L2SNCostLayer::fpropActs(int i, float, PassType pt) {  // layer.cu:1060-1071
    if (i != 0) return;
    const int n = _net->numCases(), D = _prev[1]->outputs(), h = n / 2;
    // data parallel: the data-dependent margin must be the one a single GPU would pick on the global minibatch
    // (SURVEY 8e): gather every rank's pair distances and labels, search over all of them
    const bool gather = _net->dataParallel() && _m < 0;
    const int G = _net->worldSize();
    _extCount = gather ? h * G : 0;
    if (gather) _gathered.ensure((size_t)2 * h * G);
    float *idens = _inputs[0], *data = _inputs[1];
    // Replicas that own their GPU (stream gates usable): statistics, gather, margin search and cost on the SIDE stream, as on one
    // GPU -- the side stream is idle until fc6's gradients, the main stream's gate for the result needs no hand-shake between
    // threads, and the communicator is called by the side thread (never at the same time as another thread: gatherPairsHere).  (Through the exchange thread the main
    // thread stood still 57 us per step: 28 waiting for that thread to have recorded its event, 30 inside hipStreamWaitEvent.)
    static const bool sideGather = getenv("MCN_NO_SIDE_GATHER") == nullptr;
    const bool gatherOnSide = gather && sideGather && _net->gatesUsable() && pt == PASS_TRAIN;
    _readyOnExchange = gather && !gatherOnSide && _net->exchangeThreadUsable();
    if (_readyOnExchange) {
        const int cgSeq = _net->commGatesUsable() ? _net->nextCommGateSeq() : 0;
        _readyCommGateSeq = cgSeq;
        _readyTicket = _net->forkExchangeWork([this, idens, data, D, n, h, G, cgSeq] {
            void* st = _net->stream();  // the exchange stream: statistics, gather, margin search and cost in stream order
            chk(mcn_l2sn_pair_stats(idens, data, _pairStats.p, D, n, st), "l2sn pair stats");
            _net->gatherPairsHere(_pairStats.p, _gathered.p, _pairStats.p + h, _gathered.p + (size_t)h * G, h);
            chk(mcn_l2sn_cost_from_stats(_pairStats.p, _acts, n, _m, _info.p, dCost, _gathered.p, _gathered.p + _extCount, _extCount, st), "l2sn cost");
            if (cgSeq) _net->signalCommGate(cgSeq);
            else chk(mcn_event_record(_evReady, st), "event record");
        });
        return;
    }
    // Training pass on one GPU: the gradient (one elementwise launch from the kept statistics) follows the cost on the side
    // stream right away and is the FIRST gradient its input receives this pass; the backward chain then accumulates onto it
    // (a + b == b + a: same bits as the reference order, layer.cu:1073-1082 after fc6's) instead of stopping for it.
    float* early = nullptr;
    // Opt-in (MCN_EARLY_L2SN=1): measured 20 us per step SLOWER on DeepID2 -- the cost kernel starts ~16 us after the fork
    // (thread hand-over + cross-stream wait) and runs 11 us, so the first backward kernel, which then has to wait for the
    // seeded matrix, stalls; in the reference order the wait comes one kernel later and is covered.
    static const bool earlyOn = getenv("MCN_EARLY_L2SN") != nullptr;
    if (earlyOn && !gather && !_net->dataParallel() && pt == PASS_TRAIN && _coeff != 0 && _net->fusion() && _prev[1]->isGradConsumer() &&
        _prev[1]->rcvdB() == 0)
        early = _prev[1]->getActsGrad();
    const float coeff = _coeff;
    // (the seeded form hands _evReady to the layer below, and in a data-parallel run the side thread waits for the exchange
    // stream inside this closure: both keep the event)
    const int gateSeq = (!early && (!gather || gatherOnSide) && _net->gatesUsable()) ? _net->nextGateSeq() : 0;
    _readyGateSeq = gateSeq;
    _readyTicket = _net->forkSideWork([this, idens, data, D, n, h, G, gather, gatherOnSide, early, coeff, gateSeq] {
        if (gatherOnSide) {
            chk(mcn_l2sn_pair_stats(idens, data, _pairStats.p, D, n, _net->stream()), "l2sn pair stats");
            _net->gatherPairsHere(_pairStats.p, _gathered.p, _pairStats.p + h, _gathered.p + (size_t)h * G, h);
        } else if (gather) {
            chk(mcn_l2sn_pair_stats(idens, data, _pairStats.p, D, n, _net->stream()), "l2sn pair stats");
            _net->allGather2(_pairStats.p, _gathered.p, _pairStats.p + h, _gathered.p + (size_t)h * G, h);
            _net->waitGather();
        }
        // statistics on one workgroup per 8 pairs, then margin search and cost from them on one (round 3; one launch on one
        // workgroup before: 14 us for 160 features, 35 us for DeepID2+'s 512; MCN_L2SN_ONE_LAUNCH=1 for the A/B)
        static const bool oneLaunch = getenv("MCN_L2SN_ONE_LAUNCH") != nullptr;
        if (oneLaunch && !gather) {
            chk(mcn_l2sn_cost(idens, data, _acts, D, n, _m, _info.p, _pairStats.p, dCost, nullptr, nullptr, 0, _net->stream()), "l2sn cost");
        } else {
            if (!gather) chk(mcn_l2sn_pair_stats(idens, data, _pairStats.p, D, n, _net->stream()), "l2sn pair stats");
            chk(mcn_l2sn_cost_from_stats(_pairStats.p, _acts, n, _m, _info.p, dCost, gather ? _gathered.p : nullptr,
                                         gather ? _gathered.p + _extCount : nullptr, _extCount, _net->stream()), "l2sn cost");
        }
        if (early) chk(mcn_l2sn_grad_from_stats(data, early, D, n, _pairStats.p, _info.p, 0, coeff, _net->stream()), "l2sn grad (early)");
        if (gateSeq) _net->signalGate(gateSeq);
        else chk(timedEventRecord(_evReady, _net->stream()), "event record");
    });
    _earlyGrad = early != nullptr;
    if (early) _prev[1]->seedGrad(_evReady, _readyTicket);
}

void L2SNCostLayer::bprop(PassType pt) {
    if (!_earlyGrad) {
        CostLayer::bprop(pt);
        return;
    }
    _earlyGrad = false;  // delivered during forward; pass the turn on as Layer::bpropWith would (layer.cu:150-156)
    if (_prev[1]->isGradConsumer()) _prev[1]->bprop(pt);
}

bool L2SNCostLayer::foldable(Layer* target, Fold& f) {
    if (!_readyGateSeq || _earlyGrad || _readyOnExchange || _coeff == 0 || _prev.size() != 2 || _prev[1] != target) return false;
    f.data = _prev[1]->getActs();
    f.stats = _pairStats.p;
    f.info = _info.p;
    f.coeff = _coeff;
    f.gateSeq = _readyGateSeq;
    return true;
}

void L2SNCostLayer::bpropActs(float*, int i, float scaleTargets, PassType) {  // layer.cu:1073-1082
    if (i != 1) return;
    if (_gradFolded) {  // the classifier layer's backward-data launch added it (FCLayer::bpropActs)
        _gradFolded = false;
        return;
    }
    if (_readyGateSeq && t_stream == 0) {
        // a gate may be queued before its signal has been launched: no hand-shake with the side thread, no barrier packet
        _net->awaitGate(_readyGateSeq);
        _readyGateSeq = 0;
    } else if (_readyOnExchange && _readyCommGateSeq && t_stream == 0) {
        _net->awaitCommGate(_readyCommGateSeq);
        _readyCommGateSeq = 0;
    } else {
        if (_readyOnExchange) _net->waitExchangeSubmitted(_readyTicket);  // that thread has recorded _evReady
        else _net->waitSideSubmitted(_readyTicket);
        SiteTimer st("l2sn.bprop.streamWait");
        chk(timedStreamWait(_net->stream(), _evReady), "wait event");
    }
    if (_readyOnExchange) _net->exchangeWorkJoined();  // the main stream now follows everything that task issued
    chk(mcn_l2sn_grad_from_stats(_prev[1]->getActs(), _prev[1]->getActsGrad(), _prev[1]->outputs(), _net->numCases(), _pairStats.p,
                                 _info.p, scaleTargets == 1 ? 1 : 0, _coeff, _net->stream()), "l2sn grad");
}

Fisher2CostLayer::Fisher2CostLayer(ConvNet* net, const Params& p) : CostLayer(net, p) { _alpha = (float)pfloat(p, "alpha"); }  // layer.cu:1020-1023

size_t Fisher2CostLayer::workspaceBytes(int maxCases) { return mcn_fisher2_workspace_bytes(maxCases); }

void Fisher2CostLayer::fpropActs(int i, float, PassType) {  // layer.cu:1025-1036
    if (i != 0) return;
    if (_net->dataParallel()) throw NetError("cost.fisher2 couples every pair of the minibatch: not defined for data-parallel replicas");
    float *labels = _inputs[0], *data = _inputs[1];
    const int n = _net->numCases();
    _net->deferSideWork([this, labels, data, n] {  // the value is read by nothing downstream
        chk(mcn_fisher2_cost(labels, data, _prev[1]->outputs(), n, _alpha, dCost, _net->workspace(), _net->workspaceBytes(),
                             _net->stream()), "fisher2 cost");
    }, 0);
}

void Fisher2CostLayer::bpropActs(float*, int i, float scaleTargets, PassType) {  // layer.cu:1038-1047
    if (i != 1) return;
    chk(mcn_fisher2_grad(_prev[0]->getActs(), _prev[1]->getActs(), _prev[1]->getActsGrad(), _prev[1]->outputs(), _net->numCases(), _alpha,
                         scaleTargets == 1 ? 1 : 0, _coeff, _net->workspace(), _net->workspaceBytes(), _net->stream()), "fisher2 grad");
}

void L2regCostLayer::fpropActs(int i, float, PassType) {  // layer.cu:1093-1105
    if (i != 0) return;
    float *ground = _inputs[0], *pred = _inputs[1];
    const int n = _net->numCases();
    _net->deferSideWork([this, ground, pred, n] {
        chk(mcn_l2reg_cost(ground, pred, _acts, _prev[1]->outputs(), n, dCost, _net->stream()), "l2reg cost");
    }, 0);
}

void L2regCostLayer::bpropActs(float*, int i, float scaleTargets, PassType) {  // layer.cu:1107-1116
    if (i != 1) throw NetError("l2-reg cost: gradient requested for the targets");
    chk(mcn_l2reg_grad(_prev[0]->getActs(), _prev[1]->getActs(), _prev[1]->getActsGrad(), _prev[1]->outputs(), _net->numCases(),
                       scaleTargets == 1 ? 1 : 0, _coeff, _net->stream()), "l2reg grad");
}

// ============================================================================ Uploader

void Uploader::start(int deviceID) {
    // lowest priority, i.e. out of the normal level's pool of hardware queues: with a fifth normal stream the runtime doubles
    // up (rocprofv3 on MI355X / ROCm 7.2: the exchange stream's collectives ran on the main stream's queue, in line with its kernels)
    static const bool plain = getenv("MCN_COPY_STREAM_NORMAL") != nullptr;  // (A/B switch)
    chk(plain ? mcn_stream_create(&_stream) : mcn_stream_create_priority(&_stream, -1), "copy stream");
    for (int k = 0; k < 2; k++) {
        chk(mcn_host_alloc(&_chunk[k], kChunkBytes), "pinned staging chunk");
        chk(mcn_event_create_sync(&_chunkEv[k]), "event");
    }
    _thread.start(deviceID);
}

void Uploader::stop() { _thread.stop(); }

Uploader::~Uploader() {
    stop();
    for (int k = 0; k < 2; k++) {
        if (_chunkEv[k]) mcn_event_destroy(_chunkEv[k]);
        if (_chunk[k]) mcn_host_free(_chunk[k]);
    }
    if (_stream) mcn_stream_destroy(_stream);
}

void* Uploader::acquireChunk(int& k) {
    k = _next++ & 1;
    if (_chunkUsed[k]) chk(mcn_event_sync(_chunkEv[k]), "event sync");  // the DMA out of this chunk has finished
    _chunkUsed[k] = true;
    return _chunk[k];
}

void Uploader::copy(float* dst, const float* src, size_t bytes, bool pinned) {
    if (pinned) {
        chk(mcn_h2d(dst, src, bytes, _stream), "h2d (pinned)");
        return;
    }
    for (size_t off = 0; off < bytes; off += kChunkBytes) {
        const size_t n = std::min(kChunkBytes, bytes - off);
        int k;
        void* c = acquireChunk(k);
        memcpy(c, (const char*)src + off, n);
        chk(mcn_h2d((char*)dst + off, c, n, _stream), "h2d (staged)");
        chk(mcn_event_record(_chunkEv[k], _stream), "event record");
    }
}

void Uploader::copy2d(float* dst, size_t dpitch, const float* src, size_t spitch, size_t width, size_t height, bool pinned) {
    if (pinned) {
        chk(mcn_copy2d(dst, dpitch, src, spitch, width, height, 0, _stream), "h2d 2d (pinned)");
        return;
    }
    // gather the strided rows into a staging chunk, then one dense copy per chunk
    const size_t rowsPerChunk = std::max<size_t>(1, kChunkBytes / width);
    for (size_t r0 = 0; r0 < height; r0 += rowsPerChunk) {
        const size_t rows = std::min(rowsPerChunk, height - r0);
        int k;
        char* c = (char*)acquireChunk(k);
        for (size_t r = 0; r < rows; r++) memcpy(c + r * width, (const char*)src + (r0 + r) * spitch, width);
        if (dpitch == width) chk(mcn_h2d((char*)dst + r0 * dpitch, c, rows * width, _stream), "h2d (staged)");
        else chk(mcn_copy2d((char*)dst + r0 * dpitch, dpitch, c, width, width, rows, 0, _stream), "h2d 2d (staged)");
        chk(mcn_event_record(_chunkEv[k], _stream), "event record");
    }
}

// ============================================================================ DataProvider (data.cu)
DataProvider::~DataProvider() {
    for (void* e : {_evNext, _evSlotFree, _evSliceReady[0], _evSliceReady[1], _evSliceFree[0], _evSliceFree[1]})
        if (e) mcn_event_destroy(e);
}

// Spare-slot upload on the uploader thread / copy stream; the caller keeps the host arrays alive until waitPrefetchUploaded.
void DataProvider::prefetch(const CPUData& d) {
    Uploader& up = _net->uploader();
    if (!up.running()) throw NetError("prefetchBatch: the upload thread is not running");
    if (d.numCases() <= 0) throw NetError("data batch has no cases");
    for (const HostMat& m : d.mats)
        if (m.cols != d.numCases()) throw NetError("data matrices differ in the number of cases");
    // the spare slot may hold a batch whose adoption (startBatch([])) is queued but has not run yet: wait for the swap
    while (_adoptQueued.load()) std::this_thread::yield();
    if (_nextTicket.load()) up.wait(_nextTicket.load());  // a prefetch nobody asked to adopt is simply replaced
    _nextTicket = up.push([this, d] {
        Uploader& u = _net->uploader();
        if (!_evNext) chk(mcn_event_create_sync(&_evNext), "event");
        // the spare slot held the batch trained before: the copy stream starts behind everything the compute stream had
        // queued when that batch was swapped out (adoptPrefetched)
        if (_slotFreeRecorded.load()) chk(mcn_stream_wait_event(u.stream(), _evSlotFree), "wait event");
        while (_devNext.size() < d.mats.size()) {
            _devNext.emplace_back(new DBuf());
            _tiledNext.emplace_back(new DBuf());
        }
        _dimsNext.clear();
        _numCasesNext = d.numCases();
        for (size_t i = 0; i < d.mats.size(); i++) {
            const HostMat& m = d.mats[i];
            _dimsNext.push_back(m.rows);
            _devNext[i]->ensure(m.size());
            _tiledNext[i]->ensure(m.size());
            u.copy(_devNext[i]->p, m.data, m.size() * 4, m.pinned);
            chk(mcn_tile_minibatches(_devNext[i]->p, _tiledNext[i]->p, (int)m.rows, (size_t)_numCasesNext, _minibatchSize, u.stream()),
                "tile minibatches");
        }
        chk(mcn_event_record(_evNext, u.stream()), "event record");
    });
}

void DataProvider::waitPrefetchUploaded() {
    const long t = _nextTicket.load();
    if (!t) return;
    _net->uploader().wait(t);
    if (_evNext) chk(mcn_event_sync(_evNext), "event sync");
}

// compute thread: the prefetched batch becomes the current one (the old current slot becomes the spare)
void DataProvider::adoptPrefetched() {
    if (!_nextTicket.load()) { _adoptQueued.store(false); throw NetError("no prefetched data batch"); }
    _net->uploader().wait(_nextTicket.load());  // its copies and the event record have been issued
    chk(mcn_stream_wait_event(_net->mainStream(), _evNext), "wait event");
    while (_dev.size() < _devNext.size()) {
        _dev.emplace_back(new DBuf());
        _mini.emplace_back(new DBuf());
        _tiled.emplace_back(new DBuf());
    }
    while (_devNext.size() < _dev.size()) {
        _devNext.emplace_back(new DBuf());
        _tiledNext.emplace_back(new DBuf());
    }
    std::swap(_dev, _devNext);
    std::swap(_tiled, _tiledNext);
    _dims = _dimsNext;
    _numCases = _numCasesNext;
    _onDevice = true;
    _host = CPUData();
    _nextTicket.store(0);
    if (!_evSlotFree) chk(mcn_event_create_sync(&_evSlotFree), "event");
    chk(mcn_event_record(_evSlotFree, _net->mainStream()), "event record");  // (the worker joined the side stream before it got here)
    _slotFreeRecorded.store(true);
    _adoptQueued.store(false);  // the spare slot (the batch trained before) may be overwritten from here on -- behind that event
}

// ---- streamed minibatches: the batch stays on the host (reference: batches above 200 MB, util.cuh:53 / data.cu:72-98);
// slice i+2 is copied on the copy stream while steps i and i+1 run.
void DataProvider::requestSlice(int mi) {
    const int b = mi & 1;
    Uploader& up = _net->uploader();
    const long start = (long)mi * _minibatchSize, end = std::min(start + _minibatchSize, _numCases);
    if (start >= _numCases) { _sliceTicket[b] = 0; return; }
    const CPUData host = _host;
    void* evFree = _sliceFreeRecorded[b] ? _evSliceFree[b] : nullptr;
    _sliceTicket[b] = up.push([this, host, b, start, end, evFree] {
        Uploader& u = _net->uploader();
        if (evFree) chk(mcn_stream_wait_event(u.stream(), evFree), "wait event");  // the step that read this buffer is done
        const long n = end - start;
        for (size_t i = 0; i < host.mats.size(); i++) {
            float* dst = _stream2[b][i]->p;
            u.copy2d(dst, (size_t)n * 4, host.mats[i].data + start, (size_t)_numCases * 4, (size_t)n * 4, (size_t)_dims[i], host.mats[i].pinned);
        }
        chk(mcn_event_record(_evSliceReady[b], u.stream()), "event record");
    });
}

void DataProvider::streamBegin() {
    if (!_net->uploader().running()) throw NetError("streamed minibatches need the upload thread");
    for (int b = 0; b < 2; b++) {
        if (!_evSliceReady[b]) chk(mcn_event_create_sync(&_evSliceReady[b]), "event");
        if (!_evSliceFree[b]) chk(mcn_event_create_sync(&_evSliceFree[b]), "event");
        while (_stream2[b].size() < _dims.size()) _stream2[b].emplace_back(new DBuf());
        for (size_t i = 0; i < _dims.size(); i++) _stream2[b][i]->ensure((size_t)_dims[i] * _minibatchSize);
    }
    // the compute stream may still read the buffers (previous batch): order the first refills behind it
    for (int b = 0; b < 2; b++) {
        chk(mcn_event_record(_evSliceFree[b], _net->mainStream()), "event record");
        _sliceFreeRecorded[b] = true;
    }
    requestSlice(0);
    requestSlice(1);
}

void DataProvider::streamSlice(int mi, std::vector<float*>& out) {
    const int b = mi & 1;
    if (!_sliceTicket[b]) throw NetError("streamed minibatch was not requested");
    _net->uploader().wait(_sliceTicket[b]);
    chk(mcn_stream_wait_event(_net->mainStream(), _evSliceReady[b]), "wait event");
    out.clear();
    for (size_t i = 0; i < _dims.size(); i++) out.push_back(_stream2[b][i]->p);
}

void DataProvider::streamSliceDone(int mi) {
    const int b = mi & 1;
    chk(mcn_event_record(_evSliceFree[b], _net->mainStream()), "event record");
    _sliceFreeRecorded[b] = true;
    requestSlice(mi + 2);
}

void DataProvider::setData(const CPUData& d) {
    _host = d;
    _numCases = d.numCases();
    if (_numCases <= 0) throw NetError("data batch has no cases");
    size_t bytes = 0;
    _dims.clear();
    for (const HostMat& m : d.mats) {
        if (m.cols != _numCases) throw NetError("data matrices differ in the number of cases");
        bytes += m.size() * 4;
        _dims.push_back(m.rows);
    }
    while (_dev.size() < d.mats.size()) {
        _dev.emplace_back(new DBuf());
        _mini.emplace_back(new DBuf());
        _tiled.emplace_back(new DBuf());
    }
    // The reference keeps a batch on the device below 200 MB (util.cuh:53) and otherwise uploads a
    // host-side column slice per minibatch.  With 288 GB of HBM the whole batch is resident by default.
    const char* env = getenv("MCN_MAX_DATA_ON_GPU_MB");
    const size_t limitMB = env ? (size_t)atol(env) : 16384;
    _onDevice = bytes / (1024 * 1024) < limitMB;
    if (_onDevice) {
        for (size_t i = 0; i < d.mats.size(); i++) {
            _dev[i]->ensure(d.mats[i].size());
            // (page-locked sources go out asynchronously; pageable ones are staged by the runtime inside this call)
            chk(mcn_h2d(_dev[i]->p, d.mats[i].data, d.mats[i].size() * 4, _net->stream()), "h2d data batch");
            // every minibatch slice laid out densely once per batch: the per-step slice is then a pointer
            _tiled[i]->ensure(d.mats[i].size());
            chk(mcn_tile_minibatches(_dev[i]->p, _tiled[i]->p, (int)d.mats[i].rows, (size_t)_numCases, _minibatchSize, _net->stream()),
                "tile minibatches");
        }
        chk(mcn_stream_sync(_net->stream()), "sync");
    }
}

bool DataProvider::sliceIsPointer(long start, long end) const {
    end = std::min(end, _numCases);
    if (!_onDevice || end <= start) return false;
    if (end - start == _numCases) return true;
    for (size_t i = 0; i < _dims.size(); i++)
        if (!(start % _minibatchSize == 0 && end == std::min(start + _minibatchSize, _numCases) && ((size_t)start * _dims[i]) % 4 == 0)) return false;
    return true;
}

void DataProvider::getSlice(long start, long end, std::vector<float*>& out) {
    end = std::min(end, _numCases);
    const long n = end - start;
    if (n <= 0) throw NetError("empty data slice");
    out.clear();
    for (size_t i = 0; i < _dims.size(); i++) {
        if (_onDevice && n == _numCases) {
            out.push_back(_dev[i]->p);
            continue;
        }
        if (_onDevice && start % _minibatchSize == 0 && end == std::min(start + _minibatchSize, _numCases) &&
            ((size_t)start * _dims[i]) % 4 == 0) {  // (16-byte aligned like every other activation buffer)
            out.push_back(_tiled[i]->p + (size_t)start * _dims[i]);
            continue;
        }
        _mini[i]->ensure((size_t)_dims[i] * _minibatchSize);
        if (_onDevice) {
            chk(mcn_copy2d(_mini[i]->p, n * 4, _dev[i]->p + start, _numCases * 4, n * 4, _dims[i], 2, _net->stream()), "slice data");
        } else {
            if (_host.mats.size() != _dims.size()) throw NetError("host data batch is gone");
            chk(mcn_copy2d(_mini[i]->p, n * 4, _host.mats[i].data + start, _numCases * 4, n * 4, _dims[i], 0, _net->stream()), "upload slice");
            chk(mcn_stream_sync(_net->stream()), "sync");
        }
        out.push_back(_mini[i]->p);
    }
}

double Cost::value() const {
    double v = 0;
    for (auto& kv : values) v += coeffs.at(kv.first) * kv.second.at(0);
    return v;
}

// ============================================================================ ConvNet (convnet.cu)
ConvNet::ConvNet(const std::vector<Params>& layers, int minibatchSize, int deviceID) : _minibatchSize(minibatchSize), _deviceID(deviceID) {
    if (minibatchSize <= 0) throw NetError("minibatch size must be positive");
    _fusion = getenv("MCN_NO_FUSION") == nullptr;  // debugging aid: run every layer as its own kernel(s)
    _earlyOff = getenv("MCN_NO_EARLY_SGD") != nullptr;  // (per model, like every switch a test toggles inside one process)
    _earlyMinElems = getenv("MCN_EARLY_SGD_MIN_ELEMS") ? (size_t)atol(getenv("MCN_EARLY_SGD_MIN_ELEMS")) : (size_t)8 << 20;
    for (const Params& p : layers) {  // convnet.cu:86-134
        const std::string& t = pstr(p, "type");
        Layer* l;
        if (t == "fc") l = new FCLayer(this, p);
        else if (t == "conv") l = new LocalLayer(this, p, true, true);
        else if (t == "local") l = new LocalLayer(this, p, false, false);
        else if (t == "pool") l = new PoolLayer(this, p);
        else if (t == "rnorm") l = new ResponseNormLayer(this, p, false);
        else if (t == "cmrnorm") l = new ResponseNormLayer(this, p, true);
        else if (t == "cnorm") l = new ContrastNormLayer(this, p);
        else if (t == "softmax") l = new SoftmaxLayer(this, p);
        else if (t == "eltsum") l = new EltwiseSumLayer(this, p);
        else if (t == "eltmax") l = new EltwiseMaxLayer(this, p);
        else if (t == "neuron") l = new NeuronLayer(this, p);
        else if (t == "nailbed") l = new NailbedLayer(this, p);
        else if (t == "blur") l = new GaussianBlurLayer(this, p);
        else if (t == "resize") l = new ResizeLayer(this, p);
        else if (t == "rgb2yuv") l = new ColorLayer(this, p, false);
        else if (t == "rgb2lab") l = new ColorLayer(this, p, true);
        else if (t == "shift") l = new ShiftLayer(this, p);
        else if (t == "data") { auto* d = new DataLayer(this, p); _dataLayers.push_back(d); l = d; }
        else if (t == "cost.logreg") { auto* c = new LogregCostLayer(this, p); _costs.push_back(c); l = c; }
        else if (t == "cost.sum2") { auto* c = new SumOfSquaresCostLayer(this, p); _costs.push_back(c); l = c; }
        else if (t == "cost.l2-sn") { auto* c = new L2SNCostLayer(this, p); _costs.push_back(c); l = c; }
        else if (t == "cost.fisher2") { auto* c = new Fisher2CostLayer(this, p); _costs.push_back(c); l = c; }
        else if (t == "cost.l2-reg") { auto* c = new L2regCostLayer(this, p); _costs.push_back(c); l = c; }
        else throw NetError("Unknown layer type " + t);
        _layers.emplace_back(l);
        if (phas(p, "inputs"))
            for (long i : pintv(p, "inputs")) {
                if (i < 0 || i >= (long)_layers.size() - 1) throw NetError("layer '" + l->name() + "': bad input index");
                l->addPrev(_layers[i].get());
            }
    }
    for (auto& l : _layers)
        for (Layer* p : l->prev()) p->addNext(l.get());
    _dp.reset(new DataProvider(this, minibatchSize));
}

void ConvNet::dropStepGraphs() {
    for (auto& kv : _stepGraphs) mcn_graph_destroy(kv.second.exec);
    _stepGraphs.clear();
}

void ConvNet::prefetchFromHost(const CPUData& d) { _dp->prefetch(d); }

ConvNet::~ConvNet() {
    stop();
    _uploader.stop();
    dropStepGraphs();
    _side.stop();
    _commSub.stop();
    printWaitSites();
    if (getenv("MCN_DEBUG_TIMES") && _rcclCalls) fprintf(stderr, "[mcn] all-reduce calls: %ld, host us per call: %.1f\n", (long)_rcclCalls, _rcclCallNs / 1e3 / _rcclCalls);
    if (_comm) mcn_comm_destroy(_comm);
    if (_commStream) mcn_stream_destroy(_commStream);
    if (_evGrad) mcn_event_destroy(_evGrad);
    if (_evGrad2) mcn_event_destroy(_evGrad2);
    if (_evGather) mcn_event_destroy(_evGather);
    if (_evJoinComm) mcn_event_destroy(_evJoinComm);
    if (_evComm) mcn_event_destroy(_evComm);
    for (void* e : _bucketEvents) mcn_event_destroy(e);
    if (_gradStream) mcn_stream_destroy(_gradStream);
    for (void* e : _forkEvents) mcn_event_destroy(e);
    if (_gate) mcn_gate_free(_gate);
    if (_forkGate) mcn_gate_free(_forkGate);
    for (int* g : {_commGate, _mainToComm, _sideToComm})
        if (g) mcn_gate_free(g);
    if (_tickFlag) mcn_tick_free(_tickFlag);
    if (_tickFlagSide) mcn_tick_free(_tickFlagSide);
    if (_evJoin) mcn_event_destroy(_evJoin);
    if (_stream) mcn_stream_destroy(_stream);
    for (void* e : _eventPool) mcn_event_destroy(e);
    for (void* e : _commEventPool) mcn_event_destroy(e);
}

// Fused epilogues, decided once from the graph (no counterpart in the reference, which launches
// bias add, neuron and pool-undo / neuron-gradient as separate kernels: layer.cu:249, :453-461, :734).
void ConvNet::buildFusions() {
    if (!_fusion) {  // (MCN_NO_FUSION: the per-input tables of the FC layers still have to exist)
        for (auto& up : _layers)
            if (auto* fc = dynamic_cast<FCLayer*>(up.get())) {
                fc->planMulti();
                fc->multiShapes = false;
            }
        return;
    }
    for (auto& up : _layers) {
        // forward: conv/local/fc with ONE input whose only consumer is a neuron layer sharing its activity matrix
        if (auto* w = dynamic_cast<WeightLayer*>(up.get())) {
            if (w->prev().size() == 1 && w->next().size() == 1) {
                auto* r = dynamic_cast<NeuronLayer*>(w->next()[0]);
                if (r && r->actsTarget() == 0 && r->prev().size() == 1) {
                    w->fusedNeuron = r;
                    r->fwdFused = true;
                }
            }
        }
        // conv layer (one input) whose output -- through the fused neuron, if any -- feeds a 2 x 2 / stride 2 max pool that tiles
        // the maps: the pool's output comes out of the forward launch too (MCN_NO_POOL_FUSION=1: as separate launches)
        if (auto* cl = dynamic_cast<LocalLayer*>(up.get())) {
            const bool off = getenv("MCN_NO_POOL_FUSION") != nullptr;  // (read per model: tests toggle it in one process)
            Layer* out = cl->fusedNeuron ? static_cast<Layer*>(cl->fusedNeuron) : static_cast<Layer*>(cl);
            if (!off && cl->isConv() && cl->prev().size() == 1)
                for (Layer* l : out->next()) {
                    auto* pl = dynamic_cast<PoolLayer*>(l);
                    if (pl && pl->prev().size() == 1 && pl->quadFusable(cl->filters(), cl->modulesY(), cl->modulesX())) {
                        cl->fusedPool = pl;
                        break;
                    }
                }
        }
        // FC layer with several inputs: all of them in one launch per direction; the forward neuron and, per input, the
        // gradient of a neuron layer feeding only this layer go into the epilogues
        if (auto* fc = dynamic_cast<FCLayer*>(up.get())) {
            fc->planMulti();
            if (fc->multiShapes && fc->next().size() == 1 && !fc->fusedNeuron) {
                auto* r = dynamic_cast<NeuronLayer*>(fc->next()[0]);
                if (r && r->actsTarget() == 0 && r->prev().size() == 1) {
                    fc->fusedNeuron = r;
                    r->fwdFused = true;
                }
            }
            if (fc->multiShapes)
                for (size_t k = 0; k < fc->prev().size(); k++) {
                    auto* r = dynamic_cast<NeuronLayer*>(fc->prev()[k]);
                    if (r && !r->bwdFused && r->actsGradTarget() == 0 && r->numGradProducersNext() == 1 && r->next().size() == 1 &&
                        r->prev().size() == 1 && r->spec.fusableGrad() && r->isGradConsumer()) {
                        fc->neuronAbove[k] = r;
                        r->bwdFused = true;
                    }
                }
        }
        // classifier layer directly under a softmax: stage 1 of the softmax in the epilogue of the forward product
        // (possibly through the neuron layer fused into that product above: `neuron=ident` in the DeepID cfgs)
        if (auto* fc = dynamic_cast<FCLayer*>(up.get())) {
            if (fc->prev().size() == 1 && fc->next().size() == 1) {
                Layer* below = fc->fusedNeuron ? (fc->fusedNeuron->next().size() == 1 ? fc->fusedNeuron->next()[0] : nullptr) : fc->next()[0];
                if (auto* sm = dynamic_cast<SoftmaxLayer*>(below))
                    if (sm->prev().size() == 1) fc->softmaxBelow = sm;
            }
        }
        // backward: max pool directly above a neuron layer that (a) gets gradient from nobody else,
        // (b) shares its gradient matrix with its producer and (c) has a gradient that needs only its output
        if (auto* pl = dynamic_cast<PoolLayer*>(up.get())) {
            auto* r = dynamic_cast<NeuronLayer*>(pl->prev()[0]);
            if (pl->isMax() && r && r->actsGradTarget() == 0 && r->numGradProducersNext() == 1 && r->next().size() == 1 &&
                r->spec.fusableGrad() && r->isGradConsumer()) {
                pl->fusedNeuronGrad = r;
                r->bwdFused = true;
            }
        }
    }
}

void ConvNet::start() { _thread = std::thread([this] { run(); }); }

void ConvNet::stop() {
    if (_thread.joinable()) {
        _work.enqueue([](ConvNet& n) { n._stop = true; });
        _thread.join();
    }
}

void ConvNet::run() {  // convnet.cu:147-156
    try {
        initDevice();
        WorkResult r;
        r.type = WorkResult::INIT_DONE;
        _results.enqueue(std::move(r));
    } catch (std::exception& e) {
        WorkResult r;
        r.type = WorkResult::ERROR;
        r.error = e.what();
        _results.enqueue(std::move(r));
        return;
    }
    while (!_stop) {
        Work w = _work.dequeue();
        try {
            w(*this);
        } catch (std::exception& e) {
            WorkResult r;
            r.type = WorkResult::ERROR;
            r.error = e.what();
            try {
                resetAfterError();
            } catch (std::exception& e2) {
                r.error += std::string(" (and while cleaning up: ") + e2.what() + ")";
            }
            _results.enqueue(std::move(r));
        }
    }
}

// A work item threw in the middle of a pass: nothing of that pass may leak into the next one.  The submitting threads are
// drained first (their closures hold pointers and the pass type of the failed pass), then every list, flag and counter of
// the step schedule goes back to "between steps" and the device is drained.  The model is usable again afterwards -- except
// a data-parallel replica: it has skipped collectives its peers are blocked in, so it refuses further work and the
// launcher has to tear the group down.
void ConvNet::resetAfterError() {
    mcn_flush_tick();  // (a fork whose tick no launch has carried yet: the side thread is waiting for it)
    auto drain = [](Submitter& s) {
        if (!s.running()) return;
        try {
            s.wait(s.pushed());
        } catch (...) {  // (the error that is being reported, or one that followed from it)
        }
    };
    drain(_side);
    drain(_commSub);
    // a gate the failed pass queued on the main stream may wait for a signal that will never be sent (the closure that
    // carried it was dropped): release it from the side stream, which is idle now
    if (_gate && _gateOk) mcn_stream_signal(_gate, _gateSeq, _gradStream);
    if (_commGateOk) {  // (best effort; a gate nobody can release any more ends by its own timeout)
        mcn_stream_signal(_sideToComm, _sideToCommSeq, _gradStream);
        mcn_stream_signal(_mainToComm, _mainToCommSeq, _gradStream);
        mcn_stream_signal(_commGate, _commGateSeq, _gradStream);
    }
    _deferred.clear();
    _deferredUnits = 0;
    _late.clear();
    _gradPending = false;
    _exchangeWorkPending = false;
    _earlyUpdateIssued = false;
    _earlyPos = 0;
    _pendingMainGate = 0;
    _pendingWeightGate = 0;
    _fusedTailNow = false;
    _sideWaitPending = false;
    _passNow.clear();
    _wlSeen = 0;
    _biasBatch[0].clear();
    _biasBatch[1].clear();
    {
        std::lock_guard<std::mutex> lk(_commMutex);
        _commPending = false;
        _sentUpTo = 0;
        for (Segment& sg : _segments) { sg.ready = sg.byMain = false; sg.reported = 0; }
    }
    if (_capturing) {  // a capture must be ended before the stream can be used again
        void* dead = nullptr;
        mcn_graph_end_capture(_stream, &dead, nullptr);
        if (dead) mcn_graph_destroy(dead);
    }
    _capturing = false;
    g_noHostWait = false;
    _profiling = _phases = false;
    t_stream = 0;
    for (auto& l : _layers) {
        l->reset();
        l->resetPassFlags();
        l->resetUpdateCounters();  // numUpdates = 0: the next pass starts its gradients afresh (what the failed pass added to a
                              // momentum matrix stays there: the caller decides whether to reload a checkpoint)
    }
    // the failed pass's softmax has added its per-column tie counts and the launch that clears them (queued in _late) is
    // gone: without this the next step's correctProbs would be 1 / (2 * ties)
    for (auto& l : _layers)
        if (auto* sm = dynamic_cast<SoftmaxLayer*>(l.get()))
            if (sm->costState.p) mcn_memset(sm->costState.p, 0, sm->costStateBytes, _stream);
    if (_dpOn) _failed = true;
    mcn_device_sync();  // (a sticky device error shows up in the next call's status)
    if (_gate && _gateOk) {  // a gate that did run into its timeout has been reported with this error: start clean
        mcn_memset(_gate + 1, 0, sizeof(int), _stream);
        for (int* g : {_forkGate, _commGate, _mainToComm, _sideToComm})
            if (g) mcn_memset(g + 1, 0, sizeof(int), _stream);
        mcn_stream_sync(_stream);
    }
}

void ConvNet::initDevice() {  // convnet.cu:139-145
    chk(mcn_set_device(_deviceID < 0 ? 0 : _deviceID), "set device");
    chk(mcn_stream_create(&_stream), "stream");
    if (const char* e = getenv("MCN_SIDE_UNITS")) _sideUnits = std::max(1, atoi(e));
    // opt-in: on this runtime (ROCm 7.2) a replayed step takes longer on the device than the eagerly submitted one
    // (0.400 vs 0.356 ms, DeepID2) although it needs half the host time -- DESIGN.md section 5
    _useGraphs = getenv("MCN_STEP_GRAPH") != nullptr;
    if (getenv("MCN_NO_GRAD_STREAM") == nullptr) {
        chk(mcn_stream_create(&_gradStream), "grad stream");
        chk(mcn_event_create_sync(&_evJoin), "event");
        _forkEvents.resize(32, nullptr);
        for (void*& e : _forkEvents) chk(mcn_event_create_sync(&e), "event");
        // the side thread polls: forks are stream ticks (MCN_NO_TICKS=1: events, as before)
        if (((g_hostWaitMask >> 1) & 1) && getenv("MCN_NO_TICKS") == nullptr) {
            chk(mcn_tick_alloc(&_tickFlag), "tick flag");
            chk(mcn_tick_alloc(&_tickFlagSide), "tick flag");
            // the exchange thread too (MCN_COMM_TICKS=1).  MEASURED, forced data-parallel step with one rank: 0.328 ms with,
            // 0.325 ms without -- its waits are few and the collectives behind them long; opt-in
            _commTicks = getenv("MCN_COMM_TICKS") != nullptr;
        }
        _side.start(_deviceID);
        if (getenv("MCN_NO_GATES") == nullptr) {
            chk(mcn_gate_alloc(&_gate), "gate");
            int ok = 0;
            chk(mcn_gate_selftest(_gate, _stream, _gradStream, &ok), "gate self-test");
            _gateOk = ok != 0;  // (two streams on one hardware queue: a gate would only ever time out -- events then)
            if (_gateOk && _tickFlag && getenv("MCN_NO_FORK_GATES") == nullptr) {
                chk(mcn_gate_alloc(&_forkGate), "gate");
                chk(mcn_gate_selftest(_forkGate, _gradStream, _stream, &ok), "gate self-test");
                _forkGateOk = ok != 0;
            }
            if (getenv("MCN_DEBUG_TIMES")) fprintf(stderr, "[mcn] stream gates: %s\n", _gateOk ? "on" : "off (streams share a hardware queue)");
        }
    }
    // this thread's backward-data products share the GPU with the side stream's weight gradients: the form that leaves
    // them wave slots (mcn_set_thread_option, measured on the DeepID2+ step: 0.817 vs 0.905 ms).  The same form with one
    // stream (MCN_NO_GRAD_STREAM), so that the two schedules stay bit-identical.
    chk(mcn_set_thread_option("img_acts_wide", 0), "thread option");
    _uploader.start(_deviceID);
    for (auto& l : _layers) l->copyToGPU();
    buildFusions();
    allocate(_minibatchSize);
}

void ConvNet::allocate(int maxCases) {
    for (auto& l : _layers) l->allocate(maxCases);
    size_t ws = 256;
    for (auto& l : _layers) ws = std::max(ws, l->workspaceBytes(maxCases));
    _ws.ensure((ws + 3) / 4);
    if (_gradStream) _ws2.ensure((ws + 3) / 4);
    // gradient arena in reverse layer order == the order in which backward completes them, so that
    // data-parallel buckets are contiguous ranges
    const bool dp = dataParallel();
    size_t total = 0;
    _segments.clear();
    std::vector<std::pair<Weights*, size_t>> place;
    for (int i = (int)_layers.size() - 1; i >= 0; i--) {
        auto* wl = dynamic_cast<WeightLayer*>(_layers[i].get());
        if (!wl) continue;
        std::vector<Weights*> ws_;
        wl->collectWeights(ws_);
        const size_t begin = total;
        for (Weights* w : ws_) {
            if (!w->owner() || !(w->useGrad() || dp)) continue;
            place.emplace_back(w, total);
            total += (w->size() + 3) & ~(size_t)3;
        }
        if (total > begin) _segments.push_back({wl, begin, total, false});
    }
    // a segment is complete when its layer AND every layer sharing one of its matrices have written their gradients
    for (Segment& sg : _segments) {
        sg.contributors.assign(1, sg.layer);
        for (auto& l : _layers) {
            auto* other = dynamic_cast<WeightLayer*>(l.get());
            if (!other || other == sg.layer) continue;
            bool shares = false;
            for (int k = 0; k < other->numWeights() && !shares; k++) {
                Weights& w = other->weights(k);
                if (w.owner()) continue;
                for (int j = 0; j < sg.layer->numWeights(); j++) shares = shares || w.W() == sg.layer->weights(j).W();
            }
            if (shares) sg.contributors.push_back(other);
        }
    }
    _gradArena.ensure(std::max<size_t>(total, 4));
    chk(mcn_memset(_gradArena.p, 0, _gradArena.cap * 4, _stream), "memset grads");
    for (auto& pw : place) pw.first->setGradBuf(_gradArena.p + pw.second);
    // one-launch update plan over every matrix that is updated
    std::vector<::mcn_sgd_item> items;
    for (auto& l : _layers) {
        std::vector<Weights*> ws_;
        l->collectWeights(ws_);
        for (Weights* w : ws_) {
            ::mcn_sgd_item it;
            if (w->planItem(dp, it)) items.push_back(it);
        }
    }
    _sgdCount = (int)items.size();
    _splitPlans = false;
    if (_sgdCount > 0) {
        const size_t pb = mcn_sgd_plan_bytes(_sgdCount);
        _sgdPlan.ensure((pb + 3) / 4);
        chk(mcn_sgd_plan_build(items.data(), _sgdCount, _sgdPlan.p, pb, &_sgdBlocks, _stream), "sgd plan");
    }
    // ... and the same plan with the latency-only tail launches of a step folded in (fusedTailNow)
    _fusedTailOk = false;
    const bool noFusedTail = getenv("MCN_NO_FUSED_TAIL") != nullptr;  // (read per model: tests toggle it in one process)
    if (!dp && _fusion && !_hasSharing && !noFusedTail && biasBatching() && _sgdCount > 0) {
        std::vector<::mcn_sgd_item> fi;
        std::vector<::mcn_sgd_partials> fp;
        bool any = false, expressible = true;
        for (auto& l : _layers) expressible = l->planFusedTail(maxCases) && expressible;
        for (auto& l : _layers) {
            std::vector<Weights*> ws_;
            l->collectWeights(ws_);
            for (Weights* w : ws_) {
                ::mcn_sgd_item it;
                if (!w->planItem(dp, it)) continue;
                ::mcn_sgd_partials q = {nullptr, 0, 0, 1.f};
                if (w->partials.data && it.useGrad) {
                    q.data = w->partials.data; q.count = w->partials.count; q.kind = w->partials.kind; q.scale = w->partials.scale;
                    any = true;
                } else {
                    w->partials = Weights::Partials();
                }
                fi.push_back(it);
                fp.push_back(q);
            }
        }
        if (any && expressible && (int)fi.size() == _sgdCount) {
            const size_t pb = mcn_sgd_plan_bytes(_sgdCount);
            _sgdFused.ensure((pb + 3) / 4);
            chk(mcn_sgd_plan_build_fused(fi.data(), fp.data(), _sgdCount, _sgdFused.p, pb, &_sgdFusedBlocks, _stream), "sgd plan (fused tail)");
            _sgdFusedCount = _sgdCount;
            _fusedTailOk = true;
        }
    }
    int nc = 0;
    for (CostLayer* c : _costs) nc += c->numCostValues();
    _costBuf.ensure(std::max(2 * nc, 4));
    _costHost.assign(nc, 0.0);
    nc = 0;
    for (CostLayer* c : _costs) {
        c->dCost = reinterpret_cast<double*>(_costBuf.p) + nc;
        nc += c->numCostValues();
    }
    zeroCosts();
    _allocatedCases = maxCases;
    chk(mcn_stream_sync(_stream), "sync");
}

void ConvNet::initComm(int rank, int world, const char* id128, bool force) {
    if (world < 1 || rank < 0 || rank >= world) throw NetError("bad data-parallel rank / world size");
    if (_dpOn) throw NetError("data-parallel group already initialised");
    _rank = rank;
    _world = world;
    _dpOn = world > 1 || force;
    if (_dpOn && _hasSharing)
        for (auto& l : _layers)
            if (auto* wl = dynamic_cast<WeightLayer*>(l.get()))
                for (int k = 0; k < wl->numWeights(); k++)
                    if (!wl->weights(k).owner() && !wl->weights(k).useGrad()) {
                        // layer.cu:378, :554: a second contribution to a shared FC / local matrix REPLACES the increment, momentum
                        // included; replicas exchange raw gradients and fold the momentum in afterwards, which cannot restate that
                        _dpOn = false;
                        throw NetError("layer '" + wl->name() + "': shared fc / local weights are not defined for data-parallel replicas");
                    }
    dropStepGraphs();
    if (_dpOn) {
        chk(mcn_comm_init(&_comm, world, rank, id128), "comm init");
        // The exchange stream would like a hardware queue of its own; the runtime decides.  Measured with rocprofv3 on MI355X /
        // ROCm 7.2: streams are doubled up on the hardware queues long before GPU_MAX_HW_QUEUES (with the upload stream at the
        // lowest priority the exchange stream lands on the SIDE stream's queue, otherwise on the main stream's), and forcing a
        // queue of its own -- more streams until one tests independent, or a stream of the highest priority -- over-subscribes
        // the queues: every kernel next to its work ran 2-3 times longer (forced data-parallel step 0.33 -> 0.75 ms).  So: one
        // normal stream, and the gate self-test decides whether it may be gated (independent queue) or keeps its events.
        int nr = 0, ver = 0, isRccl = 0;
        chk(mcn_comm_info(_comm, &nr, &ver, &isRccl), "comm info");
        // Ranks of the rehearsal transport (test build) SHARE one GPU and wait for each other inside their collectives: a
        // gate of one rank would spin on the device its peer needs to get there.  Spinning is for ranks that own their GPU.
        if (!isRccl && world > 1) _gateOk = _forkGateOk = false;
        chk(mcn_stream_create(&_commStream), "comm stream");
        // Opt-in (MCN_COMM_GATES=1): on every box of the pool the exchange stream shares a hardware queue, the self-test below
        // says so, and the gated exchange path has therefore never run on hardware with more than the self-test behind it.  The
        // default is what HAS run: side-stream buckets exchanged in line on the side stream (flushBuckets), events elsewhere.
        if (_gateOk && _gradStream && getenv("MCN_COMM_GATES") != nullptr && getenv("MCN_NO_COMM_GATES") == nullptr) {
            chk(mcn_gate_alloc(&_commGate), "gate");
            chk(mcn_gate_alloc(&_mainToComm), "gate");
            chk(mcn_gate_alloc(&_sideToComm), "gate");
            int a = 0, b = 0, c = 0;
            chk(mcn_gate_selftest(_commGate, _stream, _commStream, &a), "gate self-test");
            if (a) chk(mcn_gate_selftest(_mainToComm, _commStream, _stream, &b), "gate self-test");
            if (b) chk(mcn_gate_selftest(_sideToComm, _commStream, _gradStream, &c), "gate self-test");
            _commGateOk = a && b && c;
            if (getenv("MCN_DEBUG_TIMES"))
                fprintf(stderr, "[mcn] exchange stream: %s\n", _commGateOk ? "a hardware queue of its own (gates on)" : "shares a hardware queue (events)");
        }
        chk(mcn_event_create_sync(&_evGrad), "event");
        chk(mcn_event_create_sync(&_evGrad2), "event");
        chk(mcn_event_create_sync(&_evGather), "event");
        chk(mcn_event_create_sync(&_evJoinComm), "event");
        _bucketEvents.resize(16, nullptr);
        for (void*& e : _bucketEvents) chk(mcn_event_create_sync(&e), "event");
        _commSub.start(_deviceID);  // waits + RCCL calls are issued by their own thread
        chk(mcn_event_create_sync(&_evComm), "event");
        const char* env = getenv("MCN_BUCKET_MB");
        _bucketFloats = (size_t)((env ? atof(env) : 4.0) * 1024 * 1024 / 4);
        allocate(_minibatchSize);  // every weight matrix now needs a gradient buffer in the arena
        // Warm the communicator up on both streams that carry collectives (exchange stream; main stream: the in-line last
        // bucket) with the sizes of a step: RCCL builds its channels and connections inside the FIRST collectives, which can
        // take seconds with eight ranks -- longer than a stream gate waits before it reports a dead stream.  The arena is
        // all zeros here, and a sum of zeros leaves it so.
        chk(mcn_stream_sync(_stream), "stream sync");
        chk(mcn_allreduce_sum(_comm, _gradArena.p, _gradArena.cap, _commStream), "all-reduce (warm-up)");
        if (_gradArena.cap >= (size_t)64 * (world + 1)) {
            chk(mcn_comm_group_begin(), "group begin");
            chk(mcn_allgather(_comm, _gradArena.p, _gradArena.p + 64, 64, _commStream), "all-gather (warm-up)");
            chk(mcn_comm_group_end(), "group end");
        }
        chk(mcn_stream_sync(_commStream), "stream sync");
        chk(mcn_allreduce_sum(_comm, _gradArena.p, std::min<size_t>(_gradArena.cap, 8192), _stream), "all-reduce (warm-up)");
        chk(mcn_stream_sync(_stream), "stream sync");
        if (_gradStream && _gradArena.cap >= (size_t)64 * (world + 1)) {  // (the side stream carries the L2-SN gather of replicas that own their GPU)
            chk(mcn_comm_group_begin(), "group begin");
            chk(mcn_allgather(_comm, _gradArena.p, _gradArena.p + 64, 64, _gradStream), "all-gather (warm-up)");
            chk(mcn_comm_group_end(), "group end");
            chk(mcn_stream_sync(_gradStream), "stream sync");
        }
        // replicas wait for each other inside every collective (a peer's first steps page code in): a gate that depends on the
        // exchange stream must outlast that
        if (world > 1) _gateTimeoutMs = 120000;
    }
}

void ConvNet::commInfo(int& numRanks, int& version, int& isRccl) {
    if (!_comm) throw NetError("no replica group (initDataParallel has not been called)");
    chk(mcn_comm_info(_comm, &numRanks, &version, &isRccl), "comm info");
}

void ConvNet::sync() {
    joinGradWork();
    chk(mcn_stream_sync(_stream), "sync");
    for (int* g : {_gateOk ? _gate : nullptr, _forkGateOk ? _forkGate : nullptr, _commGateOk ? _commGate : nullptr,
                   _commGateOk ? _mainToComm : nullptr, _commGateOk ? _sideToComm : nullptr}) {
        if (!g) continue;
        int timedOut = 0;
        chk(mcn_gate_timed_out(g, &timedOut, _stream), "gate state");
        if (timedOut) throw NetError("a stream gate timed out: one stream never reached the point the other waited for");
    }
}

// ---- side-stream submission -----------------------------------------------------------------
// (the self-tests at start-up cover the pairs of streams that gate each other: main / side here, main / exchange in initComm)
bool ConvNet::gatesUsable() const { return _gateOk && !_capturing && !g_noHostWait && !_profiling; }
bool ConvNet::g_noHostWaitFlag() { return g_noHostWait; }
void ConvNet::signalCommGate(int seq) { chk(mcn_stream_signal(_commGate, seq, _commStream), "stream signal (exchange)"); }
void ConvNet::awaitCommGate(int seq) {
    const auto t0 = std::chrono::steady_clock::now();
    chk(mcn_stream_gate(_commGate, seq, _gateTimeoutMs, _stream), "stream gate (exchange)");
    g_eventCallMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
void ConvNet::signalGate(int seq) { chk(mcn_stream_signal(_gate, seq, stream()), "stream signal"); }
void ConvNet::awaitGate(int seq) {
    const auto t0 = std::chrono::steady_clock::now();
    chk(mcn_stream_gate(_gate, seq, _gateTimeoutMs, _stream), "stream gate");
    g_eventCallMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
void* ConvNet::stream() const { return t_stream == 1 ? _gradStream : t_stream == 2 ? _commStream : _stream; }
void* ConvNet::workspace() const { return t_stream == 1 ? _ws2.p : _ws.p; }  // (exchange-thread work uses no workspace)

void Submitter::start(int deviceID, bool* sideFlag) {
    _flag = sideFlag;
    _thread = std::thread([this, deviceID] { loop(deviceID); });
}

void Submitter::loop(int deviceID) {
    mcn_set_device(deviceID < 0 ? 0 : deviceID);
    for (;;) {
        std::function<void()> fn;
        {
            std::unique_lock<std::mutex> lk(_mutex);
            // work arrives every few tens of microseconds while a step is being submitted: poll before sleeping
            for (int spin = 0; _queue.empty() && !_stop && spin < 20000; spin++) {
                lk.unlock();
                std::this_thread::yield();
                lk.lock();
            }
            _cv.wait(lk, [this] { return !_queue.empty() || _stop; });
            if (_queue.empty()) return;  // stop requested
            fn = std::move(_queue.front());
            _queue.pop_front();
        }
        try {
            if (!_error) fn();
        } catch (...) {
            std::lock_guard<std::mutex> lk(_mutex);
            if (!_error) _error = std::current_exception();
        }
        _done.fetch_add(1, std::memory_order_release);
    }
}

void Submitter::stop() {
    if (!_thread.joinable()) return;
    {
        std::lock_guard<std::mutex> lk(_mutex);
        _stop = true;
    }
    _cv.notify_all();
    _thread.join();
}

long Submitter::push(std::function<void()> fn) {
    {
        std::lock_guard<std::mutex> lk(_mutex);
        _queue.push_back(std::move(fn));
    }
    const long t = _pushed.fetch_add(1, std::memory_order_relaxed) + 1;
    _cv.notify_one();
    return t;
}

void Submitter::wait(long ticket) {
    if (_done.load(std::memory_order_acquire) < ticket) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0; _done.load(std::memory_order_acquire) < ticket; spin++)
            if (spin > 64) std::this_thread::yield();
        g_blockedMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    std::exception_ptr e;
    {
        std::lock_guard<std::mutex> lk(_mutex);
        std::swap(e, _error);
    }
    if (e) std::rethrow_exception(e);
}

void ConvNet::waitSideSubmitted(long ticket) {
    if (ticket > 0) {
        chk(mcn_flush_tick(), "flush tick");  // (the side thread may be waiting for a tick that no launch has carried yet)
        _side.wait(ticket);
    }
}

// Bias gradients: row sums over 0.1-19 MB, one or two launches of 2-5 us per layer (eight per DeepID2 step, ~30 us of the
// side stream).  A single-GPU pass queues them and the last weight layer's bias piece (side stream) issues all of them as
// one mcn_add_sum_multi (two launches); what a pass leaves queued (its first pass, before the number of weight layers is
// known) is issued by bprop() itself.  Data parallel: a bucket of the exchange carries the bias gradients of its layers, so
// whoever sends a bucket issues what is queued first (flushBuckets) -- one call per bucket instead of one or two per layer.
bool ConvNet::biasBatching() const {
    static const bool off = getenv("MCN_NO_BIAS_BATCH") != nullptr;
    return !off && _fusion;
}
void ConvNet::queueBiasSum(float* tgt, const float* mat, int rows, int cols, float scaleSum, float* partsOut) {
    _biasBatch[t_stream == 1 ? 1 : 0].push_back({tgt, mat, rows, cols, scaleSum, partsOut});
}
void ConvNet::flushBiasSums(int which) {
    std::vector<BiasSum>& b = _biasBatch[which >= 0 ? which : (t_stream == 1 ? 1 : 0)];
    if (b.empty()) return;
    ProfScope ps(this, "net", "biasGrads");
    for (size_t i = 0; i < b.size(); i += 8) {
        const int cnt = (int)std::min<size_t>(8, b.size() - i);
        ::mcn_row_sum_item items[8];
        for (int k = 0; k < cnt; k++) {
            const BiasSum& q = b[i + k];
            items[k].tgt = q.tgt; items[k].mat = q.mat; items[k].rows = q.rows; items[k].cols = q.cols;
            items[k].scaleThis = 0.f; items[k].scaleSum = q.scaleSum;
        }
        if (_fusedTailNow) {  // the long rows' second stage happens in the update launch (their partial sums stay in the layers' buffers)
            float* parts[8];
            for (int k = 0; k < cnt; k++) parts[k] = b[i + k].partsOut;
            chk(mcn_add_sum_multi_stage1(items, cnt, parts, stream()), "bias gradients (stage 1)");
        } else if (mcn_add_sum_multi_workspace_bytes(items, cnt) <= workspaceBytes()) {
            chk(mcn_add_sum_multi(items, cnt, workspace(), workspaceBytes(), stream()), "bias gradients");
        } else {
            for (int k = 0; k < cnt; k++)
                chk(mcn_add_sum(items[k].tgt, items[k].mat, items[k].rows, items[k].cols, 1, 0.f, items[k].scaleSum, workspace(), workspaceBytes(),
                                stream()), "bias gradient");
        }
    }
    b.clear();
}

// wl is the last layer of this pass that queues a bias row sum (known from the second pass on; until then `last` decides)
bool ConvNet::flushesBiasBatch(WeightLayer* wl, bool last) {
    if (_passOrder.empty() || _passNow.size() > _passOrder.size() || !std::equal(_passNow.begin(), _passNow.end(), _passOrder.begin())) return last;
    WeightLayer* lastQueuer = nullptr;
    for (WeightLayer* l : _passOrder)
        if (l->queuesBiasSum(_fusedTailNow)) lastQueuer = l;
    return lastQueuer ? lastQueuer == wl : last;
}

bool ConvNet::lastWeightLayerOfPass(WeightLayer* wl) {
#ifdef MCN_TEST_HOOKS
    if (_failArmed) {  // (set by bprop from failNextBackwardAt)
        std::lock_guard<std::mutex> lk(_failMutex);
        if (wl->name() == _failAt) {
            _failAt.clear();
            _failArmed = false;
            throw NetError("injected failure at layer '" + wl->name() + "'");
        }
    }
#endif
    _passNow.push_back(wl);
    return ++_wlSeen == _wlPerPass;
}

// Most of the weight update, early: when a training pass reaches its LAST weight layer every other weight layer has queued
// its backward-data kernels (the only readers of its weights left in this step) on the main stream and its gradients on the
// side stream, so one launch on the side stream -- ordered after both by the fork that follows -- can update all of them
// while the main stream still computes the last layer's gradient.  The update after that covers the last layer only
// (DeepID2: conv1, 784 of 3.2 M parameters), and the main stream's gate for the side stream comes AFTER it.  Same kernel and
// records as the one-launch update (the fused-tail records where that plan is in use), so the weights are bit-identical.
// Not in data-parallel mode (the update follows the all-reduce), not with shared weights, not under capture.
// Round 2 measured this as a loss (DeepID2 0.282 -> 0.289 ms) when the join was an event wait and the early launch queued in
// front of the last layer's own side-stream pieces; with stream gates, the last layer's gradient on the main stream and the
// early launch the side stream's LAST kernel: DeepID2+ see DESIGN.md section 5.  MCN_NO_EARLY_SGD=1 switches it off.
void ConvNet::earlyUpdate() {
    _earlyPos = 0;
    if (_earlyOff || dataParallel() || _hasSharing || !_fusion || _sgdCount == 0 || _passOrder.size() < 2 || _passNow != _passOrder) return;
    if (!_gradStream || _profiling || _capturing || !gatesUsable()) return;
    if (_fusedTailOk != _fusedTailNow) return;  // (a short minibatch: the plain plan, in one launch)
    if (!_splitPlans) {
        std::vector<::mcn_sgd_item> early, late;
        std::vector<::mcn_sgd_partials> earlyP, lateP;
        for (auto& l : _layers) {
            std::vector<Weights*> ws_;
            l->collectWeights(ws_);
            const bool isLast = l.get() == static_cast<Layer*>(_passOrder.back());
            const bool visited = std::find(_passOrder.begin(), _passOrder.end(), dynamic_cast<WeightLayer*>(l.get())) != _passOrder.end();
            for (Weights* w : ws_) {
                ::mcn_sgd_item it;
                if (!w->planItem(false, it)) continue;
                ::mcn_sgd_partials q = {nullptr, 0, 0, 1.f};
                if (_fusedTailOk && w->partials.data && it.useGrad) {
                    q.data = w->partials.data; q.count = w->partials.count; q.kind = w->partials.kind; q.scale = w->partials.scale;
                }
                (visited && !isLast ? early : late).push_back(it);
                (visited && !isLast ? earlyP : lateP).push_back(q);
            }
        }
        _sgdEarlyCount = (int)early.size();
        _sgdLateCount = (int)late.size();
        _sgdEarlyElems = 0;
        for (const auto& it : early) _sgdEarlyElems += it.n;
        if (_sgdEarlyCount > 0) {
            const size_t pb = mcn_sgd_plan_bytes(_sgdEarlyCount);
            _sgdEarly.ensure((pb + 3) / 4);
            chk(mcn_sgd_plan_build_fused(early.data(), earlyP.data(), _sgdEarlyCount, _sgdEarly.p, pb, &_sgdEarlyBlocks, _stream), "sgd plan (early)");
        }
        if (_sgdLateCount > 0) {
            const size_t pb = mcn_sgd_plan_bytes(_sgdLateCount);
            _sgdLate.ensure((pb + 3) / 4);
            chk(mcn_sgd_plan_build_fused(late.data(), lateP.data(), _sgdLateCount, _sgdLate.p, pb, &_sgdLateBlocks, _stream), "sgd plan (late)");
        }
        _splitPlans = true;
    }
    // Worth it only where the update launch is long next to what the split costs (a second update launch, a second gate on the main
    // stream and one more hand-shake with the side thread, about 10 us): DeepID2+ (22 M parameters, 67 us) gains 7 us per step,
    // DeepID2 (3.2 M, 13 us) gains nothing and its traced step shows the main thread waiting for the side thread.
    if (_sgdEarlyCount == 0 || _sgdEarlyElems < _earlyMinElems) return;
    _earlyUpdateIssued = true;
    // two signals on the side stream: A in front of the early update -- every gradient piece and cost value of this step is
    // behind it, the next forward pass may overwrite the activations -- and B (joinGradWork) behind it: the weights are ready
    const int seqA = nextGateSeq();
    _earlySeqA = seqA;
    deferSideWork([this, seqA] {
        signalGate(seqA);
        chk(mcn_sgd_update_planned(_sgdEarly.p, _sgdEarlyCount, _sgdEarlyBlocks, stream()), "sgd update (early)");
    }, 0, true);
    _earlyPos = _deferred.size();  // (finishSideWork sends everything up to here to the side stream; 0 if a flush took it already)
}

void ConvNet::deferSideWork(std::function<void()> fn, int units, bool hold) {
    if (!_gradStream || _profiling) {  // profiling times every operation on the main stream
        fn();
        return;
    }
    _deferred.emplace_back(std::move(fn), units);
    _deferredUnits += units;
    if (!hold && _deferredUnits >= _sideUnits) flushSideWork(_deferred.size());
}

long ConvNet::forkSideWork(std::function<void()> fn) {
    if (!_gradStream || _profiling) {
        fn();
        return 0;
    }
    _deferred.emplace_back(std::move(fn), 0);
    flushSideWork(_deferred.size());  // (together with what was deferred before: keeps the order)
    return _capturing ? 0 : _side.pushed();
}

// the first `count` deferred items become one task of the side thread, ordered after everything the main stream
// holds at this point
// The side thread waits until the main stream has run the tick kernel that flushSideWork queued (sequence numbers
// increase, so a later tick covers an earlier one).  A main stream that dies never ticks: give up after a minute.
void ConvNet::awaitTick(int tick, volatile int* flag) {
    const auto t0 = std::chrono::steady_clock::now();
    volatile int* f = flag ? flag : _tickFlag;
    for (unsigned spin = 0; (int)(*f - tick) < 0; spin++) {
        if ((spin & 0xfffff) == 0xfffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
            throw NetError("a stream did not reach a fork point within 60 s");
    }
    g_eventCallMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void ConvNet::flushSideWork(size_t count) {
    if (count == 0) return;
    const bool ticks = _tickFlag && !_capturing && !g_noHostWait && t_stream == 0;  // (only the thread that feeds a stream ticks it)
    void* forkEvent = nullptr;
    int tick = 0;
    bool gateFork = false;
    if (ticks) {
        // a one-thread kernel that writes a sequence number the side thread polls for: seen 5 us earlier than the event,
        // and half the bubble on this stream (tools/event_poll_probe.hip)
        // ... carried by this thread's next launch: in the first instructions of the kernel itself where it has a tick slot
        // (the backward-data kernels that follow a weight layer's fork all do), else in a kernel of its own behind it
        // (MCN_TICK_KERNEL=1: always a kernel of its own, at once)
        tick = ++_tickSeq;
        static const bool ownKernel = getenv("MCN_TICK_KERNEL") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        gateFork = _forkGateOk && !ownKernel;
        if (ownKernel) chk(mcn_stream_tick(_tickFlag, tick, _stream), "stream tick");
        else chk(mcn_arm_tick_gate(_tickFlag, gateFork ? _forkGate : nullptr, tick, _stream), "arm tick");
        g_eventCallMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else {
        forkEvent = _forkEvents[_forkNext++ % _forkEvents.size()];
        // (an event may be recorded again before the side thread has waited for its previous use: that wait then covers
        // more of the main stream than it needs to, never less; 32 events make it rare)
        chk(timedEventRecord(forkEvent, _stream), "event record");
    }
    auto fns = std::make_shared<std::vector<std::function<void()>>>();
    for (size_t i = 0; i < count; i++) {
        fns->push_back(std::move(_deferred[i].first));
        _deferredUnits -= _deferred[i].second;
    }
    _deferred.erase(_deferred.begin(), _deferred.begin() + count);
    _earlyPos = _earlyPos > count ? _earlyPos - count : 0;
    _gradPending = true;
    if (_capturing) {
        // graph capture: this thread issues the side stream's work itself -- the fork event pulls the side stream into the
        // capture, and the recorded graph carries the dependency instead of a host thread
        chk(mcn_stream_wait_event(_gradStream, forkEvent), "wait event");
        t_stream = 1;
        try {
            for (auto& fn : *fns) fn();
        } catch (...) {
            t_stream = 0;
            throw;
        }
        t_stream = 0;
        return;
    }
    _side.push([this, forkEvent, tick, gateFork, fns] {
        t_stream = 1;
        const auto t0 = std::chrono::steady_clock::now();
        if (forkEvent) chk(timedStreamWait(_gradStream, forkEvent), "wait event");
        else if (gateFork) chk(mcn_stream_gate(_forkGate, tick, _gateTimeoutMs, _gradStream), "stream gate (fork)");
        else awaitTick(tick);
        const auto t1 = std::chrono::steady_clock::now();
        const uint64_t l0 = mcn_launch_ns();
        const size_t c0 = mcn_launch_count();
        for (auto& fn : *fns) fn();
        _sidePollNs += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
        _sideWorkNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
        _sideLaunchNs += (long)(mcn_launch_ns() - l0);
        _sideLaunches += (long)(mcn_launch_count() - c0);
    });
}

// Work for the exchange stream, ordered after everything the main stream holds now (data-parallel L2-SN: pair statistics,
// all-gather, margin search and cost in ONE hop to one thread -- through the side thread it took three hops and kept
// that thread from issuing the gradients of fc6 / fc5).
long ConvNet::forkExchangeWork(std::function<void()> fn) {
    _exchangeWorkPending = true;
    if (_commTicks && !_capturing && !g_noHostWait && t_stream == 0) {
        const int tick = ++_tickSeq;
        chk(mcn_stream_tick(_tickFlag, tick, _stream), "stream tick");
        return _commSub.push([this, tick, fn] {
            t_stream = 2;
            awaitTick(tick, _tickFlag);
            fn();
        });
    }
    void* forkEvent = _forkEvents[_forkNext++ % _forkEvents.size()];
    chk(timedEventRecord(forkEvent, _stream), "event record");
    return _commSub.push([this, forkEvent, fn] {
        t_stream = 2;
        chk(timedStreamWait(_commStream, forkEvent), "wait event");
        fn();
    });
}

void ConvNet::waitExchangeSubmitted(long ticket) {
    SiteTimer st("waitExchangeSubmitted");
    if (ticket > 0) _commSub.wait(ticket);
}

void ConvNet::logComm(int kind, size_t offset, size_t count) {
    std::lock_guard<std::mutex> lk(_logMutex);
    if (_commLog.size() < 65536) _commLog.push_back({kind, offset, count});
}

std::vector<ConvNet::CommRec> ConvNet::takeCommLog() {
    std::lock_guard<std::mutex> lk(_logMutex);
    std::vector<CommRec> out;
    out.swap(_commLog);
    return out;
}

std::vector<ConvNet::SegmentInfo> ConvNet::gradSegments() {
    std::vector<SegmentInfo> out;
    for (const Segment& s : _segments) out.push_back({s.layer->name(), s.begin, s.end});
    return out;
}

void* ConvNet::commEvent() {  // exchange thread only
    if (_commEventsUsed == _commEventPool.size()) {
        void* e = nullptr;
        chk(mcn_event_create(&e), "event create");
        _commEventPool.push_back(e);
    }
    return _commEventPool[_commEventsUsed++];
}

void ConvNet::commCollect(std::map<std::string, ProfEntry>& into) {  // worker thread, exchange thread idle, device drained
    for (const CommSpan& s : _commSpans) {
        float ms = 0;
        chk(mcn_event_elapsed_ms(&ms, s.e0, s.e1), "event elapsed");
        ProfEntry& e = into[s.kind == 0 ? "comm.allreduce" : "comm.allgather"];
        e.count++;
        e.ms += ms;
        ProfEntry& b = into[s.kind == 0 ? "comm.allreduce_floats" : "comm.allgather_floats"];
        b.count++;
        b.ms += (double)s.count;  // (a count carried in the ms field: floats moved per rank)
    }
    _commSpans.clear();
    _commEventsUsed = 0;
}

void ConvNet::gatherPairsHere(const float* sendA, float* recvA, const float* sendB, float* recvB, size_t countPerRank) {
    // on the calling thread's stream: the exchange stream (exchange thread) or the side stream (side thread, gated path).  No
    // lock: only one thread talks to the communicator at a time by construction -- a step's gather is issued before the side
    // thread hands the step's first bucket to the exchange thread, and after the main thread has seen the previous step's last
    // exchange issued (waitComm) -- which also makes the order of collectives the same on every rank.
    void* st = stream();
    void *e0 = nullptr, *e1 = nullptr;
    if (_commTiming) {
        e0 = commEvent();
        e1 = commEvent();
        chk(mcn_event_record(e0, st), "event record");
    }
    chk(mcn_comm_group_begin(), "group begin");
    chk(mcn_allgather(_comm, sendA, recvA, countPerRank, st), "all-gather");
    chk(mcn_allgather(_comm, sendB, recvB, countPerRank, st), "all-gather");
    chk(mcn_comm_group_end(), "group end");
    if (_commTiming) {
        chk(mcn_event_record(e1, st), "event record");
        _commSpans.push_back({1, e0, e1, 2 * countPerRank});
    }
    logComm(1, 0, countPerRank);
    logComm(1, 0, countPerRank);
}

void ConvNet::joinExchange() {
    if (!_exchangeWorkPending) return;
    SiteTimer st("joinExchange");
    _commSub.wait(_commSub.pushed());
    chk(timedEventRecord(_evJoinComm, _commStream), "event record");
    chk(timedStreamWait(_stream, _evJoinComm), "wait event");
    _exchangeWorkPending = false;
}

void ConvNet::finishSideWork(bool deferHostWait) {
    // what is still deferred has no backward-data work left to hide behind: half goes to the side stream, the
    // rest runs on the main stream next to it
    if (_earlyPos > 0) flushSideWork(_earlyPos);  // the early update and every piece queued before it: side stream
    if (_hasSharing && _gradStream && !_profiling) {
        flushSideWork(_deferred.size());  // shared matrices: every contribution on the side stream, in pass order
    } else if (_deferredUnits >= 2) {
        size_t cut = 0;
        for (int units = 0; cut < _deferred.size() && units < _deferredUnits / 2; cut++) units += _deferred[cut].second;
        flushSideWork(cut);
    }
    joinGradWork(deferHostWait);
}

// Cost VALUES nobody waits for before the batch ends: on the main stream at the very end of the pass, where it idles
// until the side stream has finished the last gradients (early on the side stream they delayed fc6's gradients).
void ConvNet::deferLateWork(std::function<void()> fn) {
    if (!_gradStream || _profiling) {
        fn();
        return;
    }
    // (measured: on the side stream right after the softmax instead, MCN_SIDE_COST=1, the step is 0-3 us slower -- the end of
    // a step is bound by the side stream's gradient tail as much as by the main stream)
    // (round 3, with the join a stream gate and the side stream's tail shortened by the fused update: the side stream is the
    // better place after all, 0.2312 -> 0.2289 ms; MCN_MAIN_COST=1 for the old placement)
    static const bool sideCost = getenv("MCN_MAIN_COST") == nullptr;
    if (dataParallel() || sideCost) {  // data parallel: the main stream's tail is the hand-over of the last bucket
        deferSideWork(std::move(fn), 0);
        return;
    }
    _late.push_back(std::move(fn));
}

void ConvNet::runLateWork() {
    if (_late.empty()) return;
    auto late = std::move(_late);
    _late.clear();
    for (auto& fn : late) fn();
}

void ConvNet::hostWaitSide() {
    if (!_sideWaitPending) return;
    _sideWaitPending = false;
    SiteTimer st("hostWaitSide");
    _side.wait(_side.pushed());
}

// deferHostWait: the caller promises to call hostWaitSide() before it touches anything the side thread's closures read or
// write (the pass state of the layers, the update counters); until then it may only queue work on the main stream.
// The first layer of a forward pass that reads weights the early update writes waits for the side stream's signal B; the layers
// in front of it (data, the pass's last weight layer -- updated on the main stream --, pools, neurons) run next to that update.
void ConvNet::beforeLayerFprop(Layer* l) {
    if (!_pendingWeightGate) return;
    if (!dynamic_cast<WeightLayer*>(l) || (!_passOrder.empty() && l == static_cast<Layer*>(_passOrder.back()))) return;
    awaitGate(_pendingWeightGate);
    _pendingWeightGate = 0;
}

// keepWeightGate: the caller is a forward pass (beforeLayerFprop places the gate for the early update's weights)
void ConvNet::joinGradWork(bool deferHostWait, bool keepWeightGate) {
    if (_pendingMainGate) {
        awaitGate(_pendingMainGate);
        _pendingMainGate = 0;
    }
    if (_pendingWeightGate && !keepWeightGate) {
        awaitGate(_pendingWeightGate);
        _pendingWeightGate = 0;
    }
    hostWaitSide();  // (left pending by an earlier join)
    if (_hasSharing && _gradStream && !_profiling && !_deferred.empty()) flushSideWork(_deferred.size());  // (never a sharer's piece on the main stream)
    if (!_deferred.empty()) {
        auto rest = std::move(_deferred);
        _deferred.clear();
        _deferredUnits = 0;
        for (auto& d : rest) d.first();  // on the main stream
    }
    runLateWork();
    joinExchange();
    if (!_gradPending) return;
    if (gatesUsable()) {
        // The side thread signals behind everything it was given; this thread queues the gate on the main stream at once --
        // a gate may be queued before its signal has been launched -- so the main stream's queue never runs dry while the
        // two threads shake hands.  (What a pass without a known last layer left in the side thread's bias batch goes
        // out on the side stream first.)
        const int seq = nextGateSeq();
        _side.push([this, seq] {
            t_stream = 1;
            flushBiasSums();
            signalGate(seq);
        });
        // (an early update is on the side stream: the main stream's gate goes behind ITS update launch, updateWeights)
        if (_earlyUpdateIssued && deferHostWait) {
            _pendingMainGate = _earlySeqA;
            _pendingWeightGate = seq;
        } else awaitGate(seq);  // (sends an armed tick first: the side thread may be waiting for it)
        _gradPending = false;
        static const bool noDefer = getenv("MCN_NO_DEFER_WAIT") != nullptr;  // (A/B switch)
        if (deferHostWait && !dataParallel() && !noDefer) _sideWaitPending = true;
        else {
            SiteTimer st("joinGradWork.waitSideThread");
            _side.wait(_side.pushed());
        }
        return;
    }
    if (!_capturing) {
        chk(mcn_flush_tick(), "flush tick");  // (as in waitSideSubmitted)
        _side.wait(_side.pushed());  // the side thread has issued everything it was given
    }
    static const bool joinTick = getenv("MCN_JOIN_TICK") != nullptr;
    if (joinTick && _tickFlagSide && !_capturing && !g_noHostWait) {
        // (experiment) this thread polls a tick of the side stream instead of queueing a wait on its own stream
        const int tick = ++_tickSeqSide;
        chk(mcn_stream_tick(_tickFlagSide, tick, _gradStream), "stream tick");
        awaitTick(tick, _tickFlagSide);
    } else {
        chk(timedEventRecord(_evJoin, _gradStream), "event record");
        chk(timedStreamWait(_stream, _evJoin), "wait event");
    }
    _gradPending = false;
}

void ConvNet::fpropCurrent(PassType pt) {  // convnet.cu:213-219
    if (_failed) throw NetError("this data-parallel replica failed in an earlier step and has left its group: destroy the model on every rank");
    joinGradWork(false, true);  // side-stream work of the previous minibatch (cost values in a test pass) reads buffers rewritten below
    for (auto& l : _layers) l->reset();
    for (DataLayer* d : _dataLayers) {
        if (d->dataIdx < 0 || d->dataIdx >= (int)_curData.size()) throw NetError("data layer '" + d->name() + "': no data matrix " + std::to_string(d->dataIdx));
        d->setSlice(_curData[d->dataIdx]);  // every data matrix is in place before anything propagates (a fused softmax reads the labels)
    }
    // Data layers that feed cost layers only (labels, attribute targets) go first: a cost layer starts when its LAST input
    // arrives, and in cfg order the labels came after the whole image tower had been walked -- the L2-SN cost, which needs
    // only fc5's output and is forked to the side stream, was forked behind fc6's product and the softmax instead of behind
    // fc5 (its cost kernel started ~40 us later than it could and the backward chain waited for it).
    static const bool labelsFirst = getenv("MCN_NO_LABELS_FIRST") == nullptr;
    auto costOnly = [](DataLayer* d) {
        for (Layer* l : d->next())
            if (l->type().compare(0, 5, "cost.") != 0) return false;
        return true;
    };
    if (labelsFirst)
        for (DataLayer* d : _dataLayers)
            if (costOnly(d)) d->fpropData(_curData[d->dataIdx], pt);
    for (DataLayer* d : _dataLayers)
        if (!labelsFirst || !costOnly(d)) d->fpropData(_curData[d->dataIdx], pt);
}

void ConvNet::fpropSlice(long start, long end, PassType pt) {
    end = std::min(end, _dp->numCases());
    _numCases = (int)(end - start);
    if (_numCases > _allocatedCases) throw NetError("minibatch larger than the allocated size");
    joinGradWork(false, true);  // the slice buffers (labels) may still be read by cost kernels of the previous minibatch
    _dp->getSlice(start, end, _curData);
    fpropCurrent(pt);
}

void ConvNet::fprop(int miniIdx, PassType pt) {  // convnet.cu:229-233
    fpropSlice((long)miniIdx * _minibatchSize, (long)(miniIdx + 1) * _minibatchSize, pt);
}

void ConvNet::bprop(PassType pt) {  // convnet.cu:206-211
#ifdef MCN_TEST_HOOKS
    {
        std::lock_guard<std::mutex> lk(_failMutex);
        _failArmed = !_failAt.empty();
    }
#endif
    // the fused tail is planned for full minibatches (the bias scale eps / numCases is part of the plan)
    _fusedTailNow = _fusedTailOk && pt == PASS_TRAIN && _numCases == _allocatedCases && !_profiling;
    if (_phases) profBegin("phase.backward_data_chain");
    for (CostLayer* c : _costs) c->bprop(pt);
    if (_phases) {
        profEnd();
        profBegin("phase.gradient_tail");
    }
    for (auto& l : _layers) l->reset();
    _wlPerPass = _wlSeen;  // (the first pass learns how many weight layers a backward pass visits)
    _wlSeen = 0;
    if (_passNow != _passOrder) {  // ... and in which order (earlyUpdate)
        _passOrder = _passNow;
        _splitPlans = false;
    }
    _passNow.clear();
    // a training pass is followed by the update launch, which reads nothing the side thread's closures touch on the host:
    // the hand-shake with that thread waits until the launch is out (updateWeights)
    finishSideWork(pt == PASS_TRAIN && !dataParallel());
    if (!_sideWaitPending) flushBiasSums(1);  // (after the join: whatever the pass left queued, on the main stream)
    flushBiasSums(0);
    if (_phases) profEnd();
    if (dataParallel() && pt == PASS_TRAIN) flushBuckets(true);
}

void ConvNet::updateWeights() {  // convnet.cu:190-194
    if (dataParallel()) {
        hostWaitSide();
        waitComm();
    }
    if (_sgdCount > 0 && _fusion) {
        ProfScope ps(this, "net", "update");
        if (_earlyUpdateIssued) {  // this pass has updated everything but its last weight layer already
            _earlyUpdateIssued = false;
            if (_sgdLateCount > 0) chk(mcn_sgd_update_planned(_sgdLate.p, _sgdLateCount, _sgdLateBlocks, _stream), "sgd update (late)");
            if (_pendingMainGate) {
                awaitGate(_pendingMainGate);
                _pendingMainGate = 0;
            }
        } else if (_fusedTailNow) {
            chk(mcn_sgd_update_planned(_sgdFused.p, _sgdFusedCount, _sgdFusedBlocks, _stream), "sgd update (fused tail)");
        } else {
            chk(mcn_sgd_update_planned(_sgdPlan.p, _sgdCount, _sgdBlocks, _stream), "sgd update");
        }
        hostWaitSide();  // (the update counters and the pass flags below belong to the side thread's closures until it is done)
        _fusedTailNow = false;
        for (auto& l : _layers) l->weightsUpdated();
    } else {
        hostWaitSide();
        for (auto& l : _layers) l->updateWeights();
    }
}

// ---- step graphs -------------------------------------------------------------------------------
bool ConvNet::stepGraphUsable(int miniIdx) {
    if (!_useGraphs || _dpOn || _profiling || _phases || !_fusion || _stepsDone == 0) return false;  // (the first step runs eagerly: it learns the pass structure)
    const long start = (long)miniIdx * _minibatchSize, end = std::min(start + _minibatchSize, _dp->numCases());
    return _dp->sliceIsPointer(start, end);
}

void ConvNet::captureStep(int miniIdx, StepGraph& g) {
    joinGradWork();  // nothing may be pending on the side stream when the capture starts
    chk(mcn_graph_begin_capture(_stream), "begin capture");
    _capturing = true;
    g_noHostWait = true;
    try {
        fprop(miniIdx, PASS_TRAIN);
        bprop(PASS_TRAIN);
        updateWeights();
        joinGradWork();
    } catch (...) {
        _capturing = false;
        g_noHostWait = false;
        void* dead = nullptr;
        mcn_graph_end_capture(_stream, &dead, nullptr);
        mcn_graph_destroy(dead);
        throw;
    }
    _capturing = false;
    g_noHostWait = false;
    chk(mcn_graph_end_capture(_stream, &g.exec, &g.nodes), "end capture");
    g.data = _curData;
    g.numCases = _numCases;
}

void ConvNet::trainStep(int miniIdx) {
    if (stepGraphUsable(miniIdx)) {
        const long start = (long)miniIdx * _minibatchSize, end = std::min(start + _minibatchSize, _dp->numCases());
        std::vector<float*> slice;
        _dp->getSlice(start, end, slice);  // pointers only
        StepGraph& g = _stepGraphs[miniIdx];
        if (g.exec && (g.data != slice || g.numCases != (int)(end - start))) {  // the batch moved or changed shape
            mcn_graph_destroy(g.exec);
            g = StepGraph();
        }
        if (!g.exec) captureStep(miniIdx, g);  // (capturing runs nothing)
        _numCases = g.numCases;
        _curData = g.data;
        chk(mcn_graph_launch(g.exec, _stream), "graph launch");
    } else {
        fprop(miniIdx, PASS_TRAIN);
        bprop(PASS_TRAIN);
        updateWeights();
    }
    _stepsDone++;
}

void ConvNet::copyToCPU() {
    sync();
    for (auto& l : _layers) l->copyToCPU();
}

// ---- data-parallel gradient exchange ---------------------------------------------------------
void ConvNet::gradReady(WeightLayer* l) {
    {
        std::lock_guard<std::mutex> lk(_commMutex);
        for (Segment& s : _segments)
            if (std::find(s.contributors.begin(), s.contributors.end(), l) != s.contributors.end()) {
                // which stream(s) hold the kernels that wrote it: the side thread issues both of a layer's pieces on the
                // side stream; on the main thread the bias piece may have gone to the side stream (last layer of a pass)
                const bool first = s.reported == 0;
                s.onSide = (first ? false : s.onSide) || t_stream == 1 || _gradPending;
                s.onMain = (first ? false : s.onMain) || t_stream != 1;
                s.byMain = (first ? false : s.byMain) || t_stream != 1;
                s.ready = ++s.reported >= (int)s.contributors.size();
            }
    }
    // Only the side thread decides where a bucket ends: it sees the layers in one fixed order on every rank.  A layer
    // finished on the main thread (the last one of the pass) is sent by the forced flush that follows it; flushing here
    // could split the buckets differently from rank to rank, and mismatched collectives hang.
    if (t_stream == 1) flushBuckets(false);
}

void ConvNet::flushBuckets(bool force) {
    std::lock_guard<std::mutex> lk(_commMutex);
    // longest run of completed segments starting at the first one not sent yet
    size_t end = _sentUpTo;
    for (Segment& s : _segments) {
        if (s.end <= _sentUpTo) continue;
        if (!force && (!s.ready || s.byMain)) break;  // (a layer finished by the main thread waits for the forced flush)
        end = s.end;
    }
    if (end == _sentUpTo) return;
    if (!force && end - _sentUpTo < _bucketFloats) return;
    if (t_stream == 1) flushBiasSums();  // (the bias gradients of this bucket's layers, queued by this thread, on this stream)
    // The last bucket of a step -- the forced flush after the join, on the thread that feeds the main stream, which by then
    // follows everything either stream wrote -- is exchanged IN LINE on the main stream: no hop to the exchange thread, no
    // cross-queue wait there and back (two of the 10-15 us waits at the very end of every step; its payload is the last
    // layers' few kilobytes).  The earlier collectives have all been issued by then, so the order of calls is the same on
    // every rank.
    static const bool inlineLast = getenv("MCN_NO_INLINE_BUCKET") == nullptr;
    if (force && inlineLast && t_stream == 0 && !_gradPending && !_capturing) {
        {
            SiteTimer st("flushBuckets.inline.waitExchangeThread");
            _commSub.wait(_commSub.pushed());
        }
        SiteTimer st2("flushBuckets.inline.allreduce_call");
        float* lbuf = _gradArena.p + _sentUpTo;
        const size_t lcount = end - _sentUpTo;
        void *e0 = nullptr, *e1 = nullptr;
        if (_commTiming) {
            e0 = commEvent();
            e1 = commEvent();
            chk(mcn_event_record(e0, _stream), "event record");
        }
        chk(mcn_allreduce_sum(_comm, lbuf, lcount, _stream), "all-reduce (last bucket, main stream)");
        if (_commTiming) {
            chk(mcn_event_record(e1, _stream), "event record");
            _commSpans.push_back({0, e0, e1, lcount});
        }
        logComm(0, _sentUpTo, lcount);
        _sentUpTo = end;
        return;
    }
    // the exchange waits for the stream(s) that hold the kernels writing this bucket (a record + wait pair costs ~16 us
    // of host time, so only where needed; force: whatever is left, written by either stream)
    bool onMain = force, onSide = force && _gradPending;
    for (Segment& sg : _segments)
        if (sg.end > _sentUpTo && sg.begin < end) {
            onMain |= sg.onMain;
            onSide |= sg.onSide;
        }
    // An exchange stream that shares a hardware queue with a compute stream (the start-up self-test: no comm gates) buys no
    // overlap over the side stream itself -- the runtime puts it on the side stream's queue -- but costs an event record, a
    // hop to the exchange thread, a queued wait there (10-15 us on that queue) and another queued wait on the main stream
    // before the update.  A bucket written by the side stream alone is then exchanged IN LINE on the side stream by the side
    // thread: stream order is the dependency, and the join the main stream makes anyway covers the exchange.
    static const bool sideExchange = getenv("MCN_NO_SIDE_EXCHANGE") == nullptr;
    if (sideExchange && !force && t_stream == 1 && !onMain && !commGatesUsable() && gatesUsable() && !_capturing && !_commTicks) {
        float* sbuf = _gradArena.p + _sentUpTo;
        const size_t scount = end - _sentUpTo;
        void *e0 = nullptr, *e1 = nullptr;
        if (_commTiming) {
            e0 = commEvent();
            e1 = commEvent();
            chk(mcn_event_record(e0, _gradStream), "event record");
        }
        const auto t0 = std::chrono::steady_clock::now();
        chk(mcn_allreduce_sum(_comm, sbuf, scount, _gradStream), "all-reduce (side stream)");
        _rcclCallNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        _rcclCalls++;
        if (_commTiming) {
            chk(mcn_event_record(e1, _gradStream), "event record");
            _commSpans.push_back({0, e0, e1, scount});
        }
        logComm(0, _sentUpTo, scount);
        _sentUpTo = end;
        return;
    }
    // the producer records (cheap), the exchange thread waits and calls RCCL (expensive)
    void* evMain = nullptr;
    void* evSide = nullptr;
    // (a stream's ticks must be queued in increasing order: only the thread that feeds a stream ticks it, anyone else
    // records an event -- the side thread sends side-stream buckets, the main thread the forced rest, so that is rare)
    int tickMain = 0, tickSide = 0, gateMain = 0, gateSide = 0;
    const bool ticks = _commTicks && !_capturing && !g_noHostWait;
    // (a stream is signalled by the thread that feeds it, so that the numbers of a word increase in stream order; the rare
    // case of the other thread -- the main thread's forced flush of what the side stream wrote -- keeps its event)
    const bool cg = commGatesUsable();
    if (onMain) {
        if (cg && t_stream == 0) {
            gateMain = ++_mainToCommSeq;
            chk(mcn_stream_signal(_mainToComm, gateMain, _stream), "stream signal");
        } else if (ticks && t_stream == 0) {
            tickMain = ++_tickSeq;
            chk(mcn_stream_tick(_tickFlag, tickMain, _stream), "stream tick");
        } else {
            evMain = _bucketEvents[_bucketNext++ % _bucketEvents.size()];
            chk(timedEventRecord(evMain, _stream), "event record");
        }
    }
    if (onSide) {
        if (cg && t_stream == 1) {
            gateSide = ++_sideToCommSeq;
            chk(mcn_stream_signal(_sideToComm, gateSide, _gradStream), "stream signal");
        } else if (ticks && t_stream == 1) {
            tickSide = ++_tickSeqSide;
            chk(mcn_stream_tick(_tickFlagSide, tickSide, _gradStream), "stream tick");
        } else {
            evSide = _bucketEvents[_bucketNext++ % _bucketEvents.size()];
            chk(timedEventRecord(evSide, _gradStream), "event record");
        }
    }
    float* buf = _gradArena.p + _sentUpTo;
    const size_t count = end - _sentUpTo;
    _commSub.push([this, evMain, evSide, tickMain, tickSide, gateMain, gateSide, buf, count] {
        if (gateMain) chk(mcn_stream_gate(_mainToComm, gateMain, _gateTimeoutMs, _commStream), "stream gate (bucket)");
        if (gateSide) chk(mcn_stream_gate(_sideToComm, gateSide, _gateTimeoutMs, _commStream), "stream gate (bucket)");
        if (evMain) chk(timedStreamWait(_commStream, evMain), "wait event");
        if (evSide) chk(timedStreamWait(_commStream, evSide), "wait event");
        if (tickMain) awaitTick(tickMain, _tickFlag);
        if (tickSide) awaitTick(tickSide, _tickFlagSide);
        const auto t0 = std::chrono::steady_clock::now();
        void *e0 = nullptr, *e1 = nullptr;
        if (_commTiming) {
            e0 = commEvent();
            e1 = commEvent();
            chk(mcn_event_record(e0, _commStream), "event record");
        }
        chk(mcn_allreduce_sum(_comm, buf, count, _commStream), "all-reduce");
        if (_commTiming) {
            chk(mcn_event_record(e1, _commStream), "event record");
            _commSpans.push_back({0, e0, e1, count});
        }
        logComm(0, (size_t)(buf - _gradArena.p), count);
        _rcclCallNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        _rcclCalls++;
    });
    _sentUpTo = end;
    _commPending = true;
}

void ConvNet::waitComm() {
    SiteTimer st("waitComm");
    std::lock_guard<std::mutex> lk(_commMutex);
    if (_commPending) {
        if (commGatesUsable()) {  // the exchange thread signals behind its last collective; the gate is queued at once
            const int seq = nextCommGateSeq();
            _commSub.push([this, seq] { t_stream = 2; signalCommGate(seq); });
            awaitCommGate(seq);
            _commSub.wait(_commSub.pushed());  // (the segment flags below are the exchange closures' until they have run)
        } else {
            _commSub.wait(_commSub.pushed());  // every exchange of this step has been issued
            chk(timedEventRecord(_evComm, _commStream), "event record");
            chk(timedStreamWait(_stream, _evComm), "wait event");
        }
        _commPending = false;
    }
    _sentUpTo = 0;
    for (Segment& s : _segments) { s.ready = s.byMain = false; s.reported = 0; }
}

void ConvNet::allGather2(const float* sendA, float* recvA, const float* sendB, float* recvB, size_t countPerRank) {
    std::lock_guard<std::mutex> lk(_commMutex);
    // both gathers in one fused RCCL launch on the exchange stream; consumers call waitGather() when they need the result
    void* ev = _bucketEvents[_bucketNext++ % _bucketEvents.size()];
    chk(timedEventRecord(ev, stream()), "event record");
    _gatherTicket = _commSub.push([this, ev, sendA, recvA, sendB, recvB, countPerRank] {
        chk(mcn_stream_wait_event(_commStream, ev), "wait event");
        chk(mcn_comm_group_begin(), "group begin");
        chk(mcn_allgather(_comm, sendA, recvA, countPerRank, _commStream), "all-gather");
        chk(mcn_allgather(_comm, sendB, recvB, countPerRank, _commStream), "all-gather");
        chk(mcn_comm_group_end(), "group end");
        logComm(1, 0, countPerRank);
        logComm(1, 0, countPerRank);
        chk(mcn_event_record(_evGather, _commStream), "event record");
    });
}

void ConvNet::waitGather() {  // on the stream the caller is enqueueing on
    _commSub.wait(_gatherTicket);  // the exchange thread has recorded _evGather
    chk(timedStreamWait(stream(), _evGather), "wait event");
}

// ---- per-operation event timing -----------------------------------------------------------------
ProfScope::ProfScope(ConvNet* net, const std::string& layer, const char* op, int idx) : _net(net->profiling() ? net : nullptr) {
    if (_net) _net->profBegin(idx >= 0 ? layer + "." + op + "[" + std::to_string(idx) + "]" : layer + "." + op);
}
ProfScope::~ProfScope() {
    if (_net) _net->profEnd();
}

void* ConvNet::profEvent() {
    if (_eventsUsed == _eventPool.size()) {
        void* e = nullptr;
        chk(mcn_event_create(&e), "event create");
        _eventPool.push_back(e);
    }
    return _eventPool[_eventsUsed++];
}

void ConvNet::profBegin(const std::string& label) {
    auto it = _profLabelIdx.find(label);
    if (it == _profLabelIdx.end()) {
        it = _profLabelIdx.emplace(label, (int)_profLabels.size()).first;
        _profLabels.push_back(label);
    }
    ProfSpan s{it->second, profEvent(), nullptr, mcn_launch_count()};
    chk(timedEventRecord(s.e0, _stream), "event record");
    _profSpans.push_back(s);
}

void ConvNet::profEnd() {
    // spans nest at most one deep in practice; close the innermost open one
    for (size_t i = _profSpans.size(); i-- > 0;)
        if (!_profSpans[i].e1) {
            if (!_phases && mcn_launch_count() == _profSpans[i].launches0 && i + 1 == _profSpans.size()) {
                _profSpans.pop_back();  // nothing was launched inside (an operation fused into its neighbour): not a kernel
                return;
            }
            _profSpans[i].e1 = profEvent();
            chk(timedEventRecord(_profSpans[i].e1, _stream), "event record");
            return;
        }
}

void ConvNet::profCollect(std::map<std::string, ProfEntry>& into) {
    sync();
    for (const ProfSpan& s : _profSpans) {
        if (!s.e1) continue;
        float ms = 0;
        chk(mcn_event_elapsed_ms(&ms, s.e0, s.e1), "event elapsed");
        ProfEntry& e = into[_profLabels[s.label]];
        e.count++;
        e.ms += ms;
    }
    _profSpans.clear();
    _eventsUsed = 0;
}

// ---- costs --------------------------------------------------------------------------------------
void ConvNet::zeroCosts() {
    joinGradWork();  // cost kernels accumulate on the side stream
    chk(mcn_memset(_costBuf.p, 0, _costBuf.cap * 4, _stream), "zero costs");
}

void ConvNet::readCosts(Cost& into, long numCases) {
    joinGradWork();
    if (!_costHost.empty()) chk(mcn_d2h(_costHost.data(), _costBuf.p, _costHost.size() * 8, _stream), "d2h costs");
    sync();
    int k = 0;
    for (CostLayer* c : _costs) {
        auto& v = into.values[c->name()];
        if (v.empty()) into.names.push_back(c->name());
        v.resize(c->numCostValues(), 0.0);
        for (int j = 0; j < c->numCostValues(); j++) v[j] += _costHost[k++];
        into.coeffs[c->name()] = c->coeff();
    }
    into.numCases += numCases;
}

double ConvNet::costValue() {
    Cost c;
    readCosts(c, _numCases);
    return c.value();
}

// ---- gradient checking (convnet.cu:257-330) -----------------------------------------------------
void ConvNet::checkGradients(std::vector<GradCheckResult>& out, long maxElems, bool central) {
    if (dataParallel()) throw NetError("gradient checking runs on a single replica");
    // The reference exits after a gradient check (convnet.cu:255); here the model stays usable, so the momentum state the
    // PASS_GC backward overwrites -- it writes the raw gradient into `inc` of every matrix without a gradient buffer of
    // its own (layer.cu:378, :554) -- is set aside first and put back afterwards.
    std::vector<Weights*> all;
    for (auto& l : _layers) l->collectWeights(all);
    std::vector<std::pair<Weights*, std::unique_ptr<DBuf>>> saved;
    for (Weights* w : all)
        if (w->getGrad(false) == w->inc()) {
            saved.emplace_back(w, std::unique_ptr<DBuf>(new DBuf()));
            saved.back().second->ensure(w->size());
            chk(mcn_d2d(saved.back().second->p, w->inc(), w->size() * 4, _stream), "d2d inc");
        }
    auto restore = [&] {
        joinGradWork();
        for (auto& sv : saved) chk(mcn_d2d(sv.first->inc(), sv.second->p, sv.first->size() * 4, _stream), "d2d inc");
        for (auto& l : _layers) l->resetUpdateCounters();  // (numUpdates back to 0: the next training pass starts a fresh gradient)
        sync();
    };
    try {
        zeroCosts();
        fprop(0, PASS_GC);
        _baseErr = costValue();
        bprop(PASS_GC);
        for (auto& l : _layers) l->checkGradients(out, maxElems, central);
    } catch (...) {
        restore();
        throw;
    }
    restore();
    zeroCosts();
}

}  // namespace mcn
