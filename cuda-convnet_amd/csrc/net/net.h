// Layer graph + device-owning worker thread of the `_ConvNet` extension.
//
// Host-side counterpart of the reference's layer.cu / convnet.cu / worker.cu /
// data.cu / cost.cu / weights.cuh, re-typed onto the C ABI of include/mi_convnet.h:
// every device operation goes through an mcn_* entry point (no HIP call here, no
// PyObject either -- see params.h).  Scheduling semantics (received-input counters,
// activity / gradient buffer sharing, scaleTargets rules, numUpdates) follow the
// reference line by line in meaning (cited at each method); what is new is
//   * fused epilogues picked at graph-build time (bias + neuron inside filterActs /
//     the FC GEMM, relu-gradient inside max-pool undo, single-pass softmax and SGD),
//   * cost accumulation in device doubles, read back once per data batch,
//   * one workspace / one gradient arena allocated up front (no allocation on the step path),
//   * data-parallel replicas: gradient buckets all-reduced over RCCL on a side stream
//     while backward continues; L2-SN margin searched over the all-gathered pairs.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <exception>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "params.h"

struct mcn_sgd_item;  // include/mi_convnet.h

namespace mcn {

enum PassType { PASS_TRAIN = 0, PASS_TEST = 1, PASS_GC = 2 };  // util.cuh:43

struct NetError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
void chk(int rc, const char* what);
extern thread_local double g_eventCallMs;
extern thread_local double g_blockedMs;  // this thread waiting for another submitting thread (Submitter::wait)
int timedEventRecord(void* event, void* stream);
int timedStreamWait(void* stream, void* event);  // throws NetError carrying mcn_last_error()

// growable device float buffer
class DBuf {
   public:
    float* p = nullptr;
    size_t cap = 0;
    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf();
    float* ensure(size_t n);
};

class ConvNet;
class Layer;
class CostLayer;

// weights.cuh:41-198
class Weights {
   public:
    Weights(const HostMat& hW, const HostMat& hInc, float eps, float wc, float mom, bool useGrad, bool devFortran);
    Weights(Weights& src, float eps);  // shared weights, weights.cuh:59-68
    size_t size() const { return _hW.size(); }
    long rows() const { return _hW.rows; }
    long cols() const { return _hW.cols; }
    float* W() { return _src ? _src->_W.p : _W.p; }
    float* inc() { return _src ? _src->_inc.p : _inc.p; }
    float* gradBuf() { return _src ? _src->_grad : _grad; }
    float* getGrad(bool dp) { return (_useGrad || dp) ? gradBuf() : inc(); }  // weights.cuh:103-106
    void setGradBuf(float* g) { _grad = g; }
    // the fused update (ConvNet::fusedTail): this matrix's gradient arrives as `count` partial pieces the update itself sums
    // (kind 1: module-chunk slabs of a conv weight gradient; kind 2: stage-1 row sums of a bias gradient, scaled by `scale`)
    struct Partials { const float* data = nullptr; int count = 0, kind = 0; float scale = 1.f; };
    Partials partials;
    bool owner() const { return _src == nullptr; }
    bool useGrad() const { return _useGrad; }
    float eps() const { return _eps; }
    float mom() const { return _mom; }
    float wc() const { return _wc; }
    int numUpdates() const { return _src ? _src->numUpdates() : _numUpdates; }
    int incNumUpdates() { return _src ? _src->incNumUpdates() : _numUpdates++; }
    void copyToGPU(void* stream);
    void copyToCPU(void* stream);
    void update(bool dp, void* stream);  // weights.cuh:150-163, one fused kernel
    bool planItem(bool dp, ::mcn_sgd_item& it);  // record for the all-matrices update; false: nothing to update
    void updated() { if (!_src && _eps > 0) _numUpdates = 0; }  // weights.cuh:152,161: only an owner that updates resets the counter
    void resetUpdates() { if (!_src) _numUpdates = 0; }         // (after a failed pass / a gradient check)
    // host <-> device image conversion for gradient checking
    void hostImage(std::vector<float>& out, void* stream);        // logical (rows, cols) row-major
    void setElement(long r, long c, float v, void* stream);
    size_t memIndex(long r, long c) const { return _devFortran ? (size_t)c * _hW.rows + r : (size_t)r * _hW.cols + c; }

   private:
    void upload(const HostMat& h, float* dst, void* stream);
    void download(const HostMat& h, const float* src, void* stream);
    HostMat _hW, _hInc;
    DBuf _W, _inc;
    float* _grad = nullptr;
    float _eps, _wc, _mom;
    bool _useGrad, _devFortran;
    int _numUpdates = 0;
    Weights* _src = nullptr;
};

class Layer {
   public:
    Layer(ConvNet* net, const Params& p);
    virtual ~Layer() {}
    // layer.cu:75-114
    void fpropFromPrev(PassType pt);
    void fprop(const std::vector<float*>& inputs, PassType pt);
    // layer.cu:116-158
    virtual void bprop(PassType pt);
    void bpropWith(float* v, PassType pt);
    void reset() { _rcvdF = _rcvdB = 0; _seeded = false; }
    virtual void resetPassFlags() {}  // forget what a pass that did not finish left behind (ConvNet::resetAfterError)
    // A cost layer may deliver its gradient into this layer's gradient matrix ahead of the backward pass, on the side
    // stream (L2-SN: as soon as its cost statistics exist).  That counts as the first received gradient (everyone after it
    // accumulates), and whoever touches the matrix first during backward waits for the event.
    void seedGrad(void* event, long sideTicket) { _seeded = true; _seedEvent = event; _seedTicket = sideTicket; _rcvdB++; }
    void awaitSeed();
    void addPrev(Layer* l) { _prev.push_back(l); }
    void addNext(Layer* l) { _next.push_back(l); _numGradProducersNext += l->isGradProducer(); }
    bool isGradConsumer();
    virtual bool isGradProducer() { return true; }
    float* getActs();
    float* getActsGrad();
    virtual size_t actsElems(int n) const { return (size_t)_outputs * n; }
    virtual void allocate(int maxCases);  // device buffers for up to maxCases columns
    virtual size_t workspaceBytes(int maxCases) { return 0; }
    virtual void updateWeights() {}
    virtual void weightsUpdated() {}
    virtual void resetUpdateCounters() {}
    virtual void copyToCPU() {}
    virtual void copyToGPU() {}
    virtual void checkGradients(std::vector<struct GradCheckResult>& out, long maxElems, bool central) {}
    virtual void collectWeights(std::vector<Weights*>& out) {}
    virtual bool planFusedTail(int numCases) { return true; }  // fill Weights::partials for a pass of numCases columns; false: not expressible

    const std::string& name() const { return _name; }
    const std::string& type() const { return _type; }
    int outputs() const { return _outputs; }
    std::vector<Layer*>& prev() { return _prev; }
    std::vector<Layer*>& next() { return _next; }
    int rcvdB() const { return _rcvdB; }
    int incRcvdB() { return ++_rcvdB; }
    int actsTarget() const { return _actsTarget; }
    int actsGradTarget() const { return _actsGradTarget; }
    int numGradProducersNext() const { return _numGradProducersNext; }

   protected:
    virtual void fpropActs(int inpIdx, float scaleTargets, PassType pt) {}
    virtual void bpropActs(float* v, int inpIdx, float scaleTargets, PassType pt) {}
    virtual void bpropCommon(float* v, PassType pt) {}
    friend class ConvNet;

    ConvNet* _net;
    std::string _name, _type;
    int _outputs = 0;
    std::vector<Layer*> _prev, _next;
    std::vector<float*> _inputs;
    int _actsTarget, _actsGradTarget;
    bool _gradConsumer, _foundGradConsumers = false;
    int _numGradProducersNext = 0, _rcvdF = 0, _rcvdB = 0;
    DBuf _ownActs, _ownActsGrad;
    bool _seeded = false;
    void* _seedEvent = nullptr;
    long _seedTicket = 0;
    float* _acts = nullptr;  // current activity matrix (own buffer, shared input, or data slice)
};

struct NeuronSpec {
    int type = 0;
    float a = 0, b = 0;
    bool needsInputs() const;  // gradient reads the neuron input (abs, square)
    bool fusableGrad() const;  // gradient needs only (dE/dy, y)
};

class NeuronLayer : public Layer {
   public:
    NeuronLayer(ConvNet* net, const Params& p);
    NeuronSpec spec;
    bool fwdFused = false;  // activation applied in the producer's epilogue
    bool bwdFused = false;  // gradient applied in the consumer's (max-pool undo) epilogue
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
};

class WeightLayer : public Layer {
   public:
    WeightLayer(ConvNet* net, const Params& p, bool useGrad, bool fortran);
    void updateWeights() override;
    void weightsUpdated() override;
    void resetUpdateCounters() override;
    bool sharesWeights() const;  // holds a matrix that another Weights object owns
    bool queuesBiasSum(bool fusedTail) const { return _biases->eps() > 0 && !(_biasInWgrad && fusedTail); }
    void copyToCPU() override;
    void copyToGPU() override;
    void checkGradients(std::vector<GradCheckResult>& out, long maxElems, bool central) override;
    void collectWeights(std::vector<Weights*>& out) override;
    bool planFusedTail(int numCases) override;
    Weights& weights(int i) { return *_weights[i]; }
    int numWeights() const { return (int)_weights.size(); }
    NeuronLayer* fusedNeuron = nullptr;

   protected:
    void bpropCommon(float* v, PassType pt) override;
    virtual void bpropBiases(float* v, PassType pt) = 0;
    virtual void biasSumShape(int numCases, int& rows, int& cols) const = 0;  // bpropBiases as a row sum of v (rows x cols)
    virtual void bpropWeights(float* v, int inpIdx, PassType pt) = 0;
    float gradScale(float eps, PassType pt) const;  // eps / (numCases * worldSize), 1 for PASS_GC
    std::vector<std::unique_ptr<Weights>> _weights;
    std::unique_ptr<Weights> _biases;
    DBuf _biasParts;  // stage-1 row sums of the bias gradient (fused update)
    bool _biasInWgrad = false;  // the weight-gradient launch of input 0 also leaves the bias gradient, in pieces (conv, strip form)
    float _wStep = 0.001f, _bStep = 0.002f;  // layer.cu:294-295
};

class SoftmaxLayer;
class FCLayer : public WeightLayer {
   public:
    FCLayer(ConvNet* net, const Params& p);
    size_t workspaceBytes(int maxCases) override;
    SoftmaxLayer* softmaxBelow = nullptr;  // the only consumer is a softmax: the forward product leaves its stage-1 partials
    // several inputs (DeepID's fc5): one launch per direction for all of them (mcn_fc_fprop_multi / mcn_fc_bprop_acts_multi)
    bool multiShapes = false;                    // every input size a multiple of 16, at most 4 distinct input layers
    std::vector<NeuronLayer*> neuronAbove;       // per input: neuron layer whose gradient the backward-data product applies
    void planMulti();
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    void bpropBiases(float* v, PassType) override;
    void biasSumShape(int numCases, int& rows, int& cols) const override { rows = _outputs; cols = numCases; }
    void bpropWeights(float* v, int, PassType) override;
    bool _fwdAllDone = false, _wgradAllDone = false;
    std::vector<char> _bwdDone;
};

class LocalLayer : public WeightLayer {
   public:
    LocalLayer(ConvNet* net, const Params& p, bool useGrad, bool conv);
    size_t workspaceBytes(int maxCases) override;
    bool planFusedTail(int numCases) override;
    class PoolLayer* fusedPool = nullptr;  // max pool over this layer's (neuron's) output written by the forward launch itself
    bool isConv() const { return _conv; }
    int filters() const { return _filters; }
    int modulesY() const { return _modulesY; }
    int modulesX() const { return _modulesX; }
   protected:
    std::vector<std::unique_ptr<DBuf>> _wgradParts;  // conv: module-chunk partials of each weight gradient (fused update)
    std::vector<int> _wgradChunks;
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    void bpropBiases(float* v, PassType) override;
    void biasSumShape(int numCases, int& rows, int& cols) const override {
        if (_conv && _sharedBiases) { rows = _filters; cols = _modules * numCases; }
        else { rows = _filters * _modules; cols = numCases; }
    }
    void bpropWeights(float* v, int, PassType) override;
    bool _conv;
    std::vector<long> _padding, _stride, _filterSize, _channels, _imgSizeY, _imgSizeX, _groups, _filterChannels;
    int _filters, _modulesY, _modulesX, _modules, _partialSum = 0, _sharedBiases = 0;
    // random-sparse connectivity (layer.cu:410-431): per input the channel table, its inverse, the re-ordered input kept from
    // forward for the weight gradient, and the backward-data product before it is folded back onto the image channels
    struct Sparse {
        std::vector<int> conns;
        DBuf dConns, dInverse, gathered, gradTmp;
    };
    std::vector<std::unique_ptr<Sparse>> _sparse;  // null: dense input
    bool sparse(int i) const { return i < (int)_sparse.size() && _sparse[i]; }
    int wideChannels(int i) const { return (int)(_groups[i] * _filterChannels[i]); }
   public:
    void allocate(int maxCases) override;
    void copyToGPU() override;
};

class SoftmaxLayer : public Layer {
   public:
    using Layer::Layer;
    size_t workspaceBytes(int maxCases) override;
    void allocate(int maxCases) override;
    DBuf part;               // stage-1 partials written by the producing FC layer's forward product
    size_t partBytes = 0;
    bool partReady = false;  // ... for the pass in flight
    void resetPassFlags() override { partReady = false; _gradFused = false; }
    DBuf costState;          // tie counters + ticket of mcn_softmax_logreg_cost (zero between calls)
    size_t costStateBytes = 0;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    bool _gradFused = false;  // this pass wrote the logreg gradient during forward
};

class EltwiseSumLayer : public Layer {
   public:
    EltwiseSumLayer(ConvNet* net, const Params& p);
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    std::vector<double> _coeffs;
};

class EltwiseMaxLayer : public Layer {
   public:
    using Layer::Layer;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
};

class DataLayer : public Layer {
   public:
    DataLayer(ConvNet* net, const Params& p);
    bool isGradProducer() override { return false; }
    void setSlice(float* slice) { _acts = slice; }
    void fpropData(float* slice, PassType pt);
    void allocate(int) override {}
    int dataIdx;
};

class PoolLayer : public Layer {
   public:
    PoolLayer(ConvNet* net, const Params& p);
    NeuronLayer* fusedNeuronGrad = nullptr;  // relu(-like) below a max pool: gradient applied in the undo kernel
    bool isMax() const { return _max; }
    // the conv / local layer above may send its gradient straight through this pool (mcn_img_acts_max_undo) when it is the
    // last of the pool's consumers to report: undoFusable = the windows tile the maps, this pool alone feeds the layer below
    bool undoFusable(int numCases, int channels, int outY, int outX) const;
    bool undoDoneAbove = false;  // ... for the pass in flight: bpropActs has nothing left to do
    // the conv layer below (through its fused neuron) may write this pool's output in its own launch (mcn_filter_acts_bias_neuron_pool):
    // quadFusable = max over the 2 x 2 / stride 2 / start 0 windows that tile a (modulesY, modulesX) map of `filters` channels
    bool quadFusable(int filters, int modulesY, int modulesX) const;
    float* ownActs() { return _actsTarget < 0 ? _ownActs.p : nullptr; }
    bool fwdDoneBelow = false;   // ... for the pass in flight: fpropActs has nothing left to do
    void resetPassFlags() override { undoDoneAbove = false; fwdDoneBelow = false; }
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    int _channels, _sizeX, _start, _stride, _outputsY, _outputsX, _imgSizeY, _imgSizeX;
    bool _max;
};

// ---- image-op layers (layer.cu:743-840, :1124-1143)
class ImageOpLayer : public Layer {
   public:
    ImageOpLayer(ConvNet* net, const Params& p);
   protected:
    int _channels, _imgSizeY, _imgSizeX;
};

class NailbedLayer : public ImageOpLayer {
   public:
    NailbedLayer(ConvNet* net, const Params& p);
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    int _start, _stride, _outputsY, _outputsX;
};

class GaussianBlurLayer : public ImageOpLayer {
   public:
    GaussianBlurLayer(ConvNet* net, const Params& p);
    void allocate(int maxCases) override;
    void copyToGPU() override;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    HostMat _hFilter;
    DBuf _filter, _tmp;  // _tmp: the horizontally blurred image (the reference blurs its output in place, layer.cu:770-771)
};

class ResizeLayer : public ImageOpLayer {
   public:
    ResizeLayer(ConvNet* net, const Params& p);
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float*, int, float, PassType) override;
    int _tgtSize;
    float _scale;
};

class ColorLayer : public ImageOpLayer {  // RGBToYUVLayer / RGBToLABLayer
   public:
    ColorLayer(ConvNet* net, const Params& p, bool lab);
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float*, int, float, PassType) override;
    bool _lab;
    int _center = 0;
};

class ShiftLayer : public ImageOpLayer {
   public:
    ShiftLayer(ConvNet* net, const Params& p);
    void allocate(int maxCases) override;
    void copyToGPU() override;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float*, int, float, PassType) override;
    HostMat _hFilter;
    DBuf _filter;
};

class ResponseNormLayer : public Layer {
   public:
    ResponseNormLayer(ConvNet* net, const Params& p, bool cross);
    void allocate(int maxCases) override;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    int _channels, _size, _imgSizeY, _imgSizeX, _blocked;
    float _scale, _pow;
    bool _cross;
    DBuf _denoms;
};

class ContrastNormLayer : public ResponseNormLayer {  // layer.cu:893-913
   public:
    ContrastNormLayer(ConvNet* net, const Params& p) : ResponseNormLayer(net, p, false) {}
    void allocate(int maxCases) override;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    DBuf _meanDiffs;
};

class CostLayer : public Layer {
   public:
    CostLayer(ConvNet* net, const Params& p);
    void bprop(PassType pt) override;  // layer.cu:929-933
    bool isGradProducer() override { return _coeff != 0; }
    float coeff() const { return _coeff; }
    virtual int numCostValues() const { return 1; }
    double* dCost = nullptr;  // device accumulators, numCostValues() doubles
    void allocate(int maxCases) override;
   protected:
    float _coeff;
    DBuf _aux;
};

class LogregCostLayer : public CostLayer {
   public:
    using CostLayer::CostLayer;
    int numCostValues() const override { return 2; }
    size_t workspaceBytes(int maxCases) override;
    size_t actsElems(int n) const override { return (size_t)n; }
    // a softmax below may finish this cost in its own launch (mcn_softmax_logreg_cost): the buffers it fills
    float* logProbsBuf() { return _ownActs.p; }
    float* correctBuf() { return _aux.p; }
    bool costDoneBySoftmax = false;  // ... for the pass in flight
    void resetPassFlags() override { costDoneBySoftmax = false; }
    void* softmaxState = nullptr;    // that softmax's counters (mcn_softmax_logreg_cost_finish)
    size_t softmaxStateBytes = 0;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
};

class SumOfSquaresCostLayer : public CostLayer {
   public:
    using CostLayer::CostLayer;
    size_t actsElems(int n) const override { return (size_t)_prev[0]->outputs() * n; }
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
};

class L2SNCostLayer : public CostLayer {
   public:
    L2SNCostLayer(ConvNet* net, const Params& p);
    ~L2SNCostLayer() override;
    void bprop(PassType pt) override;
    size_t actsElems(int n) const override { return (size_t)n; }
    void allocate(int maxCases) override;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    float _m;
    DBuf _pairStats, _info;  // what the cost launch keeps for the gradient: (dists, labels) and {margin, chosen pair, 0, 0}
    DBuf _gathered;          // data-parallel: the all-gathered margin candidates
    void* _evReady = nullptr;  // recorded after the cost launch, on whichever stream ran it
    long _readyTicket = 0;
    int _readyGateSeq = 0;  // != 0: the statistics are announced by a stream gate signal with this number, not by _evReady
    int _readyCommGateSeq = 0;  // ... on the exchange stream (data parallel)
    bool _readyOnExchange = false;
    int _extCount = 0;
    bool _earlyGrad = false;  // this pass already sent its gradient (forward, side stream)
    bool _gradFolded = false; // this pass's gradient was added by the classifier layer's backward-data launch
   public:
    void resetPassFlags() override { _earlyGrad = false; _gradFolded = false; _readyTicket = 0; _readyGateSeq = 0; _readyCommGateSeq = 0; }
    // The layer that sends `target` its FIRST gradient of the pass may add this layer's gradient -- the next one -- in the same
    // launch (mcn_fc_bprop_acts_l2sn): possible when the statistics are announced by a stream gate, which that launch waits for
    struct Fold { const float *data, *stats, *info; float coeff; int gateSeq; };
    bool foldable(Layer* target, Fold& f);
    void gradFolded() { _gradFolded = true; _readyGateSeq = 0; }
   protected:
};

class Fisher2CostLayer : public CostLayer {  // layer.cu:1017-1047
   public:
    Fisher2CostLayer(ConvNet* net, const Params& p);
    size_t actsElems(int) const override { return 4; }
    size_t workspaceBytes(int maxCases) override;
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
    float _alpha;
};

class L2regCostLayer : public CostLayer {
   public:
    using CostLayer::CostLayer;
    size_t actsElems(int n) const override { return (size_t)n; }
   protected:
    void fpropActs(int, float, PassType) override;
    void bpropActs(float* v, int, float, PassType) override;
};

// ---- data (data.cu) -----------------------------------------------------------------------
struct CPUData {
    std::vector<HostMat> mats;                 // (dims, numCases) C-contiguous
    std::shared_ptr<void> keepAlive;           // whatever owns the host buffers
    long numCases() const { return mats.empty() ? 0 : mats[0].cols; }
};

class DataProvider {
   public:
    explicit DataProvider(ConvNet* net, int minibatchSize) : _net(net), _minibatchSize(minibatchSize) {}
    ~DataProvider();
    // upload `d` into the spare slot on the copy stream while the current batch trains; adopt() swaps it in
    void prefetch(const CPUData& d);
    bool prefetchPending() const { return _nextTicket.load() != 0; }
    // caller's thread, when it queues the work item that will adopt the pending prefetch: a following prefetch() must not
    // touch the spare slot before that adoption has happened
    void claimPrefetch() { if (prefetchPending()) _adoptQueued.store(true); }
    void adoptPrefetched();
    void waitPrefetchUploaded();  // host-side: the transfer has completed (the caller may release the host arrays)
    // batch too large to keep resident: minibatch slices uploaded two ahead of the step that reads them
    bool streamed() const { return !_onDevice && !_host.mats.empty(); }
    void streamBegin();
    void streamSlice(int miniIdx, std::vector<float*>& out);   // waits for slice miniIdx, hands out its buffers
    void streamSliceDone(int miniIdx);                          // the step reading it has been issued: refill the buffer
    void setData(const CPUData& d);            // data.cu:47-65 (uploads the whole batch when it fits)
    void clearHost() { _host = CPUData(); }
    void getSlice(long start, long end, std::vector<float*>& out);  // data.cu:72-98
    int numMinibatches() const { return (int)((_numCases + _minibatchSize - 1) / _minibatchSize); }
    int minibatchSize() const { return _minibatchSize; }
    long numCases() const { return _numCases; }
    bool resident() const { return _onDevice; }
    bool sliceIsPointer(long start, long end) const;  // getSlice would hand out pointers into the tiled batch (no copy, no sync)
   private:
    ConvNet* _net;
    int _minibatchSize;
    CPUData _host;
    long _numCases = 0;
    bool _onDevice = false;
    std::vector<long> _dims;
    std::vector<std::unique_ptr<DBuf>> _dev, _mini, _tiled;  // whole batch; one copied slice; all slices, minibatch-major
    // spare slot (prefetch)
    std::vector<std::unique_ptr<DBuf>> _devNext, _tiledNext;
    std::vector<long> _dimsNext;
    long _numCasesNext = 0;
    std::atomic<long> _nextTicket{0};
    std::atomic<bool> _adoptQueued{false};
    void* _evNext = nullptr;      // recorded on the copy stream after the spare slot's upload + tiling
    void* _evSlotFree = nullptr;  // recorded on the compute stream when the batch that used a slot has been fully issued
    std::atomic<bool> _slotFreeRecorded{false};
    // streamed minibatches (double buffer)
    std::vector<std::unique_ptr<DBuf>> _stream2[2];
    void* _evSliceReady[2] = {nullptr, nullptr};
    void* _evSliceFree[2] = {nullptr, nullptr};
    long _sliceTicket[2] = {0, 0};
    bool _sliceFreeRecorded[2] = {false, false};
    void requestSlice(int miniIdx);
   public:
};

// ---- costs (cost.cu) ------------------------------------------------------------------------
struct Cost {
    long numCases = 0;
    std::vector<std::string> names;            // cfg order
    std::map<std::string, std::vector<double>> values;
    std::map<std::string, double> coeffs;
    double value() const;                      // cost.cu:64-70
};

struct GradCheckResult {
    std::string name;
    double analNorm, numNorm, relErr;
    bool fail;
};

struct ProfEntry {
    long count = 0;
    double ms = 0;
};

struct WorkResult {
    enum Type { BATCH_DONE, SYNC_DONE, GRADCHECK_DONE, INIT_DONE, ERROR } type = BATCH_DONE;
    Cost cost;
    std::vector<GradCheckResult> gradChecks;
    std::map<std::string, ProfEntry> profile;  // per-operation device time when profiling was on
    std::string error;
};

template <class T>
class Queue {  // queue.h:36-110, with the wait in a loop (the reference's `if` admits spurious wake-ups)
   public:
    void enqueue(T v) {
        { std::lock_guard<std::mutex> g(_m); _q.push_back(std::move(v)); }
        _cv.notify_one();
    }
    T dequeue() {
        std::unique_lock<std::mutex> g(_m);
        _cv.wait(g, [&] { return !_q.empty(); });
        T v = std::move(_q.front());
        _q.pop_front();
        return v;
    }
   private:
    std::mutex _m;
    std::condition_variable _cv;
    std::deque<T> _q;
};

using Work = std::function<void(ConvNet&)>;

// RAII bracket around one layer operation; a no-op unless the net is profiling
class ProfScope {
   public:
    ProfScope(ConvNet* net, const std::string& layer, const char* op, int idx = -1);
    ~ProfScope();
   private:
    ConvNet* _net;
};

// A submitting thread: runs closures in order (each issues runtime calls on one stream).  HIP scales across host
// threads that feed different streams (tools/sync_cost.hip), and a cross-stream wait or an RCCL call costs 10-60 us of
// host time -- too much for the thread that feeds the critical stream.
class Submitter {
   public:
    ~Submitter() { stop(); }
    void start(int deviceID, bool* sideFlag = nullptr);
    void stop();
    bool running() const { return _thread.joinable(); }
    long push(std::function<void()> fn);  // returns the ticket of this closure
    long pushed() const { return _pushed.load(std::memory_order_relaxed); }
    void wait(long ticket);               // host-side: until the closure with that ticket has run; rethrows its error
   private:
    void loop(int deviceID);
    std::thread _thread;
    std::mutex _mutex;
    std::condition_variable _cv;
    std::deque<std::function<void()>> _queue;
    std::atomic<long> _pushed{0}, _done{0};
    bool _stop = false;
    std::exception_ptr _error;
    bool* _flag = nullptr;
};

// Host -> device transfers on their own thread and stream (f1 of SURVEY 8f; the reference copies on the compute thread,
// data.cu:72-98 / nvmatrix.cu:141-154, and overlaps only the Python-side file loading, gpumodel.py:148-151).  Pageable
// sources go through two page-locked staging chunks (the CPU copy of chunk k+1 overlaps the DMA of chunk k); page-locked
// sources (pinnedArray) are read by DMA directly.
class Uploader {
   public:
    ~Uploader();
    void start(int deviceID);
    void stop();
    bool running() const { return _thread.running(); }
    void* stream() const { return _stream; }
    long push(std::function<void()> fn) { return _thread.push(std::move(fn)); }
    void wait(long ticket) { _thread.wait(ticket); }
    // on the uploader thread only:
    void copy(float* dst, const float* src, size_t bytes, bool pinned);
    void copy2d(float* dst, size_t dpitchBytes, const float* src, size_t spitchBytes, size_t widthBytes, size_t height, bool pinned);
   private:
    Submitter _thread;
    void* _stream = nullptr;
    void* _chunk[2] = {nullptr, nullptr};
    void* _chunkEv[2] = {nullptr, nullptr};
    bool _chunkUsed[2] = {false, false};
    int _next = 0;
    static constexpr size_t kChunkBytes = (size_t)32 << 20;
    void* acquireChunk(int& k);
};

class ConvNet {
   public:
    ConvNet(const std::vector<Params>& layers, int minibatchSize, int deviceID);
    ~ConvNet();
    void start();                               // spawns the device-owning thread (convnet.cu:147-156)
    void stop();
    Queue<Work>& workQueue() { return _work; }
    Queue<WorkResult>& resultQueue() { return _results; }

    // ---- called on the worker thread -------------------------------------------------------
    void initDevice();                          // convnet.cu:139-145
    void initComm(int rank, int world, const char* id128, bool force);  // force: run the exchange path even with one rank
    void fprop(int miniIdx, PassType pt);       // convnet.cu:229-233
    void fpropSlice(long start, long end, PassType pt);
    void fpropCurrent(PassType pt);             // convnet.cu:213-219
    void bprop(PassType pt);                    // convnet.cu:206-211
    void updateWeights();                       // convnet.cu:190-194
    void copyToCPU();
    void zeroCosts();
    void readCosts(Cost& into, long numCases);  // adds the device accumulators into `into`
    double costValue();                         // convnet.cu:247-252 (blocking)
    void checkGradients(std::vector<GradCheckResult>& out, long maxElems, bool central);
    bool checkGradient(const std::string& name, float eps, Weights& w, std::vector<GradCheckResult>& out, long maxElems, bool central);
    void sync();

    // workers (worker.cu)
    void runTraining(const CPUData& data, bool test, bool useResident);
    void runFeatures(const CPUData& data, HostMat ftrs, int layerIdx);
    void runMultiview(const CPUData& data, int numViews, int logregIdx);
    void runGradCheck(const CPUData& data, long maxElems, bool central);
    void runLoad(const CPUData& data);
    // numSteps minibatches cycling over the resident batch (benchmark / profiling loop: no host data,
    // no allocation, one cost read-back at the end).  profile: bracket every layer operation with events.
    // profile 2: only the phases of a step (forward / backward-data chain / gradient tail / update), streams as in a normal run.
    void runSteps(int numSteps, bool test, int profile);
    // One training minibatch (fprop + bprop + update).  When the minibatch is a pointer into a resident, tiled batch the
    // whole step -- both streams, every fork and join -- is captured once per minibatch index into an executable graph and
    // replayed afterwards: one submission per step instead of ~45 launches, ~10 event operations and two thread hand-overs.
    void trainStep(int miniIdx);
    void dropStepGraphs();

    // ---- services for layers ---------------------------------------------------------------
    void* stream() const;                       // the stream layer operations are issued on right now (per submitting thread)
    void* mainStream() const { return _stream; }
    // Weight / bias gradients of a layer depend only on its incoming gradient, so they run on a side stream
    // next to the backward-data chain (joined before the update, before anything overwrites their inputs).
    // Work that nothing on the backward-data chain waits for (a layer's weight / bias gradients: units = 1; the
    // cost VALUE kernels: units = 0) is deferred and sent to the side stream two layers at a time -- a cross-stream
    // dependency costs ~12 us of host time, a launch ~3 -- and what is left when backward ends is split
    // between the side stream and the main stream.
    void deferSideWork(std::function<void()> fn, int units, bool hold = false);  // flushed to the side stream once _sideUnits are waiting (hold: not by this call)
    bool lastWeightLayerOfPass(WeightLayer* wl);
    bool flushesBiasBatch(WeightLayer* wl, bool last);
    // bias gradients of a backward pass, handed to ONE mcn_add_sum_multi by the last weight layer's bias piece (single GPU)
    bool biasBatching() const;
    void queueBiasSum(float* tgt, const float* mat, int rows, int cols, float scaleSum, float* partsOut);
    // The tail of a single-GPU training step without its latency-only launches: the module-chunk reductions of the conv
    // weight gradients and the second stage of the long bias row sums happen inside the one update launch
    // (mcn_sgd_plan_build_fused); bit-identical to the separate launches.  fusedTailNow: this pass takes that path.
    bool fusedTailNow() const { return _fusedTailNow; }
    void flushBiasSums(int which = -1);  // -1: the calling thread's batch
#ifdef MCN_TEST_HOOKS
    void failNextBackwardAt(const std::string& layer) { std::lock_guard<std::mutex> lk(_failMutex); _failAt = layer; }  // (any thread)
    bool _failArmed = false;
#endif
    void commInfo(int& numRanks, int& version, int& isRccl);
    void earlyUpdate();
    void deferLateWork(std::function<void()> fn);  // main stream, at the end of the pass (joinGradWork)
    void finishSideWork(bool deferHostWait = false);   // end of backward: place the remaining work, then join
    void joinGradWork(bool deferHostWait = false, bool keepWeightGate = false);  // run whatever is still deferred on the main stream and wait for the side stream
    void beforeLayerFprop(Layer* l);
    void hostWaitSide();     // the host-side half of a join whose wait was deferred
    int numCases() const { return _numCases; }
    int worldSize() const { return _world; }
    int rank() const { return _rank; }
    bool dataParallel() const { return _dpOn; }
    void* workspace() const;
    size_t workspaceBytes() const { return _ws.cap * sizeof(float); }
    Layer& layer(int i) { return *_layers[i]; }
    int numLayers() const { return (int)_layers.size(); }
    DataProvider& dp() { return *_dp; }
    Uploader& uploader() { return _uploader; }
    void prefetchFromHost(const CPUData& d);  // any thread: hands the batch to the uploader (asynchronous)
    void waitPrefetchUploaded() { _dp->waitPrefetchUploaded(); }
    void gradReady(WeightLayer* l);             // data-parallel: layer's gradient segment is complete
    void allGather2(const float* sendA, float* recvA, const float* sendB, float* recvB, size_t countPerRank);  // starts on the exchange stream
    void waitGather();
    long forkSideWork(std::function<void()> fn);  // starts now, on the side stream; returns a ticket for waitSideSubmitted
    void waitSideSubmitted(long ticket);          // host-side: until the side thread has issued that work
    bool exchangeThreadUsable() const { return _dpOn && _gradStream && !_profiling && _commSub.running(); }
    long forkExchangeWork(std::function<void()> fn);  // on the exchange stream / thread (data parallel)
    void waitExchangeSubmitted(long ticket);
    void gatherPairsHere(const float* sendA, float* recvA, const float* sendB, float* recvB, size_t countPerRank);  // exchange or side thread, on its stream
    void joinExchange();
    void exchangeWorkJoined() { _exchangeWorkPending = false; }
    // data-parallel plan, for tests without 8 GPUs: the collectives in the order the exchange thread issued them
    // (kind 0 all-reduce over [offset, offset + count) of the gradient arena, kind 1 all-gather of count floats per rank)
    struct CommRec { int kind; size_t offset, count; };
    std::vector<CommRec> takeCommLog();
    struct SegmentInfo { std::string layer; size_t begin, end; };
    std::vector<SegmentInfo> gradSegments();
    size_t bucketFloats() const { return _bucketFloats; }
    // weightSource nets: every gradient piece on the side stream, in pass order (see WeightLayer::WeightLayer)
    void noteWeightSharing() { _hasSharing = true; }
    bool hasWeightSharing() const { return _hasSharing; }
    void setFusion(bool on) { _fusion = on; }
    bool fusion() const { return _fusion; }
    bool profiling() const { return _profiling; }
    void profBegin(const std::string& label);
    void profEnd();

   private:
    void run();
    void buildFusions();
    void allocate(int maxCases);
    void flushBuckets(bool force);
    void waitComm();

    std::vector<std::unique_ptr<Layer>> _layers;
    std::vector<DataLayer*> _dataLayers;
    std::vector<CostLayer*> _costs;
    std::unique_ptr<DataProvider> _dp;
    int _minibatchSize, _deviceID;
    int _numCases = 0;
    int _allocatedCases = 0;
    bool _fusion = true;
    bool _hasSharing = false;
    std::vector<float*> _curData;

    void* _stream = nullptr;
    void* _gradStream = nullptr;  // null: gradients stay on the main stream
    void* _evJoin = nullptr;
    bool _gradPending = false;
    // The side stream has its own submitting thread: a cross-stream dependency costs ~16 us of host time on this runtime
    // (tools/sync_cost.hip), most of it in the wait call, and two threads submit to two streams without slowing each
    // other down.  The main thread records the fork event and hands the closures over; the side thread waits for the
    // event on the side stream and issues them there.
    Submitter _side, _commSub;  // side stream and exchange stream
    Uploader _uploader;
    std::vector<void*> _forkEvents;
    size_t _forkNext = 0;
    // side stream -> main stream dependencies (the end-of-step join, the L2-SN statistics) as stream gates instead of event
    // waits (mcn_stream_gate: the waiting queue resumes one dispatch after the signal instead of 10-15 us after the event)
    int* _gate = nullptr;
    int* _forkGate = nullptr;  // main stream -> side stream: the fork ticks' device word; the side thread queues a gate on the side
                               // stream instead of polling the host word (it never blocks, and neither does the main thread for it)
    bool _forkGateOk = false;
    // data parallel: exchange stream -> main stream (all-reduced gradients, the L2-SN statistics) and the two compute
    // streams -> exchange stream (a bucket's gradients are complete), each direction with a word and a counter of its own
    int* _commGate = nullptr;      // signalled on the exchange stream (exchange thread), awaited on the main stream
    int _commGateSeq = 0;
    int* _mainToComm = nullptr;    // signalled on the main stream (main thread), awaited on the exchange stream
    int* _sideToComm = nullptr;    // signalled on the side stream (side thread), awaited on the exchange stream
    int _mainToCommSeq = 0, _sideToCommSeq = 0;
    bool _commGateOk = false;
   public:
    bool commGatesUsable() const { return _commGateOk && !_capturing && !g_noHostWaitFlag() && !_profiling; }
    int nextCommGateSeq() { return ++_commGateSeq; }
    void signalCommGate(int seq);  // exchange thread, on the exchange stream
    void awaitCommGate(int seq);   // queued on the main stream
   private:
    static bool g_noHostWaitFlag();
    int _gateSeq = 0;
    bool _gateOk = false;
    bool _sideWaitPending = false;
   public:
    bool gatesUsable() const;
    int* gateWord() const { return _gate; }          // the main / side stream gate (mcn_stream_gate), for launches that wait for it themselves
    int gateTimeoutMs() const { return _gateTimeoutMs; }
    int nextGateSeq() { return ++_gateSeq; }
    void signalGate(int seq);  // on the calling thread's stream (the side stream)
    void awaitGate(int seq);   // queued on the main stream
   private:
    int* _tickFlag = nullptr;  // main stream -> side thread forks as stream ticks (mcn_stream_tick) instead of events
    int _tickSeq = 0;
    int* _tickFlagSide = nullptr;  // side stream -> exchange thread (gradient buckets written by the side stream)
    int _tickSeqSide = 0;
    // host-side accounting of the side thread (runSteps reports it as host.side.*): ns inside launch calls, launches, ns polling
    // for fork ticks / events, ns running closures
    std::atomic<long> _sideLaunchNs{0}, _sideLaunches{0}, _sidePollNs{0}, _sideWorkNs{0};
    bool _commTicks = false;       // the exchange thread polls ticks instead of queueing event waits
    void awaitTick(int tick, volatile int* flag = nullptr);
    std::mutex _commMutex;  // gradient exchange calls may come from either submitting thread
    std::vector<void*> _bucketEvents;  // ring: producer-side records for the exchange thread to wait on
    size_t _bucketNext = 0;
    long _gatherTicket = 0;
    bool _exchangeWorkPending = false;
    void* _evJoinComm = nullptr;
    long _rcclCallNs = 0, _rcclCalls = 0;  // exchange thread only (MCN_DEBUG_TIMES)
    // startSteps(profile = 3): every collective bracketed by two events on the exchange stream (bench.py `rccl`)
    bool _commTiming = false;
    struct CommSpan { int kind; void* e0; void* e1; size_t count; };
    std::vector<CommSpan> _commSpans;     // exchange thread appends, the worker thread reads after joining it
    std::vector<void*> _commEventPool;
    size_t _commEventsUsed = 0;
    void* commEvent();
    void commCollect(std::map<std::string, ProfEntry>& into);
    std::vector<std::pair<std::function<void()>, int>> _deferred;
    std::vector<std::function<void()>> _late;
    void runLateWork();
    int _deferredUnits = 0;
    void flushSideWork(size_t count);  // first `count` deferred items -> side stream (one fork)
    DBuf _ws, _ws2;
    DBuf _gradArena;
    DBuf _costBuf;  // doubles, viewed as floats x2
    DBuf _sgdPlan;  // device records of the one-launch weight update
    int _sgdCount = 0, _sgdBlocks = 0;
    // earlyUpdate: the same records split into (all but the last weight layer of a backward pass) / (the rest)
    std::vector<WeightLayer*> _passOrder, _passNow;
    DBuf _sgdEarly, _sgdLate;
    int _sgdEarlyCount = 0, _sgdEarlyBlocks = 0, _sgdLateCount = 0, _sgdLateBlocks = 0;
    bool _splitPlans = false, _earlyUpdateIssued = false;
    size_t _sgdEarlyElems = 0, _earlyMinElems = (size_t)8 << 20;
    bool _earlyOff = false;
    size_t _earlyPos = 0;      // deferred items up to and including the early update's (0: none queued)
    int _gateTimeoutMs = 5000;   // a gate that spins longer reports a dead stream at the next sync()
    int _pendingMainGate = 0;    // signal A (in front of the early update on the side stream) the main stream still has to wait for
    int _pendingWeightGate = 0;  // signal B (behind it): before the first forward layer that reads the weights it writes
    int _earlySeqA = 0;
    std::vector<double> _costHost;
    double _baseErr = 0;

    // data-parallel state
    int _rank = 0, _world = 1;
    bool _dpOn = false;
    void* _comm = nullptr;
    void* _commStream = nullptr;
    void* _evGrad = nullptr;
    void* _evGrad2 = nullptr;
    int _sideUnits = 2, _wlSeen = 0, _wlPerPass = 0;
    void* _evGather = nullptr;
    void* _evComm = nullptr;
    struct Segment {
        WeightLayer* layer; size_t begin, end; bool ready; bool onSide = false, onMain = true, byMain = false;
        std::vector<WeightLayer*> contributors;  // the layer itself and every layer that shares one of its matrices
        int reported = 0;                        // ... of which so many have finished their gradients in this pass
    };
    std::vector<Segment> _segments;  // in arena order (reverse layer order)
    size_t _sentUpTo = 0;            // arena offset up to which all-reduce has been issued
    size_t _bucketFloats = 1 << 20;
    bool _commPending = false;
    std::mutex _logMutex;
    std::vector<CommRec> _commLog;
    void logComm(int kind, size_t offset, size_t count);

    // step graphs
    struct StepGraph { void* exec = nullptr; std::vector<float*> data; int numCases = 0; int nodes = 0; };
    std::map<int, StepGraph> _stepGraphs;  // by minibatch index
    bool _useGraphs = true, _capturing = false;
    long _stepsDone = 0;
    bool stepGraphUsable(int miniIdx);
    void captureStep(int miniIdx, StepGraph& g);

    // per-operation event timing (off on the normal path)
    bool _profiling = false, _phases = false;
    struct BiasSum { float* tgt; const float* mat; int rows, cols; float scaleSum; float* partsOut; };
    DBuf _sgdFused;
    int _sgdFusedCount = 0, _sgdFusedBlocks = 0;
    bool _fusedTailOk = false, _fusedTailNow = false;
    std::vector<BiasSum> _biasBatch[2];  // queued by closures on the main [0] / side [1] submitting thread
#ifdef MCN_TEST_HOOKS
    std::string _failAt;   // failNextBackwardAt
    std::mutex _failMutex;
#endif
    bool _failed = false;  // a data-parallel replica that lost a step: its peers wait in a collective it will never join
    void resetAfterError();
    struct ProfSpan { int label; void* e0; void* e1; size_t launches0; };
    std::vector<std::string> _profLabels;
    std::map<std::string, int> _profLabelIdx;
    std::vector<ProfSpan> _profSpans;
    std::vector<void*> _eventPool;
    size_t _eventsUsed = 0;
    void* profEvent();
    void profCollect(std::map<std::string, ProfEntry>& into);

    std::thread _thread;
    Queue<Work> _work;
    Queue<WorkResult> _results;
    bool _stop = false;
};

}  // namespace mcn
