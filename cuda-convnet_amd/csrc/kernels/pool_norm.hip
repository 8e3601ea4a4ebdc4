// Memory-bound spatial kernels: local pooling (max / avg) forward and undo,
// response normalisation within and across maps, forward and undo.
//
// Replaces conv_util.cuh:127-191 (kLocalPool), conv_util.cu:1392-1587
// (kLocalAvgUndo / kLocalMaxUndo), :829-893 (kCNorm_manyfilter), :1038-1179
// (kFCNorm / kFRNormUndo), :1592-1712 (kRNormUndoPrelims / kRNormUndo).
//
// All of them stream (filters, pixels, images) arrays whose fastest dimension is
// the image index, so every thread owns VW consecutive images of one (filter, pixel)
// and moves them with one 16-byte (VW=4) access; a wave covers 256 consecutive
// floats = two full 512-byte rows at N=128.  They are HBM-bound: no LDS needed,
// window re-reads are served by L2.
#include "common.h"

template <int VW> struct Vec;
template <> struct Vec<4> {
    f32x4 v;
    __device__ static Vec load(const float* p) { Vec r; r.v = *reinterpret_cast<const f32x4*>(p); return r; }
    __device__ static Vec splat(float x) { Vec r; r.v = (f32x4){x, x, x, x}; return r; }
    __device__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec<1> {
    float v;
    __device__ static Vec load(const float* p) { Vec r; r.v = *p; return r; }
    __device__ static Vec splat(float x) { Vec r; r.v = x; return r; }
    __device__ void store(float* p) const { *p = v; }
};

// element index -> (row, n): rows = filters*pixels, thread owns VW images
#define ROW_N_DECODE(total_rows)                                             \
    const int nv = (numImages + VW - 1) / VW;                                \
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;         \
    if (gid >= (long long)(total_rows) * nv) return;                         \
    const int row = (int)(gid / nv);                                         \
    const int n = (int)(gid % nv) * VW;

template <int VW, int MODE>
__global__ __launch_bounds__(256) void k_local_pool(const float* __restrict__ imgs, float* __restrict__ target,
                                                    int numImages, int numFilters, int imgY, int imgX, int subsX,
                                                    int startX, int strideX, int outY, int outX) {
    const int numOutputs = outY * outX;
    ROW_N_DECODE(numFilters * numOutputs)
    const int f = row / numOutputs, o = row % numOutputs;
    const int sy = startX + (o / outX) * strideX, sx = startX + (o % outX) * strideX;
    const int y0 = max(0, sy), x0 = max(0, sx), y1 = min(imgY, sy + subsX), x1 = min(imgX, sx + subsX);
    const float* src = imgs + (size_t)f * imgY * imgX * numImages + n;
    Vec<VW> acc = Vec<VW>::splat(MODE == 0 ? -2e38f : 0.f);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const Vec<VW> s = Vec<VW>::load(src + (size_t)(y * imgX + x) * numImages);
            if (MODE == 0) {
                if constexpr (VW == 4) { for (int t = 0; t < 4; t++) acc.v[t] = fmaxf(acc.v[t], s.v[t]); }
                else acc.v = fmaxf(acc.v, s.v);
            } else {
                acc.v += s.v;
            }
        }
    if (MODE == 1) acc.v = acc.v / (float)((y1 - y0) * (x1 - x0));
    acc.store(target + (size_t)row * numImages + n);
}

__device__ __forceinline__ void out_range(int px, int startX, int subsX, int strideX, int outputs, int& s, int& e) {
    s = px - startX < subsX ? 0 : 1 + (px - startX - subsX) / strideX;  // conv_util.cu:1512-1515
    e = min(outputs, 1 + (px - startX) / strideX);
}

// MODE 0: max undo, 1: avg undo.  NEURON >= 0: multiply by the gradient of the neuron
// that produced imgs in place (fused NeuronLayer::bpropActs).
template <int VW, int MODE, bool FUSE_NEURON>
__global__ __launch_bounds__(256) void k_pool_undo(const float* __restrict__ imgs, const float* __restrict__ grads,
                                                   const float* __restrict__ maxActs, float* __restrict__ target,
                                                   int numImages, int numFilters, int imgY, int imgX, int subsX,
                                                   int startX, int strideX, int outY, int outX, float sT, float sO,
                                                   int neuron, float na, float nb) {
    const int imgPixels = imgY * imgX, numOutputs = outY * outX;
    ROW_N_DECODE(numFilters * imgPixels)
    const int f = row / imgPixels, px = row % imgPixels;
    const int y = px / imgX, x = px % imgX;
    Vec<VW> acc = Vec<VW>::splat(0.f);
    Vec<VW> im = Vec<VW>::splat(0.f);
    if (MODE == 0) im = Vec<VW>::load(imgs + (size_t)row * numImages + n);
    if (x >= startX && x < startX + strideX * (outX - 1) + subsX && y >= startX && y < startX + strideX * (outY - 1) + subsX) {
        int oys, oye, oxs, oxe;
        out_range(y, startX, subsX, strideX, outY, oys, oye);
        out_range(x, startX, subsX, strideX, outX, oxs, oxe);
        for (int my = oys; my < oye; my++) {
            float rsy = 1.f;
            if (MODE == 1) rsy = fminf(imgY, startX + my * strideX + subsX) - fmaxf(0, startX + my * strideX);
            for (int mx = oxs; mx < oxe; mx++) {
                const size_t o = ((size_t)f * numOutputs + my * outX + mx) * numImages + n;
                const Vec<VW> g = Vec<VW>::load(grads + o);
                if (MODE == 0) {
                    const Vec<VW> ma = Vec<VW>::load(maxActs + o);
                    if constexpr (VW == 4) { for (int t = 0; t < 4; t++) acc.v[t] += (im.v[t] == ma.v[t] ? 1.f : 0.f) * g.v[t]; }
                    else acc.v += (im.v == ma.v ? 1.f : 0.f) * g.v;
                } else {
                    const float rsx = fminf(imgX, startX + mx * strideX + subsX) - fmaxf(0, startX + mx * strideX);
                    const float inv = 1.0f / (rsx * rsy);
                    acc.v += g.v * inv;
                }
            }
        }
    }
    float* t = target + (size_t)row * numImages + n;
    Vec<VW> out;
    out.v = sO * acc.v;
    if (FUSE_NEURON) {
        if constexpr (VW == 4) { for (int i = 0; i < 4; i++) out.v[i] = neuron_grad(neuron, out.v[i], im.v[i], 0.f, na, nb); }
        else out.v = neuron_grad(neuron, out.v, im.v, 0.f, na, nb);
    }
    if (sT != 0.f) out.v += sT * Vec<VW>::load(t).v;
    out.store(t);
}

// CROSS = false: window over pixels of the same map; true: over neighbouring maps
template <int VW, bool CROSS>
__global__ __launch_bounds__(256) void k_rnorm(const float* __restrict__ imgs, const float* __restrict__ md, float* __restrict__ denoms,
                                               float* __restrict__ target, int numImages, int numFilters, int imgY,
                                               int imgX, int size, float addScale, float powScale, int blocked) {
    const int imgPixels = imgY * imgX;
    ROW_N_DECODE(numFilters * imgPixels)
    const int f = row / imgPixels, px = row % imgPixels;
    Vec<VW> acc = Vec<VW>::splat(0.f);
    if (CROSS) {
        const int startF = blocked ? (f / size) * size : -size / 2 + f;
        const int f0 = blocked ? startF : max(0, startF), f1 = min(numFilters, startF + size);
        for (int ff = f0; ff < f1; ff++) {
            const Vec<VW> s = Vec<VW>::load(md + ((size_t)ff * imgPixels + px) * numImages + n);
            acc.v += s.v * s.v;
        }
    } else {
        const int y = px / imgX, x = px % imgX;
        const int y0 = max(0, y - size / 2), x0 = max(0, x - size / 2);
        const int y1 = min(imgY, y - size / 2 + size), x1 = min(imgX, x - size / 2 + size);
        for (int yy = y0; yy < y1; yy++)
            for (int xx = x0; xx < x1; xx++) {
                const Vec<VW> s = Vec<VW>::load(md + ((size_t)f * imgPixels + yy * imgX + xx) * numImages + n);
                acc.v += s.v * s.v;
            }
    }
    const size_t o = (size_t)row * numImages + n;
    const Vec<VW> x = Vec<VW>::load(imgs + o);
    Vec<VW> d, t;
    d.v = 1.f + addScale * acc.v;
    if constexpr (VW == 4) { for (int i = 0; i < 4; i++) t.v[i] = x.v[i] * powf(d.v[i], -powScale); }
    else t.v = x.v * powf(d.v, -powScale);
    d.store(denoms + o);
    t.store(target + o);
}

// acts := scale * outGrads * acts / denoms   (kRNormUndoPrelims conv_util.cu:1592-1605)
__global__ __launch_bounds__(256) void k_rnorm_undo_prelims(float* __restrict__ acts, const float* __restrict__ denoms,
                                                            const float* __restrict__ outGrads, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acts[i] = scale * outGrads[i] * acts[i] / denoms[i];
}

template <int VW, bool CROSS>
__global__ __launch_bounds__(256) void k_rnorm_undo(const float* __restrict__ outGrads, const float* __restrict__ denoms,
                                                    const float* __restrict__ inputs, const float* __restrict__ acts,
                                                    float* __restrict__ target, int numImages, int numFilters, int imgY,
                                                    int imgX, int size, float powScale, int blocked, float sT, float sO) {
    const int imgPixels = imgY * imgX;
    ROW_N_DECODE(numFilters * imgPixels)
    const int f = row / imgPixels, px = row % imgPixels;
    Vec<VW> acc = Vec<VW>::splat(0.f);
    if (CROSS) {
        const int startF = blocked ? (f / size) * size : -size + size / 2 + 1 + f;
        const int f0 = blocked ? startF : max(0, startF), f1 = min(numFilters, startF + size);
        for (int ff = f0; ff < f1; ff++) acc.v += Vec<VW>::load(acts + ((size_t)ff * imgPixels + px) * numImages + n).v;
    } else {
        const int y = px / imgX, x = px % imgX;
        const int y0 = max(0, y + size / 2 - size + 1), x0 = max(0, x + size / 2 - size + 1);
        const int y1 = min(imgY, y + size / 2 + 1), x1 = min(imgX, x + size / 2 + 1);
        for (int yy = y0; yy < y1; yy++)
            for (int xx = x0; xx < x1; xx++)
                acc.v += Vec<VW>::load(acts + ((size_t)f * imgPixels + yy * imgX + xx) * numImages + n).v;
    }
    const size_t o = (size_t)row * numImages + n;
    const Vec<VW> inp = Vec<VW>::load(inputs + o), og = Vec<VW>::load(outGrads + o), den = Vec<VW>::load(denoms + o);
    Vec<VW> out;
    if constexpr (VW == 4) { for (int i = 0; i < 4; i++) out.v[i] = inp.v[i] * acc.v[i] + og.v[i] * powf(den.v[i], -powScale); }
    else out.v = inp.v * acc.v + og.v * powf(den.v, -powScale);
    out.v = sO * out.v;
    if (sT != 0.f) out.v += sT * Vec<VW>::load(target + o).v;
    out.store(target + o);
}

static inline bool vec_ok(int numImages, std::initializer_list<const void*> ptrs) {
    if (numImages % 4 != 0) return false;
    for (const void* p : ptrs)
        if (p && ((uintptr_t)p % 16) != 0) return false;
    return true;
}

static inline unsigned blocks_for(long long rows, int numImages, int vw) {
    return (unsigned)((rows * ((numImages + vw - 1) / vw) + 255) / 256);
}

MCN_API int mcn_local_pool(const float* images, float* target, int numImages, int numFilters, int imgSizeY,
                           int imgSizeX, int subsX, int startX, int strideX, int outputsY, int outputsX, int mode,
                           void* stream) {
    MCN_REQUIRE(images && target, "localPool: null pointer");
    MCN_REQUIRE(numImages > 0 && numFilters > 0 && imgSizeY > 0 && imgSizeX > 0 && subsX > 0 && strideX > 0 &&
                    outputsY > 0 && outputsX > 0, "localPool: non-positive dimension");
    MCN_REQUIRE(mode == 0 || mode == 1, "localPool: mode must be 0 (max) or 1 (avg)");
    const long long rows = (long long)numFilters * outputsY * outputsX;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = vec_ok(numImages, {images, target});
    dim3 grid(blocks_for(rows, numImages, v4 ? 4 : 1));
#define POOL_ARGS images, target, numImages, numFilters, imgSizeY, imgSizeX, subsX, startX, strideX, outputsY, outputsX
    if (v4) {
        if (mode == 0) MCN_LAUNCH((k_local_pool<4, 0>), grid, dim3(256), 0, s, POOL_ARGS);
        else MCN_LAUNCH((k_local_pool<4, 1>), grid, dim3(256), 0, s, POOL_ARGS);
    } else {
        if (mode == 0) MCN_LAUNCH((k_local_pool<1, 0>), grid, dim3(256), 0, s, POOL_ARGS);
        else MCN_LAUNCH((k_local_pool<1, 1>), grid, dim3(256), 0, s, POOL_ARGS);
    }
#undef POOL_ARGS
    return MCN_LAUNCH_CHECK("localPool");
}

static int pool_undo_impl(const float* images, const float* grads, const float* maxActs, float* target, int numImages,
                          int numFilters, int imgY, int imgX, int subsX, int startX, int strideX, int outY, int outX,
                          float sT, float sO, int mode, int neuron, float na, float nb, void* stream) {
    MCN_REQUIRE(grads && target && (mode == 1 || (images && maxActs)), "poolUndo: null pointer");
    MCN_REQUIRE(numImages > 0 && numFilters > 0 && imgY > 0 && imgX > 0 && subsX > 0 && strideX > 0 && outY > 0 && outX > 0,
                "poolUndo: non-positive dimension");
    const long long rows = (long long)numFilters * imgY * imgX;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = vec_ok(numImages, {images, grads, maxActs, target});
    dim3 grid(blocks_for(rows, numImages, v4 ? 4 : 1));
#define UNDO_ARGS images, grads, maxActs, target, numImages, numFilters, imgY, imgX, subsX, startX, strideX, outY, outX, sT, sO, neuron, na, nb
    if (mode == 1) {
        if (v4) MCN_LAUNCH((k_pool_undo<4, 1, false>), grid, dim3(256), 0, s, UNDO_ARGS);
        else MCN_LAUNCH((k_pool_undo<1, 1, false>), grid, dim3(256), 0, s, UNDO_ARGS);
    } else if (neuron >= 0) {
        if (v4) MCN_LAUNCH((k_pool_undo<4, 0, true>), grid, dim3(256), 0, s, UNDO_ARGS);
        else MCN_LAUNCH((k_pool_undo<1, 0, true>), grid, dim3(256), 0, s, UNDO_ARGS);
    } else {
        if (v4) MCN_LAUNCH((k_pool_undo<4, 0, false>), grid, dim3(256), 0, s, UNDO_ARGS);
        else MCN_LAUNCH((k_pool_undo<1, 0, false>), grid, dim3(256), 0, s, UNDO_ARGS);
    }
#undef UNDO_ARGS
    return MCN_LAUNCH_CHECK("poolUndo");
}

MCN_API int mcn_local_max_undo(const float* images, const float* maxGrads, const float* maxActs, float* target,
                               int numImages, int numFilters, int imgSizeY, int imgSizeX, int subsX, int startX,
                               int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput, void* stream) {
    return pool_undo_impl(images, maxGrads, maxActs, target, numImages, numFilters, imgSizeY, imgSizeX, subsX, startX,
                          strideX, outputsY, outputsX, scaleTargets, scaleOutput, 0, -1, 0.f, 0.f, stream);
}

MCN_API int mcn_local_max_undo_neuron(const float* images, const float* maxGrads, const float* maxActs, float* target,
                                      int numImages, int numFilters, int imgSizeY, int imgSizeX, int subsX, int startX,
                                      int strideX, int outputsY, int outputsX, int neuron, float neuronA, float neuronB,
                                      void* stream) {
    MCN_REQUIRE(neuron_valid(neuron) && neuron != MCN_NEURON_ABS && neuron != MCN_NEURON_SQUARE && neuron != MCN_NEURON_SOFTRELU,
                "localMaxUndoNeuron: neuron %d needs its input; cannot be fused", neuron);
    return pool_undo_impl(images, maxGrads, maxActs, target, numImages, numFilters, imgSizeY, imgSizeX, subsX, startX,
                          strideX, outputsY, outputsX, 0.f, 1.f, 0, neuron, neuronA, neuronB, stream);
}

MCN_API int mcn_local_avg_undo(const float* avgGrads, float* target, int numImages, int numFilters, int imgSizeY,
                               int imgSizeX, int subsX, int startX, int strideX, int outputsY, int outputsX,
                               float scaleTargets, float scaleOutput, void* stream) {
    return pool_undo_impl(nullptr, avgGrads, nullptr, target, numImages, numFilters, imgSizeY, imgSizeX, subsX, startX,
                          strideX, outputsY, outputsX, scaleTargets, scaleOutput, 1, -1, 0.f, 0.f, stream);
}

// md: what the window squares and sums -- the images themselves (response norm) or images minus their local mean
// (contrast norm, conv_util.cu:2064-2066 vs :2074)
static int rnorm_impl(const float* images, const float* md, float* denoms, float* target, int numImages, int numFilters, int imgY,
                      int imgX, int size, float addScale, float powScale, int blocked, bool cross, void* stream) {
    MCN_REQUIRE(images && md && denoms && target, "responseNorm: null pointer");
    MCN_REQUIRE(numImages > 0 && numFilters > 0 && imgY > 0 && imgX > 0 && size > 0, "responseNorm: non-positive dimension");
    MCN_REQUIRE(!cross || (size <= numFilters && numFilters % 16 == 0), "responseNormCrossMap: need sizeF <= numFilters, numFilters %% 16 == 0");
    const long long rows = (long long)numFilters * imgY * imgX;
    hipStream_t s = (hipStream_t)stream;
    const bool v4 = vec_ok(numImages, {images, md, denoms, target});
    dim3 grid(blocks_for(rows, numImages, v4 ? 4 : 1));
#define RN_ARGS images, md, denoms, target, numImages, numFilters, imgY, imgX, size, addScale, powScale, blocked
    if (cross) {
        if (v4) MCN_LAUNCH((k_rnorm<4, true>), grid, dim3(256), 0, s, RN_ARGS);
        else MCN_LAUNCH((k_rnorm<1, true>), grid, dim3(256), 0, s, RN_ARGS);
    } else {
        if (v4) MCN_LAUNCH((k_rnorm<4, false>), grid, dim3(256), 0, s, RN_ARGS);
        else MCN_LAUNCH((k_rnorm<1, false>), grid, dim3(256), 0, s, RN_ARGS);
    }
#undef RN_ARGS
    return MCN_LAUNCH_CHECK("responseNorm");
}

static int rnorm_undo_impl(const float* outGrads, const float* denoms, const float* inputs, float* acts, float* target,
                           int numImages, int numFilters, int imgY, int imgX, int size, float addScale, float powScale,
                           int blocked, bool cross, float sT, float sO, void* stream) {
    MCN_REQUIRE(outGrads && denoms && inputs && acts && target, "responseNormUndo: null pointer");
    MCN_REQUIRE(numImages > 0 && numFilters > 0 && imgY > 0 && imgX > 0 && size > 0, "responseNormUndo: non-positive dimension");
    MCN_REQUIRE(numFilters % 16 == 0, "responseNormUndo: numFilters %% 16 != 0");  // conv_util.cu:2246
    const long long rows = (long long)numFilters * imgY * imgX;
    const size_t total = (size_t)rows * numImages;
    hipStream_t s = (hipStream_t)stream;
    MCN_LAUNCH(k_rnorm_undo_prelims, dim3((unsigned)std::min<size_t>(divup_sz(total, 256), 8192)), dim3(256), 0, s,
                       acts, denoms, outGrads, total, -2 * addScale * powScale);
    const bool v4 = vec_ok(numImages, {outGrads, denoms, inputs, acts, target});
    dim3 grid(blocks_for(rows, numImages, v4 ? 4 : 1));
#define RU_ARGS outGrads, denoms, inputs, acts, target, numImages, numFilters, imgY, imgX, size, powScale, blocked, sT, sO
    if (cross) {
        if (v4) MCN_LAUNCH((k_rnorm_undo<4, true>), grid, dim3(256), 0, s, RU_ARGS);
        else MCN_LAUNCH((k_rnorm_undo<1, true>), grid, dim3(256), 0, s, RU_ARGS);
    } else {
        if (v4) MCN_LAUNCH((k_rnorm_undo<4, false>), grid, dim3(256), 0, s, RU_ARGS);
        else MCN_LAUNCH((k_rnorm_undo<1, false>), grid, dim3(256), 0, s, RU_ARGS);
    }
#undef RU_ARGS
    return MCN_LAUNCH_CHECK("responseNormUndo");
}

MCN_API int mcn_response_norm(const float* images, float* denoms, float* target, int numImages, int numFilters,
                              int imgSizeY, int imgSizeX, int sizeX, float addScale, float powScale, void* stream) {
    return rnorm_impl(images, images, denoms, target, numImages, numFilters, imgSizeY, imgSizeX, sizeX, addScale, powScale, 0, false, stream);
}

MCN_API int mcn_contrast_norm(const float* images, const float* meanDiffs, float* denoms, float* target, int numImages, int numFilters,
                              int imgSizeY, int imgSizeX, int sizeX, float addScale, float powScale, void* stream) {
    return rnorm_impl(images, meanDiffs, denoms, target, numImages, numFilters, imgSizeY, imgSizeX, sizeX, addScale, powScale, 0, false, stream);
}

MCN_API int mcn_response_norm_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                                   float* target, int numImages, int numFilters, int imgSizeY, int imgSizeX, int sizeX,
                                   float addScale, float powScale, float scaleTargets, float scaleOutput, void* stream) {
    return rnorm_undo_impl(outGrads, denoms, inputs, acts, target, numImages, numFilters, imgSizeY, imgSizeX, sizeX,
                           addScale, powScale, 0, false, scaleTargets, scaleOutput, stream);
}

MCN_API int mcn_response_norm_cross_map(const float* images, float* denoms, float* target, int numImages,
                                        int numFilters, int imgPixels, int sizeF, float addScale, float powScale,
                                        int blocked, void* stream) {
    return rnorm_impl(images, images, denoms, target, numImages, numFilters, 1, imgPixels, sizeF, addScale, powScale, blocked, true, stream);
}

MCN_API int mcn_response_norm_cross_map_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                                             float* target, int numImages, int numFilters, int imgPixels, int sizeF,
                                             float addScale, float powScale, int blocked, float scaleTargets,
                                             float scaleOutput, void* stream) {
    return rnorm_undo_impl(outGrads, denoms, inputs, acts, target, numImages, numFilters, 1, imgPixels, sizeF, addScale,
                           powScale, blocked, true, scaleTargets, scaleOutput, stream);
}
