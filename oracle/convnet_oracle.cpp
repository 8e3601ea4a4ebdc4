// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// CPU restatement (plain loops, fp32, OpenMP over independent outputs) of the
// arithmetic of the cuda-convnet compute core: cudaconv2 (filterActs / imgActs /
// weightActs, pooling, response normalisation), the NVMatrix verbs the layers
// use, the cost kernels of layer_kernels.cu and the SGD update of weights.cuh.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library, and only as the checker / reported CPU baseline.
//
// PARITY UNPINNED: the reference ships no golden vectors, known-answer tests
// or fixtures for this path, its CUDA sources cannot
// be built here and its Python-2 driver cannot be imported.  The restatement is
// tied to the reference only by (i) the reference's own acceptance test, the
// finite-difference gradient check with relErr < 0.02 (convnet.cu:280-330),
// (ii) layer output shapes and first-batch costs in its training logs
// (deepid2/nohup.out:1-20,55), see tests/test_oracle_pins.py.
//
// Layouts follow the reference everywhere: activations are row-major
// (channels*rows*cols, numImages) with images contiguous; conv filters
// (filterChannels*filterPixels, numFilters); local filters
// (modules*filterChannels*filterPixels, numFilters)   (filter_acts.cu:39-43).
// Unlike the reference, every spatial routine takes (Y, X) separately so that
// rectangular images (BASELINE config 39x31x3) are defined.
//
// Each function cites the reference file:line it restates.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define ORC extern "C" __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------
// filterActs  (filter_acts.cu:56-192 colour kernel, :223-368 sparse kernel,
// host _filterActs :565-614).  targets = sT*targets + sO*(W (*) images).
// images may be a column slice of a wider matrix: imgStride = row stride.
// ---------------------------------------------------------------------------
// colorIndices != NULL: random-sparse connectivity (filter_acts.cu:401-560 filterActs_YxX_sparse_random): filter colour c of
// group g reads image channel colorIndices[g*numFilterColors + c] (shColors, :467-470); numFilterColors is given.
static void filter_acts_impl(const float* images, const float* filters, float* targets,
                             int numImages, int imgStride, int imgSizeY, int imgSizeX,
                             int numImgColors, int numFilters, int filterSize,
                             int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                             int numGroups, float scaleTargets, float scaleOutput, int conv,
                             const int* colorIndices, int numFilterColors) {
    const int fc = colorIndices ? numFilterColors : numImgColors / numGroups, fpg = numFilters / numGroups;
    const int fp = filterSize * filterSize, M = numModulesY * numModulesX;
    const int imgPixels = imgSizeY * imgSizeX;
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int m = 0; m < M; m++) {
            for (int f = 0; f < numFilters; f++) {
                const int my = m / numModulesX, mx = m % numModulesX;
                const int g = f / fpg;
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int c = 0; c < fc; c++) {
                    for (int fy = 0; fy < filterSize; fy++) {
                        const int y = paddingStart + my * moduleStride + fy;
                        if (y < 0 || y >= imgSizeY) continue;
                        for (int fx = 0; fx < filterSize; fx++) {
                            const int x = paddingStart + mx * moduleStride + fx;
                            if (x < 0 || x >= imgSizeX) continue;
                            const size_t wrow = (size_t)(conv ? 0 : m) * fc * fp + (size_t)c * fp + fy * filterSize + fx;
                            const float w = filters[wrow * numFilters + f];
                            const int ch = colorIndices ? colorIndices[g * fc + c] : g * fc + c;
                            const float* src = images + ((size_t)ch * imgPixels + (size_t)y * imgSizeX + x) * imgStride;
                            float* a = acc.data();
                            for (int n = 0; n < numImages; n++) a[n] += w * src[n];
                        }
                    }
                }
                float* t = targets + ((size_t)f * M + m) * numImages;
                if (scaleTargets == 0.f) {
                    for (int n = 0; n < numImages; n++) t[n] = scaleOutput * acc[n];
                } else {
                    for (int n = 0; n < numImages; n++) t[n] = scaleTargets * t[n] + scaleOutput * acc[n];
                }
            }
        }
    }
}

ORC void orc_filter_acts(const float* images, const float* filters, float* targets,
                         int numImages, int imgStride, int imgSizeY, int imgSizeX,
                         int numImgColors, int numFilters, int filterSize,
                         int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                         int numGroups, float scaleTargets, float scaleOutput, int conv) {
    filter_acts_impl(images, filters, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize, numModulesY,
                     numModulesX, paddingStart, moduleStride, numGroups, scaleTargets, scaleOutput, conv, nullptr, 0);
}
// conv/localFilterActsSparse (filter_acts.cu:1221-1444)
ORC void orc_filter_acts_sparse(const float* images, const float* filters, float* targets, const int* colorIndices,
                                int numImages, int imgStride, int imgSizeY, int imgSizeX,
                                int numImgColors, int numFilterColors, int numFilters, int filterSize,
                                int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                int numGroups, float scaleTargets, float scaleOutput, int conv) {
    filter_acts_impl(images, filters, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize, numModulesY,
                     numModulesX, paddingStart, moduleStride, numGroups, scaleTargets, scaleOutput, conv, colorIndices, numFilterColors);
}

// ---------------------------------------------------------------------------
// imgActs (img_acts.cu:52-188, :217-360, :392-518; host _imgActs :900-961).
// targets[c,y,x,n] = sT*targets + sO * sum_{modules covering (y,x)} sum_f
//                    hidActs[f,m,n] * W[(m),c,p(y,x,m),f]
// ---------------------------------------------------------------------------
// colorIndices != NULL: img_acts_{mediumcolor,manycolor}_sparse_rand (img_acts.cu:556-700, :749-890): targets is
// (overSample, numImgColors, imgPixels, numImages); copy s holds the contribution of groups [s*numGroups/overSample, ...),
// whose filter colours are sent to the image channels colorIndices[g*numFilterColors + c] (one permutation per copy).
static void img_acts_impl(const float* hidActs, const float* filters, float* targets,
                          int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                          int filterSize, int numModulesY, int numModulesX, int paddingStart,
                          int moduleStride, int numGroups, float scaleTargets, float scaleOutput, int conv,
                          const int* colorIndices, int numFilterColors) {
    const int fc = colorIndices ? numFilterColors : numImgColors / numGroups, fpg = numFilters / numGroups;
    const int fp = filterSize * filterSize, M = numModulesY * numModulesX;
    const int imgPixels = imgSizeY * imgSizeX;
    const int rows = colorIndices ? numGroups * numFilterColors : numImgColors;  // (copy, channel) pairs
    std::vector<int> where(rows);  // target row (copy*numImgColors + channel) -> position g*fc + c in the table
    for (int k = 0; k < rows; k++) where[colorIndices ? (k / numImgColors) * numImgColors + colorIndices[k] : k] = k;
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int cg = 0; cg < rows; cg++) {
            for (int px = 0; px < imgPixels; px++) {
                const int y = px / imgSizeX, x = px % imgSizeX;
                const int g = where[cg] / fc, c = where[cg] % fc;
                // module ranges exactly as img_acts.cu:435-440
                const int startY = y - paddingStart < filterSize ? 0 : 1 + (y - paddingStart - filterSize) / moduleStride;
                const int endY = imin(numModulesY, 1 + (y - paddingStart) / moduleStride);
                const int startX = x - paddingStart < filterSize ? 0 : 1 + (x - paddingStart - filterSize) / moduleStride;
                const int endX = imin(numModulesX, 1 + (x - paddingStart) / moduleStride);
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int my = startY; my < endY; my++) {
                    const int pyf = y - (paddingStart + my * moduleStride);
                    for (int mx = startX; mx < endX; mx++) {
                        const int pxf = x - (paddingStart + mx * moduleStride);
                        const int m = my * numModulesX + mx;
                        const size_t wrow = (size_t)(conv ? 0 : m) * fc * fp + (size_t)c * fp + pyf * filterSize + pxf;
                        const float* wr = filters + wrow * numFilters + (size_t)g * fpg;
                        for (int f = 0; f < fpg; f++) {
                            const float w = wr[f];
                            const float* h = hidActs + ((size_t)(g * fpg + f) * M + m) * numImages;
                            float* a = acc.data();
                            for (int n = 0; n < numImages; n++) a[n] += w * h[n];
                        }
                    }
                }
                float* t = targets + ((size_t)cg * imgPixels + px) * numImages;
                if (scaleTargets == 0.f) {
                    for (int n = 0; n < numImages; n++) t[n] = scaleOutput * acc[n];
                } else {
                    for (int n = 0; n < numImages; n++) t[n] = scaleTargets * t[n] + scaleOutput * acc[n];
                }
            }
        }
    }
}

ORC void orc_img_acts(const float* hidActs, const float* filters, float* targets,
                      int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilters,
                      int filterSize, int numModulesY, int numModulesX, int paddingStart,
                      int moduleStride, int numGroups, float scaleTargets, float scaleOutput, int conv) {
    img_acts_impl(hidActs, filters, targets, numImages, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize, numModulesY, numModulesX,
                  paddingStart, moduleStride, numGroups, scaleTargets, scaleOutput, conv, nullptr, 0);
}
// conv/localImgActsSparse (img_acts.cu:1929-2583): targets (overSample*numImgColors*imgPixels, numImages)
ORC void orc_img_acts_sparse(const float* hidActs, const float* filters, float* targets, const int* colorIndices,
                             int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors, int numFilters,
                             int filterSize, int numModulesY, int numModulesX, int paddingStart,
                             int moduleStride, int numGroups, float scaleTargets, float scaleOutput, int conv) {
    img_acts_impl(hidActs, filters, targets, numImages, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize, numModulesY, numModulesX,
                  paddingStart, moduleStride, numGroups, scaleTargets, scaleOutput, conv, colorIndices, numFilterColors);
}

// ---------------------------------------------------------------------------
// weightActs (weight_acts.cu:57-201, :233-380; host _weightActs :579-648).
// conv:  targets[(m/partialSum), c, p, f] = sT*targets + sO * sum_{m in chunk} sum_n img*hid
// local: partialSum == 1 and one slab per module (weight_acts.cu:966-976).
// ---------------------------------------------------------------------------
// colorIndices != NULL: conv_weight_acts_mc_mf_rand (weight_acts.cu:414-560): the image channel of filter colour c of
// group g is colorIndices[g*numFilterColors + c].
static void weight_acts_impl(const float* images, const float* hidActs, float* targets,
                             int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                             int numFilters, int filterSize, int numModulesY, int numModulesX,
                             int paddingStart, int moduleStride, int numGroups, int partialSum,
                             float scaleTargets, float scaleOutput, const int* colorIndices, int numFilterColors) {
    const int fc = colorIndices ? numFilterColors : numImgColors / numGroups, fpg = numFilters / numGroups;
    const int fp = filterSize * filterSize, M = numModulesY * numModulesX;
    const int imgPixels = imgSizeY * imgSizeX;
    if (partialSum <= 0) partialSum = M;
    const int chunks = M / partialSum;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ch = 0; ch < chunks; ch++) {
        for (int k = 0; k < fc * fp; k++) {
            const int c = k / fp, p = k % fp, fy = p / filterSize, fx = p % filterSize;
            for (int f = 0; f < numFilters; f++) {
                const int g = f / fpg;
                float s = 0.f;
                for (int m = ch * partialSum; m < (ch + 1) * partialSum; m++) {
                    const int my = m / numModulesX, mx = m % numModulesX;
                    const int y = paddingStart + my * moduleStride + fy;
                    const int x = paddingStart + mx * moduleStride + fx;
                    if (y < 0 || y >= imgSizeY || x < 0 || x >= imgSizeX) continue;
                    const int chn = colorIndices ? colorIndices[g * fc + c] : g * fc + c;
                    const float* src = images + ((size_t)chn * imgPixels + (size_t)y * imgSizeX + x) * imgStride;
                    const float* h = hidActs + ((size_t)f * M + m) * numImages;
                    float d = 0.f;
#pragma omp simd reduction(+ : d)
                    for (int n = 0; n < numImages; n++) d += src[n] * h[n];
                    s += d;
                }
                float* t = targets + ((size_t)ch * fc * fp + k) * numFilters + f;
                *t = (scaleTargets == 0.f ? 0.f : scaleTargets * *t) + scaleOutput * s;
            }
        }
    }
}

ORC void orc_weight_acts(const float* images, const float* hidActs, float* targets,
                         int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                         int numFilters, int filterSize, int numModulesY, int numModulesX,
                         int paddingStart, int moduleStride, int numGroups, int partialSum,
                         float scaleTargets, float scaleOutput) {
    weight_acts_impl(images, hidActs, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize, numModulesY,
                     numModulesX, paddingStart, moduleStride, numGroups, partialSum, scaleTargets, scaleOutput, nullptr, 0);
}
// conv/localWeightActsSparse (weight_acts.cu:978-1200)
ORC void orc_weight_acts_sparse(const float* images, const float* hidActs, float* targets, const int* colorIndices,
                                int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors,
                                int numFilters, int filterSize, int numModulesY, int numModulesX,
                                int paddingStart, int moduleStride, int numGroups, int partialSum,
                                float scaleTargets, float scaleOutput) {
    weight_acts_impl(images, hidActs, targets, numImages, imgStride, imgSizeY, imgSizeX, numImgColors, numFilters, filterSize, numModulesY,
                     numModulesX, paddingStart, moduleStride, numGroups, partialSum, scaleTargets, scaleOutput, colorIndices, numFilterColors);
}

// ---------------------------------------------------------------------------
// convLocalPool (conv_util.cuh:127-191, host :327-517).  mode 0 = max (base
// -2e38, MaxPooler :85-96), 1 = avg over the CLIPPED window (AvgPooler :72-83).
// ---------------------------------------------------------------------------
ORC void orc_local_pool(const float* imgs, float* target, int numImages, int numFilters,
                        int imgSizeY, int imgSizeX, int subsX, int startX, int strideX,
                        int outputsY, int outputsX, int mode) {
    const int imgPixels = imgSizeY * imgSizeX, numOutputs = outputsY * outputsX;
#pragma omp parallel for collapse(2) schedule(static)
    for (int f = 0; f < numFilters; f++) {
        for (int o = 0; o < numOutputs; o++) {
            const int oy = o / outputsX, ox = o % outputsX;
            const int sy = startX + oy * strideX, sx = startX + ox * strideX;
            const int y0 = imax(0, sy), x0 = imax(0, sx);
            const int y1 = imin(imgSizeY, sy + subsX), x1 = imin(imgSizeX, sx + subsX);
            const int regionSize = (y1 - y0) * (x1 - x0);
            float* t = target + ((size_t)f * numOutputs + o) * numImages;
            for (int n = 0; n < numImages; n++) t[n] = mode == 0 ? -2e38f : 0.f;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    const float* s = imgs + ((size_t)f * imgPixels + y * imgSizeX + x) * numImages;
                    if (mode == 0) for (int n = 0; n < numImages; n++) t[n] = fmaxf(t[n], s[n]);
                    else for (int n = 0; n < numImages; n++) t[n] += s[n];
                }
            if (mode == 1) for (int n = 0; n < numImages; n++) t[n] = t[n] / regionSize;
        }
    }
}

static inline void pool_out_range(int px, int startX, int subsX, int strideX, int outputs, int& s, int& e) {
    // conv_util.cu:1512-1515
    s = px - startX < subsX ? 0 : 1 + (px - startX - subsX) / strideX;
    e = imin(outputs, 1 + (px - startX) / strideX);
}

// convLocalMaxUndo (conv_util.cu:1496-1587, host :1869-1958): every pixel equal
// to the window max receives the gradient (no single-argmax tie break).
ORC void orc_local_max_undo(const float* imgs, const float* maxGrads, const float* maxActs, float* target,
                            int numImages, int numFilters, int imgSizeY, int imgSizeX, int subsX,
                            int startX, int strideX, int outputsY, int outputsX,
                            float scaleTargets, float scaleOutput) {
    const int imgPixels = imgSizeY * imgSizeX, numOutputs = outputsY * outputsX;
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int f = 0; f < numFilters; f++) {
            for (int px = 0; px < imgPixels; px++) {
                const int y = px / imgSizeX, x = px % imgSizeX;
                std::fill(acc.begin(), acc.end(), 0.f);
                if (x >= startX && x < startX + strideX * (outputsX - 1) + subsX &&
                    y >= startX && y < startX + strideX * (outputsY - 1) + subsX) {
                    int sy, ey, sx, ex;
                    pool_out_range(y, startX, subsX, strideX, outputsY, sy, ey);
                    pool_out_range(x, startX, subsX, strideX, outputsX, sx, ex);
                    const float* im = imgs + ((size_t)f * imgPixels + px) * numImages;
                    for (int my = sy; my < ey; my++)
                        for (int mx = sx; mx < ex; mx++) {
                            const size_t o = ((size_t)f * numOutputs + my * outputsX + mx) * numImages;
                            for (int n = 0; n < numImages; n++)
                                acc[n] += (im[n] == maxActs[o + n]) * maxGrads[o + n];
                        }
                }
                float* t = target + ((size_t)f * imgPixels + px) * numImages;
                if (scaleTargets == 0.f) for (int n = 0; n < numImages; n++) t[n] = scaleOutput * acc[n];
                else for (int n = 0; n < numImages; n++) t[n] = scaleTargets * t[n] + scaleOutput * acc[n];
            }
        }
    }
}

// convLocalAvgUndo (conv_util.cu:1392-1475, host :1968-2062)
ORC void orc_local_avg_undo(const float* avgGrads, float* target, int numImages, int numFilters,
                            int imgSizeY, int imgSizeX, int subsX, int startX, int strideX,
                            int outputsY, int outputsX, float scaleTargets, float scaleOutput) {
    const int imgPixels = imgSizeY * imgSizeX, numOutputs = outputsY * outputsX;
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int f = 0; f < numFilters; f++) {
            for (int px = 0; px < imgPixels; px++) {
                const int y = px / imgSizeX, x = px % imgSizeX;
                std::fill(acc.begin(), acc.end(), 0.f);
                if (x >= startX && x < startX + strideX * (outputsX - 1) + subsX &&
                    y >= startX && y < startX + strideX * (outputsY - 1) + subsX) {
                    int sy, ey, sx, ex;
                    pool_out_range(y, startX, subsX, strideX, outputsY, sy, ey);
                    pool_out_range(x, startX, subsX, strideX, outputsX, sx, ex);
                    for (int my = sy; my < ey; my++) {
                        const float rsy = fminf(imgSizeY, startX + my * strideX + subsX) - fmaxf(0, startX + my * strideX);
                        for (int mx = sx; mx < ex; mx++) {
                            const float rsx = fminf(imgSizeX, startX + mx * strideX + subsX) - fmaxf(0, startX + mx * strideX);
                            const float inv = 1.0f / (rsx * rsy);
                            const size_t o = ((size_t)f * numOutputs + my * outputsX + mx) * numImages;
                            for (int n = 0; n < numImages; n++) acc[n] += avgGrads[o + n] * inv;
                        }
                    }
                }
                float* t = target + ((size_t)f * imgPixels + px) * numImages;
                if (scaleTargets == 0.f) for (int n = 0; n < numImages; n++) t[n] = scaleOutput * acc[n];
                else for (int n = 0; n < numImages; n++) t[n] = scaleTargets * t[n] + scaleOutput * acc[n];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Response normalisation within a map (conv_util.cu:829-893 kCNorm_manyfilter
// with meanDiffs == images, host convResponseNorm :2064-2072):
//   denom = 1 + addScale * sum_{window} x^2 ; y = x * denom^-pow
// ---------------------------------------------------------------------------
// convContrastNorm (conv_util.cu:2074-2160; kCNorm* :790-1000): the window sums squares of meanDiffs, the output
// scales images.  convResponseNorm (:2064-2066) is the same call with meanDiffs = images.
ORC void orc_contrast_norm(const float* images, const float* meanDiffs, float* denoms, float* target, int numImages,
                           int numFilters, int imgSizeY, int imgSizeX, int sizeX,
                           float addScale, float powScale) {
    const int imgPixels = imgSizeY * imgSizeX;
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int f = 0; f < numFilters; f++) {
            for (int px = 0; px < imgPixels; px++) {
                const int y = px / imgSizeX, x = px % imgSizeX;
                const int y0 = imax(0, y - sizeX / 2), x0 = imax(0, x - sizeX / 2);
                const int y1 = imin(imgSizeY, y - sizeX / 2 + sizeX), x1 = imin(imgSizeX, x - sizeX / 2 + sizeX);
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int yy = y0; yy < y1; yy++)
                    for (int xx = x0; xx < x1; xx++) {
                        const float* s = meanDiffs + ((size_t)f * imgPixels + yy * imgSizeX + xx) * numImages;
                        for (int n = 0; n < numImages; n++) acc[n] += s[n] * s[n];
                    }
                const size_t o = ((size_t)f * imgPixels + px) * numImages;
                for (int n = 0; n < numImages; n++) {
                    const float d = 1.f + addScale * acc[n];
                    denoms[o + n] = d;
                    target[o + n] = images[o + n] * powf(d, -powScale);
                }
            }
        }
    }
}

ORC void orc_response_norm(const float* images, float* denoms, float* target, int numImages,
                           int numFilters, int imgSizeY, int imgSizeX, int sizeX,
                           float addScale, float powScale) {
    orc_contrast_norm(images, images, denoms, target, numImages, numFilters, imgSizeY, imgSizeX, sizeX, addScale, powScale);
}

// convResponseNormUndo (conv_util.cu:1592-1605 prelims, :1629-1712 main, host
// :2228-2412).  OVERWRITES acts with -2*addScale*pow*outGrads*acts/denoms, then
// target = sT*target + sO*(inputs * sum_window(acts') + outGrads*denoms^-pow).
// The reference's size>=6 branch launches with the add flag inverted
// (:2270-2289); this restates the intended sT/sO semantics.
ORC void orc_response_norm_undo(const float* outGrads, const float* denoms, const float* inputs, float* acts,
                                float* target, int numImages, int numFilters, int imgSizeY, int imgSizeX,
                                int sizeX, float addScale, float powScale, float scaleTargets, float scaleOutput) {
    const int imgPixels = imgSizeY * imgSizeX;
    const size_t total = (size_t)numFilters * imgPixels * numImages;
    const float sc = -2 * addScale * powScale;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < total; i++) acts[i] = sc * outGrads[i] * acts[i] / denoms[i];
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int f = 0; f < numFilters; f++) {
            for (int px = 0; px < imgPixels; px++) {
                const int y = px / imgSizeX, x = px % imgSizeX;
                const int y0 = imax(0, y + sizeX / 2 - sizeX + 1), x0 = imax(0, x + sizeX / 2 - sizeX + 1);
                const int y1 = imin(imgSizeY, y + sizeX / 2 + 1), x1 = imin(imgSizeX, x + sizeX / 2 + 1);
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int yy = y0; yy < y1; yy++)
                    for (int xx = x0; xx < x1; xx++) {
                        const float* s = acts + ((size_t)f * imgPixels + yy * imgSizeX + xx) * numImages;
                        for (int n = 0; n < numImages; n++) acc[n] += s[n];
                    }
                const size_t o = ((size_t)f * imgPixels + px) * numImages;
                for (int n = 0; n < numImages; n++) {
                    const float v = inputs[o + n] * acc[n] + outGrads[o + n] * powf(denoms[o + n], -powScale);
                    target[o + n] = (scaleTargets == 0.f ? 0.f : scaleTargets * target[o + n]) + scaleOutput * v;
                }
            }
        }
    }
}

// convResponseNormCrossMap (conv_util.cu:1038-1089 kFCNorm, host :2743-2790, :2892)
ORC void orc_response_norm_cross_map(const float* images, float* denoms, float* target, int numImages,
                                     int numFilters, int imgPixels, int sizeF, float addScale,
                                     float powScale, int blocked) {
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int f = 0; f < numFilters; f++) {
            for (int px = 0; px < imgPixels; px++) {
                const int startF = blocked ? (f / sizeF) * sizeF : -sizeF / 2 + f;
                const int f0 = blocked ? startF : imax(0, startF);
                const int f1 = imin(numFilters, startF + sizeF);
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int ff = f0; ff < f1; ff++) {
                    const float* s = images + ((size_t)ff * imgPixels + px) * numImages;
                    for (int n = 0; n < numImages; n++) acc[n] += s[n] * s[n];
                }
                const size_t o = ((size_t)f * imgPixels + px) * numImages;
                for (int n = 0; n < numImages; n++) {
                    const float d = 1.f + addScale * acc[n];
                    denoms[o + n] = d;
                    target[o + n] = images[o + n] * powf(d, -powScale);
                }
            }
        }
    }
}

// convResponseNormCrossMapUndo (conv_util.cu:1112-1179 kFRNormUndo, host :2801-2890)
ORC void orc_response_norm_cross_map_undo(const float* outGrads, const float* denoms, const float* inputs,
                                          float* acts, float* target, int numImages, int numFilters,
                                          int imgPixels, int sizeF, float addScale, float powScale,
                                          int blocked, float scaleTargets, float scaleOutput) {
    const size_t total = (size_t)numFilters * imgPixels * numImages;
    const float sc = -2 * addScale * powScale;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < total; i++) acts[i] = sc * outGrads[i] * acts[i] / denoms[i];
#pragma omp parallel
    {
        std::vector<float> acc(numImages);
#pragma omp for collapse(2) schedule(static)
        for (int f = 0; f < numFilters; f++) {
            for (int px = 0; px < imgPixels; px++) {
                const int startF = blocked ? (f / sizeF) * sizeF : -sizeF + sizeF / 2 + 1 + f;
                const int f0 = blocked ? startF : imax(0, startF);
                const int f1 = imin(numFilters, startF + sizeF);
                std::fill(acc.begin(), acc.end(), 0.f);
                for (int ff = f0; ff < f1; ff++) {
                    const float* s = acts + ((size_t)ff * imgPixels + px) * numImages;
                    for (int n = 0; n < numImages; n++) acc[n] += s[n];
                }
                const size_t o = ((size_t)f * imgPixels + px) * numImages;
                for (int n = 0; n < numImages; n++) {
                    const float v = inputs[o + n] * acc[n] + outGrads[o + n] * powf(denoms[o + n], -powScale);
                    target[o + n] = (scaleTargets == 0.f ? 0.f : scaleTargets * target[o + n]) + scaleOutput * v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Neurons (neuron.cuh:64-456; makeNeuron neuron.cu:32-75).
// type: 0 ident 1 relu 2 logistic 3 brelu[a] 4 abs 5 tanh[a,b] 6 softrelu
//       7 square 8 sqrt 9 linear[a,b]
// ---------------------------------------------------------------------------
ORC void orc_neuron_fwd(const float* in, float* out, size_t n, int type, float a, float b) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        const float x = in[i];
        float y;
        switch (type) {
            case 1: y = x < 0.f ? 0.f : x; break;
            case 2: y = 1.f / (1.f + expf(-x)); break;
            case 3: y = x < 0.f ? 0.f : (x > a ? a : x); break;
            case 4: y = fabsf(x); break;
            case 5: y = a * (2.f / (1.f + expf(x * (-2 * b))) - 1.f); break;
            case 6: y = x > 4.f ? x : logf(1.f + expf(x)); break;
            case 7: y = x * x; break;
            case 8: y = sqrtf(x); break;
            case 9: y = a * x + b; break;
            default: y = x;
        }
        out[i] = y;
    }
}

// target = (add ? target : 0) + grad(actsGrad, acts-or-inputs).  `acts` is the
// neuron output for types that use it (relu/logistic/brelu/tanh/softrelu*/sqrt)
// and `inputs` the neuron input for abs/square (neuron.cuh:241-246,370-375).
// (*) the reference's SoftRelu gradient reads its OUTPUT but treats it as the
// input (neuron.cuh:324-349); restated literally.
ORC void orc_neuron_bwd(const float* actsGrad, const float* acts, const float* inputs, float* target,
                        size_t n, int type, float a, float b, int add) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        const float g = actsGrad[i];
        float v;
        switch (type) {
            case 1: v = g * (acts[i] > 0.f); break;
            case 2: v = g * acts[i] * (1.f - acts[i]); break;
            case 3: v = g * (acts[i] > 0.f) * (acts[i] < a); break;
            case 4: v = g * (inputs[i] > 0.f ? 1.f : -1.f); break;
            case 5: { const float t = (1.f - acts[i] / a) / 2.f; v = g * (-4 * a * b) * (t * (t - 1.f)); break; }
            case 6: { const float u = acts[i]; if (u > 4.f) v = g; else { const float e = expf(u); v = g * (e / (1.f + e)); } break; }
            case 7: v = g * 2.f * inputs[i]; break;
            case 8: v = g / (2.f * acts[i]); break;
            case 9: v = g * a; break;
            default: v = g;
        }
        target[i] = add ? target[i] + v : v;
    }
}

// ---------------------------------------------------------------------------
// NVMatrix verbs
// ---------------------------------------------------------------------------
// addProduct / rightMult (nvmatrix.cu:189-239 -> cublasSgemm).  Row-major
// C[MxN] = beta*C + alpha*op(A)[MxK]*op(B)[KxN]; ta/tb say A is stored [KxM] / B [NxK].
ORC void orc_gemm(const float* A, const float* B, float* C, int M, int N, int K, int ta, int tb,
                  float alpha, float beta) {
#pragma omp parallel
    {
        std::vector<float> acc(N);
#pragma omp for schedule(static)
        for (int i = 0; i < M; i++) {
            std::fill(acc.begin(), acc.end(), 0.f);
            if (!tb) {
                for (int k = 0; k < K; k++) {
                    const float a = ta ? A[(size_t)k * M + i] : A[(size_t)i * K + k];
                    const float* b = B + (size_t)k * N;
                    float* c = acc.data();
                    for (int j = 0; j < N; j++) c[j] += a * b[j];
                }
            } else {
                for (int j = 0; j < N; j++) {
                    const float* b = B + (size_t)j * K;
                    float d = 0.f;
                    if (!ta) {
                        const float* a = A + (size_t)i * K;
#pragma omp simd reduction(+ : d)
                        for (int k = 0; k < K; k++) d += a[k] * b[k];
                    } else {
                        for (int k = 0; k < K; k++) d += A[(size_t)k * M + i] * b[k];
                    }
                    acc[j] = d;
                }
            }
            float* c = C + (size_t)i * N;
            if (beta == 0.f) for (int j = 0; j < N; j++) c[j] = alpha * acc[j];
            else for (int j = 0; j < N; j++) c[j] = beta * c[j] + alpha * acc[j];
        }
    }
}

// addVector (nvmatrix.cu:652-690): mat (rows x cols) += scale * vec, where vec
// has `rows` entries (perRow=1: one value per row, kColVectorOp on a column
// vector) or `cols` entries (perRow=0).
ORC void orc_add_vector(float* mat, const float* vec, int rows, int cols, int perRow, float scale) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) mat[(size_t)r * cols + c] += scale * (perRow ? vec[r] : vec[c]);
}

// addSum (nvmatrix.cu:896-906 -> _aggregate :700-805): tgt = scaleThis*tgt +
// scaleSum * sum over axis.  axis=1: sum each row (tgt has `rows` entries);
// axis=0: sum each column (tgt has `cols` entries).
ORC void orc_add_sum(float* tgt, const float* mat, int rows, int cols, int axis, float scaleThis, float scaleSum) {
    if (axis == 1) {
#pragma omp parallel for schedule(static)
        for (int r = 0; r < rows; r++) {
            float s = 0.f;
            const float* m = mat + (size_t)r * cols;
#pragma omp simd reduction(+ : s)
            for (int c = 0; c < cols; c++) s += m[c];
            tgt[r] = (scaleThis == 0.f ? 0.f : scaleThis * tgt[r]) + scaleSum * s;
        }
    } else {
        std::vector<float> acc(cols, 0.f);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) acc[c] += mat[(size_t)r * cols + c];
        for (int c = 0; c < cols; c++) tgt[c] = (scaleThis == 0.f ? 0.f : scaleThis * tgt[c]) + scaleSum * acc[c];
    }
}

// NVMatrix::add (nvmatrix.cu eltwise WeightedAdd): a = sa*a + sb*b
ORC void orc_axpby(float* a, const float* b, size_t n, float sa, float sb) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) a[i] = sa * a[i] + sb * b[i];
}

// NVMatrix::sum() (nvmatrix.cu:936-979): total, finished in double on the host.
ORC double orc_sum_all(const float* a, size_t n) {
    double s = 0;
    for (size_t i = 0; i < n; i++) s += a[i];
    return s;
}

// ---------------------------------------------------------------------------
// Softmax / costs
// ---------------------------------------------------------------------------
// SoftmaxLayer::fpropActs (layer.cu:585-595): per case p = exp(x-max)/sum.
// x, p are (numOut, numCases) row-major.
ORC void orc_softmax(const float* x, float* p, int numOut, int numCases) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < numCases; n++) {
        float mx = x[n];
        for (int k = 1; k < numOut; k++) mx = fmaxf(mx, x[(size_t)k * numCases + n]);
        float s = 0.f;
        for (int k = 0; k < numOut; k++) {
            const float e = expf(x[(size_t)k * numCases + n] - mx);
            p[(size_t)k * numCases + n] = e;
            s += e;
        }
        for (int k = 0; k < numOut; k++) p[(size_t)k * numCases + n] /= s;
    }
}

// computeLogregCost + kLogregCost (layer_kernels.cu:59-92, :216-239)
ORC void orc_logreg_cost(const float* labels, const float* probs, float* labelLogProbs, float* correctProbs,
                         int numOut, int numCases) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < numCases; n++) {
        float maxp = probs[n];
        for (int k = 1; k < numOut; k++) maxp = fmaxf(maxp, probs[(size_t)k * numCases + n]);
        const int label = (int)labels[n];
        const float labelp = probs[(size_t)label * numCases + n];
        labelLogProbs[n] = logf(labelp);
        if (labelp != maxp) {
            correctProbs[n] = 0.f;
        } else {
            int numMax = 0;
            for (int k = 0; k < numOut; k++) numMax += probs[(size_t)k * numCases + n] == maxp;
            correctProbs[n] = 1.0f / (float)numMax;
        }
    }
}

// computeLogregSoftmaxGrad + kLogregSoftmaxGrad (layer_kernels.cu:154-170, :287-308)
ORC void orc_logreg_softmax_grad(const float* labels, const float* probs, float* target, int numOut,
                                 int numCases, int add, float coeff) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < numOut; k++)
        for (int n = 0; n < numCases; n++) {
            const int label = (int)labels[n];
            const float v = coeff * ((label == k) - probs[(size_t)k * numCases + n]);
            float* t = target + (size_t)k * numCases + n;
            *t = add ? *t + v : v;
        }
}

// computeLogregGrad + kLogregCostGrad (layer_kernels.cu:101-118, :241-263)
ORC void orc_logreg_grad(const float* labels, const float* probs, float* target, int numOut, int numCases,
                         int add, float coeff) {
    for (int k = 0; k < numOut; k++)
        for (int n = 0; n < numCases; n++) {
            const int label = (int)labels[n];
            const float v = coeff * (label == k) / probs[(size_t)k * numCases + n];
            float* t = target + (size_t)k * numCases + n;
            *t = add ? *t + v : v;
        }
}

// computeSoftmaxGrad + kSoftmaxGrad (layer_kernels.cu:126-145, :265-285)
ORC void orc_softmax_grad(const float* acts, const float* actsGrad, float* target, int numOut, int numCases, int add) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < numCases; n++) {
        // v_k = y_k * sum_j dE/dy_j * ((j==k) - y_j) = y_k * (g_k - sum_j g_j y_j)
        for (int k = 0; k < numOut; k++) {
            float v = 0.f;
            for (int j = 0; j < numOut; j++)
                v += actsGrad[(size_t)j * numCases + n] * ((j == k) - acts[(size_t)j * numCases + n]);
            v *= acts[(size_t)k * numCases + n];
            float* t = target + (size_t)k * numCases + n;
            *t = add ? *t + v : v;
        }
    }
}

// L2-SN cost / gradient (layer_kernels.cu:482-735): see convnet_oracle_nofma.cpp (built with -ffp-contract=off).

// computeL2regCost (layer_kernels.cu:744-765): per-case sum_k (pred-gt)^2 (no 1/2)
ORC void orc_l2reg_cost(const float* ground, const float* data, float* output, int numD, int numCases) {
    for (int n = 0; n < numCases; n++) {
        float s = 0.f;
        for (int k = 0; k < numD; k++) {
            const float d = data[(size_t)k * numCases + n] - ground[(size_t)k * numCases + n];
            s += d * d;
        }
        output[n] = s;
    }
}

// computeL2regGrad (layer_kernels.cu:771-794): target (+)= coeff*(ground - data)
ORC void orc_l2reg_grad(const float* ground, const float* data, float* target, int numD, int numCases,
                        int add, float coeff) {
    const size_t n = (size_t)numD * numCases;
    for (size_t i = 0; i < n; i++) {
        const float v = (ground[i] - data[i]) * coeff;
        target[i] = add ? target[i] + v : v;
    }
}

// Weights::update (weights.cuh:150-163): useGrad: inc = mom*inc + grad;
// wc>0: inc -= wc*eps*W ; W += inc.
ORC void orc_sgd_update(float* W, float* inc, const float* grad, size_t n, float mom, float wc, float eps, int useGrad) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        float v = inc[i];
        if (useGrad) v = mom * v + 1.f * grad[i];
        if (wc > 0) v = 1.f * v + (-wc * eps) * W[i];
        inc[i] = v;
        W[i] = W[i] + v;
    }
}

ORC int orc_version() { return 1; }

// ---------------------------------------------------------------------------
// Image-op layers outside the DeepID stacks.
// ---------------------------------------------------------------------------
// convBedOfNails / convBedOfNailsUndo (conv_util.cu:461-556 kernel, host :561-680): the undo host code zeroes the
// target when scaleTargets == 0 (:673-676) and the kernel then touches the nail positions only.
ORC void orc_bed_of_nails(const float* imgs, float* target, int numImages, int numChannels, int imgSizeY, int imgSizeX,
                          int startX, int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput) {
    for (int c = 0; c < numChannels; c++)
        for (int oy = 0; oy < outputsY; oy++)
            for (int ox = 0; ox < outputsX; ox++) {
                const float* s = imgs + ((size_t)c * imgSizeY * imgSizeX + (size_t)(startX + oy * strideX) * imgSizeX + startX + ox * strideX) * numImages;
                float* t = target + ((size_t)c * outputsY * outputsX + (size_t)oy * outputsX + ox) * numImages;
                for (int n = 0; n < numImages; n++) t[n] = scaleTargets != 0 ? scaleTargets * t[n] + scaleOutput * s[n] : scaleOutput * s[n];
            }
}
ORC void orc_bed_of_nails_undo(const float* grads, float* target, int numImages, int numChannels, int imgSizeY, int imgSizeX,
                               int startX, int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput) {
    const size_t total = (size_t)numChannels * imgSizeY * imgSizeX * numImages;
    if (scaleTargets == 0) for (size_t i = 0; i < total; i++) target[i] = 0.f;
    for (int c = 0; c < numChannels; c++)
        for (int oy = 0; oy < outputsY; oy++)
            for (int ox = 0; ox < outputsX; ox++) {
                const float* g = grads + ((size_t)c * outputsY * outputsX + (size_t)oy * outputsX + ox) * numImages;
                float* t = target + ((size_t)c * imgSizeY * imgSizeX + (size_t)(startX + oy * strideX) * imgSizeX + startX + ox * strideX) * numImages;
                for (int n = 0; n < numImages; n++) t[n] = scaleTargets * t[n] + scaleOutput * g[n];  // (target was zeroed when scaleTargets == 0)
            }
}
// kGaussianBlur (conv_util.cu:356-448), restated with its own data flow: a queue of 2*radius partial outputs that every
// pixel of a line adds to (taps 0 .. 2*radius-1; the last tap enters through the symmetry f[2r] == f[0], :418), the head
// of the queue leaving as a finished output once `radius` pixels are in.  Lines run along x (horiz) or y.
ORC void orc_gaussian_blur(const float* imgs, const float* filter, float* target, int numImages, int numChannels, int imgSizeY,
                           int imgSizeX, int filterSize, int horiz, float scaleTargets, float scaleOutput) {
    const int radius = filterSize / 2, len = horiz ? imgSizeX : imgSizeY, lines = horiz ? imgSizeY : imgSizeX;
    const size_t step = (size_t)(horiz ? 1 : imgSizeX) * numImages, lineStep = (size_t)(horiz ? imgSizeX : 1) * numImages;
    std::vector<float> out(2 * radius + 1);
    for (int c = 0; c < numChannels; c++)
        for (int l = 0; l < lines; l++)
            for (int n = 0; n < numImages; n++) {
                const float* s = imgs + (size_t)c * imgSizeY * imgSizeX * numImages + l * lineStep + n;
                float* t = target + (size_t)c * imgSizeY * imgSizeX * numImages + l * lineStep + n;
                std::fill(out.begin(), out.end(), 0.f);
                int written = 0;
                auto emit = [&](float v) {
                    float* d = t + (size_t)written * step;
                    *d = scaleTargets != 0 ? scaleTargets * *d + scaleOutput * v : scaleOutput * v;
                    written++;
                };
                const int lead = radius < len ? radius : len;
                for (int col = 0; col < lead; col++) {  // :395-402
                    const float px = s[(size_t)col * step];
                    for (int r = 0; r < radius + 1 + col && r < 2 * radius; r++) out[r] += px * filter[radius + col - r];
                }
                for (int col = lead; col < len; col++) {  // :405-418 / :427-437
                    const float px = s[(size_t)col * step];
                    emit(out[0] + px * filter[0]);
                    for (int r = 1; r < 2 * radius; r++) out[r - 1] = out[r] + px * filter[r];
                    out[2 * radius - 1] = px * filter[0];
                }
                for (int r = 0; r < radius && written < len; r++) emit(out[r]);  // :420-424 / :439-442
            }
}
// kResizeBilinear (conv_util.cu:288-340), centerScale from the host code (:2427-2429)
ORC void orc_resize_bilinear(const float* imgs, float* target, int numImages, int numChannels, int imgSize, int tgtSize, float scale) {
    const float centerScale = (float)(imgSize * 0.5 - tgtSize * 0.5 * (double)scale);
    for (int c = 0; c < numChannels; c++)
        for (int pxY = 0; pxY < tgtSize; pxY++)
            for (int pxX = 0; pxX < tgtSize; pxX++) {
                const float srcPxX = fmaxf(0.0f, fminf((float)imgSize - 1.01f, (float)pxX * scale + centerScale));
                const float srcPxY = fmaxf(0.0f, fminf((float)imgSize - 1.01f, (float)pxY * scale + centerScale));
                const float u = floorf(srcPxX + 1) - srcPxX, w = srcPxY - floorf(srcPxY);
                const int p0 = (int)floorf(srcPxY) * imgSize + (int)floorf(srcPxX);
                const float* s = imgs + (size_t)c * imgSize * imgSize * numImages;
                float* t = target + ((size_t)c * tgtSize * tgtSize + (size_t)pxY * tgtSize + pxX) * numImages;
                for (int n = 0; n < numImages; n++) {
                    const float v0 = s[(size_t)p0 * numImages + n], v1 = s[(size_t)(p0 + 1) * numImages + n];
                    const float v2 = s[(size_t)(p0 + imgSize) * numImages + n], v3 = s[(size_t)(p0 + imgSize + 1) * numImages + n];
                    const float c0 = u * (v0 - v1) + v1, c1 = u * (v2 - v3) + v3;
                    t[n] = w * (c1 - c0) + c0;
                }
            }
}
// kRGBToYUV (conv_util.cu:170-196) / kRGBToLAB (:198-277)
ORC void orc_rgb_to_yuv(const float* imgs, float* target, int numImages, int imgPixels) {
    const size_t cs = (size_t)imgPixels * numImages;
    for (size_t i = 0; i < cs; i++) {
        const float R = imgs[i], G = imgs[cs + i], B = imgs[2 * cs + i];
        target[i] = 0.2126f * R + 0.7152f * G + 0.0722f * B;
        target[cs + i] = -0.09991f * R + -0.33609f * G + 0.436f * B;
        target[2 * cs + i] = 0.615f * R + -0.55861f * G + -0.05639f * B;
    }
}
static inline float orc_labf(float x) { return x > 0.0088564517f ? powf(x, 0.3333f) : 7.787037f * x + 0.13793103f; }
ORC void orc_rgb_to_lab(const float* imgs, float* target, int numImages, int imgPixels, int center) {
    const size_t cs = (size_t)imgPixels * numImages;
    for (size_t i = 0; i < cs; i++) {
        const float R = imgs[i], G = imgs[cs + i], B = imgs[2 * cs + i];
        const float X = 0.49f * R + 0.31f * G + 0.2f * B, Y = 0.17697f * R + 0.8124f * G + 0.01063f * B, Z = 0.01f * G + 0.99f * B;
        const float lx = orc_labf(X), ly = orc_labf(Y), lz = orc_labf(Z);
        target[i] = 116.0f * ly - 16.0f - (center ? 50.0f : 0.f);
        target[cs + i] = 500.0f * (lx - ly);
        target[2 * cs + i] = 200.0f * (ly - lz);
    }
}
// convShift (conv_util.cu:2902-2990): the host loop of the reference
ORC void orc_shift(const float* imgs, const float* filter, float* target, int numImages, int numChannels, int imgPixels, int tgtPixels,
                   int numFilters) {
    for (int n = 0; n < numImages; n++)
        for (int c = 0; c < numChannels; c++)
            for (int f = 0; f < numFilters; f++)
                for (int i = 0; i < tgtPixels; i++) {
                    const size_t y = (size_t)f * numChannels * tgtPixels + (size_t)c * tgtPixels + i;
                    const size_t dy = (size_t)c * imgPixels + (size_t)(int)filter[(size_t)i * numFilters + f];
                    target[y * numImages + n] = imgs[dy * numImages + n];
                }
}
