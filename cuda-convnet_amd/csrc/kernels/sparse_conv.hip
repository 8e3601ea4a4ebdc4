// Random-sparse connectivity: filter group g reads the numFilterColors image channels listed in
// colorIndices[g*numFilterColors ..], a concatenation of overSample = numGroups*numFilterColors/numImgColors permutations
// of the image channels (layer.py:830-853).
//
// Replaces conv/localFilterActsSparse (cudaconv2.cuh:80-94, filter_acts.cu:401-560, :1221-1444), conv/localImgActsSparse
// (cudaconv2.cuh:110-124, img_acts.cu:556-890, :1929-2583) and conv/localWeightActsSparse (cudaconv2.cuh:96-108,
// weight_acts.cu:414-560, :978-1200).
//
// The reference has a second family of kernels that chase the channel table inside the inner loops.  Here the table is
// applied ONCE per direction by a streaming pass (a channel is a contiguous imgPixels*numImages block, so the pass moves
// whole rows at HBM speed) and the products themselves are the ordinary grouped MFMA kernels over the re-ordered tensor:
//   forward / weight gradient:  gathered[k] = images[colorIndices[k]], k < numGroups*numFilterColors, then the dense grouped
//                               product with numImgColors' = numGroups*numFilterColors;
//   backward-data:              the dense grouped product into (numGroups*numFilterColors, pixels, images), then
//                               targets[c] = scaleTargets*targets[c] + scaleOutput * sum over the overSample permutations of
//                               the row that permutation sends to c (fixed order: deterministic).
// The layer graph keeps `gathered` between forward and the weight gradient (mcn_gather_channels is public for that).
#include "common.h"

__global__ __launch_bounds__(256) void k_gather_channels(const float* __restrict__ images, int imgStride, const int* __restrict__ idx,
                                                          float* __restrict__ target, int numImages, int numImgColors, int numIdx,
                                                          int imgPixels) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)numIdx * imgPixels * numImages) return;
    const int row = (int)(gid / numImages), n = (int)(gid % numImages);
    const int k = row / imgPixels, p = row % imgPixels;
    const int c = min(max(idx[k], 0), numImgColors - 1);
    target[(size_t)row * numImages + n] = images[((size_t)c * imgPixels + p) * imgStride + n];
}

// inverse[s*numImgColors + colorIndices[s*numImgColors + j]] = s*numImgColors + j
__global__ __launch_bounds__(256) void k_invert_channel_indices(const int* __restrict__ idx, int* __restrict__ inverse, int numImgColors,
                                                                 int total) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int s = t / numImgColors;
    inverse[s * numImgColors + min(max(idx[t], 0), numImgColors - 1)] = t;
}

__global__ __launch_bounds__(256) void k_scatter_channels_sum(const float* __restrict__ src, const int* __restrict__ inverse,
                                                               float* __restrict__ target, int numImages, int numImgColors, int overSample,
                                                               int imgPixels, float sT, float sO) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)numImgColors * imgPixels * numImages) return;
    const int row = (int)(gid / numImages), n = (int)(gid % numImages);
    const int c = row / imgPixels, p = row % imgPixels;
    float acc = 0.f;
    for (int s = 0; s < overSample; s++) acc += src[((size_t)inverse[s * numImgColors + c] * imgPixels + p) * numImages + n];
    float* t = target + (size_t)row * numImages + n;
    *t = (sT != 0.f ? sT * *t : 0.f) + sO * acc;
}

static inline unsigned blocks_for(long long elems) { return (unsigned)((elems + 255) / 256); }

static int sparse_shape_ok(const char* what, int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors, int numGroups) {
    MCN_REQUIRE(numImages > 0 && imgSizeY > 0 && imgSizeX > 0 && numImgColors > 0 && numFilterColors > 0 && numGroups > 1,
                "%s: bad argument (random-sparse connectivity needs numGroups > 1)", what);
    MCN_REQUIRE(numFilterColors % 4 == 0, "%s: numFilterColors must be divisible by 4 (layer.py:912)", what);
    MCN_REQUIRE(numImgColors % numFilterColors == 0 && ((long long)numGroups * numFilterColors) % numImgColors == 0,
                "%s: numFilterColors must divide numImgColors and numImgColors must divide numGroups*numFilterColors (layer.py:911-913)", what);
    return MCN_OK;
}

MCN_API size_t mcn_sparse_gathered_bytes(int numImages, int imgSizeY, int imgSizeX, int numFilterColors, int numGroups) {
    if (numImages <= 0 || imgSizeY <= 0 || imgSizeX <= 0 || numFilterColors <= 0 || numGroups <= 0) return 0;
    return (size_t)numGroups * numFilterColors * imgSizeY * imgSizeX * numImages * sizeof(float);
}
MCN_API size_t mcn_sparse_img_acts_workspace_bytes(int numImages, int imgSizeY, int imgSizeX, int numFilterColors, int numGroups) {
    const size_t g = mcn_sparse_gathered_bytes(numImages, imgSizeY, imgSizeX, numFilterColors, numGroups);
    return g ? g + (((size_t)numGroups * numFilterColors * sizeof(int) + 255) & ~(size_t)255) : 0;
}

MCN_API int mcn_gather_channels(const float* images, int imgStride, const int* colorIndices, float* target, int numImages, int numImgColors,
                                int numIndices, int imgPixels, void* stream) {
    MCN_REQUIRE(images && colorIndices && target && numImages > 0 && imgStride >= numImages && numImgColors > 0 && numIndices > 0 && imgPixels > 0,
                "gatherChannels: bad argument");
    MCN_REQUIRE(images != target, "gatherChannels: not in place");
    MCN_LAUNCH(k_gather_channels, dim3(blocks_for((long long)numIndices * imgPixels * numImages)), dim3(256), 0, (hipStream_t)stream,
                       images, imgStride, colorIndices, target, numImages, numImgColors, numIndices, imgPixels);
    return MCN_LAUNCH_CHECK("gatherChannels");
}
MCN_API int mcn_invert_channel_indices(const int* colorIndices, int* inverse, int numImgColors, int overSample, void* stream) {
    MCN_REQUIRE(colorIndices && inverse && numImgColors > 0 && overSample > 0, "invertChannelIndices: bad argument");
    const int total = numImgColors * overSample;
    MCN_LAUNCH(k_invert_channel_indices, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, colorIndices, inverse, numImgColors,
                       total);
    return MCN_LAUNCH_CHECK("invertChannelIndices");
}
MCN_API int mcn_scatter_channels_sum(const float* gathered, const int* inverse, float* target, int numImages, int numImgColors, int overSample,
                                     int imgPixels, float scaleTargets, float scaleOutput, void* stream) {
    MCN_REQUIRE(gathered && inverse && target && numImages > 0 && numImgColors > 0 && overSample > 0 && imgPixels > 0,
                "scatterChannelsSum: bad argument");
    MCN_REQUIRE(gathered != target, "scatterChannelsSum: not in place");
    MCN_LAUNCH(k_scatter_channels_sum, dim3(blocks_for((long long)numImgColors * imgPixels * numImages)), dim3(256), 0,
                       (hipStream_t)stream, gathered, inverse, target, numImages, numImgColors, overSample, imgPixels, scaleTargets, scaleOutput);
    return MCN_LAUNCH_CHECK("scatterChannelsSum");
}

static int filter_acts_sparse(bool conv, const float* images, const float* filters, float* targets, const int* colorIndices, void* workspace,
                              size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                              int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX, int paddingStart,
                              int moduleStride, int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    const char* what = conv ? "convFilterActsSparse" : "localFilterActsSparse";
    if (int rc = sparse_shape_ok(what, numImages, imgSizeY, imgSizeX, numImgColors, numFilterColors, numGroups)) return rc;
    MCN_REQUIRE(images && filters && targets && colorIndices && workspace, "%s: null pointer", what);
    if (workspaceBytes < mcn_sparse_gathered_bytes(numImages, imgSizeY, imgSizeX, numFilterColors, numGroups)) {
        mcn_set_error("%s: workspace too small (mcn_sparse_gathered_bytes)", what);
        return MCN_EWORKSPACE;
    }
    const int wide = numGroups * numFilterColors;
    if (int rc = mcn_gather_channels(images, imgStride, colorIndices, (float*)workspace, numImages, numImgColors, wide, imgSizeY * imgSizeX, stream))
        return rc;
    auto fn = conv ? mcn_conv_filter_acts : mcn_local_filter_acts;
    return fn((const float*)workspace, filters, targets, numImages, numImages, imgSizeY, imgSizeX, wide, numFilters, filterSize, numModulesY,
              numModulesX, paddingStart, moduleStride, numGroups, scaleTargets, scaleOutput, stream);
}
MCN_API int mcn_conv_filter_acts_sparse(const float* images, const float* filters, float* targets, const int* colorIndices, void* workspace,
                                        size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                        int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                        int paddingStart, int moduleStride, int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    return filter_acts_sparse(true, images, filters, targets, colorIndices, workspace, workspaceBytes, numImages, imgStride, imgSizeY, imgSizeX,
                              numImgColors, numFilterColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride,
                              numGroups, scaleTargets, scaleOutput, stream);
}
MCN_API int mcn_local_filter_acts_sparse(const float* images, const float* filters, float* targets, const int* colorIndices, void* workspace,
                                         size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                         int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                         int paddingStart, int moduleStride, int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    return filter_acts_sparse(false, images, filters, targets, colorIndices, workspace, workspaceBytes, numImages, imgStride, imgSizeY, imgSizeX,
                              numImgColors, numFilterColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride,
                              numGroups, scaleTargets, scaleOutput, stream);
}

static int img_acts_sparse(bool conv, const float* hidActs, const float* filters, float* targets, const int* colorIndices, void* workspace,
                           size_t workspaceBytes, int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors,
                           int numFilters, int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride, int numGroups,
                           float scaleTargets, float scaleOutput, void* stream) {
    const char* what = conv ? "convImgActsSparse" : "localImgActsSparse";
    if (int rc = sparse_shape_ok(what, numImages, imgSizeY, imgSizeX, numImgColors, numFilterColors, numGroups)) return rc;
    MCN_REQUIRE(hidActs && filters && targets && colorIndices && workspace, "%s: null pointer", what);
    if (workspaceBytes < mcn_sparse_img_acts_workspace_bytes(numImages, imgSizeY, imgSizeX, numFilterColors, numGroups)) {
        mcn_set_error("%s: workspace too small (mcn_sparse_img_acts_workspace_bytes)", what);
        return MCN_EWORKSPACE;
    }
    const int wide = numGroups * numFilterColors, overSample = wide / numImgColors;
    int* inverse = (int*)workspace;
    float* tmp = (float*)((char*)workspace + (((size_t)wide * sizeof(int) + 255) & ~(size_t)255));
    if (int rc = mcn_invert_channel_indices(colorIndices, inverse, numImgColors, overSample, stream)) return rc;
    auto fn = conv ? mcn_conv_img_acts : mcn_local_img_acts;
    if (int rc = fn(hidActs, filters, tmp, numImages, imgSizeY, imgSizeX, wide, numFilters, filterSize, numModulesY, numModulesX, paddingStart,
                    moduleStride, numGroups, 0.f, 1.f, stream))
        return rc;
    return mcn_scatter_channels_sum(tmp, inverse, targets, numImages, numImgColors, overSample, imgSizeY * imgSizeX, scaleTargets, scaleOutput,
                                    stream);
}
MCN_API int mcn_conv_img_acts_sparse(const float* hidActs, const float* filters, float* targets, const int* colorIndices, void* workspace,
                                     size_t workspaceBytes, int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors,
                                     int numFilters, int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                     int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    return img_acts_sparse(true, hidActs, filters, targets, colorIndices, workspace, workspaceBytes, numImages, imgSizeY, imgSizeX, numImgColors,
                           numFilterColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups,
                           scaleTargets, scaleOutput, stream);
}
MCN_API int mcn_local_img_acts_sparse(const float* hidActs, const float* filters, float* targets, const int* colorIndices, void* workspace,
                                      size_t workspaceBytes, int numImages, int imgSizeY, int imgSizeX, int numImgColors, int numFilterColors,
                                      int numFilters, int filterSize, int numModulesY, int numModulesX, int paddingStart, int moduleStride,
                                      int numGroups, float scaleTargets, float scaleOutput, void* stream) {
    return img_acts_sparse(false, hidActs, filters, targets, colorIndices, workspace, workspaceBytes, numImages, imgSizeY, imgSizeX, numImgColors,
                           numFilterColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups,
                           scaleTargets, scaleOutput, stream);
}

static int weight_acts_sparse(bool conv, const float* images, const float* hidActs, float* targets, const int* colorIndices, void* workspace,
                              size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                              int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX, int paddingStart,
                              int moduleStride, int numGroups, int partialSum, float scaleTargets, float scaleOutput, void* stream) {
    const char* what = conv ? "convWeightActsSparse" : "localWeightActsSparse";
    if (int rc = sparse_shape_ok(what, numImages, imgSizeY, imgSizeX, numImgColors, numFilterColors, numGroups)) return rc;
    MCN_REQUIRE(images && hidActs && targets && colorIndices && workspace, "%s: null pointer", what);
    if (workspaceBytes < mcn_sparse_gathered_bytes(numImages, imgSizeY, imgSizeX, numFilterColors, numGroups)) {
        mcn_set_error("%s: workspace too small (mcn_sparse_gathered_bytes)", what);
        return MCN_EWORKSPACE;
    }
    const int wide = numGroups * numFilterColors;
    if (int rc = mcn_gather_channels(images, imgStride, colorIndices, (float*)workspace, numImages, numImgColors, wide, imgSizeY * imgSizeX, stream))
        return rc;
    if (conv)
        return mcn_conv_weight_acts((const float*)workspace, hidActs, targets, numImages, numImages, imgSizeY, imgSizeX, wide, numFilters,
                                    filterSize, numModulesY, numModulesX, paddingStart, moduleStride, numGroups, partialSum, scaleTargets,
                                    scaleOutput, stream);
    return mcn_local_weight_acts((const float*)workspace, hidActs, targets, numImages, numImages, imgSizeY, imgSizeX, wide, numFilters, filterSize,
                                 numModulesY, numModulesX, paddingStart, moduleStride, numGroups, scaleTargets, scaleOutput, stream);
}
MCN_API int mcn_conv_weight_acts_sparse(const float* images, const float* hidActs, float* targets, const int* colorIndices, void* workspace,
                                        size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                        int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                        int paddingStart, int moduleStride, int numGroups, int partialSum, float scaleTargets,
                                        float scaleOutput, void* stream) {
    return weight_acts_sparse(true, images, hidActs, targets, colorIndices, workspace, workspaceBytes, numImages, imgStride, imgSizeY, imgSizeX,
                              numImgColors, numFilterColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride,
                              numGroups, partialSum, scaleTargets, scaleOutput, stream);
}
MCN_API int mcn_local_weight_acts_sparse(const float* images, const float* hidActs, float* targets, const int* colorIndices, void* workspace,
                                         size_t workspaceBytes, int numImages, int imgStride, int imgSizeY, int imgSizeX, int numImgColors,
                                         int numFilterColors, int numFilters, int filterSize, int numModulesY, int numModulesX,
                                         int paddingStart, int moduleStride, int numGroups, float scaleTargets, float scaleOutput,
                                         void* stream) {
    return weight_acts_sparse(false, images, hidActs, targets, colorIndices, workspace, workspaceBytes, numImages, imgStride, imgSizeY, imgSizeX,
                              numImgColors, numFilterColors, numFilters, filterSize, numModulesY, numModulesX, paddingStart, moduleStride,
                              numGroups, 0, scaleTargets, scaleOutput, stream);
}
