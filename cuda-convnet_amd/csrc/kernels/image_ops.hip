// Image-op layers of the reference outside the DeepID stacks: bed of nails (sub-sampling) forward / undo,
// separable Gaussian blur, bilinear resize, RGB -> YUV / L*a*b*, and the pixel gather of the `shift` layer.
//
// Replaces conv_util.cu:461-556 (kBedOfNails, both directions) + host :561-680, :356-448 (kGaussianBlur) + host :682-745,
// :288-340 (kResizeBilinear) + host :2416-2468, :170-196 (kRGBToYUV), :198-277 (labf, kRGBToLAB), and convShift
// (:2902-2990, which the reference runs on the HOST after copying images, filter and target back).
//
// All of them stream (channels, pixels, images) arrays whose fastest dimension is the image index: a thread owns one
// (row, image) element, a wave 64 consecutive images of a row.  HBM-bound pre-processing, no LDS.
#include "common.h"

#define ELEM_DECODE(total_rows)                                         \
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;    \
    if (gid >= (long long)(total_rows) * numImages) return;             \
    const int row = (int)(gid / numImages);                             \
    const int n = (int)(gid % numImages);

static inline unsigned elem_blocks(long long rows, int numImages) { return (unsigned)((rows * numImages + 255) / 256); }

// ---- bed of nails: target[c, oy, ox] = sT*target + sO*imgs[c, start + oy*stride, start + ox*stride]
__global__ __launch_bounds__(256) void k_bed_of_nails(const float* __restrict__ imgs, float* __restrict__ target, int numImages,
                                                      int numChannels, int imgY, int imgX, int start, int stride, int outY, int outX,
                                                      float sT, float sO) {
    ELEM_DECODE(numChannels * outY * outX)
    const int c = row / (outY * outX), o = row % (outY * outX);
    const int y = start + (o / outX) * stride, x = start + (o % outX) * stride;
    const float v = sO * imgs[((size_t)c * imgY * imgX + (size_t)y * imgX + x) * numImages + n];
    float* t = target + (size_t)row * numImages + n;
    *t = sT != 0.f ? sT * *t + v : v;
}
// undo: a thread per image pixel; pixels between the nails are zeroed when sT == 0 (the reference zeroes the matrix first,
// conv_util.cu:673-676) and left alone otherwise (its kernel touches the nails only)
__global__ __launch_bounds__(256) void k_bed_of_nails_undo(const float* __restrict__ grads, float* __restrict__ target, int numImages,
                                                           int numChannels, int imgY, int imgX, int start, int stride, int outY, int outX,
                                                           float sT, float sO) {
    ELEM_DECODE(numChannels * imgY * imgX)
    const int c = row / (imgY * imgX), px = row % (imgY * imgX);
    const int y = px / imgX - start, x = px % imgX - start;
    float v = 0.f;
    bool nail = y >= 0 && x >= 0 && y % stride == 0 && x % stride == 0 && y / stride < outY && x / stride < outX;
    if (nail) v = sO * grads[((size_t)c * outY * outX + (size_t)(y / stride) * outX + x / stride) * numImages + n];
    float* t = target + (size_t)row * numImages + n;
    if (nail) *t = (sT != 0.f ? sT * *t : 0.f) + v;
    else if (sT == 0.f) *t = 0.f;  // (sT != 0: the reference's kernel touches the nails only)
}

// ---- one direction of a separable blur with a symmetric filter of 2*radius + 1 taps, zero outside the image:
// out[x] = sum_{d=-radius..radius} img[x + d] * filter[radius + d]
__global__ __launch_bounds__(256) void k_gaussian_blur(const float* __restrict__ imgs, const float* __restrict__ filter,
                                                       float* __restrict__ target, int numImages, int numChannels, int imgY, int imgX,
                                                       int radius, int horiz, float sT, float sO) {
    ELEM_DECODE(numChannels * imgY * imgX)
    const int px = row % (imgY * imgX);
    const int y = px / imgX, x = px % imgX;
    const int pos = horiz ? x : y, len = horiz ? imgX : imgY;
    const ptrdiff_t step = (ptrdiff_t)(horiz ? 1 : imgX) * numImages;
    const float* src = imgs + (size_t)row * numImages + n;
    float acc = 0.f;
    for (int d = -radius; d <= radius; d++)
        if (pos + d >= 0 && pos + d < len) acc += src[d * step] * filter[radius + d];
    float* t = target + (size_t)row * numImages + n;
    *t = (sT != 0.f ? sT * *t : 0.f) + sO * acc;
}

// ---- bilinear resize (kResizeBilinear): source position = clamp(px*scale + centerScale, 0, imgSize - 1.01)
__global__ __launch_bounds__(256) void k_resize_bilinear(const float* __restrict__ imgs, float* __restrict__ target, int numImages,
                                                         int numChannels, int imgSize, int tgtSize, float scale, float centerScale) {
    ELEM_DECODE(numChannels * tgtSize * tgtSize)
    const int c = row / (tgtSize * tgtSize), o = row % (tgtSize * tgtSize);
    const int pxY = o / tgtSize, pxX = o % tgtSize;
    const float lim = (float)imgSize - 1.01f;
    const float sx = fmaxf(0.0f, fminf(lim, (float)pxX * scale + centerScale));
    const float sy = fmaxf(0.0f, fminf(lim, (float)pxY * scale + centerScale));
    const float u = floorf(sx + 1) - sx;
    const float w = sy - floorf(sy);
    const int p0 = (int)floorf(sy) * imgSize + (int)floorf(sx);
    const float* src = imgs + (size_t)c * imgSize * imgSize * numImages + n;
    const float v0 = src[(size_t)p0 * numImages], v1 = src[(size_t)(p0 + 1) * numImages];
    const float v2 = src[(size_t)(p0 + imgSize) * numImages], v3 = src[(size_t)(p0 + imgSize + 1) * numImages];
    const float c0 = u * (v0 - v1) + v1;
    const float c1 = u * (v2 - v3) + v3;
    target[(size_t)row * numImages + n] = w * (c1 - c0) + c0;
}

// ---- colour spaces: thread per (pixel, image), three channels in, three out
__device__ __forceinline__ float lab_f(float x) {  // conv_util.cu:198-203 (the reference raises to 0.3333 with the fast intrinsic)
    return x > 0.0088564517f ? powf(x, 0.3333f) : 7.787037f * x + 0.13793103f;
}
template <int MODE>  // 0: YUV, 1: LAB, 2: LAB centred
__global__ __launch_bounds__(256) void k_rgb_to(const float* __restrict__ imgs, float* __restrict__ target, int numImages, int imgPixels) {
    ELEM_DECODE(imgPixels)
    const size_t cs = (size_t)imgPixels * numImages, off = (size_t)row * numImages + n;
    const float R = imgs[off], G = imgs[cs + off], B = imgs[2 * cs + off];
    if (MODE == 0) {
        target[off] = 0.2126f * R + 0.7152f * G + 0.0722f * B;
        target[cs + off] = -0.09991f * R + -0.33609f * G + 0.436f * B;
        target[2 * cs + off] = 0.615f * R + -0.55861f * G + -0.05639f * B;
    } else {
        const float X = 0.49f * R + 0.31f * G + 0.2f * B;
        const float Y = 0.17697f * R + 0.8124f * G + 0.01063f * B;
        const float Z = 0.01f * G + 0.99f * B;
        const float lx = lab_f(X), ly = lab_f(Y), lz = lab_f(Z);
        target[off] = 116.0f * ly - 16.0f - (MODE == 2 ? 50.0f : 0.f);
        target[cs + off] = 500.0f * (lx - ly);
        target[2 * cs + off] = 200.0f * (ly - lz);
    }
}

// ---- shift: target[(f*C + c)*T + i] = imgs[c*imgPixels + int(filter[i*numFilters + f])]
__global__ __launch_bounds__(256) void k_shift_gather(const float* __restrict__ imgs, const float* __restrict__ filter,
                                                      float* __restrict__ target, int numImages, int numChannels, int imgPixels,
                                                      int tgtPixels, int numFilters) {
    ELEM_DECODE(numFilters * numChannels * tgtPixels)
    const int i = row % tgtPixels, c = (row / tgtPixels) % numChannels, f = row / (tgtPixels * numChannels);
    const int src = min(max((int)filter[(size_t)i * numFilters + f], 0), imgPixels - 1);  // (the reference trusts the table)
    target[(size_t)row * numImages + n] = imgs[((size_t)c * imgPixels + src) * numImages + n];
}

#define IMG_LAUNCH(kernel, rows, ...) \
    MCN_LAUNCH(kernel, dim3(elem_blocks(rows, numImages)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

MCN_API int mcn_bed_of_nails(const float* images, float* target, int numImages, int numChannels, int imgSizeY, int imgSizeX, int startX,
                             int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput, void* stream) {
    MCN_REQUIRE(images && target && numImages > 0 && numChannels > 0 && imgSizeY > 0 && imgSizeX > 0 && strideX > 0 && outputsY > 0 && outputsX > 0,
                "bedOfNails: bad argument");
    MCN_REQUIRE(startX >= 0 && startX + (outputsX - 1) * strideX < imgSizeX && startX + (outputsY - 1) * strideX < imgSizeY,
                "bedOfNails: nails outside the image");
    IMG_LAUNCH(k_bed_of_nails, (long long)numChannels * outputsY * outputsX, images, target, numImages, numChannels, imgSizeY, imgSizeX, startX,
               strideX, outputsY, outputsX, scaleTargets, scaleOutput);
    return MCN_LAUNCH_CHECK("bedOfNails");
}
MCN_API int mcn_bed_of_nails_undo(const float* actsGrad, float* target, int numImages, int numChannels, int imgSizeY, int imgSizeX, int startX,
                                  int strideX, int outputsY, int outputsX, float scaleTargets, float scaleOutput, void* stream) {
    MCN_REQUIRE(actsGrad && target && numImages > 0 && numChannels > 0 && imgSizeY > 0 && imgSizeX > 0 && strideX > 0 && outputsY > 0 &&
                outputsX > 0 && startX >= 0, "bedOfNailsUndo: bad argument");
    IMG_LAUNCH(k_bed_of_nails_undo, (long long)numChannels * imgSizeY * imgSizeX, actsGrad, target, numImages, numChannels, imgSizeY, imgSizeX,
               startX, strideX, outputsY, outputsX, scaleTargets, scaleOutput);
    return MCN_LAUNCH_CHECK("bedOfNailsUndo");
}
MCN_API int mcn_gaussian_blur(const float* images, const float* filter, float* target, int numImages, int numChannels, int imgSizeY,
                              int imgSizeX, int filterSize, int horiz, float scaleTargets, float scaleOutput, void* stream) {
    MCN_REQUIRE(images && filter && target && numImages > 0 && numChannels > 0 && imgSizeY > 0 && imgSizeX > 0, "gaussianBlur: bad argument");
    MCN_REQUIRE(filterSize >= 1 && filterSize % 2 == 1, "gaussianBlur: filterSize must be odd");
    MCN_REQUIRE(images != target, "gaussianBlur: in-place blurring is not supported (every output reads its neighbours)");
    IMG_LAUNCH(k_gaussian_blur, (long long)numChannels * imgSizeY * imgSizeX, images, filter, target, numImages, numChannels, imgSizeY, imgSizeX,
               filterSize / 2, horiz ? 1 : 0, scaleTargets, scaleOutput);
    return MCN_LAUNCH_CHECK("gaussianBlur");
}
MCN_API int mcn_resize_bilinear(const float* images, float* target, int numImages, int numChannels, int imgSize, int tgtSize, float scale,
                                void* stream) {
    MCN_REQUIRE(images && target && numImages > 0 && numChannels > 0 && imgSize >= 2 && tgtSize > 0 && scale > 0, "resizeBilinear: bad argument");
    const double centerScale = imgSize * 0.5 - tgtSize * 0.5 * (double)scale;  // conv_util.cu:2427-2429
    IMG_LAUNCH(k_resize_bilinear, (long long)numChannels * tgtSize * tgtSize, images, target, numImages, numChannels, imgSize, tgtSize, scale,
               (float)centerScale);
    return MCN_LAUNCH_CHECK("resizeBilinear");
}
MCN_API int mcn_rgb_to_yuv(const float* images, float* target, int numImages, int imgPixels, void* stream) {
    MCN_REQUIRE(images && target && numImages > 0 && imgPixels > 0, "rgbToYUV: bad argument");
    IMG_LAUNCH((k_rgb_to<0>), (long long)imgPixels, images, target, numImages, imgPixels);
    return MCN_LAUNCH_CHECK("rgbToYUV");
}
MCN_API int mcn_rgb_to_lab(const float* images, float* target, int numImages, int imgPixels, int center, void* stream) {
    MCN_REQUIRE(images && target && numImages > 0 && imgPixels > 0, "rgbToLAB: bad argument");
    if (center) IMG_LAUNCH((k_rgb_to<2>), (long long)imgPixels, images, target, numImages, imgPixels);
    else IMG_LAUNCH((k_rgb_to<1>), (long long)imgPixels, images, target, numImages, imgPixels);
    return MCN_LAUNCH_CHECK("rgbToLAB");
}
MCN_API int mcn_shift(const float* images, const float* filter, float* target, int numImages, int numChannels, int imgPixels, int tgtPixels,
                      int numFilters, void* stream) {
    MCN_REQUIRE(images && filter && target && numImages > 0 && numChannels > 0 && imgPixels > 0 && tgtPixels > 0 && numFilters > 0,
                "shift: bad argument");
    IMG_LAUNCH(k_shift_gather, (long long)numFilters * numChannels * tgtPixels, images, filter, target, numImages, numChannels, imgPixels,
               tgtPixels, numFilters);
    return MCN_LAUNCH_CHECK("shift");
}
