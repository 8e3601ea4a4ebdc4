// Element-wise / reduction / cost kernels: neurons, NVMatrix verbs, softmax,
// logistic-regression cost and gradients, the Joint identification-verification
// (L2-SN) cost and gradient, L2 regression cost, and the fused SGD update.
//
// Replaces nvmatrix_kernels.cuh:91-265 (eltwise + vector ops), :272-593
// (aggregations), layer_kernels.cu:59-170 / :482-794 and weights.cuh:150-163.
// All are HBM- or latency-bound; they use 16-byte accesses along the contiguous
// dimension and fixed-order reductions (deterministic results).
#include <algorithm>
#include <vector>
#include "common.h"

// No multiply-add contraction in this file: the reference runs every NVMatrix op of the
// cost layers as its own kernel (layer_kernels.cu:505-735), so products and sums are rounded
// separately, and the L2-SN margin / 1[d<m] decisions are discontinuous in the last bit.
// HIP's __fmul_rn/__fadd_rn are plain operators defined in a header, so they would be fused
// under the default -ffp-contract=fast: the Makefile builds this file with -ffp-contract=off.
#pragma clang fp contract(off)

#define GRID_STRIDE(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

static inline unsigned ew_blocks(size_t n, int per = 256) { return (unsigned)std::min<size_t>(divup_sz(n, per), 16384); }
static inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

// ---------------------------------------------------------------- neurons
__global__ __launch_bounds__(256) void k_neuron_fwd(const float* in, float* out, size_t n, int type, float a, float b) {
    GRID_STRIDE(i, n) out[i] = neuron_apply(type, in[i], a, b);
}
__global__ __launch_bounds__(256) void k_relu_fwd4(const f32x4* in, f32x4* out, size_t n4) {
    GRID_STRIDE(i, n4) {
        f32x4 v = in[i];
        for (int t = 0; t < 4; t++) v[t] = v[t] < 0.f ? 0.f : v[t];
        out[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_neuron_bwd(const float* g, const float* acts, const float* inputs, float* target,
                                                    size_t n, int type, float a, float b, int add) {
    GRID_STRIDE(i, n) {
        const float v = neuron_grad(type, g[i], acts ? acts[i] : 0.f, inputs ? inputs[i] : 0.f, a, b);
        target[i] = add ? target[i] + v : v;
    }
}
__global__ __launch_bounds__(256) void k_relu_bwd4(const f32x4* g, const f32x4* acts, f32x4* target, size_t n4, int add) {
    GRID_STRIDE(i, n4) {
        f32x4 v = g[i];
        const f32x4 y = acts[i];
        for (int t = 0; t < 4; t++) v[t] = v[t] * (y[t] > 0.f ? 1.f : 0.f);
        if (add) v += target[i];
        target[i] = v;
    }
}

MCN_API int mcn_neuron_fwd(const float* in, float* out, size_t n, int neuron, float a, float b, void* stream) {
    MCN_REQUIRE(in && out && neuron_valid(neuron), "neuronFwd: bad argument");
    if (n == 0) return MCN_OK;
    hipStream_t s = (hipStream_t)stream;
    if (neuron == MCN_NEURON_IDENT) {
        if (in != out) return mcn_check_hip(hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, s), "neuronFwd copy");
        return MCN_OK;
    }
    if (neuron == MCN_NEURON_RELU && n % 4 == 0 && al16(in) && al16(out))
        MCN_LAUNCH(k_relu_fwd4, dim3(ew_blocks(n / 4)), dim3(256), 0, s, (const f32x4*)in, (f32x4*)out, n / 4);
    else
        MCN_LAUNCH(k_neuron_fwd, dim3(ew_blocks(n)), dim3(256), 0, s, in, out, n, neuron, a, b);
    return MCN_LAUNCH_CHECK("neuronFwd");
}

MCN_API int mcn_neuron_bwd(const float* actsGrad, const float* acts, const float* inputs, float* target, size_t n,
                           int neuron, float a, float b, int add, void* stream) {
    MCN_REQUIRE(actsGrad && target && neuron_valid(neuron), "neuronBwd: bad argument");
    const bool needsIn = neuron == MCN_NEURON_ABS || neuron == MCN_NEURON_SQUARE;
    const bool needsAct = !needsIn && neuron != MCN_NEURON_IDENT && neuron != MCN_NEURON_LINEAR;
    MCN_REQUIRE(!needsIn || inputs, "neuronBwd: neuron %d needs its inputs", neuron);
    MCN_REQUIRE(!needsAct || acts, "neuronBwd: neuron %d needs its outputs", neuron);
    if (n == 0) return MCN_OK;
    hipStream_t s = (hipStream_t)stream;
    if (neuron == MCN_NEURON_IDENT && !add && target == actsGrad) return MCN_OK;
    if (neuron == MCN_NEURON_RELU && n % 4 == 0 && al16(actsGrad) && al16(acts) && al16(target))
        MCN_LAUNCH(k_relu_bwd4, dim3(ew_blocks(n / 4)), dim3(256), 0, s, (const f32x4*)actsGrad, (const f32x4*)acts,
                           (f32x4*)target, n / 4, add);
    else
        MCN_LAUNCH(k_neuron_bwd, dim3(ew_blocks(n)), dim3(256), 0, s, actsGrad, acts, inputs, target, n, neuron, a, b, add);
    return MCN_LAUNCH_CHECK("neuronBwd");
}

// ---------------------------------------------------------------- NVMatrix verbs
__global__ __launch_bounds__(256) void k_add_vector(float* mat, const float* vec, int rows, int cols, int perRow, float scale) {
    const size_t n = (size_t)rows * cols;
    GRID_STRIDE(i, n) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        mat[i] += scale * (perRow ? vec[r] : vec[c]);
    }
}
MCN_API int mcn_add_vector(float* mat, const float* vec, int rows, int cols, int perRow, float scale, void* stream) {
    MCN_REQUIRE(mat && vec && rows > 0 && cols > 0, "addVector: bad argument");
    MCN_LAUNCH(k_add_vector, dim3(ew_blocks((size_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, mat, vec, rows, cols, perRow, scale);
    return MCN_LAUNCH_CHECK("addVector");
}

// Row sums.  Stage 1: block (row, chunk) reduces CH columns in a fixed order -> partial[row][chunk].
// Stage 2 (or the only stage when one chunk covers the row): tgt = scaleThis*tgt + scaleSum*sum.
#define ROWSUM_CHUNK 8192
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}
// (row, chunk) of one matrix by one block
__device__ __forceinline__ void row_sum_chunk(const float* __restrict__ mat, float* __restrict__ out, int cols, int chunks, int finalStage,
                                              float scaleThis, float scaleSum, int row, int chunk, float* sh) {
    const int c0 = chunk * ROWSUM_CHUNK, c1 = min(cols, c0 + ROWSUM_CHUNK);
    const float* m = mat + (size_t)row * cols;
    float v = 0.f;
    if (cols % 4 == 0 && ((uintptr_t)mat) % 16 == 0) {
        // the whole chunk in one batch of independent 16-byte loads per thread (one memory round trip)
        const f32x4* m4 = reinterpret_cast<const f32x4*>(m + c0);
        const int n4 = (c1 - c0) / 4;
        f32x4 t[ROWSUM_CHUNK / 1024];
#pragma unroll
        for (int u = 0; u < ROWSUM_CHUNK / 1024; u++) {
            const int i = threadIdx.x + 256 * u;
            t[u] = i < n4 ? m4[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < ROWSUM_CHUNK / 1024; u++) v += (t[u][0] + t[u][1]) + (t[u][2] + t[u][3]);
    } else {
        for (int c = c0 + threadIdx.x; c < c1; c += 256) v += m[c];
    }
    const float s = block_sum256(v, sh);
    if (threadIdx.x == 0) {
        if (finalStage) out[row] = (scaleThis != 0.f ? scaleThis * out[row] : 0.f) + scaleSum * s;
        else out[(size_t)row * chunks + chunk] = s;
    }
}
__global__ __launch_bounds__(256) void k_row_sum(const float* __restrict__ mat, float* __restrict__ out, int cols, int chunks,
                                                 int finalStage, float scaleThis, float scaleSum) {
    __shared__ float sh[4];
    row_sum_chunk(mat, out, cols, chunks, finalStage, scaleThis, scaleSum, blockIdx.x, blockIdx.y, sh);
}
// short rows (cols <= 256): one wave per row, 4 rows per block
__device__ __forceinline__ void row_sum_short4(const float* __restrict__ mat, float* __restrict__ out, int rows, int cols, float scaleThis,
                                               float scaleSum, int block) {
    const int row = block * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* m = mat + (size_t)row * cols;
    float v = 0.f;
    for (int c = lane; c < cols; c += 64) v += m[c];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) out[row] = (scaleThis != 0.f ? scaleThis * out[row] : 0.f) + scaleSum * v;
}
__global__ __launch_bounds__(256) void k_row_sum_short(const float* __restrict__ mat, float* __restrict__ out, int rows, int cols,
                                                       float scaleThis, float scaleSum) {
    row_sum_short4(mat, out, rows, cols, scaleThis, scaleSum, blockIdx.x);
}
// The row sums of up to ROWSUM_MAX_ITEMS matrices in one launch (+ one more when a matrix has rows longer than a chunk): the
// bias gradients of every layer of a backward pass are row sums over 0.1-19 MB each, one to two launches of 2-5 us per
// layer in the reference's and this library's per-matrix form.  Every (matrix, row) is summed by the same code and in the
// same order as in mcn_add_sum, so the results are bit-identical to per-matrix calls.
#define ROWSUM_MAX_ITEMS 8
struct RowSumSeg {
    const float* mat;
    float* out;
    int rows, cols, chunks;  // chunks == 0: short rows (4 per block)
    int finalStage;
    int blockBase;           // first block of this segment
    float scaleThis, scaleSum;
};
struct RowSumMulti {
    RowSumSeg seg[ROWSUM_MAX_ITEMS];
    int count;
};
__global__ __launch_bounds__(256) void k_row_sums_multi(const RowSumMulti p) {
    __shared__ float sh[4];
    int i = 0;
#pragma unroll
    for (int k = 1; k < ROWSUM_MAX_ITEMS; k++)
        if (k < p.count && (int)blockIdx.x >= p.seg[k].blockBase) i = k;
    const RowSumSeg& g = p.seg[i];
    const int local = blockIdx.x - g.blockBase;
    if (g.chunks == 0) row_sum_short4(g.mat, g.out, g.rows, g.cols, g.scaleThis, g.scaleSum, local);
    else row_sum_chunk(g.mat, g.out, g.cols, g.chunks, g.finalStage, g.scaleThis, g.scaleSum, local / g.chunks, local % g.chunks, sh);
}
__global__ __launch_bounds__(256) void k_col_sum(const float* __restrict__ mat, float* __restrict__ out, int rows, int cols,
                                                 float scaleThis, float scaleSum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = 0;
    for (; r + 3 < rows; r += 4) {
        s0 += mat[(size_t)r * cols + c];
        s1 += mat[(size_t)(r + 1) * cols + c];
        s2 += mat[(size_t)(r + 2) * cols + c];
        s3 += mat[(size_t)(r + 3) * cols + c];
    }
    for (; r < rows; r++) s0 += mat[(size_t)r * cols + c];
    const float s = (s0 + s1) + (s2 + s3);
    out[c] = (scaleThis != 0.f ? scaleThis * out[c] : 0.f) + scaleSum * s;
}

MCN_API size_t mcn_add_sum_workspace_bytes(int rows, int cols, int axis) {
    if (axis != 1 || cols <= ROWSUM_CHUNK) return 0;
    return (size_t)rows * divup(cols, ROWSUM_CHUNK) * sizeof(float);
}
MCN_API int mcn_add_sum(float* tgt, const float* mat, int rows, int cols, int axis, float scaleThis, float scaleSum,
                        void* workspace, size_t workspaceBytes, void* stream) {
    MCN_REQUIRE(tgt && mat && rows > 0 && cols > 0 && (axis == 0 || axis == 1), "addSum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (axis == 0) {
        MCN_LAUNCH(k_col_sum, dim3(divup(cols, 256)), dim3(256), 0, s, mat, tgt, rows, cols, scaleThis, scaleSum);
    } else if (cols <= 256) {
        MCN_LAUNCH(k_row_sum_short, dim3(divup(rows, 4)), dim3(256), 0, s, mat, tgt, rows, cols, scaleThis, scaleSum);
    } else if (cols <= ROWSUM_CHUNK) {
        MCN_LAUNCH(k_row_sum, dim3(rows, 1), dim3(256), 0, s, mat, tgt, cols, 1, 1, scaleThis, scaleSum);
    } else {
        const int chunks = divup(cols, ROWSUM_CHUNK);
        const size_t need = (size_t)rows * chunks * sizeof(float);
        if (!workspace || workspaceBytes < need) {
            mcn_set_error("addSum: workspace %zu < %zu bytes", workspaceBytes, need);
            return MCN_EWORKSPACE;
        }
        MCN_REQUIRE(chunks <= 65535, "addSum: row too long");
        MCN_LAUNCH(k_row_sum, dim3(rows, chunks), dim3(256), 0, s, mat, (float*)workspace, cols, chunks, 0, 0.f, 1.f);
        if (chunks <= 256)
            MCN_LAUNCH(k_row_sum_short, dim3(divup(rows, 4)), dim3(256), 0, s, (const float*)workspace, tgt, rows, chunks, scaleThis, scaleSum);
        else
            MCN_LAUNCH(k_row_sum, dim3(rows, 1), dim3(256), 0, s, (const float*)workspace, tgt, chunks, 1, 1, scaleThis, scaleSum);
    }
    return MCN_LAUNCH_CHECK("addSum");
}

MCN_API size_t mcn_add_sum_multi_workspace_bytes(const mcn_row_sum_item* items, int count) {
    size_t need = 0;
    for (int i = 0; i < count; i++) need += mcn_add_sum_workspace_bytes(items[i].rows, items[i].cols, 1);
    return need;
}
static int add_sum_multi(const mcn_row_sum_item* items, int count, void* workspace, size_t workspaceBytes, void* stream, float* const* stage1Out);
MCN_API int mcn_add_sum_multi(const mcn_row_sum_item* items, int count, void* workspace, size_t workspaceBytes, void* stream) {
    return add_sum_multi(items, count, workspace, workspaceBytes, stream, nullptr);
}
// Only the first of the two launches: matrices with rows of at most 8192 columns are finished, every other matrix i leaves
// its partial row sums -- (rows_i x chunks_i) floats, chunks_i = ceil(cols_i / 8192) -- in partialsOut[i] for a consumer
// that finishes them itself (the fused update, mcn_sgd_plan_build_fused).
MCN_API int mcn_add_sum_multi_stage1(const mcn_row_sum_item* items, int count, float* const* partialsOut, void* stream) {
    MCN_REQUIRE(partialsOut, "addSumMultiStage1: null pointer");
    return add_sum_multi(items, count, nullptr, 0, stream, partialsOut);
}
static int add_sum_multi(const mcn_row_sum_item* items, int count, void* workspace, size_t workspaceBytes, void* stream, float* const* stage1Out) {
    MCN_REQUIRE(items && count > 0 && count <= ROWSUM_MAX_ITEMS, "addSumMulti: 1..%d matrices per call", ROWSUM_MAX_ITEMS);
    const size_t need = stage1Out ? 0 : mcn_add_sum_multi_workspace_bytes(items, count);
    if (need > 0 && (!workspace || workspaceBytes < need)) {
        mcn_set_error("addSumMulti: workspace %zu < %zu bytes", workspaceBytes, need);
        return MCN_EWORKSPACE;
    }
    RowSumMulti s1, s2;
    s1.count = s2.count = 0;
    int blocks1 = 0, blocks2 = 0;
    float* part = (float*)workspace;
    for (int i = 0; i < count; i++) {
        const mcn_row_sum_item& it = items[i];
        MCN_REQUIRE(it.tgt && it.mat && it.rows > 0 && it.cols > 0, "addSumMulti: bad matrix %d", i);
        RowSumSeg& g = s1.seg[s1.count++];
        g.mat = it.mat; g.rows = it.rows; g.cols = it.cols; g.blockBase = blocks1;
        g.scaleThis = it.scaleThis; g.scaleSum = it.scaleSum;
        if (it.cols <= 256) {  // as in mcn_add_sum
            g.out = it.tgt; g.chunks = 0; g.finalStage = 1;
            blocks1 += divup(it.rows, 4);
        } else if (it.cols <= ROWSUM_CHUNK) {
            g.out = it.tgt; g.chunks = 1; g.finalStage = 1;
            blocks1 += it.rows;
        } else {
            const int chunks = divup(it.cols, ROWSUM_CHUNK);
            if (stage1Out) {
                MCN_REQUIRE(stage1Out[i], "addSumMultiStage1: matrix %d needs a partials buffer", i);
                part = stage1Out[i];
            }
            g.out = part; g.chunks = chunks; g.finalStage = 0; g.scaleThis = 0.f; g.scaleSum = 1.f;
            blocks1 += it.rows * chunks;
            RowSumSeg& h = s2.seg[s2.count++];
            h.mat = part; h.out = it.tgt; h.rows = it.rows; h.cols = chunks; h.blockBase = blocks2;
            h.scaleThis = it.scaleThis; h.scaleSum = it.scaleSum; h.finalStage = 1;
            if (chunks <= 256) { h.chunks = 0; blocks2 += divup(it.rows, 4); }
            else { h.chunks = 1; blocks2 += it.rows; MCN_REQUIRE(chunks <= ROWSUM_CHUNK, "addSumMulti: row too long"); }
            part += (size_t)it.rows * chunks;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    MCN_LAUNCH(k_row_sums_multi, dim3(blocks1), dim3(256), 0, s, s1);
    if (s2.count > 0 && !stage1Out) MCN_LAUNCH(k_row_sums_multi, dim3(blocks2), dim3(256), 0, s, s2);
    return MCN_LAUNCH_CHECK("addSumMulti");
}

__global__ __launch_bounds__(256) void k_axpby(float* a, const float* b, size_t n, float sa, float sb) {
    GRID_STRIDE(i, n) a[i] = sa * a[i] + sb * b[i];
}
MCN_API int mcn_axpby(float* a, const float* b, size_t n, float sa, float sb, void* stream) {
    MCN_REQUIRE(a && b, "axpby: null pointer");
    if (n == 0) return MCN_OK;
    MCN_LAUNCH(k_axpby, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, sa, sb);
    return MCN_LAUNCH_CHECK("axpby");
}
__global__ __launch_bounds__(256) void k_scale(const float* in, float* out, size_t n, float sc) {
    GRID_STRIDE(i, n) out[i] = sc * in[i];
}
MCN_API int mcn_scale(const float* in, float* out, size_t n, float sc, void* stream) {
    MCN_REQUIRE(in && out, "scale: null pointer");
    if (n == 0) return MCN_OK;
    MCN_LAUNCH(k_scale, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, in, out, n, sc);
    return MCN_LAUNCH_CHECK("scale");
}
__global__ __launch_bounds__(256) void k_eltmax(const float* a, const float* b, float* out, size_t n) {
    GRID_STRIDE(i, n) out[i] = fmaxf(a[i], b[i]);
}
MCN_API int mcn_eltwise_max(const float* a, const float* b, float* out, size_t n, void* stream) {
    MCN_REQUIRE(a && b && out, "eltwiseMax: null pointer");
    if (n == 0) return MCN_OK;
    MCN_LAUNCH(k_eltmax, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    return MCN_LAUNCH_CHECK("eltwiseMax");
}
__global__ __launch_bounds__(256) void k_eltmax_grad(const float* g, const float* in, const float* out, float* tgt, size_t n, int add) {
    GRID_STRIDE(i, n) {
        const float v = g[i] * (out[i] == in[i] ? 1.f : 0.f);
        tgt[i] = add ? tgt[i] + v : v;
    }
}
MCN_API int mcn_eltwise_max_grad(const float* actGrad, const float* input, const float* output, float* target, size_t n,
                                 int add, void* stream) {
    MCN_REQUIRE(actGrad && input && output && target, "eltwiseMaxGrad: null pointer");
    if (n == 0) return MCN_OK;
    MCN_LAUNCH(k_eltmax_grad, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, actGrad, input, output, target, n, add);
    return MCN_LAUNCH_CHECK("eltwiseMaxGrad");
}

// total sum: single block (inputs on this path are cost vectors of <= a few thousand entries)
__global__ __launch_bounds__(256) void k_sum_all(const float* a, size_t n, double* out, int accumulate) {
    __shared__ double sh[256];
    double v = 0;
    for (size_t i = threadIdx.x; i < n; i += 256) v += (double)a[i];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0.0) + sh[0];
}
MCN_API int mcn_sum_all(const float* a, size_t n, double* outDevice, int accumulate, void* stream) {
    MCN_REQUIRE(a && outDevice, "sumAll: null pointer");
    MCN_LAUNCH(k_sum_all, dim3(1), dim3(256), 0, (hipStream_t)stream, a, n, outDevice, accumulate);
    return MCN_LAUNCH_CHECK("sumAll");
}

__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (by + r < rows && bx + tx < cols) tile[r][tx] = src[(size_t)(by + r) * cols + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < cols && by + tx < rows) dst[(size_t)(bx + r) * rows + by + tx] = tile[tx][r];
}
MCN_API int mcn_transpose(const float* src, float* dst, int rows, int cols, void* stream) {
    MCN_REQUIRE(src && dst && rows > 0 && cols > 0 && src != dst, "transpose: bad argument");
    MCN_LAUNCH(k_transpose, dim3(divup(cols, 32), divup(rows, 32)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
    return MCN_LAUNCH_CHECK("transpose");
}

// ---------------------------------------------------------------- softmax
// x, p are (numOut, numCases) row-major; the softmax runs down each column.  Stage 1: block
// (caseTile of 64, rowChunk of SM_ROWS) computes per-column (max, sum exp(x - max)) over its rows;
// stage 2: every block merges the chunk partials of its 64 columns (same fixed order in every block, so
// all rows of a column see bit-identical M and S) and writes p = exp(x - M) / S for its own rows.
#define SM_ROWS 64  // rows per chunk: K = 10575 -> 166 chunks x 2 case tiles = 332 blocks
__global__ __launch_bounds__(256) void k_softmax_partial(const float* __restrict__ x, float* __restrict__ part, int numOut, int numCases, int chunks) {
    __shared__ float shm[4][64], shs[4][64];
    const int l = threadIdx.x & 63, col = blockIdx.x * 64 + l, rl = threadIdx.x >> 6, chunk = blockIdx.y;
    const int r0 = chunk * SM_ROWS, r1 = min(numOut, r0 + SM_ROWS);
    float v[SM_ROWS / 4];
    float m = -INFINITY, s = 0.f;
#pragma unroll
    for (int i = 0; i < SM_ROWS / 4; i++) {
        const int r = r0 + rl + 4 * i;
        v[i] = (col < numCases && r < r1) ? x[(size_t)r * numCases + col] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    shm[rl][l] = m;
    __syncthreads();
    m = fmaxf(fmaxf(shm[0][l], shm[1][l]), fmaxf(shm[2][l], shm[3][l]));
#pragma unroll
    for (int i = 0; i < SM_ROWS / 4; i++) s += v[i] == -INFINITY ? 0.f : expf(v[i] - m);
    shs[rl][l] = s;
    __syncthreads();
    if (rl == 0 && col < numCases) {
        s = (shs[0][l] + shs[1][l]) + (shs[2][l] + shs[3][l]);
        part[((size_t)chunk * numCases + col) * 2] = m;
        part[((size_t)chunk * numCases + col) * 2 + 1] = s;
    }
}
// 16 columns x 16 merge lanes per block: lane g merges chunks g, g+16, ... (one batch of independent loads),
// the 16 lanes are combined through LDS in a fixed order, then the block writes its 64 rows x 16 columns.
#define SMF_COLS 16
#define SMF_ROWS 256  // rows per block of k_softmax_final
#define SMF_MAXB 16  // chunks per merge lane held in registers per batch
__device__ __forceinline__ void softmax_merge16(const f32x2* pp, int numCases, int chunks, int col, bool live, int g, int cl,
                                                float (*shm)[SMF_COLS], float (*shs)[SMF_COLS], float& M, float& S) {
    M = -INFINITY;
    S = 0.f;
    float lm = -INFINITY, ls = 0.f;
    for (int c0 = g; c0 < chunks; c0 += 16 * SMF_MAXB) {
        f32x2 v[SMF_MAXB];
#pragma unroll
        for (int u = 0; u < SMF_MAXB; u++) {
            const int c = c0 + 16 * u;
            v[u] = (live && c < chunks) ? pp[(size_t)c * numCases + col] : (f32x2){-INFINITY, 0.f};
        }
        float bm = lm;
#pragma unroll
        for (int u = 0; u < SMF_MAXB; u++) bm = fmaxf(bm, v[u][0]);
        float bs = (lm == -INFINITY) ? 0.f : ls * expf(lm - bm);
#pragma unroll
        for (int u = 0; u < SMF_MAXB; u++) bs += (v[u][0] == -INFINITY) ? 0.f : v[u][1] * expf(v[u][0] - bm);
        lm = bm;
        ls = bs;
    }
    shm[g][cl] = lm;
    shs[g][cl] = ls;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) M = fmaxf(M, shm[k][cl]);
#pragma unroll
    for (int k = 0; k < 16; k++) S += (shm[k][cl] == -INFINITY) ? 0.f : shs[k][cl] * expf(shm[k][cl] - M);
}
// labels / grad (optional): also write grad[k,n] = coeff*((label_n == k) - p[k,n]), the gradient a following logreg cost
// sends back through this softmax (kLogregSoftmaxGrad layer_kernels.cu:154-170), while p is still in registers
__global__ __launch_bounds__(256) void k_softmax_final(const float* __restrict__ x, float* __restrict__ p, const float* __restrict__ part,
                                                       int numOut, int numCases, int chunks, const float* __restrict__ labels,
                                                       float* __restrict__ grad, float coeff) {
    __shared__ float shm[16][SMF_COLS], shs[16][SMF_COLS];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4, col = blockIdx.x * SMF_COLS + cl, chunk = blockIdx.y;
    const bool live = col < numCases;
    float M, S;
    softmax_merge16(reinterpret_cast<const f32x2*>(part), numCases, chunks, col, live, g, cl, shm, shs, M, S);
    if (!live) return;
    const int label = grad ? (int)labels[col] : -1;
    const float rS = 1.0f / S;
    // a block normalises SMF_ROWS rows of its 16 columns, so that the merge above is repeated by few blocks
    const int r0 = chunk * SMF_ROWS;
    float v[SMF_ROWS / 16];
#pragma unroll
    for (int i = 0; i < SMF_ROWS / 16; i++) {
        const int r = r0 + g + 16 * i;
        v[i] = r < numOut ? x[(size_t)r * numCases + col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SMF_ROWS / 16; i++) {
        const int r = r0 + g + 16 * i;
        if (r < numOut) {
            const float pv = expf(v[i] - M) * rS;
            p[(size_t)r * numCases + col] = pv;
            if (grad) grad[(size_t)r * numCases + col] = coeff * ((label == r ? 1.f : 0.f) - pv);
        }
    }
}
// The same stage 2 with 16-byte accesses: a block of NW waves owns 32 columns (8 lanes x 4) x ROWS rows (256 rows and 8 waves
// as launched: measured 8.6 us against 9.4 us for 128 x 8, 9.3 for 128 x 4, 10.8 for 256 x 4, 11.0 for 512 x 8), so that every wave
// instruction touches whole 128-byte lines of x / p / grad (the scalar form above moves 64 bytes per row and lane group).
// Merge: lane group g (of 32) merges chunks g, g+32, ... of its 4 columns, the 32 groups are combined through LDS by 32
// threads (one column each, fixed order) and M, S are broadcast through LDS.
#define SMF4_ROWS 128
// COST: the logreg cost of the same labels in the same launch (kLogregCost layer_kernels.cu:59-92).  The largest probability
// of a column is expf(0)/S = 1/S exactly as the element that holds the maximum computes it, so every thread can count
// how many of its entries equal it; the counts meet in per-column int atomics (exact in any order) and the thread that
// holds the label's row leaves log p[label] and whether it is a maximum.  k_logreg_finish completes the cost.
struct LogregFuse {
    float* labelLogProbs;  // (numCases)
    float* correctProbs;   // (numCases)
    double* costOut;       // 2 accumulators
    int* tieCnt;           // (numCases) zero between calls
};
// Merge (round 2, second form): the column maximum first -- fmax only: three xor-shuffles over the 8 lane groups of a wave,
// the 4 waves through LDS -- then S = sum_c s_c * exp(m_c - M) with ONE exp per partial and plain adds in a fixed tree.
// (The first form merged (m, s) pairs at every level: an exp per pair per level, and a serial 32-term tail on 32 threads
// while the other 224 waited: 80 exp-equivalents on the critical wave against 40 here.)  All chunk partials of a thread are
// requested in one batch next to its x values (<= SMF4_U batches of 32 chunks stay in registers: K <= 12288 rows; beyond
// that the partials are read twice).  p = exp(x - M) * (1/S): the column's largest entry is exactly 1/S.  The exponentials of this
// kernel are the hardware's (__expf, as the reference's NVMatrix Exp operator, nvmatrix_operators.cuh:37): 16 + 12 of them per
// thread were a third of its instructions (8.7 -> 7.6 us); exp(0) is still exactly 1, so the tie count of the cost is untouched.
#define SMF4_RES 192  // chunks whose partials stay in registers
template <bool COST, int ROWS, int NW>
__global__ __launch_bounds__(NW * 64) void k_softmax_final4(const float* __restrict__ x, float* __restrict__ p, const float* __restrict__ part,
                                                        int numOut, int numCases, int chunks, const float* __restrict__ labels,
                                                        float* __restrict__ grad, float coeff, LogregFuse lf) {
    constexpr int G = 8 * NW, SMF4_U = SMF4_RES / G;  // lane groups of 8 x 4 columns; chunk batches per group
    __shared__ float shM[NW][32], shS[NW][32];
    __shared__ int shc[NW][32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cl = lane & 7, g = w * 8 + (lane >> 3), col = blockIdx.x * 32 + 4 * cl;
    const bool live = col < numCases;  // numCases % 4 == 0
    // this thread's x values first: their latency hides behind the merge of the partials
    const int r0 = blockIdx.y * ROWS;
    f32x4 v[ROWS / G];
#pragma unroll
    for (int i = 0; i < ROWS / G; i++) {
        const int r = r0 + g + G * i;
        v[i] = (live && r < numOut) ? *reinterpret_cast<const f32x4*>(x + (size_t)r * numCases + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // chunk partials of 4 consecutive columns = 8 consecutive floats (max, sum interleaved)
    const f32x4 dead = {-INFINITY, 0.f, -INFINITY, 0.f};
    const bool resident = chunks <= G * SMF4_U;
    f32x4 pa[SMF4_U], pb[SMF4_U];
    f32x4 M4 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int c0 = g; c0 < chunks; c0 += G * SMF4_U) {
#pragma unroll
        for (int u = 0; u < SMF4_U; u++) {
            const int c = c0 + G * u;
            const bool ok = live && c < chunks;
            const float* src = part + ((size_t)c * numCases + col) * 2;
            pa[u] = ok ? *reinterpret_cast<const f32x4*>(src) : dead;
            pb[u] = ok ? *reinterpret_cast<const f32x4*>(src + 4) : dead;
        }
#pragma unroll
        for (int u = 0; u < SMF4_U; u++) {
            M4[0] = fmaxf(M4[0], pa[u][0]);
            M4[1] = fmaxf(M4[1], pa[u][2]);
            M4[2] = fmaxf(M4[2], pb[u][0]);
            M4[3] = fmaxf(M4[3], pb[u][2]);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        M4[t] = fmaxf(M4[t], __shfl_xor(M4[t], 8));
        M4[t] = fmaxf(M4[t], __shfl_xor(M4[t], 16));
        M4[t] = fmaxf(M4[t], __shfl_xor(M4[t], 32));
    }
    if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 4; t++) shM[w][4 * cl + t] = M4[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++) {
        M4[t] = shM[0][4 * cl + t];
#pragma unroll
        for (int k = 1; k < NW; k++) M4[t] = fmaxf(M4[t], shM[k][4 * cl + t]);
    }
    f32x4 S4 = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = g; c0 < chunks; c0 += G * SMF4_U) {
        if (!resident) {
#pragma unroll
            for (int u = 0; u < SMF4_U; u++) {
                const int c = c0 + G * u;
                const bool ok = live && c < chunks;
                const float* src = part + ((size_t)c * numCases + col) * 2;
                pa[u] = ok ? *reinterpret_cast<const f32x4*>(src) : dead;
                pb[u] = ok ? *reinterpret_cast<const f32x4*>(src + 4) : dead;
            }
        }
#pragma unroll
        for (int u = 0; u < SMF4_U; u++) {  // (m = -inf: an absent chunk, s = 0)
            S4[0] += pa[u][0] == -INFINITY ? 0.f : pa[u][1] * __expf(pa[u][0] - M4[0]);
            S4[1] += pa[u][2] == -INFINITY ? 0.f : pa[u][3] * __expf(pa[u][2] - M4[1]);
            S4[2] += pb[u][0] == -INFINITY ? 0.f : pb[u][1] * __expf(pb[u][0] - M4[2]);
            S4[3] += pb[u][2] == -INFINITY ? 0.f : pb[u][3] * __expf(pb[u][2] - M4[3]);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        S4[t] += __shfl_xor(S4[t], 8);
        S4[t] += __shfl_xor(S4[t], 16);
        S4[t] += __shfl_xor(S4[t], 32);
    }
    if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 4; t++) shS[w][4 * cl + t] = S4[t];
    }
    __syncthreads();
    if (!COST && !live) return;
    f32x4 rS;
    int lab[4], ties[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        S4[t] = (shS[0][4 * cl + t] + shS[1][4 * cl + t]) + (shS[2][4 * cl + t] + shS[3][4 * cl + t]);
        if (NW == 8) S4[t] += (shS[4][4 * cl + t] + shS[5][4 * cl + t]) + (shS[6][4 * cl + t] + shS[7][4 * cl + t]);
        rS[t] = 1.0f / S4[t];
        lab[t] = (labels && live) ? (int)labels[col + t] : -1;
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < ROWS / G; i++) {
            const int r = r0 + g + G * i;
            if (r < numOut) {
                f32x4 pv, gv;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    pv[t] = __expf(v[i][t] - M4[t]) * rS[t];
                    gv[t] = coeff * ((lab[t] == r ? 1.f : 0.f) - pv[t]);
                    if (COST) {
                        ties[t] += pv[t] == rS[t] ? 1 : 0;
                        if (lab[t] == r) {
                            lf.labelLogProbs[col + t] = logf(pv[t]);
                            lf.correctProbs[col + t] = pv[t] == rS[t] ? 1.f : 0.f;
                        }
                    }
                }
                *reinterpret_cast<f32x4*>(p + (size_t)r * numCases + col) = pv;
                if (grad) *reinterpret_cast<f32x4*>(grad + (size_t)r * numCases + col) = gv;
            }
        }
    }
    if (!COST) return;
    // ---- logreg cost: entries equal to the column's largest probability, counted per column (exact in any order)
#pragma unroll
    for (int t = 0; t < 4; t++) {
        ties[t] += __shfl_xor(ties[t], 8);
        ties[t] += __shfl_xor(ties[t], 16);
        ties[t] += __shfl_xor(ties[t], 32);
    }
    if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 4; t++) shc[w][4 * cl + t] = ties[t];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) c += shc[k][threadIdx.x];
        const int cc = blockIdx.x * 32 + threadIdx.x;
        if (c != 0 && cc < numCases) atomicAdd(&lf.tieCnt[cc], c);
    }
}
// ... and one small block once the grid has drained: correctProbs from the hit flags and tie counts, the two cost scalars
// summed in a fixed order, the counters cleared for the next call.  (A last-block-finishes form inside the launch above
// needs a release fence per block, which on this part writes the XCD's L2 back: measured 12 -> 41 us for the launch.)
__global__ __launch_bounds__(256) void k_logreg_finish(LogregFuse lf, int numCases) {
    __shared__ double sh0[256], sh1[256];
    double a0 = 0, a1 = 0;
    for (int c = threadIdx.x; c < numCases; c += 256) {
        const float lp = lf.labelLogProbs[c], hit = lf.correctProbs[c];
        const int cnt = lf.tieCnt[c];
        const float cp = hit != 0.f ? 1.0f / (float)cnt : 0.f;
        lf.correctProbs[c] = cp;
        lf.tieCnt[c] = 0;
        a0 += (double)lp;
        a1 += 1.0 - (double)cp;
    }
    sh0[threadIdx.x] = a0;
    sh1[threadIdx.x] = a1;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {  // fixed tree
        if ((int)threadIdx.x < w) {
            sh0[threadIdx.x] += sh0[threadIdx.x + w];
            sh1[threadIdx.x] += sh1[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&lf.costOut[0], -sh0[0]);
        atomicAdd(&lf.costOut[1], sh1[0]);
    }
}
static bool softmax_final4_usable(const float* x, const float* probs, const float* part, const float* grad, int numOut, int numCases) {
    return numCases % 4 == 0 && al16(x) && al16(probs) && al16(part) && (!grad || al16(grad)) && divup(numOut, 128) <= 65535;
}
static int softmax_final_launch(const float* x, float* probs, const float* part, int numOut, int numCases, int chunks, const float* labels,
                                float* grad, float coeff, hipStream_t s, const LogregFuse* lf = nullptr) {
    static const int nwOpt = getenv("MCN_SOFTMAX_WAVES") ? atoi(getenv("MCN_SOFTMAX_WAVES")) : 8;
    static const int rowsOpt = getenv("MCN_SOFTMAX_ROWS") ? atoi(getenv("MCN_SOFTMAX_ROWS")) : 256;
    const int rows = rowsOpt == 512 ? 512 : rowsOpt == 128 ? 128 : 256, nw = nwOpt == 4 ? 4 : 8;
    const dim3 grid4(divup(numCases, 32), divup(numOut, rows));
#define SMF4_LAUNCH(COST, ROWS, NW, LF) \
    MCN_LAUNCH((k_softmax_final4<COST, ROWS, NW>), grid4, dim3(NW * 64), 0, s, x, probs, part, numOut, numCases, chunks, labels, grad, coeff, LF)
#define SMF4_PICK(COST, LF)                                     \
    if (rows == 512) SMF4_LAUNCH(COST, 512, 8, LF);             \
    else if (rows == 256 && nw == 8) SMF4_LAUNCH(COST, 256, 8, LF); \
    else if (rows == 256) SMF4_LAUNCH(COST, 256, 4, LF);        \
    else if (nw == 8) SMF4_LAUNCH(COST, 128, 8, LF);            \
    else SMF4_LAUNCH(COST, 128, 4, LF)
    if (lf) {  // (the caller has checked softmax_final4_usable)
        SMF4_PICK(true, *lf);
    } else if (softmax_final4_usable(x, probs, part, grad, numOut, numCases)) {
        SMF4_PICK(false, LogregFuse{});
    }
#undef SMF4_PICK
#undef SMF4_LAUNCH
    else
        MCN_LAUNCH(k_softmax_final, dim3(divup(numCases, SMF_COLS), divup(numOut, SMF_ROWS)), dim3(256), 0, s, x, probs, part, numOut,
                           numCases, chunks, labels, grad, coeff);
    return MCN_LAUNCH_CHECK("softmax");
}
MCN_API size_t mcn_softmax_workspace_bytes(int numOut, int numCases) {
    return (size_t)divup(numOut, SM_ROWS) * numCases * 2 * sizeof(float);
}
static int softmax_impl(const float* x, float* probs, const float* labels, float* grad, float coeff, int numOut, int numCases, void* workspace,
                        size_t workspaceBytes, void* stream) {
    MCN_REQUIRE(x && probs && numOut > 0 && numCases > 0, "softmax: bad argument");
    const int chunks = divup(numOut, SM_ROWS);
    const size_t need = mcn_softmax_workspace_bytes(numOut, numCases);
    if (!workspace || workspaceBytes < need) {
        mcn_set_error("softmax: workspace %zu < %zu bytes", workspaceBytes, need);
        return MCN_EWORKSPACE;
    }
    MCN_REQUIRE(chunks <= 65535, "softmax: too many outputs");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(divup(numCases, 64), chunks);
    MCN_LAUNCH(k_softmax_partial, grid, dim3(256), 0, s, x, (float*)workspace, numOut, numCases, chunks);
    return softmax_final_launch(x, probs, (const float*)workspace, numOut, numCases, chunks, labels, grad, coeff, s);
}
// stage 2 alone, from partials the classifier layer's forward product left (mcn_gemm_bias_neuron_softmax_part)
MCN_API int mcn_softmax_from_part(const float* x, float* probs, const float* labels, float* gradTarget, float coeff, int numOut,
                                  int numCases, const void* part, size_t partBytes, void* stream) {
    MCN_REQUIRE(x && probs && part && numOut > 0 && numCases > 0, "softmaxFromPart: bad argument");
    MCN_REQUIRE((labels != nullptr) == (gradTarget != nullptr), "softmaxFromPart: labels and gradTarget go together");
    const int chunks = divup(numOut, SM_ROWS);
    if (partBytes < mcn_softmax_workspace_bytes(numOut, numCases)) {
        mcn_set_error("softmaxFromPart: partial buffer %zu < %zu bytes", partBytes, mcn_softmax_workspace_bytes(numOut, numCases));
        return MCN_EWORKSPACE;
    }
    return softmax_final_launch(x, probs, (const float*)part, numOut, numCases, chunks, labels, gradTarget, coeff, (hipStream_t)stream);
}
// softmax (+ optional logreg gradient) + logreg cost in the stage-2 launch.  `part` != NULL: stage 1 is already there
// (mcn_gemm_bias_neuron_softmax_part), else it runs first into `workspace`.  `state`: mcn_softmax_logreg_cost_state_bytes
// bytes of device memory, zero before the first call and owned by these calls (left zero by each).
MCN_API size_t mcn_softmax_logreg_cost_state_bytes(int numCases) { return ((size_t)numCases + 4) * sizeof(int); }
// the second, one-block launch of mcn_softmax_logreg_cost(finish = 0) at a time (and on a stream) of the caller's choosing
MCN_API int mcn_softmax_logreg_cost_finish(float* labelLogProbs, float* correctProbs, double* costOut, void* state, size_t stateBytes,
                                           int numCases, void* stream) {
    MCN_REQUIRE(labelLogProbs && correctProbs && costOut && state && numCases > 0, "softmaxLogregCostFinish: bad argument");
    MCN_REQUIRE(stateBytes >= mcn_softmax_logreg_cost_state_bytes(numCases), "softmaxLogregCostFinish: state buffer too small");
    LogregFuse lf;
    lf.labelLogProbs = labelLogProbs;
    lf.correctProbs = correctProbs;
    lf.costOut = costOut;
    lf.tieCnt = (int*)state;
    MCN_LAUNCH(k_logreg_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, lf, numCases);
    return MCN_LAUNCH_CHECK("logreg finish");
}
MCN_API int mcn_softmax_logreg_cost_supported(int numOut, int numCases) {
    return numOut > 0 && numCases > 0 && numCases % 4 == 0 && divup(numOut, 128) <= 65535 && divup(numOut, SM_ROWS) <= 65535;
}
MCN_API int mcn_softmax_logreg_cost(const float* x, float* probs, const float* labels, float* gradTarget, float coeff, int numOut,
                                    int numCases, const void* part, size_t partBytes, void* workspace, size_t workspaceBytes,
                                    float* labelLogProbs, float* correctProbs, double* costOut, void* state, size_t stateBytes,
                                    int finish, void* stream) {
    MCN_REQUIRE(x && probs && labels && labelLogProbs && correctProbs && costOut && state, "softmaxLogregCost: null pointer");
    MCN_REQUIRE(mcn_softmax_logreg_cost_supported(numOut, numCases), "softmaxLogregCost: numCases must be a multiple of 4");
    MCN_REQUIRE(stateBytes >= mcn_softmax_logreg_cost_state_bytes(numCases), "softmaxLogregCost: state buffer too small");
    const int chunks = divup(numOut, SM_ROWS);
    const size_t need = mcn_softmax_workspace_bytes(numOut, numCases);
    hipStream_t s = (hipStream_t)stream;
    const void* stage1 = part;
    if (part) {
        MCN_REQUIRE(partBytes >= need, "softmaxLogregCost: partial buffer too small");
    } else {
        if (!workspace || workspaceBytes < need) {
            mcn_set_error("softmaxLogregCost: workspace %zu < %zu bytes", workspaceBytes, need);
            return MCN_EWORKSPACE;
        }
        MCN_LAUNCH(k_softmax_partial, dim3(divup(numCases, 64), chunks), dim3(256), 0, s, x, (float*)workspace, numOut, numCases, chunks);
        stage1 = workspace;
    }
    MCN_REQUIRE(softmax_final4_usable(x, probs, (const float*)stage1, gradTarget, numOut, numCases), "softmaxLogregCost: pointers must be 16-byte aligned");
    LogregFuse lf;
    lf.labelLogProbs = labelLogProbs;
    lf.correctProbs = correctProbs;
    lf.costOut = costOut;
    lf.tieCnt = (int*)state;
    const int rc = softmax_final_launch(x, probs, (const float*)stage1, numOut, numCases, chunks, labels, gradTarget, coeff, s, &lf);
    if (rc != MCN_OK || !finish) return rc;
    MCN_LAUNCH(k_logreg_finish, dim3(1), dim3(256), 0, s, lf, numCases);
    return MCN_LAUNCH_CHECK("logreg finish");
}
MCN_API int mcn_softmax(const float* x, float* probs, int numOut, int numCases, void* workspace, size_t workspaceBytes, void* stream) {
    return softmax_impl(x, probs, nullptr, nullptr, 0.f, numOut, numCases, workspace, workspaceBytes, stream);
}
MCN_API int mcn_softmax_logreg_grad(const float* x, float* probs, const float* labels, float* gradTarget, float coeff, int numOut, int numCases,
                                    void* workspace, size_t workspaceBytes, void* stream) {
    MCN_REQUIRE(labels && gradTarget, "softmaxLogregGrad: null pointer");
    return softmax_impl(x, probs, labels, gradTarget, coeff, numOut, numCases, workspace, workspaceBytes, stream);
}

// ---------------------------------------------------------------- logreg cost
// Stage 1: per (case, rowChunk): max prob and how many entries equal it.  Stage 2 (one block):
// merge chunks, read p[label], write logprob / correctProbs (kLogregCost layer_kernels.cu:59-92)
// and accumulate the two cost scalars in double.
__device__ __forceinline__ void maxcnt_merge(float& m, int& cnt, float pm, int pc) {
    if (pm > m) { m = pm; cnt = pc; } else if (pm == m) cnt += pc;
}
__global__ __launch_bounds__(256) void k_logreg_partial(const float* __restrict__ probs, float* __restrict__ part, int numOut, int numCases) {
    __shared__ float shm[4][64];
    __shared__ int shc[4][64];
    const int l = threadIdx.x & 63, col = blockIdx.x * 64 + l, rl = threadIdx.x >> 6, chunk = blockIdx.y;
    const int r0 = chunk * SM_ROWS, r1 = min(numOut, r0 + SM_ROWS);
    float v[SM_ROWS / 4];
#pragma unroll
    for (int i = 0; i < SM_ROWS / 4; i++) {
        const int r = r0 + rl + 4 * i;
        v[i] = (col < numCases && r < r1) ? probs[(size_t)r * numCases + col] : -INFINITY;
    }
    float m = -INFINITY;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < SM_ROWS / 4; i++)
        if (v[i] != -INFINITY) maxcnt_merge(m, cnt, v[i], 1);
    shm[rl][l] = m;
    shc[rl][l] = cnt;
    __syncthreads();
    if (rl == 0 && col < numCases) {
        for (int k = 1; k < 4; k++) maxcnt_merge(m, cnt, shm[k][l], shc[k][l]);
        part[((size_t)chunk * numCases + col) * 2] = m;
        part[((size_t)chunk * numCases + col) * 2 + 1] = (float)cnt;
    }
}
// 16 columns x 16 merge lanes per block (see k_softmax_final); each block adds its columns' cost into the two
// double accumulators with atomics (at most numCases/16 adds per call; order affects the 16th digit only).
__global__ __launch_bounds__(256) void k_logreg_final(const float* __restrict__ labels, const float* __restrict__ probs,
                                                      const float* __restrict__ part, float* __restrict__ labelLogProbs,
                                                      float* __restrict__ correctProbs, int numOut, int numCases, int chunks,
                                                      double* costOut) {
    __shared__ float shm[16][SMF_COLS];
    __shared__ int shc[16][SMF_COLS];
    __shared__ double sh0[SMF_COLS], sh1[SMF_COLS];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4, col = blockIdx.x * SMF_COLS + cl;
    const bool live = col < numCases;
    const f32x2* pp = reinterpret_cast<const f32x2*>(part);
    float m = -INFINITY;
    int cnt = 0;
    for (int c0 = g; c0 < chunks; c0 += 16 * SMF_MAXB) {
        f32x2 v[SMF_MAXB];
#pragma unroll
        for (int u = 0; u < SMF_MAXB; u++) {
            const int c = c0 + 16 * u;
            v[u] = (live && c < chunks) ? pp[(size_t)c * numCases + col] : (f32x2){-INFINITY, 0.f};
        }
#pragma unroll
        for (int u = 0; u < SMF_MAXB; u++)
            if (v[u][1] != 0.f) maxcnt_merge(m, cnt, v[u][0], (int)v[u][1]);
    }
    shm[g][cl] = m;
    shc[g][cl] = cnt;
    __syncthreads();
    if (g == 0) {
        double a0 = 0, a1 = 0;
        if (live) {
            for (int k = 1; k < 16; k++)
                if (shc[k][cl] != 0) maxcnt_merge(m, cnt, shm[k][cl], shc[k][cl]);
            const int label = (int)labels[col];
            const float labelp = probs[(size_t)label * numCases + col];
            const float lp = logf(labelp);
            const float cp = labelp != m ? 0.f : 1.0f / (float)cnt;
            labelLogProbs[col] = lp;
            correctProbs[col] = cp;
            a0 = (double)lp;
            a1 = 1.0 - (double)cp;
        }
        sh0[cl] = a0;
        sh1[cl] = a1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && costOut) {
        double t0 = 0, t1 = 0;
        for (int k = 0; k < SMF_COLS; k++) { t0 += sh0[k]; t1 += sh1[k]; }
        atomicAdd(&costOut[0], -t0);
        atomicAdd(&costOut[1], t1);
    }
}
MCN_API size_t mcn_logreg_cost_workspace_bytes(int numOut, int numCases) { return mcn_softmax_workspace_bytes(numOut, numCases); }
MCN_API int mcn_logreg_cost(const float* labels, const float* probs, float* labelLogProbs, float* correctProbs, int numOut,
                            int numCases, double* costOut, void* workspace, size_t workspaceBytes, void* stream) {
    MCN_REQUIRE(labels && probs && labelLogProbs && correctProbs && numOut > 0 && numCases > 0, "logregCost: bad argument");
    const int chunks = divup(numOut, SM_ROWS);
    const size_t need = mcn_logreg_cost_workspace_bytes(numOut, numCases);
    if (!workspace || workspaceBytes < need) {
        mcn_set_error("logregCost: workspace %zu < %zu bytes", workspaceBytes, need);
        return MCN_EWORKSPACE;
    }
    MCN_REQUIRE(chunks <= 65535, "logregCost: too many outputs");
    hipStream_t s = (hipStream_t)stream;
    MCN_LAUNCH(k_logreg_partial, dim3(divup(numCases, 64), chunks), dim3(256), 0, s, probs, (float*)workspace, numOut, numCases);
    MCN_LAUNCH(k_logreg_final, dim3(divup(numCases, SMF_COLS)), dim3(256), 0, s, labels, probs, (const float*)workspace,
                       labelLogProbs, correctProbs, numOut, numCases, chunks, costOut);
    return MCN_LAUNCH_CHECK("logregCost");
}

// dx[k,n] (+)= coeff*((label_n == k) - p[k,n])        (kLogregSoftmaxGrad layer_kernels.cu:154-170)
template <int MODE>  // 0: logreg-softmax grad, 1: logreg grad coeff*(label==k)/p
__global__ __launch_bounds__(256) void k_logreg_grads(const float* __restrict__ labels, const float* __restrict__ probs,
                                                      float* __restrict__ target, int numOut, int numCases, int add, float coeff) {
    const size_t n = (size_t)numOut * numCases;
    GRID_STRIDE(i, n) {
        const int k = (int)(i / numCases), c = (int)(i % numCases);
        const int label = (int)labels[c];
        const float v = MODE == 0 ? coeff * ((label == k ? 1.f : 0.f) - probs[i]) : coeff * (label == k ? 1.f : 0.f) / probs[i];
        target[i] = add ? target[i] + v : v;
    }
}
MCN_API int mcn_logreg_softmax_grad(const float* labels, const float* probs, float* target, int numOut, int numCases, int add,
                                    float coeff, void* stream) {
    MCN_REQUIRE(labels && probs && target && numOut > 0 && numCases > 0, "logregSoftmaxGrad: bad argument");
    MCN_LAUNCH((k_logreg_grads<0>), dim3(ew_blocks((size_t)numOut * numCases)), dim3(256), 0, (hipStream_t)stream, labels, probs,
                       target, numOut, numCases, add, coeff);
    return MCN_LAUNCH_CHECK("logregSoftmaxGrad");
}
MCN_API int mcn_logreg_grad(const float* labels, const float* probs, float* target, int numOut, int numCases, int add,
                            float coeff, void* stream) {
    MCN_REQUIRE(labels && probs && target && numOut > 0 && numCases > 0, "logregGrad: bad argument");
    MCN_LAUNCH((k_logreg_grads<1>), dim3(ew_blocks((size_t)numOut * numCases)), dim3(256), 0, (hipStream_t)stream, labels, probs,
                       target, numOut, numCases, add, coeff);
    return MCN_LAUNCH_CHECK("logregGrad");
}
// kSoftmaxGrad (layer_kernels.cu:126-145): O(K) per element, as in the reference (not on the configs' path)
__global__ __launch_bounds__(256) void k_softmax_grad(const float* __restrict__ acts, const float* __restrict__ g, float* __restrict__ target,
                                                      int numOut, int numCases, int add) {
    const size_t n = (size_t)numOut * numCases;
    GRID_STRIDE(i, n) {
        const int k = (int)(i / numCases), c = (int)(i % numCases);
        float v = 0.f;
        for (int j = 0; j < numOut; j++) v += g[(size_t)j * numCases + c] * ((j == k ? 1.f : 0.f) - acts[(size_t)j * numCases + c]);
        v *= acts[i];
        target[i] = add ? target[i] + v : v;
    }
}
MCN_API int mcn_softmax_grad(const float* acts, const float* actsGrad, float* target, int numOut, int numCases, int add, void* stream) {
    MCN_REQUIRE(acts && actsGrad && target && numOut > 0 && numCases > 0, "softmaxGrad: bad argument");
    MCN_LAUNCH(k_softmax_grad, dim3(ew_blocks((size_t)numOut * numCases)), dim3(256), 0, (hipStream_t)stream, acts, actsGrad,
                       target, numOut, numCases, add);
    return MCN_LAUNCH_CHECK("softmaxGrad");
}

// ---------------------------------------------------------------- L2-SN (Joint identification-verification)
// One workgroup.  Phase 1: thread j < N/2 computes label s_j and d_j = sqrt(sum_k (g-p)^2) + 1e-8 with the
// squares rounded separately and added in row order, exactly the order of the reference's
// SquaredDiff + column-sum (layer_kernels.cu:554-561, kDumbAggCols) and of the CPU oracle, so the
// data-dependent margin and every 1[d<m] decision agree bit for bit.  Phase 2: getOptimalThr
// (:482-498) as an O(h) scan per candidate, first strict maximum wins.  Phase 3: cost or gradient.
#define L2SN_MAX_PAIRS 2048
#define L2SN_TILE 4096       // floats of squared differences staged per pass: at least this many ...
#define L2SN_TILE_MAX 12288  // ... and up to this many when the 64 KB of LDS allow (DeepID2: all 160 rows in one pass)
// Pair labels and distances.  The squares are computed by all threads into an LDS tile (independent,
// coalesced loads); thread j then adds its column of the tile in row order -- the summation order of
// the reference's SquaredDiff + column-sum and of the CPU oracle.
// (j0, hw): the window of pairs this block computes -- dists / labs / sq are indexed inside the window; (0, numCases / 2) = all of them
__device__ void l2sn_stats(const float* idens, const float* data, int numD, int numCases, float* dists, float* labs, float* sq, int tile,
                           int j0, int hw) {
    const int hAll = numCases / 2;
    const int h = hw;
    idens += j0;
    data += j0;
    const int TR = max(1, tile / h);
    for (int j = threadIdx.x; j < h; j += blockDim.x) {
        labs[j] = idens[j] == idens[j + hAll] ? 1.f : -1.f;
        dists[j] = 0.f;
    }
    // 16-byte loads need h % 4 == 0 (the probe half then starts on a 16-byte boundary too)
    const bool vec = (h % 4 == 0) && (hAll % 4 == 0) && (((uintptr_t)data) % 16 == 0) && (numCases % 4 == 0);
    const int hq = h / 4;                       // quads per half row
    const int rq = threadIdx.x / max(hq, 1), jq = threadIdx.x % max(hq, 1);  // vector path: (row lane, quad)
    const int RV = max(1, (int)blockDim.x / max(hq, 1));                    // rows per pass
    for (int k0 = 0; k0 < numD; k0 += TR) {
        const int rows = min(TR, numD - k0);
        __syncthreads();
        if (vec && hq <= (int)blockDim.x) {
            // batches of 8 independent row loads per thread
            for (int kb = rq; kb < rows && rq < RV; kb += 8 * RV) {
                f32x4 gv[8], qv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int kk = kb + u * RV;
                    if (kk < rows) {
                        const float* g = data + (size_t)(k0 + kk) * numCases + 4 * jq;
                        gv[u] = *reinterpret_cast<const f32x4*>(g);
                        qv[u] = *reinterpret_cast<const f32x4*>(g + hAll);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int kk = kb + u * RV;
                    if (kk < rows) {
                        f32x4 d = gv[u] - qv[u];
#pragma unroll
                        for (int t = 0; t < 4; t++) d[t] = __fmul_rn(d[t], d[t]);
                        *reinterpret_cast<f32x4*>(sq + kk * h + 4 * jq) = d;
                    }
                }
            }
        } else {
            for (int e = threadIdx.x; e < rows * h; e += blockDim.x) {
                const int kk = e / h, j = e % h;
                const float* g = data + (size_t)(k0 + kk) * numCases + j;
                const float d = g[0] - g[hAll];
                sq[e] = __fmul_rn(d, d);
            }
        }
        __syncthreads();
        for (int jj = threadIdx.x; jj < h; jj += blockDim.x) {
            float s = dists[jj];
            int kk = 0;
            for (; kk + 8 <= rows; kk += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = sq[(kk + u) * h + jj];
#pragma unroll
                for (int u = 0; u < 8; u++) s = __fadd_rn(s, t[u]);
            }
            for (; kk < rows; kk++) s = __fadd_rn(s, sq[kk * h + jj]);
            dists[jj] = s;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < h; j += blockDim.x) dists[j] = __fadd_rn(sqrtf(dists[j]), 0.00000001f);
    __syncthreads();
}
// returns (m, idx) in sh_m[0..1]; candidates/labels of length cnt
__device__ void l2sn_threshold(const float* cd, const float* cl, int cnt, int* rates, float* sh_m) {
    // four lanes share a candidate and count a quarter of the pairs each (integer counts: any order is exact)
    const int part = threadIdx.x & 3;
    for (int i0 = 0; i0 < cnt; i0 += blockDim.x >> 2) {
        const int i = i0 + (threadIdx.x >> 2);
        const float curr = cd[min(i, cnt - 1)];
        int rate = 0;
#pragma unroll 8
        for (int j = part; j < cnt; j += 4) rate += ((cd[j] < curr && cl[j] == 1.f) || (cd[j] > curr && cl[j] == -1.f)) ? 1 : 0;
        rate += __shfl_xor(rate, 1, 64);
        rate += __shfl_xor(rate, 2, 64);
        if (part == 0 && i < cnt) rates[i] = rate;
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // first strict maximum: largest rate, lowest index among equals
        int best = 0, idx = 0;
        for (int i = threadIdx.x; i < cnt; i += 64)
            if (rates[i] > best) { best = rates[i]; idx = i; }
        if (best == 0) idx = 0;
        for (int o = 32; o > 0; o >>= 1) {
            const int ob = __shfl_down(best, o, 64), oi = __shfl_down(idx, o, 64);
            if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
        }
        if (threadIdx.x == 0) {
            sh_m[0] = cd[idx];
            sh_m[1] = (float)idx;
        }
    }
    __syncthreads();
}
// per-pair cost and its sum (one workgroup of 256 threads; layer_kernels.cu:563-604)
__device__ void l2sn_cost_tail(const float* dists, const float* labs, int h, float m, float* out, double* costOut, double* shsum) {
    double a = 0;
    for (int j = threadIdx.x; j < h; j += blockDim.x) {
        const float posMask = 0.5f * labs[j] + 0.5f, negMask1 = -0.5f * labs[j] + 0.5f;
        const float negMask2 = dists[j] < m ? 1.f : 0.f;
        const float posLoss = __fmul_rn(__fmul_rn(__fmul_rn(dists[j], dists[j]), 0.5f), posMask);
        const float t = __fadd_rn(__fmul_rn(-1.0f, dists[j]), m);
        const float negLoss = __fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(t, t), 0.5f), negMask1), negMask2);
        const float o = __fadd_rn(posLoss, negLoss);
        out[j] = o;
        a += (double)o;
    }
    shsum[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) shsum[threadIdx.x] += shsum[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && costOut) *costOut += shsum[0];
}
// MODE 0 cost (one block), 1 grad (block b owns rows [b*rowsPerBlock, ...): every block recomputes the
// pair statistics and the margin -- 80 KB of L2 reads -- instead of a second launch), 2 stats only.
template <int MODE>
__global__ __launch_bounds__(256) void k_l2sn(const float* __restrict__ idens, const float* __restrict__ data, float* __restrict__ out,
                                              int numD, int numCases, float m, int add, float coeff, float* info, float* pairStats,
                                              double* costOut, const float* extD, const float* extL, int extCount, int rowsPerBlock, int tile) {
    extern __shared__ float shf[];
    const int h = numCases / 2;
    const int cnt = extD ? extCount : h;
    float* dists = shf;              // [h]
    float* labs = shf + h;           // [h]
    float* cd = labs + h;            // [cnt] candidates when external
    float* cl = cd + (extD ? cnt : 0);
    int* rates = (int*)(cl + (extD ? cnt : 0));  // [cnt]
    float* sq = (float*)(rates + cnt);           // [L2SN_TILE]
    __shared__ float sh_m[2];
    __shared__ double shsum[256];
    l2sn_stats(idens, data, numD, numCases, dists, labs, sq, tile, 0, numCases / 2);
    if (extD) {
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) { cd[i] = extD[i]; cl[i] = extL[i]; }
    }
    __syncthreads();
    if (pairStats && blockIdx.x == 0) {
        for (int j = threadIdx.x; j < h; j += blockDim.x) { pairStats[j] = dists[j]; pairStats[h + j] = labs[j]; }
    }
    if (MODE == 2) return;
    float idx = -1.f;
    if (m < 0) {
        l2sn_threshold(extD ? cd : dists, extD ? cl : labs, cnt, rates, sh_m);
        m = sh_m[0];
        idx = sh_m[1];
    }
    if (info && threadIdx.x == 0 && blockIdx.x == 0) { info[0] = m; info[1] = idx; info[2] = 0.f; info[3] = 0.f; }
    if (MODE == 0) {
        l2sn_cost_tail(dists, labs, h, m, out, costOut, shsum);
    } else {
        const int k0 = blockIdx.x * rowsPerBlock, k1 = min(numD, k0 + rowsPerBlock);
        const int rowStep = max(1, (int)blockDim.x / h);
        const int j0 = threadIdx.x % h, r0 = threadIdx.x / h;  // one division per thread, none per element
        for (int k = k0 + r0; k < k1 && r0 < rowStep; k += rowStep)
        for (int j = j0; j < h; j += blockDim.x) {
            const float posMask = 0.5f * labs[j] + 0.5f, negMask1 = -0.5f * labs[j] + 0.5f;
            const float negMask2 = dists[j] < m ? 1.f : 0.f;
            const float r = __fadd_rn(__fmul_rn(1.0f / dists[j], m), -1.0f);
            const float s1 = data[(size_t)k * numCases + j], s2 = data[(size_t)k * numCases + j + h];
            const float diff1 = s1 - s2, diff2 = s2 - s1;
            const float gx = __fadd_rn(__fmul_rn(diff2, posMask), __fmul_rn(__fmul_rn(__fmul_rn(diff1, r), negMask1), negMask2));
            const float gy = __fadd_rn(__fmul_rn(diff1, posMask), __fmul_rn(__fmul_rn(__fmul_rn(diff2, r), negMask1), negMask2));
            float* tg = out + (size_t)k * numCases + j;
            float* tp = tg + h;
            *tg = add ? __fadd_rn(*tg, __fmul_rn(gx, coeff)) : __fmul_rn(gx, coeff);
            *tp = add ? __fadd_rn(*tp, __fmul_rn(gy, coeff)) : __fmul_rn(gy, coeff);
        }
    }
}
static int l2sn_launch(int mode, const float* idens, const float* data, float* out, int numD, int numCases, float m, int add,
                       float coeff, float* info, float* pairStats, double* costOut, const float* extD, const float* extL,
                       int extCount, void* stream) {
    MCN_REQUIRE(idens && data && (out || mode == 2) && numD > 0 && numCases > 0, "l2sn: bad argument");
    MCN_REQUIRE(numCases % 2 == 0, "l2sn: numCases must be even");  // layer_kernels.cu:511
    MCN_REQUIRE((extD == nullptr) == (extL == nullptr), "l2sn: extDists and extLabels go together");
    const int h = numCases / 2, cnt = extD ? extCount : h;
    MCN_REQUIRE(h <= L2SN_MAX_PAIRS && cnt <= L2SN_MAX_PAIRS && cnt > 0, "l2sn: too many pairs (%d, limit %d)", cnt, L2SN_MAX_PAIRS);
    const size_t fixedFloats = (size_t)2 * h + (extD ? 2 * cnt : 0) + cnt;
    const int tile = (int)std::min<size_t>(L2SN_TILE_MAX, std::max<size_t>(L2SN_TILE, (64 * 1024 - 2304) / 4 - fixedFloats));
    const size_t lds = (fixedFloats + tile) * 4;
    hipStream_t s = (hipStream_t)stream;
    const int rowsPerBlock = std::max(1, 512 / h);  // two gradient elements per thread
    const int gradBlocks = divup(numD, rowsPerBlock);
#define L2SN_ARGS idens, data, out, numD, numCases, m, add, coeff, info, pairStats, costOut, extD, extL, extCount, rowsPerBlock, tile
    if (mode == 0) MCN_LAUNCH((k_l2sn<0>), dim3(1), dim3(256), lds, s, L2SN_ARGS);
    else if (mode == 1) MCN_LAUNCH((k_l2sn<1>), dim3(gradBlocks), dim3(256), lds, s, L2SN_ARGS);
    else MCN_LAUNCH((k_l2sn<2>), dim3(1), dim3(256), lds, s, L2SN_ARGS);
#undef L2SN_ARGS
    return MCN_LAUNCH_CHECK("l2sn");
}
MCN_API int mcn_l2sn_cost(const float* idens, const float* data, float* output, int numD, int numCases, float m, float* info,
                          float* pairStats, double* costOut, const float* extDists, const float* extLabels, int extCount, void* stream) {
    return l2sn_launch(0, idens, data, output, numD, numCases, m, 0, 0.f, info, pairStats, costOut, extDists, extLabels, extCount, stream);
}
MCN_API int mcn_l2sn_grad(const float* idens, const float* data, float* dataTarget, int numD, int numCases, float m, int add,
                          float coeff, const float* extDists, const float* extLabels, int extCount, void* stream) {
    return l2sn_launch(1, idens, data, dataTarget, numD, numCases, m, add, coeff, nullptr, nullptr, nullptr, extDists, extLabels, extCount, stream);
}
// Gradient from pair statistics and a margin that an earlier l2sn cost launch left in device memory: purely
// elementwise (one gallery/probe element pair per thread), same arithmetic as k_l2sn<1>.
// one (gallery, probe) element pair: d cost / d gallery element and d cost / d probe element, before the coefficient
__device__ __forceinline__ void l2sn_grad_pair(float s1, float s2, float dist, float lab, float m, float& gx, float& gy) {
    const float posMask = 0.5f * lab + 0.5f, negMask1 = -0.5f * lab + 0.5f;
    const float negMask2 = dist < m ? 1.f : 0.f;
    const float r = __fadd_rn(__fmul_rn(1.0f / dist, m), -1.0f);
    const float diff1 = s1 - s2, diff2 = s2 - s1;
    gx = __fadd_rn(__fmul_rn(diff2, posMask), __fmul_rn(__fmul_rn(__fmul_rn(diff1, r), negMask1), negMask2));
    gy = __fadd_rn(__fmul_rn(diff1, posMask), __fmul_rn(__fmul_rn(__fmul_rn(diff2, r), negMask1), negMask2));
}
__global__ __launch_bounds__(256) void k_l2sn_grad_stats(const float* __restrict__ data, float* __restrict__ out, int numD, int numCases,
                                                          const float* __restrict__ stats, const float* __restrict__ info, int add,
                                                          float coeff) {
    const int h = numCases / 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int k = e / h, j = e - k * h;
    if (k >= numD) return;
    float gx, gy;
    l2sn_grad_pair(data[(size_t)k * numCases + j], data[(size_t)k * numCases + j + h], stats[j], stats[h + j], info[0], gx, gy);
    float* tg = out + (size_t)k * numCases + j;
    float* tp = tg + h;
    *tg = add ? __fadd_rn(*tg, __fmul_rn(gx, coeff)) : __fmul_rn(gx, coeff);
    *tp = add ? __fadd_rn(*tp, __fmul_rn(gy, coeff)) : __fmul_rn(gy, coeff);
}
MCN_API int mcn_l2sn_grad_from_stats(const float* data, float* dataTarget, int numD, int numCases, const float* pairStats,
                                     const float* info, int add, float coeff, void* stream) {
    MCN_REQUIRE(data && dataTarget && pairStats && info && numD > 0 && numCases > 0, "l2snGradFromStats: bad argument");
    MCN_REQUIRE(numCases % 2 == 0, "l2sn: numCases must be even");
    const size_t elems = (size_t)numD * (numCases / 2);
    MCN_LAUNCH(k_l2sn_grad_stats, dim3((unsigned)divup_sz(elems, 256)), dim3(256), 0, (hipStream_t)stream, data, dataTarget, numD,
                       numCases, pairStats, info, add, coeff);
    return MCN_LAUNCH_CHECK("l2snGradFromStats");
}
// The pair statistics on several workgroups: block b owns the window of L2SN_WIN pairs [b * L2SN_WIN, ...) -- a pair's squares
// are still summed by one thread in row order, so every distance keeps its bits -- and writes its part of pairStats.
#define L2SN_WIN 8
__global__ __launch_bounds__(256) void k_l2sn_stats_win(const float* __restrict__ idens, const float* __restrict__ data,
                                                        float* __restrict__ pairStats, int numD, int numCases, int tile) {
    extern __shared__ float shf[];
    const int h = numCases / 2, j0 = blockIdx.x * L2SN_WIN, hw = min(L2SN_WIN, h - j0);
    float* dists = shf;             // [L2SN_WIN]
    float* labs = shf + L2SN_WIN;   // [L2SN_WIN]
    float* sq = shf + 2 * L2SN_WIN; // [tile]
    l2sn_stats(idens, data, numD, numCases, dists, labs, sq, tile, j0, hw);
    if ((int)threadIdx.x < hw) {
        pairStats[j0 + threadIdx.x] = dists[threadIdx.x];
        pairStats[h + j0 + threadIdx.x] = labs[threadIdx.x];
    }
}
MCN_API int mcn_l2sn_pair_stats(const float* idens, const float* data, float* pairStats, int numD, int numCases, void* stream) {
    MCN_REQUIRE(pairStats, "l2snPairStats: null pointer");
    MCN_REQUIRE(idens && data && numD > 0 && numCases > 0, "l2snPairStats: bad argument");
    MCN_REQUIRE(numCases % 2 == 0, "l2sn: numCases must be even");
    const int h = numCases / 2;
    MCN_REQUIRE(h <= L2SN_MAX_PAIRS, "l2sn: too many pairs (%d, limit %d)", h, L2SN_MAX_PAIRS);
    static const bool oneBlock = getenv("MCN_L2SN_ONE_BLOCK") != nullptr;  // (A/B switch)
    if (oneBlock || h < 2 * L2SN_WIN)
        return l2sn_launch(2, idens, data, nullptr, numD, numCases, 1.f, 0, 0.f, nullptr, pairStats, nullptr, nullptr, nullptr, 0, stream);
    const int tile = L2SN_TILE_MAX;  // floats of squares staged per pass: 1536 rows of a window
    MCN_LAUNCH(k_l2sn_stats_win, dim3(divup(h, L2SN_WIN)), dim3(256), (2 * L2SN_WIN + tile) * 4, (hipStream_t)stream, idens, data, pairStats,
               numD, numCases, tile);
    return MCN_LAUNCH_CHECK("l2snPairStats");
}

// Margin search and cost from pair statistics that are already in device memory (mcn_l2sn_pair_stats): one small workgroup.
// Same arithmetic as k_l2sn<0> behind its statistics phase.
__global__ __launch_bounds__(256) void k_l2sn_cost_stats(const float* __restrict__ pairStats, float* __restrict__ out, int numCases, float m,
                                                         float* info, double* costOut, const float* extD, const float* extL, int extCount) {
    extern __shared__ float shf[];
    const int h = numCases / 2;
    const int cnt = extD ? extCount : h;
    float* dists = shf;
    float* labs = shf + h;
    float* cd = labs + h;
    float* cl = cd + (extD ? cnt : 0);
    int* rates = (int*)(cl + (extD ? cnt : 0));
    __shared__ float sh_m[2];
    __shared__ double shsum[256];
    for (int j = threadIdx.x; j < h; j += blockDim.x) { dists[j] = pairStats[j]; labs[j] = pairStats[h + j]; }
    if (extD)
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) { cd[i] = extD[i]; cl[i] = extL[i]; }
    __syncthreads();
    float idx = -1.f;
    if (m < 0) {
        l2sn_threshold(extD ? cd : dists, extD ? cl : labs, cnt, rates, sh_m);
        m = sh_m[0];
        idx = sh_m[1];
    }
    if (info && threadIdx.x == 0) { info[0] = m; info[1] = idx; info[2] = 0.f; info[3] = 0.f; }
    l2sn_cost_tail(dists, labs, h, m, out, costOut, shsum);
}
MCN_API int mcn_l2sn_cost_from_stats(const float* pairStats, float* output, int numCases, float m, float* info, double* costOut,
                                     const float* extDists, const float* extLabels, int extCount, void* stream) {
    MCN_REQUIRE(pairStats && output && numCases > 0, "l2snCostFromStats: bad argument");
    MCN_REQUIRE(numCases % 2 == 0, "l2sn: numCases must be even");
    MCN_REQUIRE((extDists == nullptr) == (extLabels == nullptr), "l2sn: extDists and extLabels go together");
    const int h = numCases / 2, cnt = extDists ? extCount : h;
    MCN_REQUIRE(h <= L2SN_MAX_PAIRS && cnt <= L2SN_MAX_PAIRS && cnt > 0, "l2sn: too many pairs (%d, limit %d)", cnt, L2SN_MAX_PAIRS);
    const size_t lds = ((size_t)2 * h + (extDists ? 2 * cnt : 0) + cnt) * 4;
    MCN_LAUNCH(k_l2sn_cost_stats, dim3(1), dim3(256), lds, (hipStream_t)stream, pairStats, output, numCases, m, info, costOut, extDists,
               extLabels, extCount);
    return MCN_LAUNCH_CHECK("l2snCostFromStats");
}

// ---------------------------------------------------------------- L2 regression (attribute towers)
__global__ __launch_bounds__(256) void k_l2reg_cost(const float* __restrict__ ground, const float* __restrict__ data, float* __restrict__ out,
                                                    int numD, int numCases, double* costOut) {
    __shared__ double sh[256];
    double a = 0;
    for (int n = threadIdx.x; n < numCases; n += 256) {
        float s = 0.f;
        for (int k = 0; k < numD; k++) {
            const float d = data[(size_t)k * numCases + n] - ground[(size_t)k * numCases + n];
            s += d * d;
        }
        out[n] = s;
        a += (double)s;
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && costOut) *costOut += sh[0];
}
MCN_API int mcn_l2reg_cost(const float* ground, const float* data, float* output, int numD, int numCases, double* costOut, void* stream) {
    MCN_REQUIRE(ground && data && output && numD > 0 && numCases > 0, "l2regCost: bad argument");
    MCN_LAUNCH(k_l2reg_cost, dim3(1), dim3(256), 0, (hipStream_t)stream, ground, data, output, numD, numCases, costOut);
    return MCN_LAUNCH_CHECK("l2regCost");
}
__global__ __launch_bounds__(256) void k_l2reg_grad(const float* ground, const float* data, float* target, size_t n, int add, float coeff) {
    GRID_STRIDE(i, n) {
        const float v = (ground[i] - data[i]) * coeff;
        target[i] = add ? target[i] + v : v;
    }
}
MCN_API int mcn_l2reg_grad(const float* ground, const float* data, float* target, int numD, int numCases, int add, float coeff, void* stream) {
    MCN_REQUIRE(ground && data && target && numD > 0 && numCases > 0, "l2regGrad: bad argument");
    const size_t n = (size_t)numD * numCases;
    MCN_LAUNCH(k_l2reg_grad, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, ground, data, target, n, add, coeff);
    return MCN_LAUNCH_CHECK("l2regGrad");
}

// ---------------------------------------------------------------- SGD (weights.cuh:150-163) in one pass
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ W, float* __restrict__ inc, const float* __restrict__ grad, size_t n,
                                             float mom, float wcEps, int useGrad, int useWc) {
    GRID_STRIDE(i, n) {
        float v = inc[i];
        const float w = W[i];
        if (useGrad) v = mom * v + 1.f * grad[i];
        if (useWc) v = 1.f * v + wcEps * w;
        inc[i] = v;
        W[i] = w + v;
    }
}
__global__ __launch_bounds__(256) void k_sgd4(f32x4* __restrict__ W, f32x4* __restrict__ inc, const f32x4* __restrict__ grad, size_t n4,
                                              float mom, float wcEps, int useGrad, int useWc) {
    GRID_STRIDE(i, n4) {
        f32x4 v = inc[i];
        const f32x4 w = W[i];
        if (useGrad) v = mom * v + grad[i];
        if (useWc) v = v + wcEps * w;
        inc[i] = v;
        W[i] = w + v;
    }
}
MCN_API int mcn_sgd_update(float* W, float* inc, const float* grad, size_t n, float mom, float wc, float eps, int useGrad, void* stream) {
    MCN_REQUIRE(W && inc && (!useGrad || grad), "sgdUpdate: null pointer");
    if (n == 0) return MCN_OK;
    hipStream_t s = (hipStream_t)stream;
    const float wcEps = -wc * eps;
    if (n % 4 == 0 && al16(W) && al16(inc) && (!useGrad || al16(grad)))
        MCN_LAUNCH(k_sgd4, dim3(ew_blocks(n / 4)), dim3(256), 0, s, (f32x4*)W, (f32x4*)inc, (const f32x4*)grad, n / 4, mom, wcEps, useGrad, wc > 0 ? 1 : 0);
    else
        MCN_LAUNCH(k_sgd, dim3(ew_blocks(n)), dim3(256), 0, s, W, inc, grad, n, mom, wcEps, useGrad, wc > 0 ? 1 : 0);
    return MCN_LAUNCH_CHECK("sgdUpdate");
}

// Weights::update for every matrix of the net in ONE launch.  The plan (one record per matrix, with the
// first block index of each) lives in device memory and is built once; block b finds its record by a scan
// over the few dozen records and updates SGD_BLOCK_ELEMS consecutive elements of that matrix.
#define SGD_BLOCK_ELEMS 2048
struct SgdRec {
    float* W;
    float* inc;
    const float* grad;
    unsigned long long n;
    float mom, wcEps;
    int useGrad, useWc;
    unsigned blockStart, pad;
    // The gradient may still be in pieces when the update starts (mcn_sgd_plan_build_fused): kind 1 = `chunks` slabs of n
    // floats (the module-chunk partials of a conv weight gradient, mcn_conv_weight_acts_partials), kind 2 = n rows of
    // `chunks` partial row sums (stage 1 of a bias gradient, mcn_add_sum_multi_stage1).  The blocks of such a record first
    // finish the sum -- with the code and in the order of the stand-alone second stages (k_reduce_chunks / k_row_sum_short),
    // so the values are bit-identical to theirs -- store it to grad and then update their elements.
    const float* part;
    int chunks, kind;
    float partScale, pad2;
};
__device__ __forceinline__ void sgd_apply_one(const SgdRec& r, size_t e, float g) {
    float v = r.inc[e];
    const float w = r.W[e];
    v = r.mom * v + 1.f * g;
    if (r.useWc) v = 1.f * v + r.wcEps * w;
    r.inc[e] = v;
    r.W[e] = w + v;
}
__global__ __launch_bounds__(256) void k_sgd_planned(const SgdRec* __restrict__ plan, int count) {
    __shared__ int which;
    __shared__ float sh[16][17];
    if (threadIdx.x == 0) {
        int t = 0;
        while (t + 1 < count && plan[t + 1].blockStart <= blockIdx.x) t++;
        which = t;
    }
    __syncthreads();
    const SgdRec r = plan[which];
    if (r.kind == 1) {  // 16 columns of a (chunks x n) stack of partial gradients: as k_reduce_chunks (conv_bwd_weights.hip)
        const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
        const size_t c = (size_t)(blockIdx.x - r.blockStart) * 16 + cl;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (c < r.n) {
            for (int q = rg; q < r.chunks; q += 512) {
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; u++) v[u] = q + 16 * u < r.chunks ? r.part[(size_t)(q + 16 * u) * r.n + c] : 0.f;
#pragma unroll
                for (int u = 0; u < 32; u += 4) {
                    s0 += v[u];
                    s1 += v[u + 1];
                    s2 += v[u + 2];
                    s3 += v[u + 3];
                }
            }
        }
        sh[rg][cl] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        for (int o = 8; o > 0; o >>= 1) {
            if (rg < o) sh[rg][cl] += sh[rg + o][cl];
            __syncthreads();
        }
        if (rg == 0 && c < r.n) {
            const float g = sh[0][cl];
            const_cast<float*>(r.grad)[c] = g;
            sgd_apply_one(r, c, g);
        }
        return;
    }
    if (r.kind == 2) {  // 4 rows of (n x chunks) partial row sums, one wave per row: as row_sum_short4
        const size_t row = (size_t)(blockIdx.x - r.blockStart) * 4 + (threadIdx.x >> 6);
        const int lane = threadIdx.x & 63;
        if (row >= r.n) return;
        const float* m = r.part + row * r.chunks;
        float v = 0.f;
        for (int c = lane; c < r.chunks; c += 64) v += m[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) {
            const float g = 0.f + r.partScale * v;
            const_cast<float*>(r.grad)[row] = g;
            sgd_apply_one(r, row, g);
        }
        return;
    }
    const size_t base = (size_t)(blockIdx.x - r.blockStart) * SGD_BLOCK_ELEMS;
#pragma unroll
    for (int rep = 0; rep < SGD_BLOCK_ELEMS / 1024; rep++) {
        const size_t i = base + (size_t)rep * 1024 + threadIdx.x * 4;
        if (i + 3 < r.n) {
            f32x4 v = *reinterpret_cast<const f32x4*>(r.inc + i);
            const f32x4 w = *reinterpret_cast<const f32x4*>(r.W + i);
            if (r.useGrad) v = r.mom * v + *reinterpret_cast<const f32x4*>(r.grad + i);
            if (r.useWc) v = v + r.wcEps * w;
            *reinterpret_cast<f32x4*>(r.inc + i) = v;
            *reinterpret_cast<f32x4*>(r.W + i) = w + v;
        } else {
            for (size_t e = i; e < r.n && e < i + 4; e++) {
                float v = r.inc[e];
                const float w = r.W[e];
                if (r.useGrad) v = r.mom * v + 1.f * r.grad[e];
                if (r.useWc) v = 1.f * v + r.wcEps * w;
                r.inc[e] = v;
                r.W[e] = w + v;
            }
        }
    }
}
MCN_API size_t mcn_sgd_plan_bytes(int count) { return (size_t)(count > 0 ? count : 1) * sizeof(SgdRec); }
static int sgd_plan_build(const mcn_sgd_item* items, const mcn_sgd_partials* parts, int count, void* devicePlan, size_t planBytes,
                          int* numBlocksOut, void* stream) {
    MCN_REQUIRE(items && devicePlan && numBlocksOut && count > 0, "sgdPlanBuild: bad argument");
    MCN_REQUIRE(planBytes >= mcn_sgd_plan_bytes(count), "sgdPlanBuild: plan buffer too small");
    std::vector<SgdRec> recs(count);
    unsigned blocks = 0;
    for (int i = 0; i < count; i++) {
        const mcn_sgd_item& it = items[i];
        MCN_REQUIRE(it.W && it.inc && (!it.useGrad || it.grad) && it.n > 0, "sgdPlanBuild: item %d: null pointer or empty", i);
        MCN_REQUIRE(al16(it.W) && al16(it.inc) && (!it.useGrad || al16(it.grad)), "sgdPlanBuild: item %d: buffers must be 16-byte aligned", i);
        SgdRec& r = recs[i];
        r.W = it.W; r.inc = it.inc; r.grad = it.grad; r.n = it.n;
        r.mom = it.mom; r.wcEps = -it.wc * it.eps; r.useGrad = it.useGrad; r.useWc = it.wc > 0 ? 1 : 0;
        r.blockStart = blocks; r.pad = 0;
        r.part = nullptr; r.chunks = 0; r.kind = 0; r.partScale = 1.f; r.pad2 = 0.f;
        if (parts && parts[i].data) {
            const mcn_sgd_partials& q = parts[i];
            MCN_REQUIRE(it.useGrad && it.grad && q.count > 0 && (q.kind == 1 || (q.kind == 2 && q.count <= 256)),
                        "sgdPlanBuildFused: item %d: bad partials (kind 1, or kind 2 with at most 256 partial sums per row)", i);
            r.part = q.data; r.chunks = q.count; r.kind = q.kind; r.partScale = q.scale;
            blocks += (unsigned)divup_sz(it.n, q.kind == 1 ? 16 : 4);
        } else {
            blocks += (unsigned)divup_sz(it.n, SGD_BLOCK_ELEMS);
        }
    }
    *numBlocksOut = (int)blocks;
    int rc = mcn_check_hip(hipMemcpyAsync(devicePlan, recs.data(), recs.size() * sizeof(SgdRec), hipMemcpyHostToDevice, (hipStream_t)stream), "sgdPlanBuild upload");
    if (rc) return rc;
    return mcn_check_hip(hipStreamSynchronize((hipStream_t)stream), "sgdPlanBuild sync");  // recs goes out of scope
}
MCN_API int mcn_sgd_plan_build(const mcn_sgd_item* items, int count, void* devicePlan, size_t planBytes, int* numBlocksOut, void* stream) {
    return sgd_plan_build(items, nullptr, count, devicePlan, planBytes, numBlocksOut, stream);
}
MCN_API int mcn_sgd_plan_build_fused(const mcn_sgd_item* items, const mcn_sgd_partials* partials, int count, void* devicePlan,
                                     size_t planBytes, int* numBlocksOut, void* stream) {
    MCN_REQUIRE(partials, "sgdPlanBuildFused: null pointer");
    return sgd_plan_build(items, partials, count, devicePlan, planBytes, numBlocksOut, stream);
}
MCN_API int mcn_sgd_update_planned(const void* devicePlan, int count, int numBlocks, void* stream) {
    MCN_REQUIRE(devicePlan && count > 0 && numBlocks > 0, "sgdUpdatePlanned: bad argument");
    MCN_LAUNCH(k_sgd_planned, dim3(numBlocks), dim3(256), 0, (hipStream_t)stream, (const SgdRec*)devicePlan, count);
    return MCN_LAUNCH_CHECK("sgdUpdatePlanned");
}

// ---------------------------------------------------------------- Fisher2 cost (layer_kernels.cu:310-479)
// Workspace (floats): E[N*N] squared distances, S[N*N] = Q + Q^T, r[N], stats[8] = {w1, w2, 1/sum(W), a, b, V, 0, 0}.
// M[y][x] = (x > y) ? (same ? w1 : -w2) : 0 and W[y][x] = (same ? alpha : 1-alpha) / sum are recomputed from the labels
// wherever they are needed instead of being stored (kLabel2Weight :32-47).
__global__ __launch_bounds__(256) void k_fisher2_dist(const float* __restrict__ data, float* __restrict__ E, int numD, int N) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= N) return;
    float dot = 0.f, si = 0.f, sj = 0.f;
    for (int d = 0; d < numD; d++) {
        const float xi = data[(size_t)d * N + i], xj = data[(size_t)d * N + j];
        dot = __fadd_rn(dot, __fmul_rn(xi, xj));
        si = __fadd_rn(si, __fmul_rn(xi, xi));
        sj = __fadd_rn(sj, __fmul_rn(xj, xj));
    }
    E[(size_t)i * N + j] = __fadd_rn(__fadd_rn(__fmul_rn(-2.0f, dot), sj), si);
}
__device__ double block_sum_double(double v, double* sh) {  // blockDim.x == 1024
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(1024) void k_fisher2_stats(const float* __restrict__ labels, const float* __restrict__ E, float* __restrict__ stats,
                                                        int N, float alpha, double* costOut) {
    __shared__ double sh[1024];
    const size_t NN = (size_t)N * N;
    double triSame = 0, same = 0;
    for (size_t e = threadIdx.x; e < NN; e += 1024) {
        const int y = (int)(e / N), x = (int)(e % N);
        const bool sm = labels[x] == labels[y];
        if (x > y) triSame += sm ? 1.0 : -1.0;
        same += sm ? 1.0 : 0.0;
    }
    const float diff = (float)block_sum_double(triSame, sh);
    const double nSame = block_sum_double(same, sh);
    const float n = N * (N - 1.0f) / 2.0f;
    const float w1 = 2.0f / (n + diff), w2 = 2.0f / (n - diff);
    const float wsum = (float)(nSame * (double)alpha + ((double)NN - nSame) * (double)(1.0f - alpha));
    const float inv = 1.0f / wsum;
    double a = 0, b = 0;
    for (size_t e = threadIdx.x; e < NN; e += 1024) {
        const int y = (int)(e / N), x = (int)(e % N);
        const bool sm = labels[x] == labels[y];
        const float m = x > y ? (sm ? w1 : -w2) : 0.f, w = __fmul_rn(sm ? alpha : 1.0f - alpha, inv);
        a += (double)__fmul_rn(m, E[e]);
        b += (double)__fmul_rn(E[e], w);
    }
    const float fa = (float)block_sum_double(a, sh), fb = (float)block_sum_double(b, sh);
    double V = 0;
    for (size_t e = threadIdx.x; e < NN; e += 1024) {
        const int y = (int)(e / N), x = (int)(e % N);
        const float w = __fmul_rn(labels[x] == labels[y] ? alpha : 1.0f - alpha, inv);
        const float c = __fadd_rn(E[e], -fb);
        V += (double)__fmul_rn(__fmul_rn(c, c), w);
    }
    const float fV = (float)block_sum_double(V, sh);
    if (threadIdx.x == 0) {
        stats[0] = w1; stats[1] = w2; stats[2] = inv; stats[3] = fa; stats[4] = fb; stats[5] = fV; stats[6] = 0.f; stats[7] = 0.f;
        if (costOut) *costOut += (double)(__fmul_rn(-fa, fa) / fV * (float)N);  // layer.cu:1031-1034: cost * numCases
    }
}
// block i: S[i][:] = Q[i][:] + Q[:][i] with Q = -4a M + 4a^2 (E.W) - 4a^2 b W (:439-440), a = a / V (:436), and
// r[i] = the two tm vectors of :443-450 / :460-464 for index i, added
__global__ __launch_bounds__(256) void k_fisher2_q(const float* __restrict__ labels, const float* __restrict__ E, const float* __restrict__ stats,
                                                   float* __restrict__ S, float* __restrict__ r, int N, float alpha) {
    __shared__ double sh[6][256];
    const int i = blockIdx.x;
    const float w1 = stats[0], w2 = stats[1], inv = stats[2], b = stats[4];
    const float a = stats[3] / stats[5];
    const float c1 = -4.0f * a, c2 = 4.0f * a * a, c3 = -4.0f * a * a * b;
    const float li = labels[i];
    double rs[6] = {0, 0, 0, 0, 0, 0};  // row sums of M, EW, W; column sums of M, EW, W
    for (int k = threadIdx.x; k < N; k += 256) {
        const bool sm = labels[k] == li;
        const float w = __fmul_rn(sm ? alpha : 1.0f - alpha, inv);
        const float mRow = k > i ? (sm ? w1 : -w2) : 0.f, mCol = i > k ? (sm ? w1 : -w2) : 0.f;  // M[i][k], M[k][i]
        const float ewRow = __fmul_rn(E[(size_t)i * N + k], w), ewCol = __fmul_rn(E[(size_t)k * N + i], w);
        const float qRow = __fadd_rn(__fadd_rn(__fmul_rn(c1, mRow), __fmul_rn(c2, ewRow)), __fmul_rn(c3, w));
        const float qCol = __fadd_rn(__fadd_rn(__fmul_rn(c1, mCol), __fmul_rn(c2, ewCol)), __fmul_rn(c3, w));
        S[(size_t)i * N + k] = __fadd_rn(qRow, qCol);
        rs[0] += mRow; rs[1] += ewRow; rs[2] += w; rs[3] += mCol; rs[4] += ewCol; rs[5] += w;
    }
    for (int q = 0; q < 6; q++) sh[q][threadIdx.x] = rs[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int q = 0; q < 6; q++) sh[q][threadIdx.x] += sh[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float d1 = 4.0f * a, d2 = -4.0f * a * a, d3 = 4.0f * a * a * b;
        const float tr = __fadd_rn(__fadd_rn(__fmul_rn(d1, (float)sh[0][0]), __fmul_rn(d2, (float)sh[1][0])), __fmul_rn(d3, (float)sh[2][0]));
        const float tc = __fadd_rn(__fadd_rn(__fmul_rn(d1, (float)sh[3][0]), __fmul_rn(d2, (float)sh[4][0])), __fmul_rn(d3, (float)sh[5][0]));
        r[i] = __fadd_rn(tr, tc);
    }
}
// target[d][n] (+)= coeff * (x[d][n] r[n] + sum_k x[d][k] S[k][n])   (S is symmetric: read along n)
__global__ __launch_bounds__(256) void k_fisher2_grad(const float* __restrict__ data, const float* __restrict__ S, const float* __restrict__ r,
                                                      float* __restrict__ target, int numD, int N, int add, float coeff) {
    const int n = blockIdx.x * 256 + threadIdx.x, d = blockIdx.y;
    if (n >= N) return;
    const float* x = data + (size_t)d * N;
    float p = 0.f;
    for (int k = 0; k < N; k++) p = fmaf(x[k], S[(size_t)k * N + n], p);
    const float g = (x[n] * r[n] + p) * coeff;
    float* t = target + (size_t)d * N + n;
    *t = add ? *t + g : g;
}
MCN_API size_t mcn_fisher2_workspace_bytes(int numCases) { return ((size_t)2 * numCases * numCases + numCases + 8) * sizeof(float); }
static int fisher2_common(const float* labels, const float* data, int numD, int N, float alpha, double* costOut, float* ws, size_t wsBytes,
                          hipStream_t s) {
    MCN_REQUIRE(labels && data && numD > 0 && N > 1, "fisher2: bad argument");
    MCN_REQUIRE(N <= 65535, "fisher2: too many cases");
    const size_t need = mcn_fisher2_workspace_bytes(N);
    if (!ws || wsBytes < need) {
        mcn_set_error("fisher2: workspace %zu < %zu bytes", wsBytes, need);
        return MCN_EWORKSPACE;
    }
    float* E = ws;
    float* stats = ws + (size_t)2 * N * N + N;
    MCN_LAUNCH(k_fisher2_dist, dim3(divup(N, 256), N), dim3(256), 0, s, data, E, numD, N);
    MCN_LAUNCH(k_fisher2_stats, dim3(1), dim3(1024), 0, s, labels, (const float*)E, stats, N, alpha, costOut);
    return MCN_LAUNCH_CHECK("fisher2");
}
MCN_API int mcn_fisher2_cost(const float* labels, const float* data, int numD, int numCases, float alpha, double* costOut,
                             void* workspace, size_t workspaceBytes, void* stream) {
    return fisher2_common(labels, data, numD, numCases, alpha, costOut, (float*)workspace, workspaceBytes, (hipStream_t)stream);
}
MCN_API int mcn_fisher2_grad(const float* labels, const float* data, float* target, int numD, int numCases, float alpha, int add,
                             float coeff, void* workspace, size_t workspaceBytes, void* stream) {
    MCN_REQUIRE(target, "fisher2Grad: null target");
    hipStream_t s = (hipStream_t)stream;
    const int rc = fisher2_common(labels, data, numD, numCases, alpha, nullptr, (float*)workspace, workspaceBytes, s);
    if (rc != MCN_OK) return rc;
    const int N = numCases;
    float* E = (float*)workspace;
    float* S = E + (size_t)N * N;
    float* r = S + (size_t)N * N;
    float* stats = r + N;
    MCN_LAUNCH(k_fisher2_q, dim3(N), dim3(256), 0, s, labels, (const float*)E, (const float*)stats, S, r, N, alpha);
    MCN_LAUNCH(k_fisher2_grad, dim3(divup(N, 256), numD), dim3(256), 0, s, data, (const float*)S, (const float*)r, target, numD, N, add,
                       coeff);
    return MCN_LAUNCH_CHECK("fisher2Grad");
}

// ---- the reduce pass of the classifier layer's K-split backward-data product (fc_gemm.hip: k_fc_tn_split leaves `splits` slabs of
// M x N) that also adds the L2-SN gradient of the same matrix: mcn_fc_bprop_acts_l2sn.  The slabs are added exactly as
// k_fc_tn_reduce adds them (eight contiguous ranges in index order, then a fixed tree), the gradient exactly as
// k_l2sn_grad_stats(add = 1) would add it to the finished element.  The statistics may come from another stream: the threads that
// need them wait for its stream gate here, after the slabs have been summed.
#include "fc_gemm.h"
typedef float f32x4_ec __attribute__((ext_vector_type(4)));
struct TnReduceL2sn {
    const float* part;
    float* C;
    int M, N, splits;
    const float *data, *stats, *info;
    float coeff;
    int* gate;
    int gateValue;
    long long gateTicks;
};
__global__ __launch_bounds__(256) void k_fc_tn_reduce_l2sn(const TnReduceL2sn p) {
    __shared__ f32x4_ec sh[8][32];
    const f32x4_ec zero = {0.f, 0.f, 0.f, 0.f};
    const size_t total4 = (size_t)p.M * p.N / 4;
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + o;
    const bool live = i < total4;
    const int per = (p.splits + 7) / 8, z0 = g * per, z1 = min(p.splits, z0 + per);
    const f32x4_ec* src = reinterpret_cast<const f32x4_ec*>(p.part) + (live ? i : 0);
    f32x4_ec s = zero;
    for (int z = z0; z < z1; z += 8) {
        f32x4_ec v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = z + u < z1 ? src[(size_t)(z + u) * total4] : zero;
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    sh[g][o] = s;
    __syncthreads();
    if (g != 0 || !live) return;
    s = ((sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o])) + ((sh[4][o] + sh[5][o]) + (sh[6][o] + sh[7][o]));
    if (p.gate) {
        const long long t0 = wall_clock64();  // 100 MHz constant clock
        for (;;) {
            const int v = __hip_atomic_load(p.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - p.gateValue) >= 0) break;
            if (wall_clock64() - t0 > p.gateTicks) {  // (as k_stream_gate: end anyway, say so)
                __hip_atomic_store(p.gate + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    const int h = p.N / 2;
    const size_t e0 = i * 4;
    const int k = (int)(e0 / p.N), n0 = (int)(e0 - (size_t)k * p.N);
    const float m = p.info[0];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int col = n0 + t, j = col < h ? col : col - h;
        float gx, gy;
        l2sn_grad_pair(p.data[(size_t)k * p.N + j], p.data[(size_t)k * p.N + j + h], p.stats[j], p.stats[h + j], m, gx, gy);
        s[t] = __fadd_rn(s[t], __fmul_rn(col < h ? gx : gy, p.coeff));
    }
    reinterpret_cast<f32x4_ec*>(p.C)[i] = s;
}
int fc_tn_reduce_l2sn_launch(const float* part, float* C, int M, int N, int splits, const FcL2snAddend& l2, hipStream_t s) {
    TnReduceL2sn p = {part, C, M, N, splits, l2.data, l2.stats, l2.info, l2.coeff, l2.gate, l2.gateValue, (long long)l2.timeoutMs * 100000ll};
    const unsigned blocks = (unsigned)divup_sz((size_t)M * N / 4, 32);
    if (p.gate && blocks > 1024) {
        // Workgroups that spin hold their wave slots: a launch large enough to fill the device would keep the very kernels it
        // waits for from starting.  (The same fold for the weight update -- thousands of workgroups -- ran into exactly that: the
        // side stream starved and the gate timed out.)  Large outputs get the one-thread gate launch in front instead.
        int rc = mcn_stream_gate(p.gate, p.gateValue, l2.timeoutMs, s);
        if (rc) return rc;
        p.gate = nullptr;
    }
    if (p.gate) mcn_flush_tick();  // (as mcn_stream_gate: whoever polls for an armed tick may be the one that signals the gate)
    MCN_LAUNCH(k_fc_tn_reduce_l2sn, dim3(blocks), dim3(256), 0, s, p);
    return MCN_LAUNCH_CHECK("gemm (fc tn reduce + l2sn grad)");
}
