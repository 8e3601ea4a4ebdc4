// Fully-connected GEMMs on fp32 MFMA (replaces NVMatrix::addProduct / rightMult,
// nvmatrix.cu:189-239, i.e. the cublasSgemm calls of FCLayer, layer.cu:355-384).
//
// Row-major C[MxN] = beta*C + alpha*op(A)*op(B) (+ bias[row], neuron).  With the
// reference's (features, cases) memory layout the three FC products are
//   fwd        Y[O x n] = W[O x I] * X[I x n]          -> NN : A k-contiguous, B n-contiguous
//   bwd-data   dX[I x n] = W^T * dY,  W stored [O x I] -> TN : A m-contiguous, B n-contiguous
//   bwd-weight dW[O x I] = dY[O x n] * X^T             -> NT : A and B k-contiguous
// Every operand is read with 16-byte loads along its contiguous dimension; the
// free index (TN: m, all: n) or the reduction index (NN: A, NT: both) of
// v_mfma_f32_16x16x4_f32 is permuted accordingly (same trick as the conv kernels),
// so no operand goes through LDS.  M here is 160..10575 with n = 128 cases: the
// launches are small, so latency is hidden inside the wave (batched loads with a
// prefetched second register set) and K is split over the waves of a workgroup
// (summed through LDS); an inter-block split with a fixed-order second pass is
// used only for very long K (fc6 backward-data).  Deterministic, no atomics.
#include <algorithm>
#include "common.h"
#include "fc_gemm.h"

struct GemmParams {
    const float* A;
    const float* B;
    float* C;        // final output
    float* part;     // split-K partials [splits][M][N] (or null)
    const float* bias;
    int M, N, K, kChunk, splits, vec, avec, cvec, neuron;
    float alpha, beta, na, nb;
};

__device__ __forceinline__ void gemm_store(const GemmParams& p, int m, int n, float v) {
    // final epilogue for one element
    float* c = p.C + (size_t)m * p.N + n;
    float o = p.alpha * v;
    if (p.beta != 0.f) o += p.beta * *c;
    if (p.bias) o += p.bias[m];
    if (p.neuron != MCN_NEURON_IDENT) o = neuron_apply(p.neuron, o, p.na, p.nb);
    *c = o;
}

// four consecutive columns n..n+3 of row m
__device__ __forceinline__ void gemm_store4(const GemmParams& p, int m, int n, f32x4 v) {
    if (p.cvec && n + 3 < p.N) {
        float* base = p.part ? p.part + ((size_t)blockIdx.z * p.M + m) * p.N + n : p.C + (size_t)m * p.N + n;
        if (!p.part) {
            v = p.alpha * v;
            if (p.beta != 0.f) v += p.beta * *reinterpret_cast<const f32x4*>(base);
            if (p.bias) v += p.bias[m];
            if (p.neuron != MCN_NEURON_IDENT) {
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = neuron_apply(p.neuron, v[t], p.na, p.nb);
            }
        }
        *reinterpret_cast<f32x4*>(base) = v;
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (n + t >= p.N) continue;
            if (p.part) p.part[((size_t)blockIdx.z * p.M + m) * p.N + n + t] = v[t];
            else gemm_store(p, m, n + t, v[t]);
        }
    }
}

__device__ __forceinline__ f32x4 ld4(const float* ptr, bool vec, int idx, int limit) {
    // 4 consecutive floats starting at logical index idx of a row with `limit` valid entries
    if (vec) return idx < limit ? *reinterpret_cast<const f32x4*>(ptr) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 r;
#pragma unroll
    for (int t = 0; t < 4; t++) r[t] = idx + t < limit ? ptr[t] : 0.f;
    return r;
}

// ---------------------------------------------------------------------------------------------
// All three kernels share one structure.  A workgroup of NW waves owns ONE C tile; its waves take the
// k-batches of the block's K range round-robin, each batch being U k-steps whose loads are all issued
// before its MFMAs while the next batch is prefetched into a second register set (M, N and the wave
// counts here are small: load latency has to be hidden inside the wave, not by occupancy).  The waves'
// accumulators are summed through LDS in a fixed halving tree and wave 0 finishes the tile
// (alpha/beta/bias/neuron, or the split-K partial).
// ---------------------------------------------------------------------------------------------
// halving tree over the waves of the block (fixed order); afterwards wave 0 holds the sums
template <int T>
__device__ __forceinline__ void gemm_wave_tree_sum(f32x4 (&acc)[T], f32x4* red, int wave, int lane, int NW) {
    for (int h = NW >> 1; h >= 1; h >>= 1) {
        __syncthreads();
        if (wave >= h && wave < 2 * h) {
#pragma unroll
            for (int t = 0; t < T; t++) red[((wave - h) * T + t) * 64 + lane] = acc[t];
        }
        __syncthreads();
        if (wave < h) {
#pragma unroll
            for (int t = 0; t < T; t++) acc[t] += red[(wave * T + t) * 64 + lane];
        }
    }
    // publish the totals so that every wave can finish a share of the tile (a single wave issuing the whole
    // epilogue would dominate the kernel: one wave retires roughly one vector instruction per 4-8 cycles)
    if (NW > 1) {
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < T; t++) red[t * 64 + lane] = acc[t];
        }
        __syncthreads();
    }
}

// ---- NN: A[MxK] row-major, B[KxN] row-major.  block tile: 16*MV rows x 64*NQ cols, k-step = 16.
template <int MV, int NQ, int U>
__global__ __launch_bounds__(512) void k_gemm_nn(const GemmParams p) {
    extern __shared__ f32x4 red[];
    constexpr int T = MV * NQ * 4;
    const int NW = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler: scalar loop control, no exec-masked loops
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int m0 = blockIdx.y * 16 * MV, n0 = blockIdx.x * 64 * NQ;
    const int kBeg = blockIdx.z * p.kChunk, kEnd = min(p.K, kBeg + p.kChunk);
    const bool vec = p.vec != 0;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow[MV];
#pragma unroll
    for (int v = 0; v < MV; v++) arow[v] = p.A + (size_t)min(m0 + 16 * v + j, p.M - 1) * p.K + 4 * kk;
    const float* bcol = p.B + n0 + 4 * j;

    f32x4 a0[U][MV], a1[U][MV], b0[U][4][NQ], b1[U][4][NQ];
    auto load = [&](int kb, f32x4(&a)[U][MV], f32x4(&b)[U][4][NQ]) {
#pragma unroll
        for (int s = 0; s < U; s++) {
            const int k16 = kb + 16 * s, ka = k16 + 4 * kk;
#pragma unroll
            for (int v = 0; v < MV; v++) a[s][v] = ld4(arow[v] + k16, vec, ka, kEnd);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int kr = ka + e;
                const float* br = bcol + (size_t)min(kr, p.K - 1) * p.N;
#pragma unroll
                for (int q = 0; q < NQ; q++) b[s][e][q] = ld4(br + 64 * q, vec, n0 + 64 * q + 4 * j, kr < kEnd ? p.N : 0);
            }
        }
    };
    auto compute = [&](const f32x4(&a)[U][MV], const f32x4(&b)[U][4][NQ]) {
#pragma unroll
        for (int s = 0; s < U; s++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int v = 0; v < MV; v++)
#pragma unroll
                    for (int q = 0; q < NQ; q++)
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            acc[(v * NQ + q) * 4 + t] = mfma16(a[s][v][e], b[s][e][q][t], acc[(v * NQ + q) * 4 + t]);
    };
    const int KB = 16 * U, kStep = KB * NW;
    const int kFirst = kBeg + wave * KB;
    if (kFirst < kEnd) load(kFirst, a0, b0);
    for (int kb = kFirst; kb < kEnd; kb += 2 * kStep) {
        const bool more1 = kb + kStep < kEnd, more2 = kb + 2 * kStep < kEnd;
        if (more1) load(kb + kStep, a1, b1);
        compute(a0, b0);
        if (more1) {
            if (more2) load(kb + 2 * kStep, a0, b0);
            compute(a1, b1);
        }
    }
    gemm_wave_tree_sum<T>(acc, red, wave, lane, NW);
#pragma unroll
    for (int v = 0; v < MV; v++)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if ((v * NQ + q) % NW != wave) continue;
            f32x4 sum[4];
#pragma unroll
            for (int t = 0; t < 4; t++) sum[t] = NW > 1 ? red[((v * NQ + q) * 4 + t) * 64 + lane] : acc[(v * NQ + q) * 4 + t];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + 16 * v + 4 * kk + r;
                if (m >= p.M) continue;
                gemm_store4(p, m, n0 + 64 * q + 4 * j, (f32x4){sum[0][r], sum[1][r], sum[2][r], sum[3][r]});
            }
        }
}

// ---- TN: A stored [KxM] (m contiguous), B[KxN].  block tile: 64 rows x 64*NQ cols, k-step = 4.
// Lane (j, kk) reads 4 consecutive m of row k0+kk; component u of that vector feeds MFMA tile u, whose row j
// is therefore m0 + 4j + u (the M index of an MFMA may be permuted freely).
template <int NQ, int U>
__global__ __launch_bounds__(512) void k_gemm_tn(const GemmParams p) {
    extern __shared__ f32x4 red[];
    constexpr int T = 4 * NQ * 4;
    const int NW = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler: scalar loop control, no exec-masked loops
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64 * NQ;
    const int kBeg = blockIdx.z * p.kChunk, kEnd = min(p.K, kBeg + p.kChunk);
    const bool vec = p.vec != 0, avec = p.avec != 0;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* ab = p.A + m0 + 4 * j;
    const float* bb = p.B + n0 + 4 * j;

    f32x4 a0[U], a1[U], b0[U][NQ], b1[U][NQ];
    auto load = [&](int kb, f32x4(&a)[U], f32x4(&b)[U][NQ]) {
#pragma unroll
        for (int s = 0; s < U; s++) {
            const int k = kb + 4 * s + kk;
            const bool kin = k < kEnd;
            const size_t kr = (size_t)min(k, p.K - 1);
            a[s] = ld4(ab + kr * p.M, avec, m0 + 4 * j, kin ? p.M : 0);
#pragma unroll
            for (int q = 0; q < NQ; q++) b[s][q] = ld4(bb + kr * p.N + 64 * q, vec, n0 + 64 * q + 4 * j, kin ? p.N : 0);
        }
    };
    auto compute = [&](const f32x4(&a)[U], const f32x4(&b)[U][NQ]) {
#pragma unroll
        for (int s = 0; s < U; s++)
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int q = 0; q < NQ; q++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[(u * NQ + q) * 4 + t] = mfma16(a[s][u], b[s][q][t], acc[(u * NQ + q) * 4 + t]);
    };
    const int KB = 4 * U, kStep = KB * NW;
    const int kFirst = kBeg + wave * KB;
    if (kFirst < kEnd) load(kFirst, a0, b0);
    for (int kb = kFirst; kb < kEnd; kb += 2 * kStep) {
        const bool more1 = kb + kStep < kEnd, more2 = kb + 2 * kStep < kEnd;
        if (more1) load(kb + kStep, a1, b1);
        compute(a0, b0);
        if (more1) {
            if (more2) load(kb + 2 * kStep, a0, b0);
            compute(a1, b1);
        }
    }
    gemm_wave_tree_sum<T>(acc, red, wave, lane, NW);
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if ((u * NQ + q) % NW != wave) continue;
            f32x4 sum[4];
#pragma unroll
            for (int t = 0; t < 4; t++) sum[t] = NW > 1 ? red[((u * NQ + q) * 4 + t) * 64 + lane] : acc[(u * NQ + q) * 4 + t];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + 4 * (4 * kk + r) + u;
                if (m >= p.M) continue;
                gemm_store4(p, m, n0 + 64 * q + 4 * j, (f32x4){sum[0][r], sum[1][r], sum[2][r], sum[3][r]});
            }
        }
}

// ---- NT: A[MxK], B stored [NxK] (both k contiguous).  block tile: 16*MV rows x 16*NV cols, k-step = 16.
template <int MV, int NV, int U>
__global__ __launch_bounds__(512) void k_gemm_nt(const GemmParams p) {
    extern __shared__ f32x4 red[];
    constexpr int T = MV * NV;
    const int NW = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform for the compiler: scalar loop control, no exec-masked loops
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int m0 = blockIdx.y * 16 * MV, n0 = blockIdx.x * 16 * NV;
    const int kBeg = blockIdx.z * p.kChunk, kEnd = min(p.K, kBeg + p.kChunk);
    const bool vec = p.vec != 0;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow[MV];
    const float* brow[NV];
#pragma unroll
    for (int v = 0; v < MV; v++) arow[v] = p.A + (size_t)min(m0 + 16 * v + j, p.M - 1) * p.K + 4 * kk;
#pragma unroll
    for (int u = 0; u < NV; u++) brow[u] = p.B + (size_t)min(n0 + 16 * u + j, p.N - 1) * p.K + 4 * kk;

    f32x4 a0[U][MV], a1[U][MV], b0[U][NV], b1[U][NV];
    auto load = [&](int kb, f32x4(&a)[U][MV], f32x4(&b)[U][NV]) {
#pragma unroll
        for (int s = 0; s < U; s++) {
            const int k16 = kb + 16 * s, ka = k16 + 4 * kk;
#pragma unroll
            for (int v = 0; v < MV; v++) a[s][v] = ld4(arow[v] + k16, vec, ka, kEnd);
#pragma unroll
            for (int u = 0; u < NV; u++) b[s][u] = ld4(brow[u] + k16, vec, ka, kEnd);
        }
    };
    auto compute = [&](const f32x4(&a)[U][MV], const f32x4(&b)[U][NV]) {
#pragma unroll
        for (int s = 0; s < U; s++)
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int v = 0; v < MV; v++)
#pragma unroll
                    for (int u = 0; u < NV; u++) acc[v * NV + u] = mfma16(a[s][v][e], b[s][u][e], acc[v * NV + u]);
    };
    const int KB = 16 * U, kStep = KB * NW;
    const int kFirst = kBeg + wave * KB;
    if (kFirst < kEnd) load(kFirst, a0, b0);
    for (int kb = kFirst; kb < kEnd; kb += 2 * kStep) {
        const bool more1 = kb + kStep < kEnd, more2 = kb + 2 * kStep < kEnd;
        if (more1) load(kb + kStep, a1, b1);
        compute(a0, b0);
        if (more1) {
            if (more2) load(kb + 2 * kStep, a0, b0);
            compute(a1, b1);
        }
    }
    gemm_wave_tree_sum<T>(acc, red, wave, lane, NW);
#pragma unroll
    for (int v = 0; v < MV; v++)
#pragma unroll
        for (int u = 0; u < NV; u++) {
            if ((v * NV + u) % NW != wave) continue;
            const f32x4 sum = NW > 1 ? red[(v * NV + u) * 64 + lane] : acc[v * NV + u];
            const int n = n0 + 16 * u + j;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + 16 * v + 4 * kk + r;
                if (m >= p.M) continue;
                if (p.part) p.part[((size_t)blockIdx.z * p.M + m) * p.N + n] = sum[r];
                else gemm_store(p, m, n, sum[r]);
            }
        }
}

// split-K partials [splits][M*N] -> C, 4 elements per thread, slabs added in index order
__global__ __launch_bounds__(256) void k_gemm_reduce(const GemmParams p) {
    const size_t total = (size_t)p.M * p.N;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    if (i + 3 < total && total % 4 == 0) {
        f32x4 s = *reinterpret_cast<const f32x4*>(p.part + i);
        int z = 1;
        for (; z + 3 < p.splits; z += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p.part + (size_t)z * total + i);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p.part + (size_t)(z + 1) * total + i);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p.part + (size_t)(z + 2) * total + i);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p.part + (size_t)(z + 3) * total + i);
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; z < p.splits; z++) s += *reinterpret_cast<const f32x4*>(p.part + (size_t)z * total + i);
#pragma unroll
        for (int t = 0; t < 4; t++) gemm_store(p, (int)((i + t) / p.N), (int)((i + t) % p.N), s[t]);
    } else {
        for (size_t e = i; e < total && e < i + 4; e++) {
            float s = 0.f;
            for (int z = 0; z < p.splits; z++) s += p.part[(size_t)z * total + e];
            gemm_store(p, (int)(e / p.N), (int)(e % p.N), s);
        }
    }
}

// Tile / wave-count / split choice.  ksteps: k-steps of the variant (16 k for NN/NT, 4 k for TN).
struct GemmPlan { int mv, nv, u, nw, splits, kChunk; };

static GemmPlan gemm_plan(int M, int N, int K, int ta, int tb) {
    GemmPlan pl;
    const int kstep = ta ? 4 : 16;
    long tiles;
    // tall outputs (fc6: 10575 rows): measured on MI355X (tools/kbench.py fc6), 32-row tiles with two waves splitting k
    // beat 16-row tiles with four (NN 15.8 -> 11.6 us, NT 20.3 -> 17.0 us)
    const bool tall = M >= 1024;
    if (tb) {          // NT: 16*mv x 16*nv
        const int ntl = divup(N, 16);
        pl.mv = 1;
        pl.nv = ntl % 5 == 0 ? 5 : ntl % 4 == 0 ? 4 : ntl % 3 == 0 ? 3 : ntl % 2 == 0 ? 2 : (ntl >= 8 ? 4 : 1);
        if (tall && ntl % 2 == 0) pl.nv = 2;
        tiles = (long)divup(M, 16 * pl.mv) * divup(N, 16 * pl.nv);
        pl.u = 2;
    } else if (ta) {   // TN: 64 x 64*nv
        pl.mv = 4;
        pl.nv = 1;
        tiles = (long)divup(M, 64) * divup(N, 64);
        pl.u = 4;
    } else {           // NN: 16*mv x 64*nv
        pl.mv = tall ? 2 : 1;
        pl.nv = (!tall && N > 64 && (long)divup(M, 16) * divup(N, 128) >= 512) ? 2 : 1;
        tiles = (long)divup(M, 16 * pl.mv) * divup(N, 64 * pl.nv);
        pl.u = 2;
    }
    const int ksteps = divup(K, kstep);
    const int batches = divup(ksteps, pl.u);
    // waves per block split the k-batches: up to 8 (512 threads keep 256 VGPRs per lane for the two register sets),
    // fewer when the grid is already large or the k loop is short
    int nw = tiles >= 1024 ? 4 : 8;
    if (tall && batches <= 6) nw = 2;
    while (nw > 1 && nw > batches) nw /= 2;
    pl.nw = nw;
    // inter-block split only when a wave would still walk many batches and the grid is small
    int splits = 1;
    const int perWave = divup(batches, nw);
    if (perWave > 4 && tiles * nw < 2048) splits = (int)std::min<long>(std::min<long>(divup(perWave, 2), 2048 / (tiles * nw) + 1), 64);
    pl.kChunk = divup(divup(K, splits), kstep * pl.u) * kstep * pl.u;
    pl.splits = divup(K, pl.kChunk);
    return pl;
}

MCN_API size_t mcn_gemm_workspace_bytes(int M, int N, int K, int transA, int transB) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const GemmPlan pl = gemm_plan(M, N, K, transA, transB);
    size_t b = pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
    // (either form may be chosen at launch, depending on the alignment of the pointers)
    if (transA && !transB && fc_tn_split_usable(nullptr, nullptr, nullptr, M, N, K)) b = std::max(b, fc_tn_split_workspace_bytes(M, N, K));
    return b;
}

// FCLayer::bpropActs (layer.cu:368-373: targets = W^T v) for a classifier layer whose input also gets the L2-SN cost's gradient
// (L2SNCostLayer::bpropActs layer.cu:1073-1082 -> computeL2SNGrad layer_kernels.cu) as its NEXT contribution: the pass that adds
// the product's K-split slabs adds that gradient to each finished element, after waiting (in the kernel) for the stream gate
// that announces the cost's statistics.  Same operations in the same order as mcn_gemm(transA) followed by a gate and
// mcn_l2sn_grad_from_stats(add = 1).
MCN_API int mcn_fc_bprop_acts_l2sn_supported(const float* weights, const float* actsGrad, const float* targets, int numInputs,
                                             int numCases, int numOutputs) {
    static const bool general = getenv("MCN_GEMM_GENERAL") != nullptr;
    return (!general && numInputs > 0 && numCases > 0 && numCases % 2 == 0 && numOutputs > 0 &&
            fc_tn_split_usable(weights, actsGrad, targets, numInputs, numCases, numOutputs)) ? 1 : 0;
}
MCN_API int mcn_fc_bprop_acts_l2sn(const float* weights, const float* actsGrad, float* targets, int numInputs, int numCases,
                                   int numOutputs, void* workspace, size_t workspaceBytes, const float* l2snData,
                                   const float* pairStats, const float* info, float coeff, int* gate, int gateValue, int timeoutMs,
                                   void* stream) {
    MCN_REQUIRE(weights && actsGrad && targets && l2snData && pairStats && info, "fcBpropActsL2sn: null pointer");
    MCN_REQUIRE(mcn_fc_bprop_acts_l2sn_supported(weights, actsGrad, targets, numInputs, numCases, numOutputs) == 1,
                "fcBpropActsL2sn: shape outside the K-split form (mcn_fc_bprop_acts_l2sn_supported)");
    MCN_REQUIRE(!gate || timeoutMs > 0, "fcBpropActsL2sn: a gate needs a timeout");
    const FcL2snAddend l2 = {l2snData, pairStats, info, coeff, gate, gateValue, timeoutMs};
    return fc_tn_split_launch(weights, actsGrad, targets, numInputs, numCases, numOutputs, 1.f, 0.f, workspace, workspaceBytes,
                              (hipStream_t)stream, &l2);
}

static int gemm_impl(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int ta, int tb,
                     float alpha, float beta, int neuron, float na, float nb, void* workspace, size_t workspaceBytes,
                     void* stream) {
    MCN_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm: bad argument");
    MCN_REQUIRE(!(ta && tb), "gemm: transA && transB is not used by any layer and is not supported");
    MCN_REQUIRE(neuron_valid(neuron), "gemm: bad neuron type");
    // classifier-layer shapes: LDS-staged / row-split forms (fc_gemm.hip)
    static const bool general = getenv("MCN_GEMM_GENERAL") != nullptr;  // (A/B switch: the general kernels for every shape)
    if (!general) {
        if (!ta && !tb && fc_nn_lds_usable(A, B, C, M, N, K))
            return fc_nn_lds_launch(A, B, bias, C, M, N, K, alpha, beta, neuron, na, nb, nullptr, (hipStream_t)stream);
        if (tb && !bias && neuron == MCN_NEURON_IDENT && fc_nt_lds_usable(A, B, C, M, N, K))
            return fc_nt_lds_launch(A, B, C, M, N, K, alpha, beta, (hipStream_t)stream);
        if (ta && !bias && neuron == MCN_NEURON_IDENT && fc_tn_split_usable(A, B, C, M, N, K))
            return fc_tn_split_launch(A, B, C, M, N, K, alpha, beta, workspace, workspaceBytes, (hipStream_t)stream);
    }
    const GemmPlan pl = gemm_plan(M, N, K, ta, tb);
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K;
    p.kChunk = pl.kChunk; p.splits = pl.splits; p.neuron = neuron;
    p.alpha = alpha; p.beta = beta; p.na = na; p.nb = nb;
    p.part = nullptr;
    if (pl.splits > 1) {
        const size_t need = (size_t)pl.splits * M * N * sizeof(float);
        if (!workspace || workspaceBytes < need) {
            mcn_set_error("gemm: workspace %zu < %zu bytes", workspaceBytes, need);
            return MCN_EWORKSPACE;
        }
        p.part = (float*)workspace;
    }
    // 16-byte loads need: NN: K%4 (A rows) and N%4 (B rows); TN: M%4 (A rows) and N%4; NT: K%4
    const bool a16 = ((uintptr_t)A % 16) == 0, b16 = ((uintptr_t)B % 16) == 0;
    p.avec = 0;
    p.cvec = N % 4 == 0 && ((uintptr_t)C % 16) == 0 && (!p.part || ((uintptr_t)p.part % 16) == 0);
    if (!ta && !tb) p.vec = a16 && b16 && K % 4 == 0 && N % 4 == 0;
    else if (ta) { p.vec = b16 && N % 4 == 0; p.avec = a16 && M % 4 == 0; }
    else p.vec = a16 && b16 && K % 4 == 0;
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(64 * pl.nw);
    if (!ta && !tb) {
        dim3 grid(divup(N, 64 * pl.nv), divup(M, 16 * pl.mv), pl.splits);
        MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
        const size_t lds = (size_t)std::max(1, pl.nw / 2) * pl.mv * pl.nv * 4 * 64 * 16;
        if (pl.mv == 2 && pl.nv == 2) MCN_LAUNCH((k_gemm_nn<2, 2, 2>), grid, block, lds, s, p);
        else if (pl.mv == 2) MCN_LAUNCH((k_gemm_nn<2, 1, 2>), grid, block, lds, s, p);
        else if (pl.nv == 2) MCN_LAUNCH((k_gemm_nn<1, 2, 2>), grid, block, lds, s, p);
        else MCN_LAUNCH((k_gemm_nn<1, 1, 2>), grid, block, lds, s, p);
    } else if (ta) {
        dim3 grid(divup(N, 64), divup(M, 64), pl.splits);
        MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
        const size_t lds = (size_t)std::max(1, pl.nw / 2) * 16 * 64 * 16;
        MCN_LAUNCH((k_gemm_tn<1, 4>), grid, block, lds, s, p);
    } else {
        dim3 grid(divup(N, 16 * pl.nv), divup(M, 16 * pl.mv), pl.splits);
        MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
        const size_t lds = (size_t)std::max(1, pl.nw / 2) * pl.mv * pl.nv * 64 * 16;
        switch (pl.nv) {
            case 5: MCN_LAUNCH((k_gemm_nt<1, 5, 2>), grid, block, lds, s, p); break;
            case 4: MCN_LAUNCH((k_gemm_nt<1, 4, 2>), grid, block, lds, s, p); break;
            case 3: MCN_LAUNCH((k_gemm_nt<1, 3, 2>), grid, block, lds, s, p); break;
            case 2: MCN_LAUNCH((k_gemm_nt<1, 2, 2>), grid, block, lds, s, p); break;
            default: MCN_LAUNCH((k_gemm_nt<1, 1, 2>), grid, block, lds, s, p); break;
        }
    }
    int rc = MCN_LAUNCH_CHECK("gemm");
    if (rc) return rc;
    if (pl.splits > 1) {
        const size_t total = (size_t)M * N;
        MCN_LAUNCH(k_gemm_reduce, dim3((unsigned)divup_sz(total, 1024)), dim3(256), 0, s, p);
        rc = MCN_LAUNCH_CHECK("gemm reduce");
    }
    return rc;
}

MCN_API int mcn_gemm(const float* A, const float* B, float* C, int M, int N, int K, int transA, int transB, float alpha,
                     float beta, void* workspace, size_t workspaceBytes, void* stream) {
    return gemm_impl(A, B, nullptr, C, M, N, K, transA, transB, alpha, beta, MCN_NEURON_IDENT, 0.f, 0.f, workspace,
                     workspaceBytes, stream);
}

// FCLayer::fpropActs under a softmax: the forward product that also leaves stage 1 of the softmax (per 64-row block and
// case: max and sum of exp(x - max)) in softmaxPart, laid out like the workspace of mcn_softmax.
MCN_API int mcn_gemm_softmax_part_supported(const float* A, const float* B, const float* C, int M, int N, int K) {
    static const bool general = getenv("MCN_GEMM_GENERAL") != nullptr;
    return (A && B && C && M > 0 && N > 0 && K > 0 && !general && fc_nn_lds_usable(A, B, C, M, N, K)) ? 1 : 0;
}
MCN_API int mcn_gemm_bias_neuron_softmax_part(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                                              int neuron, float neuronA, float neuronB, void* softmaxPart, size_t softmaxPartBytes,
                                              void* stream) {
    MCN_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && softmaxPart, "gemmSoftmaxPart: bad argument");
    MCN_REQUIRE(neuron_valid(neuron), "gemmSoftmaxPart: bad neuron type");
    MCN_REQUIRE(mcn_gemm_softmax_part_supported(A, B, C, M, N, K), "gemmSoftmaxPart: this shape has no fused form (M %d, N %d, K %d)", M, N, K);
    const size_t need = mcn_softmax_workspace_bytes(M, N);
    if (softmaxPartBytes < need) {
        mcn_set_error("gemmSoftmaxPart: partial buffer %zu < %zu bytes", softmaxPartBytes, need);
        return MCN_EWORKSPACE;
    }
    return fc_nn_lds_launch(A, B, bias, C, M, N, K, 1.f, 0.f, neuron, neuronA, neuronB, (float*)softmaxPart, (hipStream_t)stream);
}

MCN_API int mcn_gemm_bias_neuron(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, float beta,
                                 int neuron, float neuronA, float neuronB, void* workspace, size_t workspaceBytes, void* stream) {
    return gemm_impl(A, B, bias, C, M, N, K, 0, 0, 1.f, beta, neuron, neuronA, neuronB, workspace, workspaceBytes, stream);
}
