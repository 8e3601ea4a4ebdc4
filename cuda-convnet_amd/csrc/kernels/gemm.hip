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
// so no operand goes through LDS.  M here is 160..10575 with n = 128 cases, so
// occupancy, not tile size, is the lever: the host picks tiles / split-K so that
// the launch has >= 1 wave per SIMD; split-K partials are reduced in a fixed
// order by a second kernel (deterministic, no atomics).
#include <algorithm>
#include "common.h"

struct GemmParams {
    const float* A;
    const float* B;
    float* C;        // final output
    float* part;     // split-K partials [splits][M][N] (or null)
    const float* bias;
    int M, N, K, kChunk, splits, vec, neuron;
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

__device__ __forceinline__ f32x4 ld4(const float* ptr, bool vec, int idx, int limit) {
    // 4 consecutive floats starting at logical index idx of a row with `limit` valid entries
    if (vec) return idx < limit ? *reinterpret_cast<const f32x4*>(ptr) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 r;
#pragma unroll
    for (int t = 0; t < 4; t++) r[t] = idx + t < limit ? ptr[t] : 0.f;
    return r;
}

// ---- NN: A[MxK] row-major, B[KxN] row-major.  wave: 16*MV rows x 64*NQ cols.
template <int MV, int NQ>
__global__ __launch_bounds__(256) void k_gemm_nn(GemmParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int mt = blockIdx.y * 4 + wave;
    const int m0 = mt * 16 * MV;
    if (m0 >= p.M) return;
    const int n0 = blockIdx.x * 64 * NQ;
    const int kBeg = blockIdx.z * p.kChunk, kEnd = min(p.K, kBeg + p.kChunk);
    const bool vec = p.vec != 0;
    f32x4 acc[MV][NQ * 4];
#pragma unroll
    for (int v = 0; v < MV; v++)
#pragma unroll
        for (int t = 0; t < NQ * 4; t++) acc[v][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow[MV];
#pragma unroll
    for (int v = 0; v < MV; v++) arow[v] = p.A + (size_t)min(m0 + 16 * v + j, p.M - 1) * p.K + 4 * kk;
    for (int k16 = kBeg; k16 < kEnd; k16 += 16) {
        const int ka = k16 + 4 * kk;
        f32x4 a[MV];
#pragma unroll
        for (int v = 0; v < MV; v++) a[v] = ld4(arow[v] + k16, vec, ka, kEnd);
        f32x4 b[4][NQ];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int kr = ka + s;
            const float* br = p.B + (size_t)min(kr, p.K - 1) * p.N + n0 + 4 * j;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                b[s][q] = ld4(br + 64 * q, vec, n0 + 64 * q + 4 * j, kr < kEnd ? p.N : 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int v = 0; v < MV; v++)
#pragma unroll
                for (int q = 0; q < NQ; q++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[v][q * 4 + t] = mfma16(a[v][s], b[s][q][t], acc[v][q * 4 + t]);
    }
#pragma unroll
    for (int v = 0; v < MV; v++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + 16 * v + 4 * kk + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int q = 0; q < NQ; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int n = n0 + 64 * q + 4 * j + t;
                    if (n >= p.N) continue;
                    if (p.part) p.part[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[v][q * 4 + t][r];
                    else gemm_store(p, m, n, acc[v][q * 4 + t][r]);
                }
        }
}

// ---- TN: A stored [KxM] (m contiguous), B[KxN].  wave: 16*MV rows (vector of MV per lane) x 64*NQ cols.
template <int MV, int NQ>
__global__ __launch_bounds__(256) void k_gemm_tn(GemmParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int mt = blockIdx.y * 4 + wave;
    const int m0 = mt * 16 * MV;
    if (m0 >= p.M) return;
    const int n0 = blockIdx.x * 64 * NQ;
    const int kBeg = blockIdx.z * p.kChunk, kEnd = min(p.K, kBeg + p.kChunk);
    const bool vec = p.vec != 0;
    f32x4 acc[MV][NQ * 4];
#pragma unroll
    for (int v = 0; v < MV; v++)
#pragma unroll
        for (int t = 0; t < NQ * 4; t++) acc[v][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = kBeg; k0 < kEnd; k0 += 4) {
        const int k = k0 + kk;
        const bool kin = k < kEnd;
        const float* ar = p.A + (size_t)min(k, p.K - 1) * p.M + m0 + MV * j;
        float a[MV];
#pragma unroll
        for (int u = 0; u < MV; u++) a[u] = (kin && m0 + MV * j + u < p.M) ? ar[u] : 0.f;
        const float* br = p.B + (size_t)min(k, p.K - 1) * p.N + n0 + 4 * j;
        f32x4 b[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) b[q] = ld4(br + 64 * q, vec, n0 + 64 * q + 4 * j, kin ? p.N : 0);
#pragma unroll
        for (int u = 0; u < MV; u++)
#pragma unroll
            for (int q = 0; q < NQ; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[u][q * 4 + t] = mfma16(a[u], b[q][t], acc[u][q * 4 + t]);
    }
#pragma unroll
    for (int u = 0; u < MV; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + MV * (4 * kk + r) + u;
            if (m >= p.M) continue;
#pragma unroll
            for (int q = 0; q < NQ; q++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int n = n0 + 64 * q + 4 * j + t;
                    if (n >= p.N) continue;
                    if (p.part) p.part[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[u][q * 4 + t][r];
                    else gemm_store(p, m, n, acc[u][q * 4 + t][r]);
                }
        }
}

// ---- NT: A[MxK], B stored [NxK] (both k contiguous).  wave: 16*MV rows x 16*NV cols.
template <int MV, int NV>
__global__ __launch_bounds__(256) void k_gemm_nt(GemmParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int mt = blockIdx.y * 4 + wave;
    const int m0 = mt * 16 * MV;
    if (m0 >= p.M) return;
    const int n0 = blockIdx.x * 16 * NV;
    const int kBeg = blockIdx.z * p.kChunk, kEnd = min(p.K, kBeg + p.kChunk);
    const bool vec = p.vec != 0;
    f32x4 acc[MV][NV];
#pragma unroll
    for (int v = 0; v < MV; v++)
#pragma unroll
        for (int u = 0; u < NV; u++) acc[v][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow[MV];
    const float* brow[NV];
#pragma unroll
    for (int v = 0; v < MV; v++) arow[v] = p.A + (size_t)min(m0 + 16 * v + j, p.M - 1) * p.K + 4 * kk;
#pragma unroll
    for (int u = 0; u < NV; u++) brow[u] = p.B + (size_t)min(n0 + 16 * u + j, p.N - 1) * p.K + 4 * kk;
    for (int k16 = kBeg; k16 < kEnd; k16 += 16) {
        const int ka = k16 + 4 * kk;
        f32x4 a[MV], b[NV];
#pragma unroll
        for (int v = 0; v < MV; v++) a[v] = ld4(arow[v] + k16, vec, ka, kEnd);
#pragma unroll
        for (int u = 0; u < NV; u++) b[u] = ld4(brow[u] + k16, vec, ka, kEnd);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int v = 0; v < MV; v++)
#pragma unroll
                for (int u = 0; u < NV; u++) acc[v][u] = mfma16(a[v][t], b[u][t], acc[v][u]);
    }
#pragma unroll
    for (int v = 0; v < MV; v++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + 16 * v + 4 * kk + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const int n = n0 + 16 * u + j;
                if (n >= p.N) continue;
                if (p.part) p.part[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[v][u][r];
                else gemm_store(p, m, n, acc[v][u][r]);
            }
        }
}

__global__ __launch_bounds__(256) void k_gemm_reduce(GemmParams p) {
    const size_t total = (size_t)p.M * p.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < p.splits; z++) s += p.part[(size_t)z * total + i];
        gemm_store(p, (int)(i / p.N), (int)(i % p.N), s);
    }
}

struct GemmPlan { int mv, nv, splits, kChunk; };

static GemmPlan gemm_plan(int M, int N, int K, int ta, int tb) {
    GemmPlan pl;
    const long target = 1024;
    if (!tb) {
        // NN / TN: cols in quads of 64
        int nq = N > 64 ? 2 : 1;
        int mv = ta ? 4 : 2;
        auto waves = [&](int mv_, int nq_) { return (long)divup(M, 16 * mv_) * divup(N, 64 * nq_); };
        if (waves(mv, nq) < target && nq == 2) nq = 1;
        while (waves(mv, nq) < target && mv > 1) mv /= 2;
        pl.mv = mv; pl.nv = nq;
        const long w = waves(mv, nq);
        int splits = (int)std::min<long>(std::max<long>(1, target / std::max<long>(w, 1)), std::max(1, K / 64));
        pl.kChunk = divup(divup(K, splits), 16) * 16;
        pl.splits = divup(K, pl.kChunk);
    } else {
        int mv = 3, nv = 2;
        auto waves = [&](int mv_, int nv_) { return (long)divup(M, 16 * mv_) * divup(N, 16 * nv_); };
        while (waves(mv, nv) < target && mv > 1) mv--;
        if (waves(mv, nv) < target && nv == 2) nv = 1;
        pl.mv = mv; pl.nv = nv;
        const long w = waves(mv, nv);
        int splits = (int)std::min<long>(std::max<long>(1, target / std::max<long>(w, 1)), std::max(1, K / 64));
        pl.kChunk = divup(divup(K, splits), 16) * 16;
        pl.splits = divup(K, pl.kChunk);
    }
    return pl;
}

MCN_API size_t mcn_gemm_workspace_bytes(int M, int N, int K, int transA, int transB) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const GemmPlan pl = gemm_plan(M, N, K, transA, transB);
    return pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
}

static int gemm_impl(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int ta, int tb,
                     float alpha, float beta, int neuron, float na, float nb, void* workspace, size_t workspaceBytes,
                     void* stream) {
    MCN_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm: bad argument");
    MCN_REQUIRE(!(ta && tb), "gemm: transA && transB is not used by any layer and is not supported");
    MCN_REQUIRE(neuron_valid(neuron), "gemm: bad neuron type");
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
    // 16-byte loads need: NN: K%4 (A rows) and N%4 (B rows); TN: N%4; NT: K%4
    const bool a16 = ((uintptr_t)A % 16) == 0, b16 = ((uintptr_t)B % 16) == 0;
    if (!ta && !tb) p.vec = a16 && b16 && K % 4 == 0 && N % 4 == 0;
    else if (ta) p.vec = b16 && N % 4 == 0;
    else p.vec = a16 && b16 && K % 4 == 0;
    hipStream_t s = (hipStream_t)stream;
    if (!tb) {
        dim3 grid(divup(N, 64 * pl.nv), divup(divup(M, 16 * pl.mv), 4), pl.splits);
        MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
#define LAUNCH(KERN, MVv, NQv) hipLaunchKernelGGL((KERN<MVv, NQv>), grid, dim3(256), 0, s, p)
        if (!ta) {
            if (pl.mv == 2 && pl.nv == 2) LAUNCH(k_gemm_nn, 2, 2);
            else if (pl.mv == 2) LAUNCH(k_gemm_nn, 2, 1);
            else if (pl.nv == 2) LAUNCH(k_gemm_nn, 1, 2);
            else LAUNCH(k_gemm_nn, 1, 1);
        } else {
            if (pl.mv == 4 && pl.nv == 2) LAUNCH(k_gemm_tn, 4, 2);
            else if (pl.mv == 4) LAUNCH(k_gemm_tn, 4, 1);
            else if (pl.mv == 2 && pl.nv == 2) LAUNCH(k_gemm_tn, 2, 2);
            else if (pl.mv == 2) LAUNCH(k_gemm_tn, 2, 1);
            else if (pl.nv == 2) LAUNCH(k_gemm_tn, 1, 2);
            else LAUNCH(k_gemm_tn, 1, 1);
        }
    } else {
        dim3 grid(divup(N, 16 * pl.nv), divup(divup(M, 16 * pl.mv), 4), pl.splits);
        MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
        if (pl.mv == 3 && pl.nv == 2) LAUNCH(k_gemm_nt, 3, 2);
        else if (pl.mv == 3) LAUNCH(k_gemm_nt, 3, 1);
        else if (pl.mv == 2 && pl.nv == 2) LAUNCH(k_gemm_nt, 2, 2);
        else if (pl.mv == 2) LAUNCH(k_gemm_nt, 2, 1);
        else if (pl.nv == 2) LAUNCH(k_gemm_nt, 1, 2);
        else LAUNCH(k_gemm_nt, 1, 1);
#undef LAUNCH
    }
    int rc = MCN_LAUNCH_CHECK("gemm");
    if (rc) return rc;
    if (pl.splits > 1) {
        const size_t total = (size_t)M * N;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)std::min<size_t>(divup_sz(total, 256), 4096)), dim3(256), 0, s, p);
        rc = MCN_LAUNCH_CHECK("gemm reduce");
    }
    return rc;
}

MCN_API int mcn_gemm(const float* A, const float* B, float* C, int M, int N, int K, int transA, int transB, float alpha,
                     float beta, void* workspace, size_t workspaceBytes, void* stream) {
    return gemm_impl(A, B, nullptr, C, M, N, K, transA, transB, alpha, beta, MCN_NEURON_IDENT, 0.f, 0.f, workspace,
                     workspaceBytes, stream);
}

MCN_API int mcn_gemm_bias_neuron(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, float beta,
                                 int neuron, float neuronA, float neuronB, void* workspace, size_t workspaceBytes, void* stream) {
    return gemm_impl(A, B, bias, C, M, N, K, 0, 0, 1.f, beta, neuron, neuronA, neuronB, workspace, workspaceBytes, stream);
}
