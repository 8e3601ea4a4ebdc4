// FC-layer GEMMs for the shapes of the classifier layer (fc6: 4213 / 10575 outputs x 160 / 512 inputs, 128 cases):
// LDS-staged forms of the three products of FCLayer (layer.cu:355-384, NVMatrix::addProduct nvmatrix.cu:189-239).
//
// The general kernels of gemm.hip give every wave its own operand stream from global memory, which for a tall W
// (thousands of rows) against a small X (inputs x cases, 80 KB) means that every wave fetches X again: PMC showed
// 37.6 MB of traffic for 12.3 MB of data on the backward-data product and the launches ran at 12-18 % of the MFMA
// rate.  Here
//   * forward  Y[O x n] = W[O x I] X[I x n] (+ bias, neuron)  and
//     weights  dW[O x I] = a dY[O x n] X[I x n]^T + b dW
//     stage X ONCE per workgroup in LDS (conflict-free for ds_read_b128: see the layouts below); a wave owns 16 rows
//     of W / dY, whose fragments are all requested before the first MFMA, and every X fragment it needs comes from LDS;
//   * backward-data dX[I x n] = W^T dY (reduction over the O = thousands of rows) splits the rows over the grid, with
//     output tiles of 80 x 64 so that the partial slabs stay small (64 x 80 KB instead of the operands being re-read
//     per 64 x 64 tile), and a fixed-order second pass adds the slabs.
// The forward kernel can also leave, per 64-row block, the per-case (max, sum exp(x - max)) of its rows: stage 1 of the
// softmax that follows the classifier layer (SoftmaxLayer::fpropActs layer.cu:585-595), which then needs no pass of
// its own over the (classes x cases) matrix.
// v_mfma_f32_16x16x4_f32 throughout (exact fp32), operand index permutations as in gemm.hip.
#include <algorithm>
#include "common.h"
#include "fc_gemm.h"

struct FcParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    float* smPart;   // NN only: (rowBlocks, N, 2) softmax stage-1 partials, or null
    float* part;     // TN only: split-K partial slabs
    int M, N, K;
    int kChunk, splits;
    int neuron;
    float alpha, beta, na, nb;
    McnTick tick;    // NN / TN split: stored by the launch's first thread (MCN_TICK_PROLOGUE)
    int xcdGroup;    // TN split: launch order remapped so that the tiles of a K chunk share an XCD
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
#define F4ZERO ((f32x4){0.f, 0.f, 0.f, 0.f})

// ---------------------------------------------------------------------------------------------
// NN, tall A.  Workgroup = 4 waves = 64 rows of A x one 128-column tile of B/C.  K is walked in chunks of 16*S rows of B
// (S = 10: K = 160 in one chunk, 80 KB of LDS).  LDS image of the chunk: plain [k][128] floats; a fragment read is
// lane (j, kk) -> row 16s + 4kk + e, columns 4j..4j+3 (+64): within each 16-lane group of a ds_read_b128 the lanes of one
// kk cover 256 contiguous bytes and rows are 512 B apart, i.e. all 64 banks once -- conflict-free without padding.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_fc_nn_lds(const FcParams p) {
    extern __shared__ f32x4 lds4[];
    MCN_TICK_PROLOGUE(p);
    float* lds = reinterpret_cast<float*>(lds4);
    constexpr int KC = 16 * S, NLD = KC * 32 / 256;  // staging loads (16 B) per thread and chunk
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int m0 = blockIdx.x * 64 + 16 * wave, n0 = blockIdx.y * 128;
    const float* arow = p.A + (size_t)min(m0 + j, p.M - 1) * p.K + 4 * kk;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = F4ZERO;
    float biasv[4];  // requested first: the latency hides behind the whole k loop instead of stalling the epilogue
#pragma unroll
    for (int r = 0; r < 4; r++) biasv[r] = p.bias ? p.bias[min(m0 + 4 * kk + r, p.M - 1)] : 0.f;

    for (int kc0 = 0; kc0 < p.K; kc0 += KC) {
        // this wave's A fragments of the chunk and this thread's share of the B chunk: all requested at once
        f32x4 a[S], st[NLD];
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int k = kc0 + 16 * s + 4 * kk;
            a[s] = k < p.K ? ldg4(arow + kc0 + 16 * s) : F4ZERO;  // K % 4 == 0
        }
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 5, c4 = idx & 31;
            const int k = kc0 + row, n = n0 + 4 * c4;
            st[i] = (k < p.K && n < p.N) ? ldg4(p.B + (size_t)k * p.N + n) : F4ZERO;  // N % 4 == 0
        }
        if (kc0 > 0) __syncthreads();  // the previous chunk's fragments have been read
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            *reinterpret_cast<f32x4*>(lds + 4 * idx) = st[i];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; s++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float* br = lds + (16 * s + 4 * kk + e) * 128 + 4 * j;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(br), b1 = *reinterpret_cast<const f32x4*>(br + 64);
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = mfma16(a[s][e], b0[t], acc[t]);
#pragma unroll
                for (int t = 0; t < 4; t++) acc[4 + t] = mfma16(a[s][e], b1[t], acc[4 + t]);
            }
    }

    // D row 4kk + r <-> row m0 + 4kk + r ; tile (q, t), lane j <-> column n0 + 64q + 4j + t
    f32x4 v[2][4];
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + 4 * kk + r, n = n0 + 64 * q + 4 * j;
            f32x4 o = p.alpha * (f32x4){acc[4 * q][r], acc[4 * q + 1][r], acc[4 * q + 2][r], acc[4 * q + 3][r]};
            if (m < p.M && n < p.N) {
                float* c = p.C + (size_t)m * p.N + n;
                if (p.beta != 0.f) o += p.beta * ldg4(c);
                o += biasv[r];
                if (p.neuron != MCN_NEURON_IDENT) {
#pragma unroll
                    for (int t = 0; t < 4; t++) o[t] = neuron_apply(p.neuron, o[t], p.na, p.nb);
                }
                *reinterpret_cast<f32x4*>(c) = o;
            } else {
                o = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // not a row / column of C: leaves the softmax partial alone
            }
            v[q][r] = o;
        }
    if (!p.smPart) return;

    // softmax stage 1 over this block's 64 rows: per column max, then sum exp(x - max); fixed order
    __syncthreads();  // everyone is done with the B chunk
    float* shm = lds;             // [4 waves][128]
    float* shs = lds + 4 * 128;   // [4 waves][128]
    f32x4 mx[2], sm[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        f32x4 m4 = v[q][0];
#pragma unroll
        for (int r = 1; r < 4; r++)
#pragma unroll
            for (int t = 0; t < 4; t++) m4[t] = fmaxf(m4[t], v[q][r][t]);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            m4[t] = fmaxf(m4[t], __shfl_xor(m4[t], 16));
            m4[t] = fmaxf(m4[t], __shfl_xor(m4[t], 32));
        }
        if (kk == 0) *reinterpret_cast<f32x4*>(shm + wave * 128 + 64 * q + 4 * j) = m4;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const float* c = shm + 64 * q + 4 * j;
        f32x4 m4 = ldg4(c);
#pragma unroll
        for (int w = 1; w < 4; w++) {
            const f32x4 o = ldg4(c + 128 * w);
#pragma unroll
            for (int t = 0; t < 4; t++) m4[t] = fmaxf(m4[t], o[t]);
        }
        mx[q] = m4;
        f32x4 s4 = F4ZERO;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int t = 0; t < 4; t++) s4[t] += v[q][r][t] == -INFINITY ? 0.f : __expf(v[q][r][t] - m4[t]);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            s4[t] += __shfl_xor(s4[t], 16);
            s4[t] += __shfl_xor(s4[t], 32);
        }
        sm[q] = s4;
        if (kk == 0) *reinterpret_cast<f32x4*>(shs + wave * 128 + 64 * q + 4 * j) = s4;
    }
    __syncthreads();
    if (wave == 0 && kk == 0) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float* c = shs + 64 * q + 4 * j;
            const f32x4 s4 = (ldg4(c) + ldg4(c + 128)) + (ldg4(c + 256) + ldg4(c + 384));
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int n = n0 + 64 * q + 4 * j + t;
                if (n < p.N) {
                    float* o = p.smPart + ((size_t)blockIdx.x * p.N + n) * 2;
                    o[0] = mx[q][t];
                    o[1] = s4[t];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// NT, tall A:  C[M x I] = alpha A[M x K] B[I x K]^T + beta C  (dW = dY X^T; K = cases).  Workgroup = 4 waves = 64 rows of
// A x one tile of NT*16 columns (rows of B).  The B tile is staged in LDS as [i][32 x 16-byte chunks] (K chunk of 128),
// chunk c of row i at slot c ^ h(i), h(i) = ((i >> 2) ^ (i << 2)) & 15: a fragment read is lane (j, kk) -> row i(j),
// chunk 4s + kk, and h spreads the 16 rows of a tile over the 16 bank groups for both column orders used below.
// Column order: tiles come in groups of four, tile u of group g <-> columns 64g + 4j + u, so that a lane holds four
// consecutive columns of a row and the epilogue moves 16 bytes per lane; left-over tiles <-> columns base + 16u + j.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int nt_swz(int i) { return ((i >> 2) ^ (i << 2)) & 15; }

template <int NT>
__global__ __launch_bounds__(256) void k_fc_nt_lds(const FcParams p) {
    extern __shared__ f32x4 lds4[];
    float* lds = reinterpret_cast<float*>(lds4);
    constexpr int NC = 16 * NT, G = NT / 4, NLD = NC * 32 / 256;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    const int m0 = blockIdx.x * 64 + 16 * wave, i0 = blockIdx.y * NC;
    const float* arow = p.A + (size_t)min(m0 + j, p.M - 1) * p.K + 4 * kk;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = F4ZERO;
    int irow[NT], isw[NT];  // LDS row of this lane's B fragment per tile, and its swizzle
#pragma unroll
    for (int t = 0; t < NT; t++) {
        irow[t] = t < 4 * G ? 64 * (t / 4) + 4 * j + (t % 4) : 64 * G + 16 * (t - 4 * G) + j;
        isw[t] = nt_swz(irow[t]);
    }

    // beta != 0 (momentum: dW = mom * dW + ...): the C tile is requested before the k loop, not after it
    f32x4 cpre[G > 0 ? G : 1][4];
    float cpre1[NT - 4 * G > 0 ? NT - 4 * G : 1][4];
    if (p.beta != 0.f) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float* crow = p.C + (size_t)min(m0 + 4 * kk + r, p.M - 1) * p.N + i0;
#pragma unroll
            for (int g = 0; g < G; g++) cpre[g][r] = i0 + 64 * g + 4 * j < p.N ? ldg4(crow + 64 * g + 4 * j) : F4ZERO;
#pragma unroll
            for (int t = 4 * G; t < NT; t++) cpre1[t - 4 * G][r] = i0 + irow[t] < p.N ? crow[irow[t]] : 0.f;
        }
    }

    for (int kc0 = 0; kc0 < p.K; kc0 += 128) {
        f32x4 a[8], st[NLD];
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int k = kc0 + 16 * s + 4 * kk;
            a[s] = k < p.K ? ldg4(arow + kc0 + 16 * s) : F4ZERO;  // K % 4 == 0
        }
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 5, c = idx & 31;
            const int gi = i0 + row, k = kc0 + 4 * c;
            st[i] = (gi < p.N && k < p.K) ? ldg4(p.B + (size_t)gi * p.K + k) : F4ZERO;
        }
        if (kc0 > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 5, c = idx & 31;
            *reinterpret_cast<f32x4*>(lds + (row * 32 + (c ^ nt_swz(row))) * 4) = st[i];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; s++) {
            f32x4 b[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) b[t] = *reinterpret_cast<const f32x4*>(lds + (irow[t] * 32 + ((4 * s + kk) ^ isw[t])) * 4);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = mfma16(a[s][e], b[t][e], acc[t]);  // NT independent chains in a row
        }
    }

#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = m0 + 4 * kk + r;
        if (m >= p.M) continue;
        float* crow = p.C + (size_t)m * p.N + i0;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int i = 64 * g + 4 * j;
            if (i0 + i >= p.N) continue;  // N % 4 == 0
            f32x4 o = p.alpha * (f32x4){acc[4 * g][r], acc[4 * g + 1][r], acc[4 * g + 2][r], acc[4 * g + 3][r]};
            if (p.beta != 0.f) o += p.beta * cpre[g][r];
            *reinterpret_cast<f32x4*>(crow + i) = o;
        }
#pragma unroll
        for (int t = 4 * G; t < NT; t++) {
            const int i = irow[t];
            if (i0 + i >= p.N) continue;
            float o = p.alpha * acc[t][r];
            if (p.beta != 0.f) o += p.beta * cpre1[t - 4 * G][r];
            crow[i] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// TN, long K:  C[M x N] = alpha A[K x M]^T B[K x N] + beta C  (dX = W^T dY; K = thousands of classes).  Workgroup =
// 4 waves owns an (MG*16 rows) x 64 columns tile of C for one K chunk; its waves take the chunk's k-steps (4 rows of A
// and B each) round-robin in batches of U, all loads of a batch in flight before its MFMAs and the next batch
// prefetched.  Row order: lane (j, kk) reads 4 consecutive m of row k0 + kk -> component u feeds tile u, whose row j is
// m = 64h + 4j + u (h-th group of 64 rows); MG % 4 left-over tiles read one m per lane (m = base + 16u + j).
// The waves' sums are combined through LDS in a fixed order; slab z of `part` gets the chunk's tile, and
// k_fc_tn_reduce adds the slabs in index order.
// ---------------------------------------------------------------------------------------------
template <int MG, int U>
__global__ __launch_bounds__(256) void k_fc_tn_split(const FcParams p) {
    extern __shared__ f32x4 red[];
    MCN_TICK_PROLOGUE(p);
    constexpr int H = MG / 4, R = MG % 4, T = MG * 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    // the tiles of one K chunk read the same rows of A and B: one XCD per chunk (mcn_xcd_group; fc6: 37 -> 23 MB per launch)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcdGroup) {
        int t;
        mcn_xcd_group((int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)), (int)(gridDim.x * gridDim.y), (int)gridDim.z, bz, t);
        by = t / (int)gridDim.x;
        bx = t - by * (int)gridDim.x;
    }
    const int mb = by * (16 * MG), nb = bx * 64;
    const int kBeg = bz * p.kChunk, kEnd = min(p.K, kBeg + p.kChunk);
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; t++) acc[t] = F4ZERO;
    const bool nin = nb + 4 * j < p.N;  // N % 4 == 0
    bool min4[H > 0 ? H : 1], min1[R > 0 ? R : 1];
#pragma unroll
    for (int h = 0; h < H; h++) min4[h] = mb + 64 * h + 4 * j < p.M;  // M % 4 == 0
#pragma unroll
    for (int u = 0; u < R; u++) min1[u] = mb + 64 * H + 16 * u + j < p.M;
    const float* ab = p.A + mb;
    const float* bb = p.B + nb + 4 * j;

    f32x4 a40[U][H > 0 ? H : 1], a41[U][H > 0 ? H : 1], b0[U], b1[U];
    float a10[U][R > 0 ? R : 1], a11[U][R > 0 ? R : 1];
    auto load = [&](int kb, f32x4(&a4)[U][H > 0 ? H : 1], float(&a1)[U][R > 0 ? R : 1], f32x4(&b)[U]) {
#pragma unroll
        for (int s = 0; s < U; s++) {
            const int k = kb + 4 * s + kk;
            const bool kin = k < kEnd;
            const size_t kr = (size_t)min(k, p.K - 1);
#pragma unroll
            for (int h = 0; h < H; h++) a4[s][h] = (kin && min4[h]) ? ldg4(ab + kr * p.M + 64 * h + 4 * j) : F4ZERO;
#pragma unroll
            for (int u = 0; u < R; u++) a1[s][u] = (kin && min1[u]) ? ab[kr * p.M + 64 * H + 16 * u + j] : 0.f;
            b[s] = (kin && nin) ? ldg4(bb + kr * p.N) : F4ZERO;
        }
    };
    auto compute = [&](const f32x4(&a4)[U][H > 0 ? H : 1], const float(&a1)[U][R > 0 ? R : 1], const f32x4(&b)[U]) {
#pragma unroll
        for (int s = 0; s < U; s++) {
#pragma unroll
            for (int h = 0; h < H; h++)
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[(4 * h + u) * 4 + t] = mfma16(a4[s][h][u], b[s][t], acc[(4 * h + u) * 4 + t]);
#pragma unroll
            for (int u = 0; u < R; u++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[(4 * H + u) * 4 + t] = mfma16(a1[s][u], b[s][t], acc[(4 * H + u) * 4 + t]);
        }
    };
    constexpr int KB = 4 * U;
    const int kStep = KB * 4;
    const int kFirst = kBeg + wave * KB;
    if (kFirst < kEnd) load(kFirst, a40, a10, b0);
    for (int kb = kFirst; kb < kEnd; kb += 2 * kStep) {
        const bool more1 = kb + kStep < kEnd, more2 = kb + 2 * kStep < kEnd;
        if (more1) load(kb + kStep, a41, a11, b1);
        compute(a40, a10, b0);
        if (more1) {
            if (more2) load(kb + 2 * kStep, a40, a10, b0);
            compute(a41, a11, b1);
        }
    }

    // fixed halving tree over the four waves, totals published so that every wave stores a share of the tile
    for (int hh = 2; hh >= 1; hh >>= 1) {
        __syncthreads();
        if (wave >= hh && wave < 2 * hh) {
#pragma unroll
            for (int t = 0; t < T; t++) red[((wave - hh) * T + t) * 64 + lane] = acc[t];
        }
        __syncthreads();
        if (wave < hh) {
#pragma unroll
            for (int t = 0; t < T; t++) acc[t] += red[(wave * T + t) * 64 + lane];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < T; t++) red[t * 64 + lane] = acc[t];
    }
    __syncthreads();
    float* slab = p.part + (size_t)bz * p.M * p.N;
    if (!nin) return;
    // tile (mt, t): D row 4kk + r <-> lane-row index jA = 4kk + r of the A operand; column j <-> n = nb + 4j + t
#pragma unroll
    for (int mt = 0; mt < MG; mt++) {
        if ((mt & 3) != wave) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int jA = 4 * kk + r;
            const int m = mt < 4 * H ? mb + 64 * (mt / 4) + 4 * jA + (mt % 4) : mb + 64 * H + 16 * (mt - 4 * H) + jA;
            if (m >= p.M) continue;
            const f32x4 o = {red[(mt * 4 + 0) * 64 + lane][r], red[(mt * 4 + 1) * 64 + lane][r], red[(mt * 4 + 2) * 64 + lane][r],
                             red[(mt * 4 + 3) * 64 + lane][r]};
            *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + nb + 4 * j) = o;
        }
    }
}

// slabs [splits][M*N] -> C = alpha * sum + beta * C.  The classifier's gradient is 160 x 128: one thread per 16-byte column
// group would be 20 blocks and eight dependent round trips (5.6 us of pure latency).  Eight threads share a column group
// instead: thread (o, g) adds the contiguous slab range g in index order with all its loads in flight, the eight ranges
// meet in LDS and are added in a fixed tree -- 160 blocks, one round trip for up to 64 slabs.
__global__ __launch_bounds__(256) void k_fc_tn_reduce(const FcParams p) {
    __shared__ f32x4 sh[8][32];
    const size_t total4 = (size_t)p.M * p.N / 4;
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + o;
    const bool live = i < total4;
    const int per = (p.splits + 7) / 8, z0 = g * per, z1 = min(p.splits, z0 + per);
    const f32x4* src = reinterpret_cast<const f32x4*>(p.part) + (live ? i : 0);
    f32x4 s = F4ZERO;
    int z = z0;
    for (; z + 7 < z1; z += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = src[(size_t)(z + u) * total4];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    if (z < z1) {  // (the last range of 61 slabs holds 5)
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = z + u < z1 ? src[(size_t)(z + u) * total4] : F4ZERO;
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    sh[g][o] = s;
    __syncthreads();
    if (g != 0 || !live) return;
    s = ((sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o])) + ((sh[4][o] + sh[5][o]) + (sh[6][o] + sh[7][o]));
    f32x4* c = reinterpret_cast<f32x4*>(p.C) + i;
    s = p.alpha * s;
    if (p.beta != 0.f) s += p.beta * *c;
    *c = s;
}

// ---------------------------------------------------------------------------------------------
// host side: which shapes take these forms (everything else stays with gemm.hip)
// ---------------------------------------------------------------------------------------------
static inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

bool fc_nn_lds_usable(const float* A, const float* B, const float* C, int M, int N, int K) {
    return M >= 512 && N % 4 == 0 && K % 4 == 0 && K <= 4096 && al16(A) && al16(B) && al16(C);
}
bool fc_nt_lds_usable(const float* A, const float* B, const float* C, int M, int N, int K) {
    return M >= 512 && N % 16 == 0 && N >= 64 && K % 4 == 0 && K <= 1024 && al16(A) && al16(B) && al16(C);
}
// TN with a long reduction and a small output
struct TnPlan { int mg, splits, kChunk; };
static TnPlan tn_plan(int M, int N, int K) {
    TnPlan pl;
    pl.mg = (M % 80 == 0 || M <= 80) && M > 64 ? 5 : 4;
    const long tiles = (long)divup(M, 16 * pl.mg) * divup(N, 64);
    static const long target = getenv("MCN_TN_BLOCKS") ? atol(getenv("MCN_TN_BLOCKS")) : 256;
    long s = std::max<long>(1, target / tiles);
    s = std::min<long>(s, std::max(1, K / 64));
    pl.kChunk = divup(divup(K, (int)s), 16) * 16;
    pl.splits = divup(K, pl.kChunk);
    return pl;
}
bool fc_tn_split_usable(const float* A, const float* B, const float* C, int M, int N, int K) {
    return K >= 2048 && (long)M * N <= 512 * 512 && M % 4 == 0 && N % 4 == 0 && al16(A) && al16(B) && al16(C);
}
size_t fc_tn_split_workspace_bytes(int M, int N, int K) { return (size_t)tn_plan(M, N, K).splits * M * N * sizeof(float); }

static FcParams fc_params(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, float alpha, float beta,
                          int neuron, float na, float nb) {
    FcParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.smPart = nullptr; p.part = nullptr;
    p.M = M; p.N = N; p.K = K; p.kChunk = K; p.splits = 1;
    p.neuron = neuron; p.alpha = alpha; p.beta = beta; p.na = na; p.nb = nb;
    p.tick = {nullptr, 0};
    p.xcdGroup = 0;
    return p;
}

int fc_nn_lds_launch(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, float alpha, float beta,
                     int neuron, float na, float nb, float* smPart, hipStream_t s) {
    FcParams p = fc_params(A, B, bias, C, M, N, K, alpha, beta, neuron, na, nb);
    p.smPart = smPart;
    dim3 grid(divup(M, 64), divup(N, 128));
    MCN_REQUIRE(grid.y <= 65535, "gemm: grid too large");
    p.tick = mcn_take_tick(s);  // (after the last check: a tick taken is a tick sent)
    // 160-row chunks when they waste nothing (K = 160), else 128
    if (K % 160 == 0 || (K < 160 && K > 128)) MCN_LAUNCH((k_fc_nn_lds<10>), grid, dim3(256), 160 * 128 * 4, s, p);
    else MCN_LAUNCH((k_fc_nn_lds<8>), grid, dim3(256), 128 * 128 * 4, s, p);
    return MCN_LAUNCH_CHECK("gemm (fc nn)");
}

int fc_nt_lds_launch(const float* A, const float* B, float* C, int M, int N, int K, float alpha, float beta, hipStream_t s) {
    FcParams p = fc_params(A, B, nullptr, C, M, N, K, alpha, beta, MCN_NEURON_IDENT, 0.f, 0.f);
    // column tile: 160 when it divides (fc6: 160 inputs), else 128 (DeepID2+: 512 inputs), else 64
    const int nt = N % 160 == 0 ? 10 : (N % 128 == 0 || N > 128) ? 8 : 4;
    dim3 grid(divup(M, 64), divup(N, 16 * nt));
    MCN_REQUIRE(grid.y <= 65535, "gemm: grid too large");
    const size_t lds = (size_t)16 * nt * 128 * 4;
    if (nt == 10) MCN_LAUNCH((k_fc_nt_lds<10>), grid, dim3(256), lds, s, p);
    else if (nt == 8) MCN_LAUNCH((k_fc_nt_lds<8>), grid, dim3(256), lds, s, p);
    else MCN_LAUNCH((k_fc_nt_lds<4>), grid, dim3(256), lds, s, p);
    return MCN_LAUNCH_CHECK("gemm (fc nt)");
}

int fc_tn_split_launch(const float* A, const float* B, float* C, int M, int N, int K, float alpha, float beta, void* workspace,
                       size_t workspaceBytes, hipStream_t s, const FcL2snAddend* l2) {
    const TnPlan pl = tn_plan(M, N, K);
    const size_t need = (size_t)pl.splits * M * N * sizeof(float);
    if (!workspace || workspaceBytes < need || !al16(workspace)) {
        mcn_set_error("gemm: workspace %zu < %zu bytes", workspaceBytes, need);
        return MCN_EWORKSPACE;
    }
    FcParams p = fc_params(A, B, nullptr, C, M, N, K, alpha, beta, MCN_NEURON_IDENT, 0.f, 0.f);
    p.part = (float*)workspace;
    p.kChunk = pl.kChunk;
    p.splits = pl.splits;
    static const int xcdGroupEnv = getenv("MCN_TN_XCD_GROUP") ? atoi(getenv("MCN_TN_XCD_GROUP")) : 1;
    p.xcdGroup = xcdGroupEnv;
    dim3 grid(divup(N, 64), divup(M, 16 * pl.mg), pl.splits);
    MCN_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "gemm: grid too large");
    p.tick = mcn_take_tick(s);
    if (pl.mg == 5) MCN_LAUNCH((k_fc_tn_split<5, 6>), grid, dim3(256), 2 * 20 * 64 * 16, s, p);
    else MCN_LAUNCH((k_fc_tn_split<4, 6>), grid, dim3(256), 2 * 16 * 64 * 16, s, p);
    int rc = MCN_LAUNCH_CHECK("gemm (fc tn)");
    if (rc) return rc;
    // (the reduce pass that also adds the L2-SN gradient lives in eltwise_cost.hip: that file is built without floating-point
    // contraction, and the gradient has to keep the bits of k_l2sn_grad_stats)
    if (l2) return fc_tn_reduce_l2sn_launch(p.part, C, M, N, pl.splits, *l2, s);
    MCN_LAUNCH(k_fc_tn_reduce, dim3((unsigned)divup_sz((size_t)M * N / 4, 32)), dim3(256), 0, s, p);
    return MCN_LAUNCH_CHECK("gemm (fc tn reduce)");
}

// =============================================================================================
// A hidden FC layer with several inputs (DeepID's fc5 reads local4 AND pool3, layer.cu:355-366 loops over the inputs
// with one cublasSgemm each): small outputs (160 x 128 cases), long reductions, launches that are pure latency.
// One launch per direction for all inputs.
// =============================================================================================
#define FCM_MAXSEG 4
struct FcMultiParams {
    const float* A[FCM_MAXSEG];   // forward: W_i [M x K_i];  backward: W_i [K x M_i] (the same memory, read the other way)
    const float* B[FCM_MAXSEG];   // forward: X_i [K_i x N]
    float* C[FCM_MAXSEG];         // backward: dX_i [M_i x N]
    const float* nacts[FCM_MAXSEG];  // backward: output of the neuron that produced X_i (gradient applied in the epilogue), or null
    int K[FCM_MAXSEG];            // forward: K_i;  backward: M_i
    int steps[FCM_MAXSEG];        // forward: K_i / 16;  backward: 32-row tiles of segment i
    float beta[FCM_MAXSEG];
    int ntype[FCM_MAXSEG];
    float na[FCM_MAXSEG], nb[FCM_MAXSEG];
    int nseg, totalSteps;
    const float* bias;
    const float* dY;              // backward: [K x N]
    float* Y;                     // forward: [M x N]
    int M, N, Kb;                 // forward: M, N;  backward: Kb = K (outputs of the layer), N
    int neuron;
    float fa, fb, fbeta;
    McnTick tick;                 // backward-data: stored by the launch's first thread (MCN_TICK_PROLOGUE)
    int xcdGroup;                 // forward / backward-data: the column tiles of one row tile on one XCD (mcn_xcd_group)
};

// ---- forward: Y[M x N] = sum_i W_i X_i (+ bias, neuron).  One workgroup of 8 waves per 16 x 16 tile of Y; the waves split the
// concatenated reduction (every K_i a multiple of 16) into 8 contiguous ranges and request their fragments in batches of U
// 16-deep steps (next batch in flight while the current one multiplies); fixed-order tree over the waves through LDS.
template <int U>
__global__ __launch_bounds__(512) void k_fc_fprop_multi(const FcMultiParams p) {
    __shared__ f32x4 red[4 * 64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kk = lane >> 4, j = lane & 15;
    // the column tiles of a row tile read the same 16 weight rows -- the large operand: one XCD per row tile
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcdGroup) mcn_xcd_group((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.x, (int)gridDim.y, by, bx);
    const int m0 = by * 16, n0 = bx * 16;
    const int per = (p.totalSteps + 7) / 8;
    const int sBeg = wave * per, sEnd = min(p.totalSteps, sBeg + per);
    const int mrow = min(m0 + j, p.M - 1);
    const bool nin = n0 + j < p.N;
    const int ncol = min(n0 + j, p.N - 1);
    f32x4 acc = F4ZERO;
    f32x4 a0[U], a1[U];
    float b0[U][4], b1[U][4];
    auto load = [&](int s0, f32x4(&a)[U], float(&b)[U][4]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = s0 + u;
            if (s < sEnd) {  // wave-uniform
                int seg = 0, ks = s;
                while (ks >= p.steps[seg]) { ks -= p.steps[seg]; seg++; }
                const float* ap = p.A[seg] + (size_t)mrow * p.K[seg] + 16 * ks + 4 * kk;
                a[u] = ldg4(ap);
                const float* bp = p.B[seg] + (size_t)(16 * ks + 4 * kk) * p.N + ncol;
#pragma unroll
                for (int e = 0; e < 4; e++) b[u][e] = nin ? bp[(size_t)e * p.N] : 0.f;
            } else {
                a[u] = F4ZERO;
#pragma unroll
                for (int e = 0; e < 4; e++) b[u][e] = 0.f;
            }
        }
    };
    auto compute = [&](const f32x4(&a)[U], const float(&b)[U][4]) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc = mfma16(a[u][e], b[u][e], acc);
    };
    if (sBeg < sEnd) load(sBeg, a0, b0);
    for (int s = sBeg; s < sEnd; s += 2 * U) {
        const bool more1 = s + U < sEnd, more2 = s + 2 * U < sEnd;
        if (more1) load(s + U, a1, b1);
        compute(a0, b0);
        if (more1) {
            if (more2) load(s + 2 * U, a0, b0);
            compute(a1, b1);
        }
    }
    // fixed halving tree over the 8 waves
    for (int hh = 4; hh >= 1; hh >>= 1) {
        __syncthreads();
        if (wave >= hh && wave < 2 * hh) red[(wave - hh) * 64 + lane] = acc;
        __syncthreads();
        if (wave < hh) acc += red[wave * 64 + lane];
    }
    if (wave != 0 || !nin) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = m0 + 4 * kk + r;
        if (m >= p.M) continue;
        float* c = p.Y + (size_t)m * p.N + n0 + j;
        float o = acc[r];
        if (p.fbeta != 0.f) o += p.fbeta * *c;
        if (p.bias) o += p.bias[m];
        if (p.neuron != MCN_NEURON_IDENT) o = neuron_apply(p.neuron, o, p.fa, p.fb);
        *c = o;
    }
}

// ---- backward-data: dX_i[M_i x N] = beta_i dX_i + W_i^T dY  (K = outputs of the layer, short: 160).  One WAVE per
// 32 x 32 tile of one dX_i: lane (j, kk) reads 2 consecutive m of W row k0 + kk and 2 consecutive n of dY row k0 + kk
// (component u / t feeds tile (u, t): rows m0 + 2j + u, columns n0 + 2j + t), ALL k-steps requested up front in batches of
// U, no LDS, no barrier.  Optional epilogue: the gradient of the neuron that produced X_i, from its output.
template <int U>
__global__ __launch_bounds__(64) void k_fc_bprop_multi(const FcMultiParams p) {
    MCN_TICK_PROLOGUE(p);
    const int lane = threadIdx.x, kk = lane >> 4, j = lane & 15;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcdGroup) mcn_xcd_group((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)gridDim.x, (int)gridDim.y, by, bx);  // (as in the forward kernel)
    int seg = 0, tile = by;
    while (tile >= p.steps[seg]) { tile -= p.steps[seg]; seg++; }
    const int M = p.K[seg], m0 = tile * 32, n0 = bx * 32;
    const float* A = p.A[seg];
    const bool min2 = m0 + 2 * j < M, nin2 = n0 + 2 * j < p.N;  // M, N even
    const float* ab = A + min(m0 + 2 * j, M - 2);
    const float* bb = p.dY + min(n0 + 2 * j, p.N - 2);
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = F4ZERO;
    // epilogue operands first
    const float beta = p.beta[seg];
    f32x2 cold[2][4], nact[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + 2 * (4 * kk + r) + u;
            const size_t off = (size_t)min(m, M - 1) * p.N + min(n0 + 2 * j, p.N - 2);
            cold[u][r] = beta != 0.f ? *reinterpret_cast<const f32x2*>(p.C[seg] + off) : (f32x2){0.f, 0.f};
            nact[u][r] = p.nacts[seg] ? *reinterpret_cast<const f32x2*>(p.nacts[seg] + off) : (f32x2){1.f, 1.f};
        }
    const int ksteps = (p.Kb + 3) / 4;
    f32x2 a0[U], a1[U], b0[U], b1[U];
    auto load = [&](int s0, f32x2(&a)[U], f32x2(&b)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = 4 * (s0 + u) + kk;
            const bool kin = k < p.Kb;
            const size_t kr = (size_t)min(k, p.Kb - 1);
            a[u] = (kin && min2) ? *reinterpret_cast<const f32x2*>(ab + kr * M) : (f32x2){0.f, 0.f};
            b[u] = (kin && nin2) ? *reinterpret_cast<const f32x2*>(bb + kr * p.N) : (f32x2){0.f, 0.f};
        }
    };
    auto compute = [&](const f32x2(&a)[U], const f32x2(&b)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc[0] = mfma16(a[u][0], b[u][0], acc[0]);
            acc[1] = mfma16(a[u][0], b[u][1], acc[1]);
            acc[2] = mfma16(a[u][1], b[u][0], acc[2]);
            acc[3] = mfma16(a[u][1], b[u][1], acc[3]);
        }
    };
    load(0, a0, b0);
    for (int s = 0; s < ksteps; s += 2 * U) {
        const bool more1 = s + U < ksteps, more2 = s + 2 * U < ksteps;
        if (more1) load(s + U, a1, b1);
        compute(a0, b0);
        if (more1) {
            if (more2) load(s + 2 * U, a0, b0);
            compute(a1, b1);
        }
    }
    if (!nin2) return;
    const int nt = p.ntype[seg];
    // tile (u, t): D row 4kk + r <-> lane-row jA = 4kk + r of the A operand, i.e. m = m0 + 2 jA + u; column j <-> n0 + 2j + t
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + 2 * (4 * kk + r) + u;
            if (m >= M) continue;
            f32x2 o = {acc[2 * u][r], acc[2 * u + 1][r]};
            if (p.nacts[seg]) {
#pragma unroll
                for (int t = 0; t < 2; t++) o[t] = neuron_grad(nt, o[t], nact[u][r][t], 0.f, p.na[seg], p.nb[seg]);
            }
            if (beta != 0.f) o += beta * cold[u][r];
            *reinterpret_cast<f32x2*>(p.C[seg] + (size_t)m * p.N + n0 + 2 * j) = o;
        }
}

bool fc_fprop_multi_usable(const float* const* W, const float* const* X, const int* K, int count, const float* Y, int M, int N) {
    if (count < 1 || count > FCM_MAXSEG || M < 1 || N < 1 || (long)divup(M, 16) * divup(N, 16) > 4096) return false;
    long total = 0;
    for (int i = 0; i < count; i++) {
        if (K[i] <= 0 || K[i] % 16 != 0 || !al16(W[i]) || !X[i]) return false;
        total += K[i];
    }
    return total >= 16 && Y != nullptr;
}

int fc_fprop_multi_launch(const float* const* W, const float* const* X, const int* K, int count, const float* bias, float* Y, int M,
                          int N, float beta, int neuron, float na, float nb, hipStream_t s) {
    FcMultiParams p = {};
    static const int fcmXcd = getenv("MCN_FCM_XCD_GROUP") ? atoi(getenv("MCN_FCM_XCD_GROUP")) : 3;  // bit 0: forward, bit 1: backward-data
    p.xcdGroup = fcmXcd & 1;
    p.nseg = count;
    p.totalSteps = 0;
    for (int i = 0; i < count; i++) {
        p.A[i] = W[i]; p.B[i] = X[i]; p.K[i] = K[i]; p.steps[i] = K[i] / 16;
        p.totalSteps += p.steps[i];
    }
    for (int i = count; i < FCM_MAXSEG; i++) p.steps[i] = 1 << 30;
    p.bias = bias; p.Y = Y; p.M = M; p.N = N; p.neuron = neuron; p.fa = na; p.fb = nb; p.fbeta = beta;
    dim3 grid(divup(N, 16), divup(M, 16));
    MCN_REQUIRE(grid.y <= 65535, "fcFpropMulti: grid too large");
    // a wave's whole share in two register sets of 9 steps when it fits (fc5: 139 steps / 8 waves = 18): every load is in
    // flight before the first MFMA, one round trip per wave; longer reductions walk batches of 6
    // (16 waves with ONE register set of 9 steps each -- a single round trip per wave, 84 VGPRs -- measured: fc5 10.2 -> 10.3 us alone,
    // the step unchanged; the second round trip is not what this launch spends its time on.  Not kept.)
    if ((p.totalSteps + 7) / 8 <= 18) MCN_LAUNCH((k_fc_fprop_multi<9>), grid, dim3(512), 0, s, p);
    else MCN_LAUNCH((k_fc_fprop_multi<6>), grid, dim3(512), 0, s, p);
    return MCN_LAUNCH_CHECK("fcFpropMulti");
}

bool fc_bprop_multi_usable(const float* const* W, float* const* dX, const int* Mi, int count, const float* dY, int K, int N) {
    if (count < 1 || count > FCM_MAXSEG || K < 1 || N < 2 || N % 2 != 0 || !dY || ((uintptr_t)dY % 8) != 0) return false;
    long tiles = 0;
    for (int i = 0; i < count; i++) {
        if (Mi[i] < 2 || Mi[i] % 2 != 0 || !W[i] || !dX[i] || ((uintptr_t)W[i] % 8) != 0 || ((uintptr_t)dX[i] % 8) != 0) return false;
        tiles += divup(Mi[i], 32);
    }
    return tiles <= 65535 && K <= 4096;
}

int fc_bprop_multi_launch(const float* const* W, float* const* dX, const int* Mi, const float* beta, const float* const* nacts,
                          const int* ntype, const float* na, const float* nb, int count, const float* dY, int K, int N, hipStream_t s) {
    FcMultiParams p = {};
    static const int fcmXcd = getenv("MCN_FCM_XCD_GROUP") ? atoi(getenv("MCN_FCM_XCD_GROUP")) : 3;
    p.xcdGroup = (fcmXcd >> 1) & 1;
    p.nseg = count;
    int tiles = 0;
    for (int i = 0; i < count; i++) {
        p.A[i] = W[i]; p.C[i] = dX[i]; p.K[i] = Mi[i]; p.steps[i] = divup(Mi[i], 32); p.beta[i] = beta[i];
        p.nacts[i] = nacts ? nacts[i] : nullptr;
        p.ntype[i] = ntype ? ntype[i] : MCN_NEURON_IDENT;
        p.na[i] = na ? na[i] : 0.f;
        p.nb[i] = nb ? nb[i] : 0.f;
        if (p.nacts[i]) MCN_REQUIRE(((uintptr_t)p.nacts[i] % 8) == 0 && neuron_valid(p.ntype[i]), "fcBpropActsMulti: bad neuron operand");
        tiles += p.steps[i];
    }
    for (int i = count; i < FCM_MAXSEG; i++) p.steps[i] = 1 << 30;
    p.dY = dY; p.Kb = K; p.N = N;
    dim3 grid(divup(N, 32), tiles);
    p.tick = mcn_take_tick(s);
    // up to 160 outputs (fc5): the whole reduction in two register sets of 20 steps, every load in flight before the first
    // MFMA (two round trips instead of four: 0.2475 -> 0.2446 ms per DeepID2 step); longer reductions walk batches of 10
    if ((K + 3) / 4 <= 40) MCN_LAUNCH((k_fc_bprop_multi<20>), grid, dim3(64), 0, s, p);
    else MCN_LAUNCH((k_fc_bprop_multi<10>), grid, dim3(64), 0, s, p);
    return MCN_LAUNCH_CHECK("fcBpropActsMulti");
}

// ---- backward-weights: dW_i[M x I_i] = beta_i dW_i + alpha_i dY X_i^T  (reduction over the cases, short: 128).  One WAVE per
// 32 x 32 tile of one dW_i: lane (j, kk) reads 4 consecutive cases of dY rows m0 + j / m0 + 16 + j and of X_i rows
// i0 + 2j / i0 + 2j + 1 (column j of tile t <-> input i0 + 2j + t: a lane stores two consecutive inputs of a row), every
// case step of a batch requested up front, the momentum tile (beta_i dW_i) before them; no LDS, no barrier.
// FcMultiParams: B[i] = X_i, C[i] = dW_i, K[i] = I_i, steps[i] = 32-column tiles of input i, dY, M, N = cases, na[i] = alpha_i.
template <int U>
__global__ __launch_bounds__(64) void k_fc_wgrad_multi(const FcMultiParams p) {
    const int lane = threadIdx.x, kk = lane >> 4, j = lane & 15;
    int seg = 0, tile = blockIdx.y;
    while (tile >= p.steps[seg]) { tile -= p.steps[seg]; seg++; }
    const int I = p.K[seg], i0 = tile * 32, m0 = blockIdx.x * 32;
    const float* X = p.B[seg];
    float* C = p.C[seg];
    const float alpha = p.na[seg], beta = p.beta[seg];
    const bool iin = i0 + 2 * j < I;  // I even
    const float* a0p = p.dY + (size_t)min(m0 + j, p.M - 1) * p.N + 4 * kk;
    const float* a1p = p.dY + (size_t)min(m0 + 16 + j, p.M - 1) * p.N + 4 * kk;
    const float* b0p = X + (size_t)min(i0 + 2 * j, I - 1) * p.N + 4 * kk;
    const float* b1p = X + (size_t)min(i0 + 2 * j + 1, I - 1) * p.N + 4 * kk;
    f32x2 cold[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = min(m0 + 16 * u + 4 * kk + r, p.M - 1);
            cold[u][r] = (beta != 0.f && iin) ? *reinterpret_cast<const f32x2*>(C + (size_t)m * I + i0 + 2 * j) : (f32x2){0.f, 0.f};
        }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = F4ZERO;
    const int steps = (p.N + 15) / 16;  // N % 4 == 0
    for (int s0 = 0; s0 < steps; s0 += U) {
        f32x4 a0[U], a1[U], b0[U], b1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = 16 * (s0 + u) + 4 * kk;
            const bool kin = s0 + u < steps && k < p.N;
            a0[u] = kin ? ldg4(a0p + 16 * (s0 + u)) : F4ZERO;
            a1[u] = kin ? ldg4(a1p + 16 * (s0 + u)) : F4ZERO;
            b0[u] = kin ? ldg4(b0p + 16 * (s0 + u)) : F4ZERO;
            b1[u] = kin ? ldg4(b1p + 16 * (s0 + u)) : F4ZERO;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc[0] = mfma16(a0[u][e], b0[u][e], acc[0]);
                acc[1] = mfma16(a0[u][e], b1[u][e], acc[1]);
                acc[2] = mfma16(a1[u][e], b0[u][e], acc[2]);
                acc[3] = mfma16(a1[u][e], b1[u][e], acc[3]);
            }
    }
    if (!iin) return;
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + 16 * u + 4 * kk + r;
            if (m >= p.M) continue;
            f32x2 o = alpha * (f32x2){acc[2 * u][r], acc[2 * u + 1][r]};
            if (beta != 0.f) o += beta * cold[u][r];
            *reinterpret_cast<f32x2*>(C + (size_t)m * I + i0 + 2 * j) = o;
        }
}

bool fc_wgrad_multi_usable(const float* dY, const float* const* X, float* const* dW, const int* Ii, int count, int M, int N) {
    if (count < 1 || count > FCM_MAXSEG || M < 1 || N < 4 || N % 4 != 0 || !al16(dY)) return false;
    long tiles = 0;
    for (int i = 0; i < count; i++) {
        if (Ii[i] < 2 || Ii[i] % 2 != 0 || !al16(X[i]) || !dW[i] || ((uintptr_t)dW[i] % 8) != 0) return false;
        tiles += divup(Ii[i], 32);
    }
    return tiles <= 65535 && divup(M, 32) <= 65535;
}

int fc_wgrad_multi_launch(const float* dY, const float* const* X, float* const* dW, const int* Ii, const float* alpha, const float* beta,
                          int count, int M, int N, hipStream_t s) {
    FcMultiParams p = {};
    p.nseg = count;
    int tiles = 0;
    for (int i = 0; i < count; i++) {
        p.B[i] = X[i]; p.C[i] = dW[i]; p.K[i] = Ii[i]; p.steps[i] = divup(Ii[i], 32); p.beta[i] = beta[i]; p.na[i] = alpha[i];
        tiles += p.steps[i];
    }
    for (int i = count; i < FCM_MAXSEG; i++) p.steps[i] = 1 << 30;
    p.dY = dY; p.M = M; p.N = N;
    dim3 grid(divup(M, 32), tiles);
    MCN_LAUNCH((k_fc_wgrad_multi<8>), grid, dim3(64), 0, s, p);
    return MCN_LAUNCH_CHECK("fcBpropWeightsMulti");
}

MCN_API int mcn_fc_bprop_weights_multi_supported(const float* actsGrad, const float* const* inputs, float* const* targets,
                                                 const int* numInputs, int count, int numOutputs, int numCases) {
    return (actsGrad && inputs && targets && numInputs &&
            fc_wgrad_multi_usable(actsGrad, inputs, targets, numInputs, count, numOutputs, numCases)) ? 1 : 0;
}
MCN_API int mcn_fc_bprop_weights_multi(const float* actsGrad, const float* const* inputs, float* const* targets, const int* numInputs,
                                       const float* scaleOutput, const float* scaleTargets, int count, int numOutputs, int numCases,
                                       void* stream) {
    MCN_REQUIRE(actsGrad && inputs && targets && numInputs && scaleOutput && scaleTargets, "fcBpropWeightsMulti: null pointer");
    MCN_REQUIRE(fc_wgrad_multi_usable(actsGrad, inputs, targets, numInputs, count, numOutputs, numCases),
                "fcBpropWeightsMulti: unsupported shape (even input sizes, case count a multiple of 4, 16-byte aligned operands, at most %d inputs)",
                FCM_MAXSEG);
    return fc_wgrad_multi_launch(actsGrad, inputs, targets, numInputs, scaleOutput, scaleTargets, count, numOutputs, numCases,
                                 (hipStream_t)stream);
}
MCN_API int mcn_fc_fprop_multi_supported(const float* const* weights, const float* const* inputs, const int* numInputs, int count,
                                         const float* acts, int numOutputs, int numCases) {
    return (weights && inputs && numInputs && fc_fprop_multi_usable(weights, inputs, numInputs, count, acts, numOutputs, numCases)) ? 1 : 0;
}
MCN_API int mcn_fc_fprop_multi(const float* const* weights, const float* const* inputs, const int* numInputs, int count,
                               const float* biases, float* acts, int numOutputs, int numCases, int neuron, float neuronA, float neuronB,
                               void* stream) {
    MCN_REQUIRE(weights && inputs && numInputs && acts, "fcFpropMulti: null pointer");
    MCN_REQUIRE(neuron_valid(neuron), "fcFpropMulti: bad neuron type %d", neuron);
    MCN_REQUIRE(fc_fprop_multi_usable(weights, inputs, numInputs, count, acts, numOutputs, numCases),
                "fcFpropMulti: unsupported shape (every input size must be a multiple of 16, at most %d inputs)", FCM_MAXSEG);
    return fc_fprop_multi_launch(weights, inputs, numInputs, count, biases, acts, numOutputs, numCases, 0.f, neuron, neuronA, neuronB,
                                 (hipStream_t)stream);
}
MCN_API int mcn_fc_bprop_acts_multi_supported(const float* const* weights, float* const* targets, const int* numInputs, int count,
                                              const float* actsGrad, int numOutputs, int numCases) {
    return (weights && targets && numInputs && fc_bprop_multi_usable(weights, targets, numInputs, count, actsGrad, numOutputs, numCases)) ? 1 : 0;
}
MCN_API int mcn_fc_bprop_acts_multi(const float* const* weights, float* const* targets, const int* numInputs, const float* scaleTargets,
                                    const float* const* neuronActs, const int* neuronTypes, const float* neuronA, const float* neuronB,
                                    int count, const float* actsGrad, int numOutputs, int numCases, void* stream) {
    MCN_REQUIRE(weights && targets && numInputs && scaleTargets && actsGrad, "fcBpropActsMulti: null pointer");
    MCN_REQUIRE(fc_bprop_multi_usable(weights, targets, numInputs, count, actsGrad, numOutputs, numCases),
                "fcBpropActsMulti: unsupported shape (even input sizes and case count, at most %d inputs)", FCM_MAXSEG);
    return fc_bprop_multi_launch(weights, targets, numInputs, scaleTargets, neuronActs, neuronTypes, neuronA, neuronB, count, actsGrad,
                                 numOutputs, numCases, (hipStream_t)stream);
}
