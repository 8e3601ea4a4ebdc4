// TEST INFRASTRUCTURE ONLY -- part of the CPU oracle (see convnet_oracle.cpp header;
// PARITY UNPINNED).  This translation unit is compiled with -ffp-contract=off:
// computeL2SNCost / computeL2SNGrad run every NVMatrix op as its own kernel
// (layer_kernels.cu:505-735), so no multiply-add is ever fused in the reference,
// and the data-dependent margin makes the result discontinuous in the last bit.
#include <cmath>
#include <cstddef>
#include <vector>

#define ORC extern "C" __attribute__((visibility("default")))

// Pair distances shared by L2-SN cost and grad (layer_kernels.cu:517-561):
// gallery = cols [0,N/2), probe = cols [N/2,N); s_j = +1 if ids equal else -1;
// d_j = sqrt(sum_k (g-p)^2) + 1e-8.  The 'k' sum is a sequential fmaf chain in
// row order with separately rounded squares and adds -- the order of the reference's
// SquaredDiff pass followed by a column sum (kDumbAggCols walks the rows serially);
// the HIP kernel uses the same order so that d (and therefore the data-dependent
// margin m and every 1[d<m] decision) is bit-identical.
static void l2sn_dists(const float* idens, const float* data, int numD, int numCases,
                       std::vector<float>& labels, std::vector<float>& dists) {
    const int h = numCases / 2;
    labels.resize(h);
    dists.resize(h);
    for (int j = 0; j < h; j++) {
        labels[j] = idens[j] == idens[j + h] ? 1.f : -1.f;
        float s = 0.f;
        for (int k = 0; k < numD; k++) {
            const float d = data[(size_t)k * numCases + j] - data[(size_t)k * numCases + j + h];
            s = s + d * d;
        }
        dists[j] = sqrtf(s) + 0.00000001f;
    }
}

// getOptimalThr (layer_kernels.cu:482-498): first strict maximum, init m = d_0
ORC float orc_l2sn_optimal_thr(const float* dists, const float* labels, int nCases, int* idxOut) {
    float m = dists[0];
    int maxRate = 0, idx = 0;
    for (int i = 0; i < nCases; i++) {
        const float curr = dists[i];
        int rate = 0;
        for (int j = 0; j < nCases; j++)
            if ((dists[j] < curr && labels[j] == 1) || (dists[j] > curr && labels[j] == -1)) rate++;
        if (rate > maxRate) { maxRate = rate; m = curr; idx = i; }
    }
    if (idxOut) *idxOut = idx;
    return m;
}

// computeL2SNCost (layer_kernels.cu:505-604).  output has N/2 entries.
// Returns the margin used; mIdx (optional) = index of the chosen candidate or -1.
// distsOut/labelsOut (optional, N/2 each) expose the intermediates for tests.
ORC float orc_l2sn_cost(const float* idens, const float* data, float* output, int numD, int numCases,
                        float m, int* mIdx, float* distsOut, float* labelsOut) {
    std::vector<float> labels, dists;
    l2sn_dists(idens, data, numD, numCases, labels, dists);
    const int h = numCases / 2;
    int idx = -1;
    if (m < 0) m = orc_l2sn_optimal_thr(dists.data(), labels.data(), h, &idx);
    if (mIdx) *mIdx = idx;
    for (int j = 0; j < h; j++) {
        const float posMask = 0.5f * labels[j] + 0.5f, negMask1 = -0.5f * labels[j] + 0.5f;
        const float negMask2 = dists[j] < m ? 1.f : 0.f;
        const float posLoss = dists[j] * dists[j] * 0.5f * posMask;
        const float t = -1.0f * dists[j] + m;
        const float negLoss = t * t * 0.5f * negMask1 * negMask2;
        output[j] = posLoss + negLoss;
        if (distsOut) distsOut[j] = dists[j];
        if (labelsOut) labelsOut[j] = labels[j];
    }
    return m;
}

// computeL2SNGrad (layer_kernels.cu:620-735).  dataTarget (numD, numCases).
ORC float orc_l2sn_grad(const float* idens, const float* data, float* dataTarget, int numD, int numCases,
                        float m, int add, float coeff) {
    std::vector<float> labels, dists;
    l2sn_dists(idens, data, numD, numCases, labels, dists);
    const int h = numCases / 2;
    if (m < 0) m = orc_l2sn_optimal_thr(dists.data(), labels.data(), h, nullptr);
    for (int j = 0; j < h; j++) {
        const float posMask = 0.5f * labels[j] + 0.5f, negMask1 = -0.5f * labels[j] + 0.5f;
        const float negMask2 = dists[j] < m ? 1.f : 0.f;
        const float r = (1.0f / dists[j]) * m + (-1.0f);  // Reciprocal, MultByScalar(m), AddScalar(-1)
        for (int k = 0; k < numD; k++) {
            const float s1 = data[(size_t)k * numCases + j], s2 = data[(size_t)k * numCases + j + h];
            const float diff1 = s1 - s2, diff2 = s2 - s1;
            const float gx = diff2 * posMask + diff1 * r * negMask1 * negMask2;
            const float gy = diff1 * posMask + diff2 * r * negMask1 * negMask2;
            float* tg = dataTarget + (size_t)k * numCases + j;
            float* tp = dataTarget + (size_t)k * numCases + j + h;
            *tg = add ? *tg + gx * coeff : gx * coeff;
            *tp = add ? *tp + gy * coeff : gy * coeff;
        }
    }
    return m;
}


// ---------------------------------------------------------------------------------------------
// Fisher2 cost (layer_kernels.cu:310-479), every NVMatrix step restated on dense (N, N) matrices.
// X = data (numD, N).  Whole-matrix sums (NVMatrix::sum, an unspecified GPU reduction order in the
// reference) are accumulated in double and rounded to the float the reference keeps.
namespace {
struct Fisher2 {
    int N;
    std::vector<float> M, W, E;  // (N, N) row-major: row y, column x as in kLabel2Weight (:32-47)
    float a, b, V;               // sum(M.E), sum(W.E), sum(W.(E-b)^2)
};
float sumf(const std::vector<float>& v) {
    double s = 0;
    for (float x : v) s += x;
    return (float)s;
}
void label2weight(const float* labels, int N, bool tri, float w1, float w2, std::vector<float>& out) {  // :32-47
    out.assign((size_t)N * N, 0.f);
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++)
            out[(size_t)y * N + x] = (!tri || x > y) ? (labels[x] == labels[y] ? w1 : w2) : 0.f;
}
void fisher2_common(const float* labels, const float* data, int numD, int N, float alpha, Fisher2& f) {
    f.N = N;
    label2weight(labels, N, true, 1.0f, -1.0f, f.M);  // :326 / :392
    const float diff = sumf(f.M);
    const float n = N * (N - 1.0f) / 2.0f;
    const float w1 = 2.0f / (n + diff), w2 = 2.0f / (n - diff);
    label2weight(labels, N, true, w1, -w2, f.M);               // :332
    label2weight(labels, N, false, alpha, 1.0f - alpha, f.W);  // :336
    const float inv = 1.0f / sumf(f.W);
    for (float& w : f.W) w *= inv;  // :338
    // E = |x_i|^2 + |x_j|^2 - 2 x_i.x_j (:340-354): the -2 X^T X product first, then the row vector, then the column vector
    std::vector<float> s1(N);
    for (int j = 0; j < N; j++) {
        float s = 0.f;
        for (int d = 0; d < numD; d++) s = s + data[(size_t)d * N + j] * data[(size_t)d * N + j];
        s1[j] = s;
    }
    f.E.assign((size_t)N * N, 0.f);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            float dot = 0.f;
            for (int d = 0; d < numD; d++) dot = dot + data[(size_t)d * N + i] * data[(size_t)d * N + j];
            f.E[(size_t)i * N + j] = (-2.0f * dot + s1[j]) + s1[i];
        }
    double a = 0, b = 0, V = 0;
    for (size_t i = 0; i < f.E.size(); i++) {
        a += (double)(f.M[i] * f.E[i]);
        b += (double)(f.E[i] * f.W[i]);
    }
    f.a = (float)a;
    f.b = (float)b;
    for (size_t i = 0; i < f.E.size(); i++) {
        const float c = f.E[i] + -f.b;  // addScalar(-m) (:363 / :433)
        V += (double)((c * c) * f.W[i]);
    }
    f.V = (float)V;
}
}  // namespace

// computeFisher2Cost (:310-369): -a^2 / b with a = sum(M.E), b = sum(W.(E - sum(E.W))^2)
ORC float orc_fisher2_cost(const float* labels, const float* data, int numD, int numCases, float alpha) {
    Fisher2 f;
    fisher2_common(labels, data, numD, numCases, alpha, f);
    return -f.a * f.a / f.V;
}

// computeFisher2Grad (:371-479): target (+)= coeff * (G_X + G1_Y)
ORC void orc_fisher2_grad(const float* labels, const float* data, float* target, int numD, int numCases, float alpha, int add,
                          float coeff) {
    Fisher2 f;
    const int N = numCases;
    fisher2_common(labels, data, numD, N, alpha, f);
    const float a = f.a / f.V;  // :436
    // EM = -4a M + 4a^2 (E.W) - 4a^2 b W (:439-440)
    std::vector<float> EM((size_t)N * N), EW((size_t)N * N);
    for (size_t i = 0; i < EM.size(); i++) {
        EW[i] = f.E[i] * f.W[i];
        EM[i] = (-4.0f * a) * f.M[i] + (4.0f * a * a) * EW[i];
        EM[i] = EM[i] + (-4.0f * a * a * f.b) * f.W[i];
    }
    auto rowsum = [&](const std::vector<float>& m, int y) { double s = 0; for (int x = 0; x < N; x++) s += m[(size_t)y * N + x]; return (float)s; };
    auto colsum = [&](const std::vector<float>& m, int x) { double s = 0; for (int y = 0; y < N; y++) s += m[(size_t)y * N + x]; return (float)s; };
    std::vector<float> tr(N), tc(N);
    for (int k = 0; k < N; k++) {
        // tm.add(tew, 4a, -4a^2); tm.add(tw, 4a^2 b)  (:443-450 with row sums, :460-464 with column sums)
        tr[k] = (4.0f * a) * rowsum(f.M, k) + (-4.0f * a * a) * rowsum(EW, k);
        tr[k] = tr[k] + (4.0f * a * a * f.b) * rowsum(f.W, k);
        tc[k] = (4.0f * a) * colsum(f.M, k) + (-4.0f * a * a) * colsum(EW, k);
        tc[k] = tc[k] + (4.0f * a * a * f.b) * colsum(f.W, k);
    }
    for (int d = 0; d < numD; d++)
        for (int n = 0; n < N; n++) {
            const float* x = data + (size_t)d * N;
            float p1 = 0.f, p2 = 0.f;  // (X EM^T)[d][n] (:454-456) and (X EM)[d][n] (:468)
            for (int k = 0; k < N; k++) {
                p1 = p1 + x[k] * EM[(size_t)n * N + k];
                p2 = p2 + x[k] * EM[(size_t)k * N + n];
            }
            const float xx = x[n] * tr[n] + p1, yy = x[n] * tc[n] + p2;
            const float g = (xx + yy) * coeff;
            float* t = target + (size_t)d * N + n;
            *t = add ? *t + g : g;
        }
}
