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

