// LDS-staged / row-split GEMM forms for the classifier-layer shapes (fc_gemm.hip); gemm.hip dispatches to them.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

bool fc_nn_lds_usable(const float* A, const float* B, const float* C, int M, int N, int K);
bool fc_nt_lds_usable(const float* A, const float* B, const float* C, int M, int N, int K);
bool fc_tn_split_usable(const float* A, const float* B, const float* C, int M, int N, int K);
size_t fc_tn_split_workspace_bytes(int M, int N, int K);
int fc_nn_lds_launch(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, float alpha, float beta,
                     int neuron, float na, float nb, float* smPart, hipStream_t s);
int fc_nt_lds_launch(const float* A, const float* B, float* C, int M, int N, int K, float alpha, float beta, hipStream_t s);
// the L2-SN cost's gradient of C, added by the reduce pass of the TN form (mcn_fc_bprop_acts_l2sn)
struct FcL2snAddend {
    const float *data, *stats, *info;
    float coeff;
    int* gate;  // stream gate announcing stats / info, or null
    int gateValue, timeoutMs;
};
// slabs [splits][M x N] -> C = sum (index order, as k_fc_tn_reduce with alpha 1, beta 0) + the addend (eltwise_cost.hip)
int fc_tn_reduce_l2sn_launch(const float* part, float* C, int M, int N, int splits, const FcL2snAddend& l2, hipStream_t s);
int fc_tn_split_launch(const float* A, const float* B, float* C, int M, int N, int K, float alpha, float beta, void* workspace,
                       size_t workspaceBytes, hipStream_t s, const FcL2snAddend* l2 = nullptr);

// all inputs of a hidden FC layer in one launch (forward / backward-data)
bool fc_fprop_multi_usable(const float* const* W, const float* const* X, const int* K, int count, const float* Y, int M, int N);
int fc_fprop_multi_launch(const float* const* W, const float* const* X, const int* K, int count, const float* bias, float* Y, int M,
                          int N, float beta, int neuron, float na, float nb, hipStream_t s);
bool fc_bprop_multi_usable(const float* const* W, float* const* dX, const int* Mi, int count, const float* dY, int K, int N);
int fc_bprop_multi_launch(const float* const* W, float* const* dX, const int* Mi, const float* beta, const float* const* nacts,
                          const int* ntype, const float* na, const float* nb, int count, const float* dY, int K, int N, hipStream_t s);
