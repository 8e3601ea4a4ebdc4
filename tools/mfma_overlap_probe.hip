// Does v_mfma_f32_16x16x4_f32 tolerate a vDst that PARTIALLY overlaps SrcC on gfx950?  (LLVM only forbids it for
// destinations wider than 4 VGPRs; tools/check_mfma_overlap.py forbids it always.)  Chains of dependent MFMAs whose
// destination slides down / up by two registers per step, against the same chain on disjoint, identical tuples.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* in, float* out, int mode) {
    const int l = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    in += (size_t)(w & 63) * 768;
    out += (size_t)w * 256;
    float a0 = in[l], a1 = in[64 + l], a2 = in[128 + l], a3 = in[192 + l];
    float b0 = in[256 + l], b1 = in[320 + l], b2 = in[384 + l], b3 = in[448 + l];
    f32x4 c = {in[512 + l], in[576 + l], in[640 + l], in[704 + l]};
    f32x4 d;
    if (mode == 0) {  // reference: same tuple every step (tied)
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, d, 0, 0, 0);
    } else if (mode == 1) {  // slide DOWN by 2: [46:49] -> [44:47] -> [42:45] -> [40:43] -> [38:41]
        asm volatile(
            "v_mov_b32 v46, %4\n v_mov_b32 v47, %5\n v_mov_b32 v48, %6\n v_mov_b32 v49, %7\n s_nop 4\n"
            "v_mfma_f32_16x16x4_f32 v[44:47], %8, %12, v[46:49]\n"
            "v_mfma_f32_16x16x4_f32 v[42:45], %9, %13, v[44:47]\n"
            "v_mfma_f32_16x16x4_f32 v[40:43], %10, %14, v[42:45]\n"
            "v_mfma_f32_16x16x4_f32 v[38:41], %11, %15, v[40:43]\n"
            "s_nop 15\n s_nop 15\n"
            "v_mov_b32 %0, v38\n v_mov_b32 %1, v39\n v_mov_b32 %2, v40\n v_mov_b32 %3, v41\n"
            : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
            : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
            : "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    } else if (mode == 2) {  // slide UP by 2: [38:41] -> [40:43] -> ... -> [46:49]
        asm volatile(
            "v_mov_b32 v38, %4\n v_mov_b32 v39, %5\n v_mov_b32 v40, %6\n v_mov_b32 v41, %7\n s_nop 4\n"
            "v_mfma_f32_16x16x4_f32 v[40:43], %8, %12, v[38:41]\n"
            "v_mfma_f32_16x16x4_f32 v[42:45], %9, %13, v[40:43]\n"
            "v_mfma_f32_16x16x4_f32 v[44:47], %10, %14, v[42:45]\n"
            "v_mfma_f32_16x16x4_f32 v[46:49], %11, %15, v[44:47]\n"
            "s_nop 15\n s_nop 15\n"
            "v_mov_b32 %0, v46\n v_mov_b32 %1, v47\n v_mov_b32 %2, v48\n v_mov_b32 %3, v49\n"
            : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
            : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
            : "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    } else if (mode == 3) {  // shift by 2 down then up, single instructions with independent filler between (no forwarding)
        asm volatile(
            "v_mov_b32 v46, %4\n v_mov_b32 v47, %5\n v_mov_b32 v48, %6\n v_mov_b32 v49, %7\n s_nop 4\n"
            "v_mfma_f32_16x16x4_f32 v[44:47], %8, %12, v[46:49]\n s_nop 15\n s_nop 15\n"
            "v_mfma_f32_16x16x4_f32 v[46:49], %9, %13, v[44:47]\n s_nop 15\n s_nop 15\n"
            "v_mfma_f32_16x16x4_f32 v[44:47], %10, %14, v[46:49]\n s_nop 15\n s_nop 15\n"
            "v_mfma_f32_16x16x4_f32 v[46:49], %11, %15, v[44:47]\n s_nop 15\n s_nop 15\n"
            "v_mov_b32 %0, v46\n v_mov_b32 %1, v47\n v_mov_b32 %2, v48\n v_mov_b32 %3, v49\n"
            : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
            : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
            : "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    }
    for (int r = 0; r < 4; r++) out[r * 64 + l] = d[r];
}

int main() {
    const int NW = 2048 * 8;
    std::vector<float> h(768 * 64);
    srand(1);
    for (auto& x : h) x = (rand() % 2001 - 1000) / 500.0f;
    float *din, *dout;
    hipMalloc(&din, 768 * 64 * 4);
    hipMalloc(&dout, (size_t)NW * 256 * 4);
    hipMemcpy(din, h.data(), 768 * 64 * 4, hipMemcpyHostToDevice);
    std::vector<float> ref((size_t)NW * 256), got((size_t)NW * 256);
    int bad_total = 0;
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < (mode ? 20 : 1); rep++) {
            hipLaunchKernelGGL(probe, dim3(2048), dim3(512), 0, 0, din, dout, mode);
            hipMemcpy((mode ? got : ref).data(), dout, (size_t)NW * 256 * 4, hipMemcpyDeviceToHost);
            if (mode) {
                int bad = 0;
                for (size_t i = 0; i < (size_t)NW * 256; i++) bad += got[i] != ref[i];
                if (bad && rep == 0) printf("mode %d: %d values differ (e.g. got %g want %g)\n", mode, bad, got[0], ref[0]);
                bad_total += bad;
                if (bad) break;
            }
        }
        if (mode) printf("mode %d done\n", mode);
    }
    printf(bad_total ? "PARTIAL OVERLAP GIVES WRONG RESULTS\n" : "partial overlap: results identical\n");
    return 0;
}
