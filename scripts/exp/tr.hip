// Probe for the gfx950 LDS transpose read used by the k-major GEMM operands (csrc/gemm.hip, Dma::frag):
//   ds_read_b64_tr_b16  (builtin __builtin_amdgcn_ds_read_tr16_b64_v4i16)
// Fills an LDS image [16 k-rows][16 columns] of 16-bit values v(k, c) = 100 * k + c, lets every lane of one wave issue the
// read with the address pattern the GEMM uses, and prints what each lane received.  Observed on MI355X (round 1):
//   lane l passes the address of element (k = l_i >> 2 within a 4-row block ..., c = (l_i & 3) * 4), l_i = l & 15, and RECEIVES
//   the four values of column (l & 15) at rows (l >> 4) * 4 + {0, 1, 2, 3} of the 4 x 16 block its 16-lane group addressed --
//   i.e. a hardware 4x16 -> 16x4 transpose: exactly the "8 consecutive k for one output row" MFMA fragment after two reads.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_test scripts/exp/tr.hip && /tmp/tr_test
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void probe(short* out) {
    __shared__ __attribute__((aligned(16))) short img[16 * 16];
    const int lane = threadIdx.x;
    for (int t = lane; t < 256; t += 64) img[t] = (short)(100 * (t / 16) + (t % 16));
    __syncthreads();
    const int i = lane & 15;
    // 16 lanes address a 4 x 16 block: row (i >> 2) of the block selected by (lane >> 4), 4-element column group (i & 3)
    const short* p = img + ((lane >> 4) * 4 + (i >> 2)) * 16 + (i & 3) * 4;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

int main() {
    short* d;
    short h[256];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            printf(" %4d", h[l * 4 + j]);
            const int want = 100 * ((l >> 4) * 4 + j) + (l & 15);       // column l & 15, rows (l >> 4) * 4 + j
            bad += h[l * 4 + j] != want;
        }
        printf("\n");
    }
    printf(bad ? "MISMATCH vs the documented transpose semantics (%d)\n" : "transpose semantics as documented (%d mismatches)\n", bad);
    return bad != 0;
}
