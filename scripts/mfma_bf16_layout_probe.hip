#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// hypothesis: lane l holds A[l%32][8*(l/32)+j], B[l%32][8*(l/32)+j]; acc[r] = C[(r&3)+8*(r>>2)+4*(l>>5)][l&31], C = A.B^T
__global__ void k(const float *A, const float *B, float *C)
{
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = (__bf16)A[(l % 32) * 16 + 8 * (l / 32) + j];
        b[j] = (__bf16)B[(l % 32) * 16 + 8 * (l / 32) + j];
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
int main()
{
    float hA[32 * 16], hB[32 * 16], hC[32 * 32];
    for (int i = 0; i < 32 * 16; i++) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            float ref = 0;
            for (int kk = 0; kk < 16; kk++) ref += hA[i * 16 + kk] * hB[j * 16 + kk];
            if (ref != hC[i * 32 + j]) { if (bad < 5) printf("mismatch C[%d][%d] = %g ref %g\n", i, j, hC[i * 32 + j], ref); bad++; }
        }
    printf("layout check: %d mismatches\n", bad);
    return bad != 0;
}
