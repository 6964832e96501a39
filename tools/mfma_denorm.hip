// tools/mfma_denorm.hip -- does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (debug probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    a[0] = (_Float16)a_val;   // A[row][k=8h]
    b[0] = (_Float16)b_val;
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float tests[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1024.0f}, {1.0e-6f, 1024.0f}, {6.0e-8f, 16384.0f}, {3.0e-5f, 3.0e-5f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const float ref = 2.0f * (float)(_Float16)t[0] * (float)(_Float16)t[1];  // two k-groups (h=0,1) hit element 0? no: only lanes' own k
        printf("a=%g b=%g  mfma=%.9g  expected(one term)=%.9g\n", t[0], t[1], h, ref / 2);
    }
    return 0;
}
