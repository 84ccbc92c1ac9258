// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs, and does the fp32 -> fp16 conversion keep them?
// (Decides how an fp16 two-piece split of fp32 operands has to scale its pieces.)   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float* in, float* out) {
    // in[0] = a (fp32), in[1] = b (fp32): every A element = fp16(a), every B element = fp16(b)
    const _Float16 a = (_Float16)in[0], b = (_Float16)in[1];
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = a; B[i] = b; }
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a; out[2] = (float)b; }
}

int main() {
    float *din, *dout;
    hipMalloc(&din, 8); hipMalloc(&dout, 12);
    const float cases[][2] = {{9.5367431640625e-07f, 1024.f},      // a = 2^-20 (fp16 subnormal), b = 2^10: sum of 16 = 2^-6 = 0.015625
                              {5.9604644775390625e-08f, 16384.f},  // a = 2^-24 (smallest subnormal), b = 2^14: 16 * 2^-10 = 0.015625
                              {3.0517578125e-05f, 1.f},            // a = 2^-15 (subnormal), 16 * 2^-15 = 4.8828125e-4
                              {6.103515625e-05f, 1.f},             // a = 2^-14 (smallest NORMAL): 9.765625e-4
                              {1.5f, 6.1e-6f}};                    // b subnormal
    for (auto& c : cases) {
        hipMemcpy(din, c, 8, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(din, dout);
        float o[3];
        hipMemcpy(o, dout, 12, hipMemcpyDeviceToHost);
        printf("a=%.10e b=%.10e  fp16(a)=%.10e fp16(b)=%.10e  mfma sum16=%.10e  expected=%.10e\n", c[0], c[1], o[1], o[2], o[0], 16.0 * (double)o[1] * (double)o[2]);
    }
    return 0;
}
