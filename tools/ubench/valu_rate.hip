// Issue cost of the vector instructions the split-fp16 staging uses, one wave (and 2 / 3 waves per SIMD) at a time.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
template <int OP>
__global__ void k(float* out, uint64_t* cyc, int iters) {
    float x[8], y[8];
    unsigned h[8], l[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i, y[i] = 1.f + i, h[i] = threadIdx.x + i, l[i] = i;
    float s = 2.f;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(s));
                if (OP == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h[i]) : "v"(x[i]), "v"(s));
                if (OP == 2) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[i]) : "v"(x[i]), "v"(s), "v"(h[i]));
                if (OP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(x[i]), "v"(y[i]));
                if (OP == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(s));
                if (OP == 5) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(x[i]) : "v"(y[i]), "v"(s));
                if (OP == 6) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(y[i]) : "v"(h[i]), "v"(x[i]));
                if (OP == 7) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(*(double*)&x[i & 6]) : "v"(*(double*)&y[i & 6]), "v"(*(double*)&y[(i + 2) & 6]));
                if (OP == 8) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(s));
                if (OP == 9) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[i]) : "v"(y[i]), "v"(s));
            }
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += x[i] + y[i] + h[i] + l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, uint64_t* cyc) {
    const int iters = 2000;
    for (int waves : {1, 2, 3}) {      // waves per SIMD (blocks of 256 threads = one wave per SIMD each, `waves` blocks per CU)
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256 * waves), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s waves/SIMD %d: %6.2f counter ticks / instr (wave 0), %7.3f ms  -> %.2f ns / instr / wave\n", name, waves,
               (double)c / ((double)iters * REP), ms, ms * 1e6 / ((double)iters * REP));
    }
}

int main() {
    float* out;
    uint64_t* cyc;
    hipMalloc(&out, 256 * 3 * 256 * 4);
    hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_fma_mixlo_f16", out, cyc);
    run<2>("v_fma_mixhi_f16 (3 src)", out, cyc);
    run<3>("v_cvt_pk_f16_f32", out, cyc);
    run<4>("v_med3_f32", out, cyc);
    run<5>("v_max3_f32 |.|", out, cyc);
    run<6>("v_fma_mix_f32", out, cyc);
    run<7>("v_pk_mul_f32", out, cyc);
    run<8>("v_mul_f32", out, cyc);
    run<9>("v_cndmask_b32", out, cyc);
    return 0;
}
