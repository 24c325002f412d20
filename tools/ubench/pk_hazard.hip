// Attempt at a minimal reproduction of the "zero lanes" hazard of round 4: with SLP vectorisation the epilogue of
// conv_h2_kernel computed y = acc * descale + bias as `v_pk_fma_f32 ... op_sel:[0,1,1]` on register pairs a ds_read_b128 had just
// returned (the per-row tables live in LDS), next to matrix instructions of the other waves; on a loaded chip (768-sample launches)
// the low result came back as exactly 0.0 in lanes 48..63 a few hundred times per launch, non-deterministically.
//
// This kernel isolates the ingredients: every wave runs a chain of v_mfma_f32_32x32x16_f16 into 4 accumulator tiles, then reads a
// (descale, bias) table from LDS with ds_read_b128 and IMMEDIATELY (one s_waitcnt lgkmcnt(0), no other instruction) feeds the
// returned register pairs to v_pk_fma_f32 with the op_sel pattern the compiler chose; the same products are formed with scalar
// v_fma_f32 from a second, earlier read of the table; any lane where the two differ is counted.  Launched like the decoder convs:
// 15360 workgroups of 256 threads, 3 resident per CU, 1 GB of stores.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_hazard pk_hazard.hip ; run: ./pk_hazard [repeats]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 3) void k(float* __restrict__ out, unsigned* __restrict__ bad, unsigned* __restrict__ zero_lanes, int stages) {
    __shared__ __attribute__((aligned(16))) float tab[4 * 64];      // [row][ds0, bias0, ds1, bias1]
    __shared__ __attribute__((aligned(16))) unsigned char xl[16896];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 256; i += 256) tab[i] = 0.5f + 0.001f * i + 0.01f * (blockIdx.x & 7);
    for (int i = threadIdx.x; i < 16896 / 4; i += 256) reinterpret_cast<unsigned*>(xl)[i] = 0x3c003c00u + ((i * 2654435761u) & 0x03ff03ffu);
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h16x8 a = *reinterpret_cast<const h16x8*>(xl + lane * 16);
    for (int st = 0; st < stages; ++st) {
        const h16x8 b0 = *reinterpret_cast<const h16x8*>(xl + ((lane + 64 * ((st + wave) & 7)) * 16));
        const h16x8 b1 = *reinterpret_cast<const h16x8*>(xl + ((lane + 64 * ((st + wave + 3) & 7)) * 16 + 8192));
        for (int r = 0; r < 3; ++r) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a, acc[3], 0, 0, 0);
        }
    }
    // the epilogue under test, per accumulator tile and row pair
    unsigned nbad = 0, nzero = 0;
    float* const o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 64;
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const f32x4 ref = *reinterpret_cast<const f32x4*>(&tab[4 * (row & 63)]);      // early read: scalar reference operands
            float y0s = fmaf(acc[j][r], ref[0], ref[1]);
            float y1s = fmaf(acc[j][r + 1], ref[2], ref[3]);
            float y0p, y1p;
            // the late read feeds the packed instruction straight away
            asm volatile(
                "ds_read_b128 v[40:43], %4\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_mov_b32 v44, %2\n"
                "v_mov_b32 v45, %3\n"
                "v_pk_fma_f32 v[46:47], v[44:45], v[40:41], v[42:43] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
                "v_mov_b32 %0, v46\n"
                "v_mov_b32 %1, v47\n"
                : "=v"(y0p), "=v"(y1p)
                : "v"(acc[j][r]), "v"(acc[j][r + 1]), "v"((unsigned)(reinterpret_cast<uintptr_t>(&tab[4 * (row & 63)]) & 0xffffu))
                : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
            // packed semantics used here: lo = a.lo * b.lo + c.hi ... (whatever the op_sel selects), so compare against the same selection
            const float e0 = fmaf(acc[j][r], ref[0], ref[3]), e1 = fmaf(acc[j][r + 1], ref[1], ref[2]);
            if (y0p != e0 || y1p != e1) ++nbad;
            if ((y0p == 0.f && e0 != 0.f) || (y1p == 0.f && e1 != 0.f)) ++nzero;
            o[j * 16 + r] = y0s + y0p;
            o[j * 16 + r + 1] = y1s + y1p;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (nzero) atomicAdd(zero_lanes + (lane >> 4), nzero);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int blocks = 15360;
    float* out;
    unsigned *bad, *zl;
    hipMalloc(&out, (size_t)blocks * 256 * 64 * 4);
    hipMalloc(&bad, 4);
    hipMalloc(&zl, 16);
    hipMemset(bad, 0, 4);
    hipMemset(zl, 0, 16);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, bad, zl, 8);
    hipDeviceSynchronize();
    unsigned hb, hz[4];
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(hz, zl, 16, hipMemcpyDeviceToHost);
    printf("launches %d x %d workgroups: lanes where v_pk_fma_f32 on fresh ds_read_b128 results differs from the expected value: %u; exact zeros by lane quarter: %u %u %u %u\n",
           reps, blocks, hb, hz[0], hz[1], hz[2], hz[3]);
    return 0;
}
