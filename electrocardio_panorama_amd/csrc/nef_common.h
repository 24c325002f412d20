// Shared helpers for the gfx950 kernels of libnefnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nefnet_hip.h"

#define NEF_WAVE 64

#define NEF_REQUIRE(cond, code) \
    do {                        \
        if (!(cond)) return (code); \
    } while (0)

// hipGetLastError() is sticky across the process: drop whatever an earlier (foreign) call left behind so that the
// status returned by an entry point reflects only its own launches.
#define NEF_ENTER() (void)hipGetLastError()

static inline int nef_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NEF_OK : (int)e;
}

static inline int64_t nef_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Diagnostics switches (A/B kernel forms, timing experiments) are read from the environment ONLY under NEF_DIAG=1: a stray variable
// cannot change which kernel a production run takes.  Product switches (NEF_H2, NEF_LIB, ...) are documented in README.md.
#include <stdlib.h>
static inline const char* nef_diag_env(const char* name) {
    const char* d = getenv("NEF_DIAG");
    return (d && d[0] == '1') ? getenv(name) : nullptr;
}

// Compute units of the CURRENT device (256 on MI355X); queried once per device, idempotent -> thread-safe.
static inline int nef_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    int n = __atomic_load_n(&cached[dev & 63], __ATOMIC_ACQUIRE);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&cached[dev & 63], n, __ATOMIC_RELEASE);
    }
    return n;
}

// Raise a kernel's dynamic-LDS limit once per DEVICE.  `done` is a per-kernel bit mask (bit = device ordinal); the call is
// idempotent, so two host threads racing on the first launch both succeed -- the entry points stay re-entrant and the
// library keeps no other state.
static inline int nef_ensure_dyn_lds(const void* fn, size_t lds, unsigned long long* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
    return 0;
}

// Grid for an HBM-bound elementwise pass: enough blocks to fill 256 CUs, grid-stride beyond.
static inline int nef_stream_grid(int64_t work_items, int block) {
    int64_t g = nef_cdiv(work_items, block);
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float nef_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double nef_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves).  `sm` must hold >= 4 doubles.  All threads get the result.
__device__ __forceinline__ double nef_block_sum_d(double v, double* sm) {
    v = nef_wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// Counter-based keep decision for in-kernel dropout: uniform in [0,1) from (seed, dense element index).  One 64-bit mix
// (two 64-bit multiplies: ~90 % of the decision's cost) serves TWO neighbouring elements -- elements 2k and 2k+1 take the
// bit fields [40,64) and [16,40) of the same word -- so the Winograd epilogues, which own aligned output pairs, hash once
// per pair (measured on the K = 3 encoder convs: the dropout epilogue cost 12 % of the launch with one mix per element).
__device__ __forceinline__ uint64_t nef_rng_mix(uint64_t seed, uint64_t pair) {
    uint64_t z = pair + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float nef_rng_uniform(uint64_t seed, uint64_t idx) {
    const uint64_t z = nef_rng_mix(seed, idx >> 1);
    return (float)((idx & 1) ? ((z >> 16) & 0xFFFFFFull) : (z >> 40)) * (1.0f / 16777216.0f);
}
// both elements of the aligned pair (even_idx, even_idx + 1)
__device__ __forceinline__ void nef_rng_uniform2(uint64_t seed, uint64_t even_idx, float& u0, float& u1) {
    const uint64_t z = nef_rng_mix(seed, even_idx >> 1);
    u0 = (float)(z >> 40) * (1.0f / 16777216.0f);
    u1 = (float)((z >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------------------
// Buffer-descriptor loads (gfx950 `buffer_load_* ... offen`).  One wave-uniform 128-bit descriptor per tile, a
// 32-bit per-lane byte offset (VGPR) and a wave-uniform byte offset (SGPR): a whole tile of rows costs ONE address
// register per lane, and lanes whose offset is NEF_OOB get 0.0 from the hardware range check -- the conv halo /
// zero padding comes for free, with no branches around the loads.
// ------------------------------------------------------------------------------------------------------------
typedef unsigned nef_u32x4 __attribute__((ext_vector_type(4)));
typedef float nef_f32x4 __attribute__((ext_vector_type(4)));
#define NEF_OOB 0x80000000u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t nef_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFC, 0x00020000);
}
// descriptor with a run-time extent: `bytes` = 0 turns every load through it into a no-op that returns 0.0 without a memory
// access -- a branch-free way to switch a burst of loads off (see conv_wino4_kernel: a branch around the burst makes the
// compiler's s_waitcnt bookkeeping conservative and the burst synchronous)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t nef_rsrc_n(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float nef_buf_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
typedef float nef_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ nef_f32x2 nef_buf_f32x2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(nef_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void nef_buf_store_f32x4(nef_f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nef_u32x4, v), r, (int)voff, (int)soff, 0);      // NEF_OOB lanes store nothing
}
__device__ __forceinline__ nef_f32x4 nef_buf_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(nef_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
