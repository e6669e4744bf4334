// Probe: cost per IDCT term of (a) the DPP-broadcast loop used by k_idct_color and (b) a loop that takes (table row offset,
// coefficient) pairs from SGPRs (scalar loads of a list in global memory) and reads the table with ds_read_addtid_b32 (M0 + lane*4).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/idct_terms tools/probes/idct_terms.hip && /tmp/idct_terms
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define NT 24            // terms per block (multiple of 8)
#define BLOCKS_PER_WAVE 256

template <int I> __device__ __forceinline__ uint32_t row_bc(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + I, 0xF, 0xF, true); }
#define G4(E, I) { \
    const float l0 = *reinterpret_cast<const float*>(lut_b + (row_bc<I>(E.x) + lane4)); \
    const float l1 = *reinterpret_cast<const float*>(lut_b + (row_bc<I + 1>(E.x) + lane4)); \
    const float l2 = *reinterpret_cast<const float*>(lut_b + (row_bc<I + 2>(E.x) + lane4)); \
    const float l3 = *reinterpret_cast<const float*>(lut_b + (row_bc<I + 3>(E.x) + lane4)); \
    acc = __fadd_rn(acc, __fmul_rn(__uint_as_float(row_bc<I>(E.y)), l0)); \
    acc = __fadd_rn(acc, __fmul_rn(__uint_as_float(row_bc<I + 1>(E.y)), l1)); \
    acc = __fadd_rn(acc, __fmul_rn(__uint_as_float(row_bc<I + 2>(E.y)), l2)); \
    acc = __fadd_rn(acc, __fmul_rn(__uint_as_float(row_bc<I + 3>(E.y)), l3)); }

// (a) lists in global memory as uint2 {row*256, float bits}; each 16-lane row loads 16 entries into registers
__global__ void __launch_bounds__(512) k_dpp(const float* __restrict__ lut, const uint2* __restrict__ lists, float* __restrict__ out)
{
    __shared__ float s_lut[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 512) s_lut[i] = lut[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 8 + (threadIdx.x >> 6), lane4 = lane * 4, li = lane & 15;
    const char* lut_b = reinterpret_cast<const char*>(s_lut);
    float tot = 0.f;
    for (int b = 0; b < BLOCKS_PER_WAVE; b++) {
        const uint2* L = lists + ((size_t)wave * BLOCKS_PER_WAVE + b) * NT;
        float acc = 0.f;
        for (int r = 0; r < NT; r += 16) {
            const uint2 e = L[r + li < NT ? r + li : 0];
            G4(e, 0) G4(e, 4) if (r + 8 >= NT) break; G4(e, 8) G4(e, 12)
        }
        tot += acc;
    }
    out[(size_t)wave * 64 + lane] = tot;
}

// (b) eight entries at a time from SGPRs
__device__ __forceinline__ float terms8(float acc, const uint32_t (&e)[16])
{
    float l0, l1, l2, l3, l4, l5, l6, l7;
    asm volatile(
        "s_mov_b32 m0, %9\n\ts_nop 0\n\tds_read_addtid_b32 %1\n\t"
        "s_mov_b32 m0, %11\n\ts_nop 0\n\tds_read_addtid_b32 %2\n\t"
        "s_mov_b32 m0, %13\n\ts_nop 0\n\tds_read_addtid_b32 %3\n\t"
        "s_mov_b32 m0, %15\n\ts_nop 0\n\tds_read_addtid_b32 %4\n\t"
        "s_mov_b32 m0, %17\n\ts_nop 0\n\tds_read_addtid_b32 %5\n\t"
        "s_mov_b32 m0, %19\n\ts_nop 0\n\tds_read_addtid_b32 %6\n\t"
        "s_mov_b32 m0, %21\n\ts_nop 0\n\tds_read_addtid_b32 %7\n\t"
        "s_mov_b32 m0, %23\n\ts_nop 0\n\tds_read_addtid_b32 %8\n\t"
        "s_waitcnt lgkmcnt(7)\n\tv_mul_f32 %1, %10, %1\n\tv_add_f32 %0, %0, %1\n\t"
        "s_waitcnt lgkmcnt(6)\n\tv_mul_f32 %2, %12, %2\n\tv_add_f32 %0, %0, %2\n\t"
        "s_waitcnt lgkmcnt(5)\n\tv_mul_f32 %3, %14, %3\n\tv_add_f32 %0, %0, %3\n\t"
        "s_waitcnt lgkmcnt(4)\n\tv_mul_f32 %4, %16, %4\n\tv_add_f32 %0, %0, %4\n\t"
        "s_waitcnt lgkmcnt(3)\n\tv_mul_f32 %5, %18, %5\n\tv_add_f32 %0, %0, %5\n\t"
        "s_waitcnt lgkmcnt(2)\n\tv_mul_f32 %6, %20, %6\n\tv_add_f32 %0, %0, %6\n\t"
        "s_waitcnt lgkmcnt(1)\n\tv_mul_f32 %7, %22, %7\n\tv_add_f32 %0, %0, %7\n\t"
        "s_waitcnt lgkmcnt(0)\n\tv_mul_f32 %8, %24, %8\n\tv_add_f32 %0, %0, %8"
        : "+v"(acc), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(l4), "=&v"(l5), "=&v"(l6), "=&v"(l7)
        : "s"(e[0]), "s"(e[1]), "s"(e[2]), "s"(e[3]), "s"(e[4]), "s"(e[5]), "s"(e[6]), "s"(e[7]),
          "s"(e[8]), "s"(e[9]), "s"(e[10]), "s"(e[11]), "s"(e[12]), "s"(e[13]), "s"(e[14]), "s"(e[15])
        : "memory");
    return acc;
}
__global__ void __launch_bounds__(512) k_sgpr(const float* __restrict__ lut, const uint32_t* __restrict__ lists, float* __restrict__ out, uint32_t lds_base_probe)
{
    __shared__ float s_lut[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 512) s_lut[i] = lut[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 8 + (threadIdx.x >> 6));
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)s_lut);        // LDS offset of the table
    asm volatile("" :: "s"(base) : "memory");
    float tot = 0.f;
    for (int b = 0; b < BLOCKS_PER_WAVE; b++) {
        const uint32_t* L = lists + ((size_t)wave * BLOCKS_PER_WAVE + b) * NT * 2;
        float acc = 0.f;
        uint32_t e[NT / 8][16];
        #pragma unroll
        for (int r = 0; r < NT / 8; r++) {
            #pragma unroll
            for (int q = 0; q < 16; q++) e[r][q] = __builtin_amdgcn_readfirstlane(L[r * 16 + q]) + ((q & 1) ? 0u : base);
        }
        #pragma unroll
        for (int r = 0; r < NT / 8; r++) acc = terms8(acc, e[r]);
        tot += acc;
    }
    out[(size_t)wave * 64 + lane] = tot;
}


// (c) as (b), software-pipelined: the multiply of term i of the current group sits between "s_mov m0" and the table read of
//     term i of the NEXT group (it is the wait state that hazard needs), so a term costs 1 SALU + 2 VALU + 1 LDS issue.
__device__ __forceinline__ void reads8(float (&l)[8], const uint32_t (&e)[16])
{
    asm volatile(
        "s_mov_b32 m0, %8\n\ts_nop 0\n\tds_read_addtid_b32 %0\n\t" "s_mov_b32 m0, %9\n\ts_nop 0\n\tds_read_addtid_b32 %1\n\t"
        "s_mov_b32 m0, %10\n\ts_nop 0\n\tds_read_addtid_b32 %2\n\t" "s_mov_b32 m0, %11\n\ts_nop 0\n\tds_read_addtid_b32 %3\n\t"
        "s_mov_b32 m0, %12\n\ts_nop 0\n\tds_read_addtid_b32 %4\n\t" "s_mov_b32 m0, %13\n\ts_nop 0\n\tds_read_addtid_b32 %5\n\t"
        "s_mov_b32 m0, %14\n\ts_nop 0\n\tds_read_addtid_b32 %6\n\t" "s_mov_b32 m0, %15\n\ts_nop 0\n\tds_read_addtid_b32 %7"
        : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3]), "=&v"(l[4]), "=&v"(l[5]), "=&v"(l[6]), "=&v"(l[7])
        : "s"(e[0]), "s"(e[2]), "s"(e[4]), "s"(e[6]), "s"(e[8]), "s"(e[10]), "s"(e[12]), "s"(e[14]) : "memory");
}
// math of group (l, e) while reading group (ln, en)
__device__ __forceinline__ float step8(float acc, float (&l)[8], const uint32_t (&e)[16], float (&ln)[8], const uint32_t (&en)[16])
{
    asm volatile(
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %17\n\tv_mul_f32 %1, %25, %1\n\tds_read_addtid_b32 %9\n\tv_add_f32 %0, %0, %1\n\t"
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %18\n\tv_mul_f32 %2, %26, %2\n\tds_read_addtid_b32 %10\n\tv_add_f32 %0, %0, %2\n\t"
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %19\n\tv_mul_f32 %3, %27, %3\n\tds_read_addtid_b32 %11\n\tv_add_f32 %0, %0, %3\n\t"
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %20\n\tv_mul_f32 %4, %28, %4\n\tds_read_addtid_b32 %12\n\tv_add_f32 %0, %0, %4\n\t"
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %21\n\tv_mul_f32 %5, %29, %5\n\tds_read_addtid_b32 %13\n\tv_add_f32 %0, %0, %5\n\t"
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %22\n\tv_mul_f32 %6, %30, %6\n\tds_read_addtid_b32 %14\n\tv_add_f32 %0, %0, %6\n\t"
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %23\n\tv_mul_f32 %7, %31, %7\n\tds_read_addtid_b32 %15\n\tv_add_f32 %0, %0, %7\n\t"
        "s_waitcnt lgkmcnt(7)\n\ts_mov_b32 m0, %24\n\tv_mul_f32 %8, %32, %8\n\tds_read_addtid_b32 %16\n\tv_add_f32 %0, %0, %8"
        : "+v"(acc), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7]),
          "=&v"(ln[0]), "=&v"(ln[1]), "=&v"(ln[2]), "=&v"(ln[3]), "=&v"(ln[4]), "=&v"(ln[5]), "=&v"(ln[6]), "=&v"(ln[7])
        : "s"(en[0]), "s"(en[2]), "s"(en[4]), "s"(en[6]), "s"(en[8]), "s"(en[10]), "s"(en[12]), "s"(en[14]),
          "s"(e[1]), "s"(e[3]), "s"(e[5]), "s"(e[7]), "s"(e[9]), "s"(e[11]), "s"(e[13]), "s"(e[15]) : "memory");
    return acc;
}
__device__ __forceinline__ float math8(float acc, float (&l)[8], const uint32_t (&e)[16])
{
    asm volatile(
        "s_waitcnt lgkmcnt(7)\n\tv_mul_f32 %1, %9, %1\n\tv_add_f32 %0, %0, %1\n\t" "s_waitcnt lgkmcnt(6)\n\tv_mul_f32 %2, %10, %2\n\tv_add_f32 %0, %0, %2\n\t"
        "s_waitcnt lgkmcnt(5)\n\tv_mul_f32 %3, %11, %3\n\tv_add_f32 %0, %0, %3\n\t" "s_waitcnt lgkmcnt(4)\n\tv_mul_f32 %4, %12, %4\n\tv_add_f32 %0, %0, %4\n\t"
        "s_waitcnt lgkmcnt(3)\n\tv_mul_f32 %5, %13, %5\n\tv_add_f32 %0, %0, %5\n\t" "s_waitcnt lgkmcnt(2)\n\tv_mul_f32 %6, %14, %6\n\tv_add_f32 %0, %0, %6\n\t"
        "s_waitcnt lgkmcnt(1)\n\tv_mul_f32 %7, %15, %7\n\tv_add_f32 %0, %0, %7\n\t" "s_waitcnt lgkmcnt(0)\n\tv_mul_f32 %8, %16, %8\n\tv_add_f32 %0, %0, %8"
        : "+v"(acc), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7])
        : "s"(e[1]), "s"(e[3]), "s"(e[5]), "s"(e[7]), "s"(e[9]), "s"(e[11]), "s"(e[13]), "s"(e[15]) : "memory");
    return acc;
}
__global__ void __launch_bounds__(512) k_sgpr2(const float* __restrict__ lut, const uint32_t* __restrict__ lists, float* __restrict__ out)
{
    __shared__ float s_lut[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 512) s_lut[i] = lut[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 8 + (threadIdx.x >> 6));
    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)s_lut);        // LDS offset of the table (0 here)
    asm volatile("" :: "s"(base) : "memory");
    float tot = 0.f;
    const uint32_t* L = lists + (size_t)wave * BLOCKS_PER_WAVE * NT * 2;
    uint32_t e[NT / 8][16], en[NT / 8][16];
    #pragma unroll
    for (int r = 0; r < NT / 8; r++) { _Pragma("unroll") for (int q = 0; q < 16; q++) e[r][q] = __builtin_amdgcn_readfirstlane(L[r * 16 + q]); }
    for (int b = 0; b < BLOCKS_PER_WAVE; b++) {
        const uint32_t* Ln = L + (size_t)(b + 1 < BLOCKS_PER_WAVE ? b + 1 : b) * NT * 2;
        #pragma unroll
        for (int r = 0; r < NT / 8; r++) { _Pragma("unroll") for (int q = 0; q < 16; q++) en[r][q] = __builtin_amdgcn_readfirstlane(Ln[r * 16 + q]); }   // next block's list
        float acc = 0.f, la[8], lb[8];
        reads8(la, e[0]);
        acc = step8(acc, la, e[0], lb, e[1]);
        acc = step8(acc, lb, e[1], la, e[2]);
        acc = math8(acc, la, e[2]);
        tot += acc;
        #pragma unroll
        for (int r = 0; r < NT / 8; r++) { _Pragma("unroll") for (int q = 0; q < 16; q++) e[r][q] = en[r][q]; }
    }
    out[(size_t)wave * 64 + lane] = tot;
}

int main()
{
    const int wgs = 256 * 12, waves = wgs * 8;
    std::vector<float> lut(4096); for (int i = 0; i < 4096; i++) lut[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.5f;
    const size_t nent = (size_t)waves * BLOCKS_PER_WAVE * NT;
    std::vector<uint32_t> L(nent * 2);
    uint32_t x = 12345;
    for (size_t i = 0; i < nent; i++) { x = x * 1664525u + 1013904223u; const uint32_t row = 1 + (x >> 8) % 63; x = x * 1664525u + 1013904223u; const float c = (float)((int)((x >> 10) % 200) - 100);
        L[2 * i] = row * 256u; memcpy(&L[2 * i + 1], &c, 4); }
    float *d_lut, *d_o1, *d_o2; uint32_t* d_L;
    CHECK(hipMalloc(&d_lut, 16384)); CHECK(hipMalloc(&d_L, nent * 8)); CHECK(hipMalloc(&d_o1, (size_t)waves * 256)); CHECK(hipMalloc(&d_o2, (size_t)waves * 256));
    CHECK(hipMemcpy(d_lut, lut.data(), 16384, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_L, L.data(), nent * 8, hipMemcpyHostToDevice));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        float ms1, ms2;
        hipEventRecord(a); hipLaunchKernelGGL(k_dpp, dim3(wgs), dim3(512), 0, 0, d_lut, (const uint2*)d_L, d_o1); hipEventRecord(b); CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms1, a, b);
        hipEventRecord(a); hipLaunchKernelGGL(k_sgpr, dim3(wgs), dim3(512), 0, 0, d_lut, d_L, d_o2, 0u); hipEventRecord(b); CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms2, a, b);
        const double terms = (double)nent;
        float ms3; hipEventRecord(a); hipLaunchKernelGGL(k_sgpr2, dim3(wgs), dim3(512), 0, 0, d_lut, d_L, d_o2); hipEventRecord(b); CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms3, a, b);
        printf("pipelined sgpr %.3f ms (%.2f cycles/term/CU at 2.4 GHz)\n", ms3, ms3 * 1e-3 * 2.4e9 / (terms / 256));
        printf("dpp  %.3f ms  %.3f ns/term-wave   sgpr %.3f ms  %.3f ns/term-wave   ratio %.2f\n", ms1, ms1 * 1e6 / terms, ms2, ms2 * 1e6 / terms, ms1 / ms2);
    }
    std::vector<float> o1((size_t)waves * 64), o2((size_t)waves * 64);
    CHECK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < o1.size(); i++) if (memcmp(&o1[i], &o2[i], 4)) bad++;
    printf("mismatching outputs: %zu of %zu (first: %g vs %g)\n", bad, o1.size(), o1[0], o2[0]);
    return 0;
}
