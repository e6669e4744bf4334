// Probe (round 6): cycles per step of the pair form's term loop with the coefficient delivered three ways --
//   P0  the round-5 loop: coefficient as a DPP operand of the two multiplies, row word as a DPP operand of the address add
//       (2 x 4.2 + 4.2 + 2 x 2.2 = 17 vector cycles per step, one LDS instruction);
//   P1  coefficients of two terms as one BROADCAST ds_read_b64 (all lanes of a half read the same eight bytes), plain multiplies
//       (2 x 2.2 + 4.2 + 2 x 2.2 = 13 vector cycles, 1.5 LDS instructions);
//   P2  coefficient AND row word of a term as one broadcast ds_read_b64, plain address add (11 vector cycles, 2 LDS instructions).
// The rounds of P1 / P2 come from tools/gen/gen_pair_round.py (the production rounds of P1 are the same text).  All three run the same
// synthetic lists; sums must agree bit for bit.
//   python tools/gen/gen_pair_round.py /tmp/pair_round_probe.inc --probe
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I/tmp -o /tmp/idct_bcast tools/probes/idct_bcast.hip && /tmp/idct_bcast
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "pair_round_probe.inc"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ROUNDS 4096                     // rounds per wave
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_u8_t*)p; }
__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { return *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)a; }
#define BC(I) " row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
#define S_(x) #x
#define XS(x) S_(x)

// ---- P0: as tools/probes/idct_quad.hip / the round-5 tree ------------------------------------------------------------------------------
#define P_AD(K) "v_add_u32_dpp %[ad], %[rw], %[lo]" BC(K)
#define P_RD(R) "ds_read_b64 v[" XS(V_##R) ":" XS(V_##R##_1) "], %[ad]\n\t"
#define P_MUL(R, K) "v_mul_f32_dpp v" XS(V_##R) ", %[ey], v" XS(V_##R) BC(K) "v_mul_f32_dpp v" XS(V_##R##_1) ", %[ey], v" XS(V_##R##_1) BC(K)
#define P_ADD(R) "v_add_f32 %[a0], %[a0], v" XS(V_##R) "\n\t" "v_add_f32 %[a1], %[a1], v" XS(V_##R##_1) "\n\t"
#define V_A0 48
#define V_A0_1 49
#define V_A1 50
#define V_A1_1 51
#define V_A2 52
#define V_A2_1 53
#define V_A3 54
#define V_A3_1 55
#define V_B0 56
#define V_B0_1 57
#define V_B1 58
#define V_B1_1 59
#define V_B2 60
#define V_B2_1 61
#define V_B3 62
#define V_B3_1 63
#define P_STEP(X, Y, K, KN) "s_waitcnt lgkmcnt(3)\n\t" P_MUL(X, K) P_AD(KN) P_RD(Y) P_ADD(X)
#define P_LAST(X, K, C) "s_waitcnt lgkmcnt(" #C ")\n\t" P_MUL(X, K) P_ADD(X)
#define PAIR_ROUND_P0() asm volatile( \
    P_AD(0) P_RD(A0) P_AD(1) P_RD(A1) P_AD(2) P_RD(A2) P_AD(3) P_RD(A3) \
    P_STEP(A0, B0, 0, 4) P_STEP(A1, B1, 1, 5) P_STEP(A2, B2, 2, 6) P_STEP(A3, B3, 3, 7) \
    P_STEP(B0, A0, 4, 8) P_STEP(B1, A1, 5, 9) P_STEP(B2, A2, 6, 10) P_STEP(B3, A3, 7, 11) \
    P_STEP(A0, B0, 8, 12) P_STEP(A1, B1, 9, 13) P_STEP(A2, B2, 10, 14) P_STEP(A3, B3, 11, 15) \
    P_LAST(B0, 12, 3) P_LAST(B1, 13, 2) P_LAST(B2, 14, 1) P_LAST(B3, 15, 0) \
    : [a0] "+v"(a0), [a1] "+v"(a1), [ad] "=&v"(ad) : [rw] "v"(rw), [ey] "v"(ey), [lo] "v"(lo) \
    : "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63")

// lists of a wave, per half (512 bytes): P0 / P1: 64 coefficients (fp32), 64 row words; P2: 64 entries {coefficient, row word}
template <int FORM, int OCC>
__global__ void __launch_bounds__(256, OCC) k_terms(const float* __restrict__ lut, const float* __restrict__ ey_g, const uint32_t* __restrict__ rw_g, float* __restrict__ out, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    float* s_lut = reinterpret_cast<float*>(s_dyn);
    for (int i = threadIdx.x; i < 4096; i += 256) s_lut[i] = lut[i];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 31;
    uint32_t* lst = reinterpret_cast<uint32_t*>(s_dyn + 16384 + wave * 1024);
    if (lane < 16) for (int h = 0; h < 2; h++) {
        if (FORM == 2) { lst[h * 128 + lane * 2] = __float_as_uint(ey_g[h * 16 + lane]); lst[h * 128 + lane * 2 + 1] = rw_g[h * 16 + lane]; }
        else { lst[h * 128 + lane] = __float_as_uint(ey_g[h * 16 + lane]); lst[h * 128 + 64 + lane] = rw_g[h * 16 + lane]; }
    }
    __syncthreads();
    const uint32_t a_half = lds_addr(lst) + (lane >> 5) * 512u, a_ey = a_half + (lane & 15u) * 4u, a_rw = a_half + 256u + (lane & 15u) * 4u, lo = l * 8u;
    float a0 = 0.f, a1 = 0.f; uint32_t ad;
    uint32_t nl = 16; asm volatile("" : "+s"(nl));
    #pragma nounroll
    for (int r = 0; r < rounds; r++) {
        if (FORM == 0) { const float ey = __uint_as_float(lds_r32(a_ey)); const uint32_t rw = lds_r32(a_rw); PAIR_ROUND_P0(); }
        if (FORM == 1) { uint32_t rw; asm volatile(PAIR_ROUND_ASM_0 : [acc0] "+v"(a0), [acc1] "+v"(a1), [ad] "=&v"(ad), [rw] "=&v"(rw) : [ah] "v"(a_half), [arw] "v"(a_rw), [l8] "v"(lo), [nl] "s"(nl) : "scc", PAIR_ROUND_CLOBBERS); }
        if (FORM == 2) { asm volatile(ENTRY_ROUND_ASM_0 : [acc0] "+v"(a0), [acc1] "+v"(a1), [ad] "=&v"(ad) : [ah] "v"(a_half), [l8] "v"(lo), [nl] "s"(nl) : "scc", ENTRY_ROUND_CLOBBERS); }
    }
    out[(size_t)(blockIdx.x * 256 + threadIdx.x) * 2] = a0; out[(size_t)(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = a1;
}

template <class K> static double run(K kern, int wgs, const float* lut, const float* ey, const uint32_t* rw, float* out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 16384 + 4096, 0, lut, ey, rw, out, ROUNDS); CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) { hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 16384 + 4096, 0, lut, ey, rw, out, ROUNDS); hipEventRecord(b); CHECK(hipEventSynchronize(b)); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
    CHECK(hipGetLastError());
    return best;
}
int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<float> lut(4096), ey(32); std::vector<uint32_t> rw(32);
    for (int i = 0; i < 4096; i++) lut[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.5f;
    uint32_t x = 777; for (int i = 0; i < 32; i++) { x = x * 1664525u + 1013904223u; ey[i] = (float)((int)((x >> 10) % 200) - 100); x = x * 1664525u + 1013904223u; rw[i] = (1u + (x >> 8) % 63u) * 256u; }
    float *d_lut, *d_ey, *d_out; uint32_t* d_rw;
    CHECK(hipMalloc(&d_lut, 16384)); CHECK(hipMalloc(&d_ey, 128)); CHECK(hipMalloc(&d_rw, 128)); CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 16));
    CHECK(hipMemcpy(d_lut, lut.data(), 16384, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_ey, ey.data(), 128, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rw, rw.data(), 128, hipMemcpyHostToDevice));
    const double clk = prop.clockRate * 1e3;
    printf("%d CUs at %.2f GHz; %d rounds of 16 steps per wave; cycles per step and SIMD (a workgroup of 4 waves puts one wave on each SIMD)\n", cus, clk / 1e9, ROUNDS);
    for (int occ : { 4, 6, 8 }) {
        const int wgs = cus * occ;
        double m0 = 0, m1 = 0, m2 = 0;
        if (occ == 4) { m0 = run(k_terms<0, 4>, wgs, d_lut, d_ey, d_rw, d_out); m1 = run(k_terms<1, 4>, wgs, d_lut, d_ey, d_rw, d_out); m2 = run(k_terms<2, 4>, wgs, d_lut, d_ey, d_rw, d_out); }
        if (occ == 6) { m0 = run(k_terms<0, 6>, wgs, d_lut, d_ey, d_rw, d_out); m1 = run(k_terms<1, 6>, wgs, d_lut, d_ey, d_rw, d_out); m2 = run(k_terms<2, 6>, wgs, d_lut, d_ey, d_rw, d_out); }
        if (occ == 8) { m0 = run(k_terms<0, 8>, wgs, d_lut, d_ey, d_rw, d_out); m1 = run(k_terms<1, 8>, wgs, d_lut, d_ey, d_rw, d_out); m2 = run(k_terms<2, 8>, wgs, d_lut, d_ey, d_rw, d_out); }
        const double steps_per_simd = (double)ROUNDS * 16 * occ;
        printf("%d waves per SIMD:  P0 (DPP coefficient) %6.2f   P1 (coefficient pairs from LDS) %6.2f   P2 (entries from LDS) %6.2f\n", occ,
               m0 * 1e-3 * clk / steps_per_simd, m1 * 1e-3 * clk / steps_per_simd, m2 * 1e-3 * clk / steps_per_simd);
    }
    std::vector<float> o0(512), o1(512), o2(512);
    hipLaunchKernelGGL((k_terms<0, 8>), dim3(1), dim3(256), 16384 + 4096, 0, d_lut, d_ey, d_rw, d_out, 3); CHECK(hipMemcpy(o0.data(), d_out, 2048, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((k_terms<1, 8>), dim3(1), dim3(256), 16384 + 4096, 0, d_lut, d_ey, d_rw, d_out, 3); CHECK(hipMemcpy(o1.data(), d_out, 2048, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((k_terms<2, 8>), dim3(1), dim3(256), 16384 + 4096, 0, d_lut, d_ey, d_rw, d_out, 3); CHECK(hipMemcpy(o2.data(), d_out, 2048, hipMemcpyDeviceToHost));
    int b1 = 0, b2 = 0; for (int i = 0; i < 512; i++) { b1 += memcmp(&o0[i], &o1[i], 4) != 0; b2 += memcmp(&o0[i], &o2[i], 4) != 0; }
    printf("sums of one workgroup, three rounds: P1 differs from P0 in %d of 512, P2 in %d of 512 (sample %.6f)\n", b1, b2, o0[5]);
    return 0;
}
