// Probe: the issue cost of one step of the back end's term loop in the PAIR form of the tree (two blocks per wave: lane = two
// samples, one ds_read_b64, two DPP multiplies, two adds, one DPP address add per term of both blocks) against a QUAD form (four blocks per wave:
// lane = four samples of one 16-lane row = one block; per term of all four blocks ONE DPP move of the coefficient into a VGPR, one DPP address add,
// one ds_read_b128, four plain multiplies, four plain adds).  Both loops are the 16-step software-pipelined rounds of k_idct_color (four table
// reads in flight, lists of 16 coefficients / row words per round read from LDS), run on synthetic lists of NT terms per block with the table in
// LDS; what is measured is shader cycles per step and per block-term at 4 / 6 / 8 waves per SIMD.  tools/term_padding.py has the other half of the
// question: how many steps each form executes on the bench pictures (lock-step padding).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/idct_quad tools/probes/idct_quad.hip && /tmp/idct_quad
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NT 16                           // terms per block and list (one round)
#define ROUNDS 4096                     // rounds per wave
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_u8_t*)p; }
__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { return *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)a; }
#define BC(I) " row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
#define S_(x) #x
#define XS(x) S_(x)

// ---- pair form: table pairs in v[48:63] ---------------------------------------------------------------------------------------------------
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
#define PAIR_ROUND() asm volatile( \
    P_AD(0) P_RD(A0) P_AD(1) P_RD(A1) P_AD(2) P_RD(A2) P_AD(3) P_RD(A3) \
    P_STEP(A0, B0, 0, 4) P_STEP(A1, B1, 1, 5) P_STEP(A2, B2, 2, 6) P_STEP(A3, B3, 3, 7) \
    P_STEP(B0, A0, 4, 8) P_STEP(B1, A1, 5, 9) P_STEP(B2, A2, 6, 10) P_STEP(B3, A3, 7, 11) \
    P_STEP(A0, B0, 8, 12) P_STEP(A1, B1, 9, 13) P_STEP(A2, B2, 10, 14) P_STEP(A3, B3, 11, 15) \
    P_LAST(B0, 12, 3) P_LAST(B1, 13, 2) P_LAST(B2, 14, 1) P_LAST(B3, 15, 0) \
    : [a0] "+v"(a0), [a1] "+v"(a1), [ad] "=&v"(ad) : [rw] "v"(rw), [ey] "v"(ey), [lo] "v"(lo) \
    : "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63")

template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_pair(const float* __restrict__ lut, const float* __restrict__ ey_g, const uint32_t* __restrict__ rw_g, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    float* s_lut = reinterpret_cast<float*>(s_dyn);
    for (int i = threadIdx.x; i < 4096; i += 256) s_lut[i] = lut[i];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 31;
    uint32_t* lst = reinterpret_cast<uint32_t*>(s_dyn + 16384 + wave * 1024);      // per wave, per half: 16 coefficients, then 16 row words (two lists x 2 halves)
    if (lane < 16) { lst[lane] = __float_as_uint(ey_g[lane]); lst[64 + lane] = rw_g[lane]; lst[128 + lane] = __float_as_uint(ey_g[16 + lane]); lst[192 + lane] = rw_g[16 + lane]; }
    __syncthreads();
    const uint32_t a_half = lds_addr(lst) + (lane >> 5) * 512u, a_ey = a_half + (lane & 15u) * 4u, a_rw = a_half + 256u + (lane & 15u) * 4u, lo = l * 8u;
    float a0 = 0.f, a1 = 0.f; uint32_t ad;
    #pragma nounroll
    for (int r = 0; r < ROUNDS; r++) {
        const float ey = __uint_as_float(lds_r32(a_ey)); const uint32_t rw = lds_r32(a_rw);
        PAIR_ROUND();
    }
    out[(size_t)(blockIdx.x * 256 + threadIdx.x) * 2] = a0; out[(size_t)(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = a1;
}

// ---- quad form: table quads in v[32:63] (eight in flight would need 32 registers; four in flight = v[48:63]) -------------------------------
#define Q_AD(K) "v_add_u32_dpp %[ad], %[rw], %[lo]" BC(K)
#define Q_RD(R) "ds_read_b128 v[" XS(V_##R) ":" XS(V_##R##_3) "], %[ad]\n\t"
#define Q_BC(K) "v_mov_b32_dpp %[c], %[ey]" BC(K)
#define Q_MUL(R) "v_mul_f32 v" XS(V_##R) ", %[c], v" XS(V_##R) "\n\t" "v_mul_f32 v" XS(V_##R##_1) ", %[c], v" XS(V_##R##_1) "\n\t" "v_mul_f32 v" XS(V_##R##_2) ", %[c], v" XS(V_##R##_2) "\n\t" "v_mul_f32 v" XS(V_##R##_3) ", %[c], v" XS(V_##R##_3) "\n\t"
#define Q_ADD(R) "v_add_f32 %[a0], %[a0], v" XS(V_##R) "\n\t" "v_add_f32 %[a1], %[a1], v" XS(V_##R##_1) "\n\t" "v_add_f32 %[a2], %[a2], v" XS(V_##R##_2) "\n\t" "v_add_f32 %[a3], %[a3], v" XS(V_##R##_3) "\n\t"
#define V_QA 48
#define V_QA_1 49
#define V_QA_2 50
#define V_QA_3 51
#define V_QB 52
#define V_QB_1 53
#define V_QB_2 54
#define V_QB_3 55
#define V_QC 56
#define V_QC_1 57
#define V_QC_2 58
#define V_QC_3 59
#define V_QD 60
#define V_QD_1 61
#define V_QD_2 62
#define V_QD_3 63
// four reads in flight in four register quads, each re-used as soon as its products have been added
#define Q_STEP(X, K, KN) "s_waitcnt lgkmcnt(3)\n\t" Q_BC(K) Q_MUL(X) Q_AD(KN) Q_ADD(X) Q_RD(X)
#define Q_LAST(X, K, C) "s_waitcnt lgkmcnt(" #C ")\n\t" Q_BC(K) Q_MUL(X) Q_ADD(X)
#define QUAD_ROUND() asm volatile( \
    Q_AD(0) Q_RD(QA) Q_AD(1) Q_RD(QB) Q_AD(2) Q_RD(QC) Q_AD(3) Q_RD(QD) \
    Q_STEP(QA, 0, 4) Q_STEP(QB, 1, 5) Q_STEP(QC, 2, 6) Q_STEP(QD, 3, 7) Q_STEP(QA, 4, 8) Q_STEP(QB, 5, 9) Q_STEP(QC, 6, 10) Q_STEP(QD, 7, 11) \
    Q_STEP(QA, 8, 12) Q_STEP(QB, 9, 13) Q_STEP(QC, 10, 14) Q_STEP(QD, 11, 15) \
    Q_LAST(QA, 12, 3) Q_LAST(QB, 13, 2) Q_LAST(QC, 14, 1) Q_LAST(QD, 15, 0) \
    : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [ad] "=&v"(ad), [c] "=&v"(c) : [rw] "v"(rw), [ey] "v"(ey), [lo] "v"(lo) \
    : "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63")

template <int OCC>
__global__ void __launch_bounds__(256, OCC) k_quad(const float* __restrict__ lut, const float* __restrict__ ey_g, const uint32_t* __restrict__ rw_g, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    float* s_lut = reinterpret_cast<float*>(s_dyn);
    for (int i = threadIdx.x; i < 4096; i += 256) s_lut[i] = lut[i];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane & 15, row = lane >> 4;
    uint32_t* lst = reinterpret_cast<uint32_t*>(s_dyn + 16384 + wave * 1024);      // per wave, per 16-lane row (block): 16 coefficients, 16 row words
    if (lane < 16) for (int q = 0; q < 4; q++) { lst[q * 32 + lane] = __float_as_uint(ey_g[(q & 1) * 16 + lane]); lst[q * 32 + 16 + lane] = rw_g[(q & 1) * 16 + lane]; }
    __syncthreads();
    const uint32_t a_ey = lds_addr(lst) + row * 128u + l * 4u, a_rw = a_ey + 64u, lo = l * 16u;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, c; uint32_t ad;
    #pragma nounroll
    for (int r = 0; r < ROUNDS; r++) {
        const float ey = __uint_as_float(lds_r32(a_ey)); const uint32_t rw = lds_r32(a_rw);
        QUAD_ROUND();
    }
    float* o = out + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4; o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
}

template <class K> static double run(K kern, int wgs, const float* lut, const float* ey, const uint32_t* rw, float* out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 16384 + 4096, 0, lut, ey, rw, out); CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) { hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 16384 + 4096, 0, lut, ey, rw, out); hipEventRecord(b); CHECK(hipEventSynchronize(b)); float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; }
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
    const double clk = prop.clockRate * 1e3;                       // Hz
    printf("%d CUs at %.2f GHz; %d rounds of 16 steps per wave; cycles are per SIMD (a workgroup of 4 waves puts one wave on each SIMD)\n", cus, clk / 1e9, ROUNDS);
    for (int occ : { 4, 6, 8 }) {
        const int wgs = cus * occ;                                  // occ waves per SIMD
        double mp = 0, mq = 0;
        if (occ == 4) { mp = run(k_pair<4>, wgs, d_lut, d_ey, d_rw, d_out); mq = run(k_quad<4>, wgs, d_lut, d_ey, d_rw, d_out); }
        if (occ == 6) { mp = run(k_pair<6>, wgs, d_lut, d_ey, d_rw, d_out); mq = run(k_quad<6>, wgs, d_lut, d_ey, d_rw, d_out); }
        if (occ == 8) { mp = run(k_pair<8>, wgs, d_lut, d_ey, d_rw, d_out); mq = run(k_quad<8>, wgs, d_lut, d_ey, d_rw, d_out); }
        const double steps_per_simd = (double)ROUNDS * 16 * occ;    // steps issued on one SIMD
        const double cp = mp * 1e-3 * clk / steps_per_simd, cq = mq * 1e-3 * clk / steps_per_simd;
        printf("%d waves per SIMD:  pair step %6.2f cycles = %5.2f per block-term   |   quad step %6.2f cycles = %5.2f per block-term\n", occ, cp, cp / 2, cq, cq / 4);
    }
    // the pair form's sums of wave 0 against the quad form's (same lists: block A / B of the pair = rows 0 / 1 of the quad): bit for bit
    std::vector<float> op(256 * 2), oq(256 * 4);
    hipLaunchKernelGGL(k_pair<8>, dim3(1), dim3(256), 16384 + 4096, 0, d_lut, d_ey, d_rw, d_out); CHECK(hipMemcpy(op.data(), d_out, op.size() * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_quad<8>, dim3(1), dim3(256), 16384 + 4096, 0, d_lut, d_ey, d_rw, d_out); CHECK(hipMemcpy(oq.data(), d_out, oq.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int blk = 0; blk < 2; blk++) for (int s = 0; s < 64; s++) {
        const float p = op[((blk * 32 + s / 2) * 2) + (s & 1)], q = oq[((blk * 16 + s / 4) * 4) + (s & 3)];
        if (memcmp(&p, &q, 4)) bad++;
    }
    printf("pair against quad, 128 sums of wave 0: %d differ\n", bad);
    return 0;
}
