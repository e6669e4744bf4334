// Probe: IDCT term loops that keep the 64 x 64 cosine table in REGISTERS (lane = output sample, v[64:127] = the 64 rows) and
// pick the row of each term through the VGPR-index mode (s_set_gpr_idx_idx -> M0), against the production loop of
// k_idct_color (table in LDS, row offset and coefficient broadcast as DPP operands: 3 VALU + 1 LDS read per term).
//   (a) k_dpp      production loop
//   (d) k_idx_rl   rows packed four to a dword in a VGPR, one v_readlane per four terms, coefficient by DPP row broadcast:
//                  2.25 VALU + 1.75 SALU per term, no LDS
//   (e) k_idx_sl   (row, coefficient) pairs arrive in SGPRs by scalar loads (a list another kernel wrote to memory):
//                  2 VALU + 1 SALU per term, no LDS
// The add is issued as v_fma_f32 acc, t, 1.0, acc: t * 1.0 is exact, so the result is the separately rounded sum; its src1 is a
// constant, which the index mode (SRC1_REL) leaves alone.  All three must agree bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/idct_terms2 tools/probes/idct_terms2.hip && /tmp/idct_terms2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NT 24            // terms per block (multiple of 8)
#define BLOCKS_PER_WAVE 256
#define THREADS 256

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

__global__ void __launch_bounds__(THREADS) k_dpp(const float* __restrict__ lut, const uint2* __restrict__ lists, float* __restrict__ out, uint32_t nsets)
{
    __shared__ float s_lut[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += THREADS) s_lut[i] = lut[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6), lane4 = lane * 4, li = lane & 15;
    const char* lut_b = reinterpret_cast<const char*>(s_lut);
    float tot = 0.f;
    for (int b = 0; b < BLOCKS_PER_WAVE; b++) {
        const uint2* L = lists + ((size_t)(wave % nsets) * BLOCKS_PER_WAVE + b) * NT;
        float acc = 0.f;
        for (int r = 0; r < NT; r += 16) {
            const uint2 e = L[r + li < NT ? r + li : 0];
            G4(e, 0) G4(e, 4) if (r + 8 >= NT) break; G4(e, 8) G4(e, 12)
        }
        tot += acc;
    }
    out[(size_t)wave * 64 + lane] = tot;
}

#define CLOB64 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
               "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
__device__ __forceinline__ void table_to_regs(const float* __restrict__ lut, uint32_t lane)
{
    for (uint32_t r = 0; r < 64; r++) {                       // v[64 + r] <- lut[r][lane], written through the destination index
        const float v = lut[r * 64 + lane]; const uint32_t rs = __builtin_amdgcn_readfirstlane(r);
        asm volatile("s_set_gpr_idx_on %1, 8\n\tv_mov_b32 v64, %0\n\ts_set_gpr_idx_off" :: "v"(v), "s"(rs) : CLOB64, "m0");
    }
}

// (d) four terms: packed row bytes in lane Q of `rows`, coefficients in lanes I..I+3 of every 16-lane row of `ey`
#define T4D(Q, I0, I1, I2, I3) \
    "v_readlane_b32 s40, %[rows], " #Q "\n\t" \
    "s_set_gpr_idx_idx s40\n\tv_mul_f32_dpp %[t0], %[ey], v64 row_newbcast:" #I0 " row_mask:0xf bank_mask:0xf\n\ts_lshr_b32 s41, s40, 8\n\tv_fma_f32 %[acc], %[t0], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx s41\n\tv_mul_f32_dpp %[t1], %[ey], v64 row_newbcast:" #I1 " row_mask:0xf bank_mask:0xf\n\ts_lshr_b32 s42, s40, 16\n\tv_fma_f32 %[acc], %[t1], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx s42\n\tv_mul_f32_dpp %[t0], %[ey], v64 row_newbcast:" #I2 " row_mask:0xf bank_mask:0xf\n\ts_lshr_b32 s43, s40, 24\n\tv_fma_f32 %[acc], %[t0], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx s43\n\tv_mul_f32_dpp %[t1], %[ey], v64 row_newbcast:" #I3 " row_mask:0xf bank_mask:0xf\n\tv_fma_f32 %[acc], %[t1], 1.0, %[acc]\n\t"
#define D8(QA, QB, I) \
    asm volatile("s_nop 1\n\ts_set_gpr_idx_on %[z], 2\n\t" T4D(QA, I, I + 1, I + 2, I + 3) T4D(QB, I + 4, I + 5, I + 6, I + 7) "s_set_gpr_idx_off" \
                 : [acc] "+v"(acc), [t0] "=&v"(t0), [t1] "=&v"(t1) : [rows] "v"(rows), [ey] "v"(ey), [z] "s"(zero) : CLOB64, "m0", "scc", "s40", "s41", "s42", "s43")
// macro arguments are pasted as tokens: spell the lane numbers out
#define T4D_(Q, I0, I1, I2, I3) T4D(Q, I0, I1, I2, I3)
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_num_vgpr(56))) k_idx_rl(const float* __restrict__ lut, const float* __restrict__ coefs,
                                                                                          const uint32_t* __restrict__ rows4, float* __restrict__ out, uint32_t nsets)
{
    const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6), li = lane & 15;
    table_to_regs(lut, lane);
    const uint32_t zero = __builtin_amdgcn_readfirstlane(0u);
    float tot = 0.f;
    for (int b = 0; b < BLOCKS_PER_WAVE; b++) {
        const size_t blk = (size_t)(wave % nsets) * BLOCKS_PER_WAVE + b;
        const float* C = coefs + blk * NT;
        const uint32_t rows = rows4[blk * (NT / 4) + (lane < NT / 4 ? lane : 0)];
        float acc = 0.f, t0, t1;
        {
            const float ey = C[li];
            asm volatile("s_nop 1\n\ts_set_gpr_idx_on %[z], 2\n\t" T4D(0, 0, 1, 2, 3) T4D(1, 4, 5, 6, 7) T4D(2, 8, 9, 10, 11) T4D(3, 12, 13, 14, 15) "s_set_gpr_idx_off"
                         : [acc] "+v"(acc), [t0] "=&v"(t0), [t1] "=&v"(t1) : [rows] "v"(rows), [ey] "v"(ey), [z] "s"(zero) : CLOB64, "m0", "scc", "s40", "s41", "s42", "s43");
        }
        {
            const float ey = C[16 + (li < NT - 16 ? li : 0)];
            asm volatile("s_nop 1\n\ts_set_gpr_idx_on %[z], 2\n\t" T4D(4, 0, 1, 2, 3) T4D(5, 4, 5, 6, 7) "s_set_gpr_idx_off"
                         : [acc] "+v"(acc), [t0] "=&v"(t0), [t1] "=&v"(t1) : [rows] "v"(rows), [ey] "v"(ey), [z] "s"(zero) : CLOB64, "m0", "scc", "s40", "s41", "s42", "s43");
        }
        tot += acc;
    }
    out[(size_t)wave * 64 + lane] = tot;
}

// (e) eight terms from sixteen SGPRs {row, coefficient bits} x 8
#define E8 \
    "s_set_gpr_idx_on %[r0], 2\n\tv_mul_f32 %[t0], %[c0], v64\n\t" \
    "s_set_gpr_idx_idx %[r1]\n\tv_mul_f32 %[t1], %[c1], v64\n\tv_fma_f32 %[acc], %[t0], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx %[r2]\n\tv_mul_f32 %[t0], %[c2], v64\n\tv_fma_f32 %[acc], %[t1], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx %[r3]\n\tv_mul_f32 %[t1], %[c3], v64\n\tv_fma_f32 %[acc], %[t0], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx %[r4]\n\tv_mul_f32 %[t0], %[c4], v64\n\tv_fma_f32 %[acc], %[t1], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx %[r5]\n\tv_mul_f32 %[t1], %[c5], v64\n\tv_fma_f32 %[acc], %[t0], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx %[r6]\n\tv_mul_f32 %[t0], %[c6], v64\n\tv_fma_f32 %[acc], %[t1], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_idx %[r7]\n\tv_mul_f32 %[t1], %[c7], v64\n\tv_fma_f32 %[acc], %[t0], 1.0, %[acc]\n\t" \
    "s_set_gpr_idx_off\n\tv_fma_f32 %[acc], %[t1], 1.0, %[acc]"
__device__ __forceinline__ float terms8(float acc, const uint32_t (&e)[16])
{
    float t0, t1;
    asm volatile(E8 : [acc] "+v"(acc), [t0] "=&v"(t0), [t1] "=&v"(t1)
                 : [r0] "s"(e[0]), [c0] "s"(e[1]), [r1] "s"(e[2]), [c1] "s"(e[3]), [r2] "s"(e[4]), [c2] "s"(e[5]), [r3] "s"(e[6]), [c3] "s"(e[7]),
                   [r4] "s"(e[8]), [c4] "s"(e[9]), [r5] "s"(e[10]), [c5] "s"(e[11]), [r6] "s"(e[12]), [c6] "s"(e[13]), [r7] "s"(e[14]), [c7] "s"(e[15]) : CLOB64, "m0");
    return acc;
}
__device__ __forceinline__ void load16(uint32_t (&e)[16], const uint32_t* __restrict__ p)
{
    #pragma unroll
    for (int q = 0; q < 16; q++) e[q] = __builtin_amdgcn_readfirstlane(p[q]);
}
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_num_vgpr(56))) k_idx_sl(const float* __restrict__ lut, const uint32_t* __restrict__ lists, float* __restrict__ out, uint32_t nsets)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6));
    table_to_regs(lut, lane);
    const uint32_t* L = lists + (size_t)(wave % nsets) * BLOCKS_PER_WAVE * NT * 2;        // chunk c (8 terms) at L + 16 c
    float tot = 0.f;
    uint32_t A[16], B[16];
    load16(A, L);
    constexpr int NCH = BLOCKS_PER_WAVE * (NT / 8);
    for (int c = 0; c < NCH; c += 6) {                       // two blocks (six chunks) per trip: the chunk after the one in work is always in flight
        float acc = 0.f;
        load16(B, L + 16 * (c + 1)); acc = terms8(acc, A);
        load16(A, L + 16 * (c + 2)); acc = terms8(acc, B);
        load16(B, L + 16 * (c + 3)); acc = terms8(acc, A);
        tot += acc; acc = 0.f;
        load16(A, L + 16 * (c + 4)); acc = terms8(acc, B);
        load16(B, L + 16 * (c + 5)); acc = terms8(acc, A);
        load16(A, L + 16 * (c + 6 < NCH ? c + 6 : c)); acc = terms8(acc, B);
        tot += acc;
    }
    out[(size_t)wave * 64 + lane] = tot;
}

int main(int argc, char** argv)
{
    const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 16;
    uint32_t nsets = argc > 2 ? (uint32_t)atoi(argv[2]) : 0u;   // distinct list sets (0: one per wave, streamed from HBM; small: L2 resident)
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, wgs = cus * wgs_per_cu, waves = wgs * (THREADS / 64);
    std::vector<float> lut(4096); for (int i = 0; i < 4096; i++) lut[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f - 0.5f;
    const size_t nblk = (size_t)waves * BLOCKS_PER_WAVE, nent = nblk * NT;
    std::vector<uint32_t> La(nent * 2), Le(nent * 2), R4(nent / 4); std::vector<float> C(nent);
    uint32_t x = 12345;
    for (size_t i = 0; i < nent; i++) {
        x = x * 1664525u + 1013904223u; const uint32_t row = 1 + (x >> 8) % 63; x = x * 1664525u + 1013904223u; const float c = (float)((int)((x >> 10) % 200) - 100);
        La[2 * i] = row * 256u; memcpy(&La[2 * i + 1], &c, 4);
        Le[2 * i] = row; memcpy(&Le[2 * i + 1], &c, 4);
        C[i] = c; R4[i / 4] = (i % 4 ? R4[i / 4] : 0u) | (row << (8 * (i % 4)));
    }
    float *d_lut, *d_C, *d_o[3]; uint32_t *d_La, *d_Le, *d_R4;
    CHECK(hipMalloc(&d_lut, 16384)); CHECK(hipMalloc(&d_La, nent * 8)); CHECK(hipMalloc(&d_Le, nent * 8 + 4096)); CHECK(hipMalloc(&d_C, nent * 4 + 256)); CHECK(hipMalloc(&d_R4, nent + 256));
    for (auto& o : d_o) CHECK(hipMalloc(&o, (size_t)waves * 256));
    CHECK(hipMemcpy(d_lut, lut.data(), 16384, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_La, La.data(), nent * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_Le, Le.data(), nent * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_C, C.data(), nent * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_R4, R4.data(), nent, hipMemcpyHostToDevice));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double terms = (double)nent; if (!nsets) nsets = (uint32_t)waves;
    printf("%d CUs, %d workgroups of %d threads, %d terms per block, %.0f M terms, %u distinct list sets\n", cus, wgs, THREADS, NT, terms / 1e6, nsets);
    for (int rep = 0; rep < 3; rep++) {
        float ms[3];
        hipEventRecord(a); hipLaunchKernelGGL(k_dpp, dim3(wgs), dim3(THREADS), 0, 0, d_lut, (const uint2*)d_La, d_o[0], nsets); hipEventRecord(b); CHECK(hipGetLastError()); CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms[0], a, b);
        hipEventRecord(a); hipLaunchKernelGGL(k_idx_rl, dim3(wgs), dim3(THREADS), 0, 0, d_lut, d_C, d_R4, d_o[1], nsets); hipEventRecord(b); CHECK(hipGetLastError()); CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms[1], a, b);
        hipEventRecord(a); hipLaunchKernelGGL(k_idx_sl, dim3(wgs), dim3(THREADS), 0, 0, d_lut, d_Le, d_o[2], nsets); hipEventRecord(b); CHECK(hipGetLastError()); CHECK(hipEventSynchronize(b)); hipEventElapsedTime(&ms[2], a, b);
        const char* nm[3] = { "(a) dpp + LDS table   ", "(d) idx, readlane rows", "(e) idx, scalar lists " };
        for (int k = 0; k < 3; k++) printf("%s %8.3f ms  %6.2f cycles per term and CU at 2.4 GHz\n", nm[k], ms[k], ms[k] * 1e-3 * 2.4e9 / (terms / cus));
    }
    std::vector<float> o[3]; for (int k = 0; k < 3; k++) { o[k].resize((size_t)waves * 64); CHECK(hipMemcpy(o[k].data(), d_o[k], o[k].size() * 4, hipMemcpyDeviceToHost)); }
    for (int k = 1; k < 3; k++) { size_t bad = 0; for (size_t i = 0; i < o[0].size(); i++) if (memcmp(&o[0][i], &o[k][i], 4)) bad++;
        printf("variant %d vs (a): %zu of %zu outputs differ (first: %g vs %g)\n", k, bad, o[0].size(), o[0][0], o[k][0]); }
    return 0;
}
