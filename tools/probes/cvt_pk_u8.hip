// Probe: does v_cvt_pk_u8_f32 equal "clamp to [0,255] then truncate" on gfx950?  (candidate for the DIB byte packing)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* x, unsigned* a, unsigned* b, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    a[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u) & 255u;
    b[i] = (unsigned)(int)__builtin_amdgcn_fmed3f(x[i], 0.0f, 255.0f);
}
int main()
{
    const int n = 1 << 22; float* hx = new float[n];
    for (int i = 0; i < n; i++) hx[i] = -64.0f + (float)i * (400.0f / n);
    hx[0] = -0.0f; hx[1] = 254.99998f; hx[2] = 255.00002f; hx[3] = 0.99999994f; hx[4] = -0.5f; hx[5] = 255.5f; hx[6] = 1e9f; hx[7] = -1e9f;
    float* dx; unsigned *da, *db; hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, n);
    unsigned* ha = new unsigned[n]; unsigned* hb = new unsigned[n];
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    long bad = 0; for (int i = 0; i < n; i++) if (ha[i] != hb[i]) { if (bad < 8) printf("x=%.8f pk=%u ref=%u\n", hx[i], ha[i], hb[i]); bad++; }
    printf("mismatches: %ld of %d\n", bad, n);
    return 0;
}
