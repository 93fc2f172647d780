// Layout probe for v_mfma_f32_4x4x1_16B_f32 on gfx950: which (block, row, column) each lane's A / B operand and each of its four result
// registers belong to.  A = one-hot in lane la, B = one-hot in lane lb: the non-zero results tell the mapping.  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float *da, *db, *dd, ha[64], hb[64], hd[256];
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    // general check: A[l] = 1 + l, B[l] = 100 + l  -> d[l][r] should be A[lane of (block, row r)] * B[l]
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + l; }
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            // find (la) with ha[la] * hb[lb] == hd for lb = l
            int found = -1;
            for (int la = 0; la < 64; ++la) if (ha[la] * hb[l] == hd[l * 4 + r]) found = la;
            if (l < 8 || l % 16 == 0) printf("lane %2d reg %d = A[lane %2d] * B[lane %2d]\n", l, r, found, l);
            if (found != (l / 4) * 4 + r) ok = 0;
        }
    printf("hypothesis D[lane l][reg r] = A[4*(l/4) + r] * B[l]: %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
