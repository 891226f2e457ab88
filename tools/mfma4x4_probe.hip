// Empirical lane maps of v_mfma_f64_4x4x4_4b_f64 (and its CBSZ/ABID A-broadcast) on gfx950:
// for every (A lane, B lane) one-hot pair, which D lanes become non-zero?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void k(const double *a, const double *b, double *d, int pairs) {
    int l = threadIdx.x;
    for (int p = blockIdx.x; p < pairs; p += gridDim.x) {
        double acc = 0.0;
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[p * 64 + l], b[p * 64 + l], acc, CBSZ, ABID, 0);
        d[p * 64 + l] = acc;
    }
}
template <int CBSZ, int ABID> void probe() {
    const int pairs = 64 * 64;
    std::vector<double> ha(pairs * 64, 0.0), hb(pairs * 64, 0.0), hd(pairs * 64);
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) { ha[(la * 64 + lb) * 64 + la] = 1.0; hb[(la * 64 + lb) * 64 + lb] = 1.0; }
    double *da, *db, *dd;
    hipMalloc(&da, ha.size() * 8); hipMalloc(&db, hb.size() * 8); hipMalloc(&dd, hd.size() * 8);
    hipMemcpy(da, ha.data(), ha.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(256), dim3(64), 0, 0, da, db, dd, pairs);
    hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost);
    printf("# cbsz=%d abid=%d : lines 'la lb -> d lanes'\n", CBSZ, ABID);
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            bool any = false;
            for (int l = 0; l < 64; ++l) if (hd[(la * 64 + lb) * 64 + l] != 0.0) any = true;
            if (!any) continue;
            printf("%d %d ->", la, lb);
            for (int l = 0; l < 64; ++l) if (hd[(la * 64 + lb) * 64 + l] != 0.0) printf(" %d", l);
            printf("\n");
        }
    hipFree(da); hipFree(db); hipFree(dd);
}
int main() { probe<0, 0>(); probe<2, 0>(); probe<2, 1>(); probe<2, 3>(); probe<1, 1>(); return 0; }
