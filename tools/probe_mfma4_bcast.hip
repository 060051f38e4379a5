// Does v_mfma_f64_4x4x4_4b_f64 honour the A-operand broadcast controls (CBSZ / ABID)? With CBSZ = 2 the A block ABID
// should serve all four blocks: D_b = A_abid * B_b + C_b. Checked on random operands against the plain lane layout
//   A[b][i][k] <- lane 16k + 4b + i,   B[b][k][j] <- lane 16k + 4b + j,   D[b][i][j] -> lane 16i + 4b + j   (probe_mfma4.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
template <int CBSZ, int ABID> __global__ void run(const double* a, const double* b, double* d)
{
    const int lane = threadIdx.x;
    d[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], 0.0, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID> int check(const std::vector<double>& A, const std::vector<double>& B, double* da, double* db, double* dd)
{
    hipLaunchKernelGGL((run<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<double> D(64);
    hipMemcpy(D.data(), dd, 64 * sizeof(double), hipMemcpyDeviceToHost);
    // candidate semantics: source block of A for destination block b
    int ok_bcast = 1, ok_plain = 1;
    for (int b = 0; b < 4; ++b)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double e_plain = 0, e_bcast = 0;
                const int group = CBSZ == 0 ? b : ((b >> CBSZ) << CBSZ) + (ABID & ((1 << CBSZ) - 1));
                for (int k = 0; k < 4; ++k) {
                    e_plain += A[16 * k + 4 * b + i] * B[16 * k + 4 * b + j];
                    e_bcast += A[16 * k + 4 * group + i] * B[16 * k + 4 * b + j];
                }
                const double got = D[16 * i + 4 * b + j];
                if (std::fabs(got - e_plain) > 1e-12) ok_plain = 0;
                if (std::fabs(got - e_bcast) > 1e-12) ok_bcast = 0;
            }
    printf("cbsz %d abid %d: plain %s, broadcast-of-block-%d %s\n", CBSZ, ABID, ok_plain ? "MATCH" : "no", ABID, ok_bcast ? "MATCH" : "no");
    return ok_bcast;
}
int main()
{
    std::vector<double> A(64), B(64);
    srand(3);
    for (int l = 0; l < 64; ++l) { A[l] = rand() / (double)RAND_MAX - 0.5; B[l] = rand() / (double)RAND_MAX - 0.5; }
    double *da, *db, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, A.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(db, B.data(), 512, hipMemcpyHostToDevice);
    int ok = 1;
    check<0, 0>(A, B, da, db, dd);
    ok &= check<2, 0>(A, B, da, db, dd);
    ok &= check<2, 1>(A, B, da, db, dd);
    ok &= check<2, 2>(A, B, da, db, dd);
    ok &= check<2, 3>(A, B, da, db, dd);
    check<1, 0>(A, B, da, db, dd);
    check<1, 1>(A, B, da, db, dd);
    printf(ok ? "BROADCAST SUPPORTED\n" : "BROADCAST NOT AS EXPECTED\n");
    return 0;
}
