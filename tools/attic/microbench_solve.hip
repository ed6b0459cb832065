// Diagnostic build: where does a chain step spend its cycles?  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVBA_STAMPS -I vinsat_amd/csrc tools/microbench_solve.hip -o /tmp/mb
#include <cstdio>
#include <vector>
#include <random>
#include "../vinsat_amd/csrc/vba_solve.hip"

using namespace vba;
#ifndef PIVOT_MB
#define PIVOT_MB false
#endif

__global__ __launch_bounds__(64) void k_time(const double* bands, const double* rhs, int n, double* Xs, double* zs, double* x, long long* st) {
    __shared__ double blk[2][256];
    bool zp = false;
    const BandSource src{bands, rhs};
    chain_solve<PIVOT_MB, false>(src, n, 1e-4, Xs, zs, x, blk, threadIdx.x, zp, st);
}

int main() {
    const int n = 500;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> N01;
    std::vector<double> bands(n * 243), rhs(n * 9);
    for (int i = 0; i < n; ++i) {
        double G[9][9];
        for (auto& r : G) for (auto& v : r) v = N01(rng);
        for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) {
            double s = 0; for (int k = 0; k < 9; ++k) s += G[a][k] * G[b][k];
            bands[i * 243 + 81 + a * 9 + b] = s + (a == b ? 20.0 : 0.0);
            bands[i * 243 + a * 9 + b] = i > 0 ? 0.3 * N01(rng) : 0.0;
            bands[i * 243 + 162 + a * 9 + b] = i < n - 1 ? 0.3 * N01(rng) : 0.0;
        }
        for (int a = 0; a < 9; ++a) rhs[i * 9 + a] = N01(rng);
    }
    double *db, *dr, *dX, *dz, *dx; long long* ds;
    hipMalloc(&db, bands.size() * 8); hipMalloc(&dr, rhs.size() * 8); hipMalloc(&dX, n * 81 * 8); hipMalloc(&dz, n * 9 * 8); hipMalloc(&dx, n * 9 * 8); hipMalloc(&ds, 64);
    hipMemcpy(db, bands.data(), bands.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dr, rhs.data(), rhs.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_time, dim3(1), dim3(64), 0, 0, db, dr, n, dX, dz, dx, ds);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[3]; hipMemcpy(st, ds, 24, hipMemcpyDeviceToHost);
        printf("n=%d kernel %.3f ms; forward %lld cyc (%.0f/block), backward %lld cyc (%.0f/block)\n", n, ms, st[1] - st[0], double(st[1] - st[0]) / n, st[2] - st[1], double(st[2] - st[1]) / n);
    }
    // residual check
    std::vector<double> x(n * 9); hipMemcpy(x.data(), dx, n * 72, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < n; ++i) for (int a = 0; a < 9; ++a) {
        double s = -rhs[i * 9 + a];
        for (int b = 0; b < 9; ++b) {
            s += (bands[i * 243 + 81 + a * 9 + b] + (a == b ? (double)(float)1e-4 : 0.0)) * x[i * 9 + b];
            if (i > 0) s += bands[i * 243 + a * 9 + b] * x[(i - 1) * 9 + b];
            if (i < n - 1) s += bands[i * 243 + 162 + a * 9 + b] * x[(i + 1) * 9 + b];
        }
        worst = fmax(worst, fabs(s));
    }
    printf("max residual %.3e\n", worst);
    return 0;
}
