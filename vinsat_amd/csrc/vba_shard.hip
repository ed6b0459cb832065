// vba_shard.hip -- glue kernels of the observation-sharded multi-GPU mode.
//
// Every rank owns a contiguous slice of the observation rows.  Per BA() call three small device buffers are
// exchanged with all-gathers (RCCL, issued by the host side on the same stream) and reduced in rank order, so
// all ranks hold bit-identical normal equations and take the same LM decisions:
//   |r| keys (2 m_local doubles)  ->  exact global lower median
//   partial = [sum w J^T J (21 n) | sum w J^T r (6 n) | max raw weight | sum |r_obs|]
//   trial   = [sum |w r_obs'| local, sqrt(Sigma) sum |r_pred'|]
#include "vba_device.h"
#include "vba_launch.h"

namespace vba {

__global__ __launch_bounds__(256) void k_shard_pack(DevView V, double* out) {
    __shared__ double red[4];
    const int n = V.n[0];
    const int cnt = 27 * n;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt; e += gridDim.x * 256)
        out[e] = e < 21 * n ? V.Hraw[e] : V.braw[e - 21 * n];
    if (blockIdx.x == 0) {
        double s = 0.0;
        for (int b = threadIdx.x; b < V.nblk_obs; b += 256) s += V.part_init[b];
        const double t = block_sum<256>(s, red);
        if (threadIdx.x == 0) {
            out[cnt] = bits_f64(V.sc[0].wmax_bits[V.par]);
            out[cnt + 1] = t;
        }
    }
}

// with_sum == 0: the slot of sum |r_obs| is not in use (carried keys: the sum came with the trial's exchange)
__global__ __launch_bounds__(256) void k_shard_reduce(DevView V, const double* all, int ranks, int with_sum) {
    const int n = V.n[0];
    const int cnt = 27 * n;
    const int64_t stride = cnt + 2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt; e += gridDim.x * 256) {
        double s = 0.0;
        for (int q = 0; q < ranks; ++q) s += all[q * stride + e];   // fixed rank order
        if (e < 21 * n) V.Hraw[e] = s; else V.braw[e - 21 * n] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double mx = 0.0, sa = 0.0;
        for (int q = 0; q < ranks; ++q) {
            mx = fmax(mx, all[q * stride + cnt]);
            sa += all[q * stride + cnt + 1];
        }
        V.sc[0].wmax_bits[V.par] = f64_bits(mx);
        if (with_sum) V.sc[0].sum_in[V.par] = sa;
    }
}

__global__ __launch_bounds__(64) void k_shard_trial_sum(DevView V, double* out) {
    const int lane = threadIdx.x;
    double so = 0.0, sd = 0.0;
    for (int b = lane; b < V.nblk_obs; b += 64) so += V.part_trial[b];
    for (int b = lane; b < V.nblk_dyn + (V.prm.initialize ? 0 : V.nblk_long); b += 64) sd += V.part_trial[V.nblk_obs + b];
    so = wave_sum(so);
    sd = wave_sum(sd);
    if (lane == 0) { out[0] = so; out[1] = sd; }
}

void launch_shard_pack(const DevView& V, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_pack, dim3(64), dim3(256), 0, s, V, out);
}
void launch_shard_reduce(const DevView& V, const double* all, int ranks, hipStream_t s, int with_sum) {
    hipLaunchKernelGGL(k_shard_reduce, dim3(64), dim3(256), 0, s, V, all, ranks, with_sum);
}
void launch_shard_trial_sum(const DevView& V, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_shard_trial_sum, dim3(1), dim3(64), 0, s, V, out);
}

}  // namespace vba
