// vba_api.hip -- C ABI of libvinsat_ba.so (see include/vinsat_ba.h): context, options, uploads, states, diagnostics.
#include "vba_context.h"

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}



void fill_params(StepParams& p, int iter, int initialize) {
    // BA_filtering.py:22: alpha = min(max(1 - (2*(iter/5) - 1), 1), 2);  :26: Sigma = min(10000*(iter+1)**2, 1000000)
    double alpha = 1.0 - (2.0 * ((double)iter / 5.0) - 1.0);
    alpha = std::min(std::max(alpha, 1.0), 2.0);
    const double it1 = (double)iter + 1.0;
    const double sigma = std::min(10000.0 * it1 * it1, 1000000.0);
    p.alpha = alpha;
    p.am2 = std::fabs(alpha - 2.0);
    p.expo = alpha / 2.0 - 1.0;
    p.alpha_is_2 = alpha == 2.0;
    p.sigma = sigma;
    p.sqrt_sigma = std::sqrt(sigma);
    p.initialize = initialize ? 1 : 0;
    p.iter = iter;
    p.pad = 0;
}

int check_window(vba_handle h, int window) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (window < 0 || window >= h->W) return fail(VBA_EINVAL, "window index out of range");
    return VBA_OK;
}

// k_decide wrote the outcome of the trial into mapped host memory; waiting for the stream is all that is needed
int read_heads(vba_handle h) {
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int w = 0; w < h->W; ++w)
        if (h->h_head[w].flags & 64u)       // k_solve_resident: a consumer block gave up waiting for its producers
            return fail(VBA_ESTATE, "resident solve: a block of window " + std::to_string(w) + " timed out waiting for its producers (vba_set_fusion bits 5, 6)");
    return VBA_OK;
}

const volatile WinHead* head(vba_handle h, int w) { return h->h_head + w; }

// The second stream of handles with many windows carries the dynamics factor beside the streaming observation kernels.
// VBA_AUX_PRIO (diagnostic): 1 = highest priority, -1 = lowest, unset / 0 = default.
hipError_t create_aux_stream(hipStream_t* s) {
    // VBA_AUX_CUMASK (diagnostic): hex word repeated over the 8 x 32 compute units, e.g. 11111111 = every fourth one
    if (const char* m = std::getenv("VBA_AUX_CUMASK")) {
        const uint32_t word = (uint32_t)std::strtoul(m, nullptr, 16);
        uint32_t mask[8];
        for (auto& x : mask) x = word;
        return hipExtStreamCreateWithCUMask(s, 8, mask);
    }
    const char* e = std::getenv("VBA_AUX_PRIO");
    const int want = e ? std::atoi(e) : 0;
    if (want == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, want > 0 ? greatest : least);
}

// A call that vba_iterate_resident enqueued speculatively (iterate_pipelined) and that the caller did not ask for after all
// -- or that anything else than the next resident call is about to disturb: wait for it and forget it.  It has run its
// front and its first trial but nobody decided it: its input states, the result the caller holds, are intact (call
// parity), its trial states and everything keyed to them are dropped.  `boundary`: the caller left the resident loop
// (uploads, new states): remember not to speculate behind a call with that iter again.
int settle(vba_handle h, bool boundary) {
    if (!h) return VBA_OK;
    if (!h->spec.valid) {
        // the caller left the resident loop behind a call that had speculated nothing: what it does next is not "what follows
        // that iter" (learning across a window boundary made every second window waste a speculated call)
        if (boundary) h->prev_res_iter = -1;
        return VBA_OK;
    }
    HIPCHK(hipSetDevice(h->device));
    launch_reset_calls(h->V, h->stream);            // (also clears a missed warm select the dropped call may have recorded)
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int w = 0; w < h->W; ++w) h->h_head[w].flags = 0;
    h->par = (h->chain_par0 + h->spec.c) & 1;      // S[par]: the input of the speculated call = the last result
    h->carry_ok = 0;                                // its trial consumed the carried keys and left its own
    h->need_hist_reset = true;
    h->hist_dirty = false;
    h->spec.valid = false;
    h->spec_discards++;
    if (boundary && h->prev_res_iter >= 0) h->pred_iter[h->prev_res_iter & 63] = -2;
    h->prev_res_iter = -1;
    return VBA_OK;
}

int ready(vba_handle h) {
    for (int w = 0; w < h->W; ++w)
        if (!h->have_obs[w] || !h->have_win[w] || !h->have_state[w])
            return fail(VBA_ESTATE, "window " + std::to_string(w) + " is missing observations, pose constants or states");
    if (h->reg)
        for (int w = 0; w < h->W; ++w)
            if (!h->have_prior[w]) return fail(VBA_ESTATE, "window " + std::to_string(w) + " has no prior (vba_upload_prior)");
    return VBA_OK;
}



static int vba_set_accumulate_lanes(vba_handle h, int lanes);
static int vba_set_trial_tiles(vba_handle h, int tiles);
int vba_set_solver(vba_handle h, int chunk);

int vba_version(void) { return 210; }     // 2.1: vba_set_chunk_waves, fusion bits 2..4, warm select mode 3

const char* vba_last_error(void) { return g_err.c_str(); }

int vba_device_count(int* count) {
    if (!count) return fail(VBA_EINVAL, "null count");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return VBA_OK;
}

// Where the mode switches of a handle lie, chosen from the round-4 sweeps over W = 1 .. 4096 windows with the kernel set and the
// solver forced (tools/mode_sweep.py; DESIGN.md section 7, bench.py "batched_sweep").  C3 windows (500 poses / 50 000 rows),
// thousands of BA calls per second:
//   W            1    2    4    8   12   16   22   28   32   48   64  128  256  512 1024 2048 4096
//   latency     22   40   71  114  144  168  187  201  204    .  184  189  202  204    .    .    .   (best fusion mask, below)
//   bandwidth    .   23   38   64  112  139  167  194  212  245  273  314  342  361  367    .  369   (partitioned solve)
//   ... walk     .    2    4    9    .   17   23    .   34    .   63  114  187  277  366  435  479   (four windows per wavefront)
// (as measured when the switches were placed; with the tiled trial kernel and the chunk rules that followed the latency row reads
// 22 / 41 / 74 / 125 / 154 / 177 / 201 / 210 at W = 1 .. 28, bench.py "batched_sweep")
// and C2 windows (100 poses / 5 000 rows): latency 107 / 379 / 606 / 943 / 1230 / 1446 / 1541 at W = 4 / 16 / 32 / 64 / 128 / 256 /
// 512 against bandwidth 51 / 199 / 360 / 662 / 1145 / 1649 / 1931; walk against partitioned solve 954 : 1649 at 256 windows,
// 2183 : 2071 at 1024, 3227 : 2259 at 4096.
// * Kernel set: what fills the chip is rows AND windows -- the latency-mode kernels won up to 31 C3 windows, ~180 C2 windows, ~13
//   C4 windows in that sweep (38 / ~190 / ~20 with the latency set as it is now, see default_latency_mode below; until round 4 the
//   switch was at 16 windows whatever their size).
// * Solver: the sequential walk is a latency chain per window (~2.2 ms at 500 poses whatever the window count) and pays only
//   once ~1000 windows share it; below that the chains are cut into chunks (until round 4 the walk took over at 128 windows:
//   3.3 ms per step at 256 windows where the partitioned solve needs 0.75).
// * Inside latency mode the fusions that trade instructions for launches hold only while launches are what a call costs:
//   up to 175 000 rows mask 15 (the trial kernel forms the step, the chunk elimination its blocks); up to 450 000 rows 14 (the
//   trial kernel reads a step that a launch of its own formed: every observation block re-forming the steps of its poses costs
//   more than that launch as soon as a few windows share the chip); beyond 12 (the assembly is a launch of its own as well).
constexpr int kLatWindowsCap = 192;             // (33 MB of bin buckets per C3 window; per-window prologues)
constexpr int kPartitionedWindowsMax = 1023;
// The crossover measured at three window sizes -- ~180 windows of 5 000 rows, 31 of 50 000, ~13 of 200 000 (C4: latency 55.1 / 64.0 /
// 65.7 k it/s at W = 8 / 12 / 16 against bandwidth 50.0 / 62.9 / 73.7) -- lies on W* = 31 (50 000 / m)^0.7: between "by windows"
// (exponent 0) and "by rows" (exponent 1), because both the per-window prologues and the rows fill the chip.
// Re-measured at the end of round 4, after the tiled trial kernel and the chunk rules had made the latency set 4 .. 20 % faster
// (k it/s, latency : bandwidth): C3 212 : 205 at 32 windows, 216 : 213 at 36, 223 : 226 at 40; C4 80.3 : 73.3 at 16, 79.7 : 80.0 at
// 20; 200 poses / 20 000 rows 502 : 480 at 64, 544 : 594 at 96; C2 1389 : 1251 at 160, 1399 : 1403 at 192 -- W* = 38 (50 000 / m)^0.7
// for windows up to C3's size, 38 (50 000 / m)^0.46 beyond.
static bool default_latency_mode(int windows, int64_t m_max) {
    const double x = m_max <= 50000 ? 0.7 : 0.46;
    return windows == 1 || (windows <= kLatWindowsCap && (double)windows <= 38.0 * std::pow(50000.0 / (double)m_max, x));
}
static int default_fusion(bool lat, int windows, int n_max, int64_t m_max) {
    const double rows = (double)windows * (double)m_max;
    // one window (round 4, with the trial kernel's observation blocks of several tiles, vba_set_trial_tiles): mask 15 : 14 = 19.8 : 19.6
    // k it/s at 75 000 rows, 19.3 : 18.8 at 100 000, 17.1 : 17.2 at 150 000, 15.4 : 17.4 at 200 000 (C4), 10.1 : 12.2 at 500 000 rows /
    // 2004 poses (C5), 15.4 : 16.3 at 100 000 rows / 2000 poses
    if (windows == 1) return (lat && (m_max >= 150000 || n_max >= 1500)) ? 14 : 15;
    return !lat ? 15 : rows <= 175e3 ? 15 : rows <= 450e3 ? 14 : 12;
}

int vba_create(int device, int windows, int n_max, int64_t m_max, vba_handle* out) {
    return vba_create_mode(device, windows, n_max, m_max, -1, out);
}

int vba_create_mode(int device, int windows, int n_max, int64_t m_max, int mode, vba_handle* out) {
    if (!out) return fail(VBA_EINVAL, "null out");
    *out = nullptr;
    if (mode < -1 || mode > 1) return fail(VBA_EINVAL, "mode must be -1 (automatic), 0 (bandwidth-mode kernels) or 1 (latency-mode kernels)");
    const bool lat = mode == -1 ? default_latency_mode(windows, m_max) : mode == 1;
    if (windows < 1 || n_max < 2 || m_max < 1) return fail(VBA_EINVAL, "need windows >= 1, n_max >= 2, m_max >= 1");
    if (m_max > (int64_t)1 << 30) return fail(VBA_EINVAL, "m_max too large");
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt == 0) return fail(VBA_ENODEV, "no HIP device visible");
    if (device < 0 || device >= cnt) return fail(VBA_EINVAL, "device index out of range");
    HIPCHK(hipSetDevice(device));
    HIPCHK(configure_solver_device());
    vba_context* h = new (std::nothrow) vba_context();
    if (!h) return fail(VBA_ENOMEM, "host allocation failed");
    h->device = device;
    h->W = windows;
    h->n_max = n_max;
    h->m_max = m_max;
    const size_t W = windows, N = n_max, M = (size_t)m_max;
    const int nblk_obs = (int)((M + kObsBlock - 1) / kObsBlock);
    const int nblk_dyn = (int)((N + kObsBlock - 1) / kObsBlock);
    const int nblk_dyn16 = (int)((N - 1 + 14) / 15);      // pose-chain blocks of the 16-lanes-per-pose geometry (vba_set_fusion bit 0)
    const int trial_stride = nblk_obs + std::max(nblk_dyn, nblk_dyn16) + kLongCap;     // (+ the slots of the long edges, vba_long.hip)
    size_t bytes = 0;
    auto need = [&](size_t b) { bytes += ((b + 255) & ~size_t(255)) + 256; };
    need(W * 4); need(W * 4); need(W * sizeof(WinScalars));
    const size_t m_pad = (M + 31) & ~size_t(31);
    const size_t obs_stride = (6 * m_pad + m_pad / 2 + (N + 2) / 2 + 31) & ~size_t(31);     // doubles
    need(W * obs_stride * 8);
    need(W * N * 10 * 8); need(W * N * 10 * 8);
    need(W * N * 4 * 8); need(W * N * 4 * 8); need(W * N * 4);
    const int nblk_pred = (int)((N * kDynLanes + 255) / 256);
    const int pred_stride = nblk_pred + kLongCap;
    need(W * 2 * pred_stride * 8); need(W * 2 * pred_stride * 8); need(W * 81 * 8);
    const size_t long_pool_cap = windows <= kLongPoolFewWindows ? kLongPoolFew : kLongPool;
    need(W * kLongCap * 4); need(W * 4); need(W * kLongCap * 4); need(W * 2 * long_pool_cap * 6 * 8);
    need(W * N * 36 * 8); need(W * N * 6 * 8);
    need(W * 2 * M * 8); need(2 * W * M * 8); need(W * 2 * M * 8); need(W * nblk_obs * 8); need(W * trial_stride * 8); need(W * nblk_obs * 8);
    need(W * kHistStride * 4);
    // bin buckets of the carried keys (latency mode only): capacity ~6x the count of the densest warm bin -- the bin of the
    // median holds ~0.13 % of the keys with 1/256-binade bins, half of that with 1/512 (see warm_shift below)
    // ... and 2^46 (1/64 of a binade, range [c / 2^16, c * 2^16)) for handles of many windows: the flush of a block's
    // histogram costs a global atomic per bin it touched, coarser bins contend in LDS and lengthen the list the single pass
    // (k_select_warm) compacts and k_select_finish ranks -- about half a per cent of the keys here.  Swept 44 .. 51 at
    // 4096 x C3: 10.53 / 9.94 / 9.85 / 9.87 / 9.90 / 9.94 / 10.03 ms per step for 44 .. 50 (exact two-pass select: 10.12)
    const int warm_shift = !lat ? 46 : (2 * m_max <= 300000 ? 44 : 43);
    int bucket_cap = 0;
    if (lat) {
        const double expect = 2.0 * (double)m_max * (warm_shift == 44 ? 0.0013 : 0.00065) * 6.0;
        bucket_cap = 256;
        while (bucket_cap < expect && bucket_cap < 4096) bucket_cap *= 2;
        need(W * 2 * (size_t)kSelBins * bucket_cap * 8);
    }
    const size_t per_pose = 2 * (21 + 6) + 2 * (6 + 36 + 6 + 1 + 3 + 9 + 9 + 9) + 243 + 9 + 81 + 9 + 9;
    need(W * N * per_pose * 8 + 16 * 256);
    need(W * N * 171 * 8);
    need(W * N * (171 + 171 + 81 + 9 + 9) * 8 + 6 * 256);
    need(W * N * (171 * 3 + 9) * 8 + 6 * 256);              // second level (over-sized: p_max <= n_max / 2 + 1)
    need(W * (N + 8) * 4);                                  // flags of the resident solve (over-sized: chunks + groups + 1 <= n_max / 2 + 8)
    bytes += 1 << 16;
    if (hipMalloc(&h->arena.base, bytes) != hipSuccess) {
        delete h;
        return fail(VBA_ENOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
    }
    h->arena.size = bytes;
    // (hipMemset of device memory returns before the fill has run, and the handle's streams are non-blocking: without the wait the
    // fill of a 50 GB arena was still running when the first windows were uploaded and zeroed their observations again)
    if (hipMemset(h->arena.base, 0, bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
        hipFree(h->arena.base);
        delete h;
        return fail(VBA_EHIP, "hipMemset of the device arena failed");
    }
    Arena& A = h->arena;
    DevView& V = h->V;
    V.W = windows; V.n_max = n_max; V.m_max = m_max; V.nblk_obs = nblk_obs; V.nblk_dyn = nblk_dyn;
    V.n = h->d_n = A.take<int>(W);
    V.m = h->d_m = A.take<int>(W);
    V.sc = A.take<WinScalars>(W);
    h->d_obs = A.take<double>(W * obs_stride);
    h->m_pad = (int64_t)m_pad;
    V.obs_stride = (int64_t)obs_stride;
    V.ox = h->d_obs; V.oy = V.ox + m_pad; V.oz = V.oy + m_pad; V.ou = V.oz + m_pad; V.ov = V.ou + m_pad; V.oconf = V.ov + m_pad;
    V.opose = reinterpret_cast<const int*>(V.oconf + m_pad);
    V.pose_ptr = V.opose + m_pad;
    h->S[0] = A.take<double>(W * N * 10); h->S[1] = A.take<double>(W * N * 10);
    V.states = V.states_prev = h->S[0]; V.states_new = h->S[1];
    V.part_pred = A.take<double>(W * 2 * pred_stride); V.part_prior = A.take<double>(W * 2 * pred_stride); V.nblk_pred = nblk_pred;
    V.pred_stride = pred_stride;
    V.long_idx = h->d_long_idx = A.take<int>(W * kLongCap); V.n_long = h->d_n_long = A.take<int>(W);
    V.nblk_long = 0;
    V.long_off = h->d_long_off = A.take<int>(W * kLongCap);
    V.long_pool = A.take<double>(W * 2 * long_pool_cap * 6);
    V.long_pool_cap = (int)long_pool_cap;
    V.lastD = A.take<double>(W * 81);
    V.intr = h->d_intr = A.take<double>(W * N * 4);
    V.cumrot = h->d_cumrot = A.take<double>(W * N * 4);
    V.steps = h->d_steps = A.take<int>(W * N);
    V.prior_H = h->d_prior_H = A.take<double>(W * N * 36);
    V.prior_x = h->d_prior_x = A.take<double>(W * N * 6);
    V.reg = 0;
    V.absr = A.take<double>(W * 2 * M); V.wraw = h->wraw2 = A.take<double>(2 * W * M); V.ckeys = A.take<double>(W * 2 * M);
    V.acc_lanes = 8;    // set after construction by vba_set_accumulate_lanes(h, 0)
    V.part_init = A.take<double>(W * nblk_obs); V.part_trial = A.take<double>(W * trial_stride);
    V.trial_stride = trial_stride;
    V.part_next = A.take<double>(W * nblk_obs);
    V.hist = A.take<unsigned>(W * kHistStride);
    V.wbucket = bucket_cap ? A.take<double>(W * 2 * (size_t)kSelBins * bucket_cap) : nullptr;
    V.bucket_cap = bucket_cap;
    h->bucket_cap_alloc = bucket_cap;
    V.sel_inline = 0;
    V.Hraw = h->Hraw2 = A.take<double>(2 * W * N * 21); V.braw = h->braw2 = A.take<double>(2 * W * N * 6);
    // (the pose-chain factor's outputs: per call parity as well, see wraw2)
    V.xhat = h->dyn2[0] = A.take<double>(2 * W * N * 6); V.Phi = h->dyn2[1] = A.take<double>(2 * W * N * 36); V.rorb = h->dyn2[2] = A.take<double>(2 * W * N * 6);
    V.fatt = h->dyn2[3] = A.take<double>(2 * W * N); V.qgrad = h->dyn2[4] = A.take<double>(2 * W * N * 3);
    V.Hd = h->dyn2[5] = A.take<double>(2 * W * N * 9); V.Hu = h->dyn2[6] = A.take<double>(2 * W * N * 9); V.Hl = h->dyn2[7] = A.take<double>(2 * W * N * 9);
    V.bands = A.take<double>(W * N * 243); V.rhs = A.take<double>(W * N * 9);
    V.Xs = A.take<double>(W * N * 81); V.zs = A.take<double>(W * N * 9); V.dpose = A.take<double>(W * N * 9);
    V.p_max = n_max / 2 + 1;
    const size_t PM = V.p_max;
    V.csol = A.take<double>(W * N * 171);
    V.cL = A.take<double>(W * PM * 171); V.cR = A.take<double>(W * PM * 171);
    V.rXs = A.take<double>(W * PM * 81); V.rzs = A.take<double>(W * PM * 9); V.rx = A.take<double>(W * PM * 9);
    V.csol2 = A.take<double>(W * PM * 171); V.cL2 = A.take<double>(W * PM * 171); V.cR2 = A.take<double>(W * PM * 171);
    V.rx2 = A.take<double>(W * PM * 9);
    V.res_stride = n_max + 8;
    V.res_flags = A.take<unsigned>(W * (N + 8));
    V.resident = 0;
    V.m_total = 0; V.abs_all = nullptr; V.abs_all_count = 0;
    V.hop = 0; V.pivot = 0; V.call = -1; V.emit = 0; V.carry = 0; V.dyn_in_acc = 0;
    V.par = 0; V.fold = 0; V.redo = 0; V.fused_trial = 0; V.pending_only = 0; V.warm_force_miss = 0;
    V.lat = lat ? 1 : 0;       // latency mode: few windows cannot fill the chip, the kernel COUNT of a call is what costs
    V.trial_tiles = 1;          // set after construction by vba_set_trial_tiles(h, 0)
    // warm bins: 2^44 bit patterns (1/256 of a binade, range [c/16, c*8)) while a bin of the median's density stays short,
    // 2^43 (1/512, [c/4, c*2)) for the big windows
    V.warm_shift = warm_shift;
    V.chunk = 0; V.chunk2 = 0;      // set after construction by vba_set_solver(h, -1)
    {   // every device array a kernel may touch must have been carved: a null here would fault on the GPU
        const void* must[] = {V.n, V.m, V.sc, V.ox, V.oy, V.oz, V.ou, V.ov, V.oconf, V.opose, V.pose_ptr, V.states,
                              V.states_new, V.states_prev, V.intr, V.cumrot, V.steps, V.prior_H, V.prior_x, V.absr, V.wraw, V.ckeys, V.part_init, V.part_next,
                              V.part_pred, V.part_prior, V.lastD, V.long_idx, V.n_long, V.long_off, V.long_pool,
                              V.part_trial, V.hist, V.Hraw, V.braw, V.xhat, V.Phi, V.rorb, V.fatt, V.qgrad, V.Hd, V.Hu, V.Hl,
                              V.bands, V.rhs, V.Xs, V.zs, V.dpose, V.csol, V.cL, V.cR, V.rXs, V.rzs, V.rx, V.csol2, V.cL2, V.cR2, V.rx2, V.res_flags};
        bool ok = A.used <= A.size;
        for (const void* q : must) ok = ok && q != nullptr;
        if (!ok) {
            hipFree(A.base);
            delete h;
            return fail(VBA_ENOMEM, "internal: device arena mis-carved");
        }
    }
    // the second stream carries the dynamics factor beside the observation kernels and exists only where that is done (many
    // windows): a process maps its streams onto a few hardware queues, and every idle stream of another handle is one more
    // to share them with
    for (int k = 0; k < 64; ++k) h->pred_iter[k] = h->pred_init[k] = -1;
    h->V.host_states = nullptr;
    h->fusion = default_fusion(lat, windows, n_max, m_max);
    if ((windows == 1 && (hipEventCreateWithFlags(&h->ev_first, hipEventDisableTiming) != hipSuccess ||
                          hipHostMalloc((void**)&h->h_states_map, (size_t)2 * n_max * 10 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                          hipHostGetDevicePointer((void**)&h->V.host_states, h->h_states_map, 0) != hipSuccess)) ||
        hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
        (!lat && create_aux_stream(&h->aux_stream) != hipSuccess) ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_stage, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_up[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_up[1], hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&h->h_up[0], std::max(obs_stride, 9 * N + 160) * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_up[1], std::max(obs_stride, 9 * N + 160) * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_stage, ((size_t)n_max * 10 + 1) * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_back, (size_t)n_max * 10 * sizeof(double) + sizeof(WinScalars), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_head, W * sizeof(WinHead), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->V.host_head, h->h_head, 0) != hipSuccess) {
        vba_destroy(h);
        return fail(VBA_EHIP, "stream/event/pinned allocation failed");
    }
    h->stream = h->own_stream;
    h->n.assign(W, 0); h->m.assign(W, 0); h->n_long.assign(W, 0);
    h->have_obs.assign(W, 0); h->have_win.assign(W, 0); h->have_state.assign(W, 0); h->have_prior.assign(W, 0);
    h->perm.resize(W);
    vba_set_accumulate_lanes(h, 0);
    vba_set_trial_tiles(h, 0);
    vba_set_solver(h, -1);
    // warm select everywhere; the accept test is folded into the next call's first kernel only in latency mode (with many
    // windows the decide launch is 20 us of a 10 ms step, and every block of the select would repeat the test)
    h->warm_enabled = 1;
    h->fold_enabled = lat;
    if (const char* e = std::getenv("VBA_X_FOLD")) h->fold_enabled = std::atoi(e) != 0;     // (experiment knob)
#ifdef VBA_VARIANTS
    if (const char* e = std::getenv("VBA_CR_LEVELS")) h->cr_levels = std::atoi(e) == 3 ? 3 : 2;     // (three levels in front: measured slower, comparison build only)
#endif
    *out = h;
    return VBA_OK;
}

int vba_has_variants(void) {
#ifdef VBA_VARIANTS
    return 1;
#else
    return 0;
#endif
}

int vba_get_mode(vba_handle h, int* mode, int* chunk) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (mode) *mode = h->V.lat;
    if (chunk) *chunk = h->V.chunk;
    return VBA_OK;
}

int vba_destroy(vba_handle h) {
    if (!h) return VBA_OK;
    (void)settle(h);
    watch_stop(h);
    hipSetDevice(h->device);
    (void)vba_sh_comm_destroy(h);
    if (h->own_stream) { hipStreamSynchronize(h->own_stream); hipStreamDestroy(h->own_stream); }
    if (h->aux_stream) { hipStreamSynchronize(h->aux_stream); hipStreamDestroy(h->aux_stream); }
    for (hipEvent_t e : h->cprof.ev) if (e) hipEventDestroy(e);
    for (auto& ge : h->graphs) if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
    if (h->ev_first) hipEventDestroy(h->ev_first);
    if (h->h_states_map) hipHostFree(h->h_states_map);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->ev_stage) hipEventDestroy(h->ev_stage);
    for (int k = 0; k < 2; ++k) {
        if (h->ev_up[k]) hipEventDestroy(h->ev_up[k]);
        if (h->h_up[k]) hipHostFree(h->h_up[k]);
    }
    if (h->h_stage) hipHostFree(h->h_stage);
    if (h->h_back) hipHostFree(h->h_back);
    if (h->h_head) hipHostFree(h->h_head);
    if (h->d_dbg) hipFree(h->d_dbg);
    if (h->arena.base) hipFree(h->arena.base);
    delete h;
    return VBA_OK;
}

int vba_set_solver2(vba_handle h, int chunk, int chunk2) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (chunk < 2 || chunk > 60 || (chunk2 != 0 && chunk2 != -1 && (chunk2 < 2 || chunk2 > 60)))
        return fail(VBA_EINVAL, "chunk sizes must be in [2, 60] (chunk2 = 0: single level, -1: cyclic reduction)");
    if (chunk2 == -1 && (h->n_max + chunk - 1) / chunk - 1 > 128)
        return fail(VBA_EINVAL, "chunk2 = -1 needs at most 128 separators: chunk >= ceil(n_max / 129)");
    h->V.chunk = chunk;
    h->V.chunk2 = chunk2;
    h->no_pack = 0;
    return VBA_OK;
}

int vba_set_solver(vba_handle h, int chunk) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    h->V.chunk2 = 0;
    if (chunk == -1) {      // default: many windows supply their own parallelism (one wave walks each chain);
        h->no_pack = 0;     // otherwise the chain is cut into chunks and the reduced system over the (at most 64)
                            // separators is solved by cyclic reduction in one workgroup; very long chains: two levels
        if (h->W > kPartitionedWindowsMax || h->n_max < 8) { h->V.chunk = 0; return VBA_OK; }
        // <= 64 separators while that keeps the chunks at <= 8 poses, else up to 128 (their first reduction level runs
        // on its own CUs either way, see k_cr_level0)
        const int c64 = (h->n_max + 64) / 65, c128 = (h->n_max + 128) / 129;
        int c1 = std::max(std::min(c64, std::max(8, c128)), 2);
        // bandwidth mode: the chunks are there for throughput, not for the shortest chain -- fewer separators (less redundant
        // work in the reduced system) win: chunks of 12 measured +7 .. +9 % at 100 poses (256 .. 1000 windows) over the latency
        // rule's chunks of 2, +2 % at 500 poses over chunks of 8 (4 / 8 / 16 / 14 all slower; tools/attic/chunk_bw.sh)
        if (!h->V.lat) c1 = std::max(c1, std::min(12, std::max(2, h->n_max / 3)));
        // latency mode, several windows: the chunk elimination holds 256 registers, i.e. 1024 two-wave blocks are one round of
        // the chip and block 1025 waits for a second one (18 C3 windows of 63 chunks: 26.1 us against 17.7 at 15 windows) --
        // chunks of up to 12 poses where that keeps the elimination in one round (172.8 -> 180.4 k it/s at 18 windows,
        // 194.9 -> 202.3 at 22; no difference where it does not fit either way)
        // latency mode (re-measured in round 4 with the kernels as they are now): chunks of 8 also where 64 separators would allow
        // shorter ones -- one window of 100 poses (C2) 24.8 k it/s against 23.4 k with chunks of 2, 16 such windows 367 k against 319 k,
        // 64 windows 903 k against 749 k; 200 poses: on par with the old rule's 4 for one window, +5 % from 16 windows on
        if (h->V.lat) c1 = std::max(c1, std::min(8, std::max(2, h->n_max / 3)));
        if (h->V.lat && h->W > 1) {
            int c = c1;
            while (c < 12 && (int64_t)h->W * ((h->n_max + c - 1) / c) > 1024) ++c;
            if ((int64_t)h->W * ((h->n_max + c - 1) / c) <= 1024) c1 = c;
        }
        if (c1 <= 60) {
            h->V.chunk = c1;
            h->V.chunk2 = -1;
        } else {
            const int c3 = std::min(std::max((int)std::ceil(std::cbrt((double)h->n_max)), 2), 60);
            h->V.chunk = c3;
            h->V.chunk2 = c3;
        }
        return VBA_OK;
    }
    if (chunk == -2) {      // sequential, one window per wavefront (no packing): diagnostic / comparison
        h->V.chunk = 0;
        h->no_pack = 1;
        return VBA_OK;
    }
#ifndef VBA_VARIANTS
    if (chunk == -3) return fail(VBA_EINVAL, "three windows per wavefront (k_solve_packed) is a comparison variant that this build does not carry (make VARIANTS=1)");
#endif
    if (chunk == -3) {      // sequential, three windows per wavefront whenever the pose counts allow it
        h->V.chunk = 0;
        h->no_pack = 0;
        h->pack_min = 3;
        return VBA_OK;
    }
    if (chunk < 0 || chunk == 1 || chunk > 60) return fail(VBA_EINVAL, "chunk must be -3, -2, -1, 0 or in [2, 60]");
    h->V.chunk = chunk;
    h->no_pack = 0;
    return VBA_OK;
}

int vba_set_integrator(vba_handle h, int hop100) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    h->V.hop = hop100 ? 1 : 0;
    h->V.nblk_long = h->V.hop ? 0 : *std::max_element(h->n_long.begin(), h->n_long.end());
    return VBA_OK;
}

static int vba_set_accumulate_lanes(vba_handle h, int lanes) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (lanes == 0) {
        const double avg = (double)h->m_max / (double)h->n_max;
        int G = 4;
        while (G < 64 && avg / G > 16.0) G *= 2;     // measured on C3 x 1024: G = 8 (12.5 rows per lane) is the fastest
        // few windows: spend idle lanes on shorter per-lane loops (latency) instead of fewer shuffles (throughput)
        // (limit re-measured in round 4 with the handle's other defaults as they are now: 3 windows of 500 poses 57.4 k it/s with 32
        // lanes against 52.8 k with 16, 6 windows 93.0 against 91.2 with 16 instead of 8; 2, 4, 8, 12 windows unchanged)
        while (G < 64 && (int64_t)h->W * h->n_max * G * 2 <= 49152) G *= 2;
        lanes = G;
    }
    if (lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32 && lanes != 64) return fail(VBA_EINVAL, "lanes must be 0, 4, 8, 16, 32 or 64");
    h->V.acc_lanes = lanes;
    return VBA_OK;
}

// Tiles of 256 rows per observation block of the latency-mode trial kernel (plain geometry: fusion bit 0 off).  Measured on the
// chained 20-call schedule (k it/s, tiles 1 / 2 / 4 / 8): one C4 window (782 tiles) 15.3 / 16.4 / 17.4 / 16.2, one C5 window
// (1954 tiles) 9.8 / 11.5 / 12.2 / 12.1, 8 C3 windows (1568) 113.9 / 120.6 / 121.9, 22 C3 windows (4312) 185 / 197 / 195; one C3
// window (196, with that fusion mask) 19.7 / 20.6 / 19.8.  Results do not depend on it (see k_trial).
static int vba_set_trial_tiles(vba_handle h, int tiles) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (tiles == 0) {
        const int64_t blocks = (int64_t)h->W * h->V.nblk_obs;
        tiles = blocks >= 600 ? 4 : (blocks >= 150 ? 2 : 1);
    }
    if (tiles != 1 && tiles != 2 && tiles != 4 && tiles != 8) return fail(VBA_EINVAL, "tiles must be 0 (automatic), 1, 2, 4 or 8");
    h->V.trial_tiles = tiles;
    return VBA_OK;
}

static int vba_set_key_carry(vba_handle h, int on) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    h->carry_enabled = on != 0;
    h->carry_ok = false;
    return VBA_OK;
}

static int vba_set_fusion(vba_handle h, int mask) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (mask < 0 || mask > 127) return fail(VBA_EINVAL, "mask must be in [0, 127]");
#ifndef VBA_VARIANTS
    if (mask & (16 | 32 | 64)) return fail(VBA_EINVAL, "mask bits 4 .. 6 select comparison variants that this build does not carry (make VARIANTS=1)");
#endif
    h->fusion = mask;
    h->fusion_auto = false;
    return VBA_OK;
}

static int vba_set_chunk_waves(vba_handle h, int waves) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (waves != 1 && waves != 2) return fail(VBA_EINVAL, "waves must be 1 or 2");
    h->chunk_waves = waves;
    return VBA_OK;
}

static int vba_set_warm_select(vba_handle h, int on) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    h->warm_enabled = on == 2 ? 2 : (on != 0);
    h->inline_select = on != 3;     // 3: warm select as its own kernel (k_select_warm), the round-2 mid-point; comparison / tests
    return VBA_OK;
}

static int vba_set_warm_shift(vba_handle h, int shift) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (shift < 42 || shift > 51) return fail(VBA_EINVAL, "shift must be in [42, 51]");
    h->V.warm_shift = shift;
    h->carry_ok = 0;            // a histogram binned with another width cannot be resolved
    return VBA_OK;
}

static int vba_set_bucket_cap(vba_handle h, int cap) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (!h->bucket_cap_alloc) return fail(VBA_ESTATE, "this handle has no bin buckets (16 windows or more)");
    if (cap != 0 && (cap < 8 || cap > h->bucket_cap_alloc)) return fail(VBA_EINVAL, "cap must be 0 (default) or in [8, allocated capacity]");
    h->V.bucket_cap = cap ? cap : h->bucket_cap_alloc;
    h->carry_ok = 0;            // buckets filled with another stride are not addressable any more
    return VBA_OK;
}

int vba_set_host_watch(vba_handle h, int slot, const void* live, const void* copy, int64_t bytes) {
    if (!h || slot < 0 || slot >= 8) return fail(VBA_EINVAL, "bad argument (8 watch slots)");
    if (live && (!copy || bytes < 1)) return fail(VBA_EINVAL, "a watched buffer needs its reference copy and a size");
    watch_quiesce(h);
    h->watch[slot].live = live;
    h->watch[slot].copy = live ? copy : nullptr;
    h->watch[slot].bytes = live ? (size_t)bytes : 0;
    return VBA_OK;
}

static int vba_set_pipeline(vba_handle h, int on) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc = settle(h)) return rc;
    h->pipeline = on != 0;
    return VBA_OK;
}

int vba_pipeline_stats(vba_handle h, int* hits, int* discards) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (hits) *hits = h->spec_hits;
    if (discards) *discards = h->spec_discards;
    return VBA_OK;
}

int vba_warm_select_misses(vba_handle h, int* count) {
    if (!h || !count) return fail(VBA_EINVAL, "null argument");
    *count = h->warm_misses;
    return VBA_OK;
}

static int vba_set_pivoting(vba_handle h, int always) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    h->pivot_mode = always ? 1 : 0;
    return VBA_OK;
}

int vba_solver_fallbacks(vba_handle h, int* count) {
    if (!h || !count) return fail(VBA_EINVAL, "null argument");
    *count = h->fallbacks;
    return VBA_OK;
}

int vba_set_stream(vba_handle h, void* hip_stream, int external) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    hipStreamSynchronize(h->stream);
    h->stream = external ? (hipStream_t)hip_stream : h->own_stream;
    return VBA_OK;
}

int vba_upload_observations(vba_handle h, int window, int n, int64_t m, const double* xyz, const double* uv,
                            const double* conf, const int64_t* ii) {
    if (int rc = check_window(h, window)) return rc;
    if (int rc_settle = settle(h, true)) return rc_settle;
    if (!xyz || !uv || !conf || !ii) return fail(VBA_EINVAL, "null observation array");
    h->carry_ok = false;
    if (n < 2 || n > h->n_max) return fail(VBA_EINVAL, "n out of range (need 2 <= n <= n_max)");
    if (m < 1 || m > h->m_max) return fail(VBA_EINVAL, "m out of range (need 1 <= m <= m_max)");
    if (h->have_win[window] && h->n[window] != n) { h->have_win[window] = 0; h->have_state[window] = 0; h->have_prior[window] = 0; }   // a new window: re-upload its constants
    HIPCHK(hipSetDevice(h->device));
    // stable counting sort by pose: the reference's segment sums run in input order inside a pose
    std::vector<int> ptr(n + 1, 0);
    for (int64_t k = 0; k < m; ++k) {
        if (ii[k] < 0 || ii[k] >= n) return fail(VBA_EINVAL, "ii[" + std::to_string(k) + "] outside [0, n)");
        ptr[ii[k] + 1]++;
    }
    for (int i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
    std::vector<int64_t>& perm = h->perm[window];
    perm.assign(m, 0);
    {
        std::vector<int> cur(ptr.begin(), ptr.end() - 1);
        for (int64_t k = 0; k < m; ++k) perm[cur[ii[k]]++] = k;
    }
    // the whole block of the window -- six coordinate arrays, pose index, CSR -- is packed into pinned memory and goes up
    // with ONE asynchronous copy on the handle's stream; the pose / row counts follow in a one-thread kernel
    const int ub = h->up_next;
    h->up_next ^= 1;
    HIPCHK(hipEventSynchronize(h->ev_up[ub]));      // the previous copy out of this buffer has left it
    double* blk = h->h_up[ub];
    const size_t mp = (size_t)h->m_pad;
    double *x = blk, *y = x + mp, *z = y + mp, *u = z + mp, *v = u + mp, *c = v + mp;
    int* pose = reinterpret_cast<int*>(c + mp);
    int* cptr = pose + mp;
    for (int64_t s = 0; s < m; ++s) {
        const int64_t k = perm[s];
        x[s] = xyz[3 * k]; y[s] = xyz[3 * k + 1]; z[s] = xyz[3 * k + 2];
        u[s] = uv[2 * k]; v[s] = uv[2 * k + 1];
        c[s] = conf[k];
        pose[s] = (int)ii[k];
    }
    std::memcpy(cptr, ptr.data(), (size_t)(n + 1) * sizeof(int));
    const int mi = (int)m;
    const size_t used = (size_t)(reinterpret_cast<char*>(cptr + n + 1) - reinterpret_cast<char*>(blk));
    HIPCHK(hipMemcpyAsync(h->d_obs + (size_t)window * h->V.obs_stride, blk, used, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_up[ub], h->stream));
    launch_set_counts(h->V, window, n, mi, h->stream);
    HIPCHK(hipGetLastError());
    h->n[window] = n;
    h->m[window] = mi;
    h->have_obs[window] = 1;
    h->V.n_min = *std::min_element(h->n.begin(), h->n.end());
    return VBA_OK;
}

int vba_upload_window(vba_handle h, int window, int n, const double* intrinsics, const double* cumrot_last,
                      const int64_t* time_idx) {
    if (int rc = check_window(h, window)) return rc;
    if (int rc_settle = settle(h, true)) return rc_settle;
    if (!intrinsics || !cumrot_last || !time_idx) return fail(VBA_EINVAL, "null pose-constant array");
    h->carry_ok = false;
    if (n < 2 || n > h->n_max) return fail(VBA_EINVAL, "n out of range (need 2 <= n <= n_max)");
    if (h->have_obs[window] && h->n[window] != n) { h->have_obs[window] = 0; h->have_state[window] = 0; h->have_prior[window] = 0; }   // a new window: re-upload its rows
    HIPCHK(hipSetDevice(h->device));
    std::vector<int> steps(n);
    for (int i = 0; i + 1 < n; ++i) {
        const int64_t d = time_idx[i + 1] - time_idx[i];
        if (d < 1 || d > 100000000) return fail(VBA_EINVAL, "time_idx must be strictly increasing");
        steps[i] = (int)d;
    }
    steps[n - 1] = 1;   // BA_utils.py:75
    // long gaps (vba_long.hip): the first kLongCap edges of more than kLongGap steps are marked by a NEGATIVE step count and
    // listed; the kernels that walk the chain leave them to k_long_factor / k_long_trial (with the hop integrator the sign is
    // ignored and nothing is long)
    // ... each with room for its chain (header + G sub-chunk start states) in the window's pool; a gap the pool has no room for
    // stays an ordinary edge
    int long_list[2 * kLongCap + 1];
    int nl = 0, pool_used = 0;
    for (int i = 0; i + 1 < n && nl < kLongCap; ++i)
        if (steps[i] > kLongGap) {
            const int need_states = 3 + long_plan(steps[i]).G + 48;     // header, sub-chunk start states, eight 6x6 partial products (vba_long.hip)
            if (pool_used + need_states > h->V.long_pool_cap) continue;
            long_list[kLongCap + 1 + nl] = pool_used;
            pool_used += need_states;
            long_list[nl++] = i;
            steps[i] = -steps[i];
        }
    long_list[kLongCap] = nl;
    const size_t pb = (size_t)window * h->n_max;
    const int ub = h->up_next;
    h->up_next ^= 1;
    HIPCHK(hipEventSynchronize(h->ev_up[ub]));
    double* blk = h->h_up[ub];                      // holds max(obs_stride, 9 n_max + 32) doubles
    std::memcpy(blk, intrinsics, (size_t)n * 32);
    std::memcpy(blk + (size_t)n * 4, cumrot_last, (size_t)n * 32);
    std::memcpy(blk + (size_t)n * 8, steps.data(), (size_t)n * 4);
    double* lblk = blk + (size_t)n * 8 + (size_t)(n + 1) / 2;      // (the staging block holds max(obs_stride, 9 n_max + 160) doubles)
    std::memcpy(lblk, long_list, sizeof(long_list));
    HIPCHK(hipMemcpyAsync(h->d_intr + pb * 4, blk, (size_t)n * 32, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_cumrot + pb * 4, blk + (size_t)n * 4, (size_t)n * 32, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_steps + pb, blk + (size_t)n * 8, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    if (nl) {
        HIPCHK(hipMemcpyAsync(h->d_long_idx + (size_t)window * kLongCap, lblk, (size_t)nl * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_long_off + (size_t)window * kLongCap, reinterpret_cast<const int*>(lblk) + kLongCap + 1, (size_t)nl * 4,
                              hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipMemcpyAsync(h->d_n_long + window, reinterpret_cast<const int*>(lblk) + kLongCap, 4, hipMemcpyHostToDevice, h->stream));
    // (the carried chains of the window's long edges belong to the steps that were just replaced: all ones = NaN start states, which
    // no state ever equals)
    if (nl || h->n_long[window])
        HIPCHK(hipMemsetAsync(h->V.long_pool + (size_t)window * 2 * h->V.long_pool_cap * 6, 0xFF,
                              (size_t)2 * h->V.long_pool_cap * 6 * sizeof(double), h->stream));
    HIPCHK(hipEventRecord(h->ev_up[ub], h->stream));
    h->n_long[window] = nl;
    h->V.nblk_long = h->V.hop ? 0 : *std::max_element(h->n_long.begin(), h->n_long.end());     // (the <= 100 s hops of predict_gpu: no gap is long)
    launch_set_counts(h->V, window, n, -1, h->stream);
    HIPCHK(hipGetLastError());
    h->n[window] = n;
    h->have_win[window] = 1;
    return VBA_OK;
}

int vba_upload_prior(vba_handle h, int window, int n, const double* states_prior, const double* hessian_state) {
    if (int rc = check_window(h, window)) return rc;
    if (int rc_settle = settle(h, true)) return rc_settle;
    if (!states_prior || !hessian_state) return fail(VBA_EINVAL, "null prior array");
    if (!h->have_obs[window] && !h->have_win[window]) return fail(VBA_ESTATE, "upload the window before its prior");
    if (n != h->n[window]) return fail(VBA_EINVAL, "the prior needs one row per pose of the window");
    HIPCHK(hipSetDevice(h->device));
    std::vector<double> xp((size_t)n * 6);
    for (int i = 0; i < n; ++i) {
        const double* s = states_prior + (size_t)i * 10;
        double* o = xp.data() + (size_t)i * 6;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[7]; o[4] = s[8]; o[5] = s[9];
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t pb = (size_t)window * h->n_max;
    HIPCHK(hipMemcpy(h->d_prior_x + pb * 6, xp.data(), xp.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_prior_H + pb * 36, hessian_state, (size_t)n * 36 * 8, hipMemcpyHostToDevice));
    h->have_prior[window] = 1;
    return VBA_OK;
}

int vba_set_prior(vba_handle h, int on) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    h->reg = on != 0;
    return VBA_OK;
}

int vba_set_states(vba_handle h, int window, const double* states, double lamda) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h, true)) return rc_settle;
    if (!states) return fail(VBA_EINVAL, "null states");
    h->carry_ok = 0;
    double* S = h->S[h->par];           // the input buffer of the next call
    if (window == -1) {     // the same states for every window (all windows must have the same number of poses)
        const int n = h->n[0];
        for (int w = 0; w < h->W; ++w) {
            if (!h->have_obs[w] && !h->have_win[w]) return fail(VBA_ESTATE, "upload the windows before their states");
            if (h->n[w] != n) return fail(VBA_EINVAL, "window = -1 needs equal pose counts");
        }
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipEventSynchronize(h->ev_stage));
        std::memcpy(h->h_stage, states, (size_t)n * 80);
        HIPCHK(hipMemcpyAsync(S, h->h_stage, (size_t)n * 80, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipEventRecord(h->ev_stage, h->stream));
        DevView V = h->V;
        V.states = S;
        V.par = h->par;
        launch_broadcast_states(V, n, lamda, h->stream);
        HIPCHK(hipGetLastError());
        for (int w = 0; w < h->W; ++w) h->have_state[w] = 1;
        return VBA_OK;
    }
    if (int rc = check_window(h, window)) return rc;
    if (!h->have_obs[window] && !h->have_win[window]) return fail(VBA_ESTATE, "upload the window before its states");
    HIPCHK(hipSetDevice(h->device));
    const int n = h->n[window];
    // through a pinned staging buffer and asynchronously on the handle's stream: the call returns as soon as the
    // caller's array has been read, and the next call's kernels queue up behind the copy instead of behind two
    // blocking transfers
    HIPCHK(hipEventSynchronize(h->ev_stage));
    std::memcpy(h->h_stage, states, (size_t)n * 80);
    h->h_stage[(size_t)h->n_max * 10] = lamda;
    HIPCHK(hipMemcpyAsync(S + (size_t)window * h->n_max * 10, h->h_stage, (size_t)n * 80, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(&h->V.sc[window].lam[h->par], h->h_stage + (size_t)h->n_max * 10, 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipEventRecord(h->ev_stage, h->stream));
    h->have_state[window] = 1;
    return VBA_OK;
}

// what a finished call left in the window's scalars, seen from the parity `par` of the NEXT call
void unpack_scalars(const WinScalars* sc, int par, double* lamda, double* last_hessian, int* n_trials, unsigned* flags) {
    if (lamda) *lamda = sc->lam[par];
    if (last_hessian) std::memcpy(last_hessian, sc->last_hessian, 81 * 8);
    if (n_trials) *n_trials = sc->n_trials;
    if (flags) *flags = sc->fl[par ^ 1] & 7u;       // the public bits (vinsat_ba.h)
}

int vba_get_states(vba_handle h, int window, double* states, double* lamda, double* last_hessian, int* n_trials,
                   unsigned* flags) {
    if (int rc = check_window(h, window)) return rc;
    if (int rc_settle = settle(h)) return rc_settle;
    if (!h->have_state[window]) return fail(VBA_ESTATE, "no states uploaded");
    HIPCHK(hipSetDevice(h->device));
    const int n = h->n[window];
    // both pieces through pinned memory behind the queued work, one wait for the lot
    WinScalars* sc = reinterpret_cast<WinScalars*>(h->h_back + (size_t)h->n_max * 10);
    if (states) HIPCHK(hipMemcpyAsync(h->h_back, h->S[h->par] + (size_t)window * h->n_max * 10, (size_t)n * 80, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(sc, h->V.sc + window, sizeof(WinScalars), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (states) std::memcpy(states, h->h_back, (size_t)n * 80);
    unpack_scalars(sc, h->par, lamda, last_hessian, n_trials, flags);
    return VBA_OK;
}

int vba_set_states_all(vba_handle h, const double* states, const double* lamda) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h, true)) return rc_settle;
    if (!states || !lamda) return fail(VBA_EINVAL, "null states / lamda");
    for (int w = 0; w < h->W; ++w)
        if (!h->have_obs[w] && !h->have_win[w]) return fail(VBA_ESTATE, "upload the windows before their states");
    HIPCHK(hipSetDevice(h->device));
    h->carry_ok = 0;
    const size_t per = (size_t)h->n_max * 10;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->S[h->par], states, (size_t)h->W * per * sizeof(double), hipMemcpyHostToDevice));
    std::vector<WinScalars> sc((size_t)h->W);
    HIPCHK(hipMemcpy(sc.data(), h->V.sc, sc.size() * sizeof(WinScalars), hipMemcpyDeviceToHost));
    for (int w = 0; w < h->W; ++w) sc[w].lam[h->par] = lamda[w];
    HIPCHK(hipMemcpy(h->V.sc, sc.data(), sc.size() * sizeof(WinScalars), hipMemcpyHostToDevice));
    for (int w = 0; w < h->W; ++w) h->have_state[w] = 1;
    return VBA_OK;
}

int vba_get_states_all(vba_handle h, double* states, double* lamda, double* last_hessian, int* n_trials, unsigned* flags) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    for (int w = 0; w < h->W; ++w)
        if (!h->have_state[w]) return fail(VBA_ESTATE, "no states uploaded");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t per = (size_t)h->n_max * 10;
    if (states) HIPCHK(hipMemcpy(states, h->S[h->par], (size_t)h->W * per * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<WinScalars> sc((size_t)h->W);
    HIPCHK(hipMemcpy(sc.data(), h->V.sc, sc.size() * sizeof(WinScalars), hipMemcpyDeviceToHost));
    for (int w = 0; w < h->W; ++w)
        unpack_scalars(&sc[w], h->par, lamda ? lamda + w : nullptr, last_hessian ? last_hessian + (size_t)w * 81 : nullptr,
                       n_trials ? n_trials + w : nullptr, flags ? flags + w : nullptr);
    return VBA_OK;
}


// the settings a caller of BA() never needs, behind one entry point (include/vinsat_ba.h: VBA_OPT_*)
int vba_set_option(vba_handle h, int option, int value) {
    switch (option) {
        case VBA_OPT_ACCUMULATE_LANES: return vba_set_accumulate_lanes(h, value);
        case VBA_OPT_TRIAL_TILES: return vba_set_trial_tiles(h, value);
        case VBA_OPT_KEY_CARRY: return vba_set_key_carry(h, value);
        case VBA_OPT_WARM_SELECT: return vba_set_warm_select(h, value);
        case VBA_OPT_WARM_SHIFT: return vba_set_warm_shift(h, value);
        case VBA_OPT_BUCKET_CAP: return vba_set_bucket_cap(h, value);
        case VBA_OPT_FUSION: return vba_set_fusion(h, value);
        case VBA_OPT_CHUNK_WAVES: return vba_set_chunk_waves(h, value);
        case VBA_OPT_PIVOTING: return vba_set_pivoting(h, value);
        case VBA_OPT_PIPELINE: return vba_set_pipeline(h, value);
        case VBA_OPT_SCHEDULE_GRAPH: return vba_set_schedule_graph(h, value);
        case VBA_OPT_CHAIN_PROFILE: return vba_set_chain_profile(h, value);
        default: return fail(VBA_EINVAL, "unknown option (VBA_OPT_*)");
    }
}

int vba_last_step_ms(vba_handle h, float* ms) {
    if (!h || !ms) return fail(VBA_EINVAL, "null argument");
    if (!h->stepped) return fail(VBA_ESTATE, "no step has run");
    *ms = h->last_ms;
    return VBA_OK;
}

int vba_debug_fetch(vba_handle h, int window, int what, double* out, int64_t capacity, int64_t* count) {
    if (int rc = check_window(h, window)) return rc;
    if (int rc_settle = settle(h)) return rc_settle;
    if (!out || !count) return fail(VBA_EINVAL, "null output");
    if (!h->stepped) return fail(VBA_ESTATE, "no step has run");
    if (h->last_pipelined)
        return fail(VBA_ESTATE, "the last call was a pipelined vba_iterate_resident: the call speculated behind it has reused its scratch "
                                "(maximum weight, step, trial states); switch the pipeline off (vba_set_pipeline(h, 0)) to inspect intermediates");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int n = h->n[window];
    const int64_t m = h->m[window];
    const size_t pb = (size_t)window * h->n_max;
    // the view of the call that ran last: its input states are in the buffer of the other parity by now
    DevView V;
    {
        CallSpec c;
        c.iter = h->last_iter; c.initialize = h->last_init; c.call = -1; c.par = h->par ^ 1;
        view_for_call(h, V, c);
    }
    auto copy = [&](const double* src, int64_t cnt) -> int {
        if (cnt > capacity) return fail(VBA_EINVAL, "debug buffer too small");
        HIPCHK(hipMemcpy(out, src, cnt * 8, hipMemcpyDeviceToHost));
        *count = cnt;
        return VBA_OK;
    };
    WinScalars sc;
    HIPCHK(hipMemcpy(&sc, V.sc + window, sizeof(sc), hipMemcpyDeviceToHost));
    double wmax;
    std::memcpy(&wmax, &sc.wmax_bits[V.par], 8);
    switch (what) {
#ifdef VBA_RESIDENT_STAMPS
        case 100:           // diagnostic build: the wall-clock stamps of the last k_solve_resident launch (raw 64-bit words)
            return copy(V.cR2 + (size_t)window * V.res_stride * 4, (int64_t)V.res_stride * 4);
        case 102: {         // ... and along k_trial (g_ostamps, vba_obs.hip): 64 raw words
            if (capacity < 64) return fail(VBA_EINVAL, "debug buffer too small");
            fetch_ostamps(reinterpret_cast<unsigned long long*>(out));
            *count = 64;
            return VBA_OK;
        }
        case 101: {         // ... and of one thread along the solve kernels (g_kstamps, vba_solve.hip): 128 raw words
            if (capacity < 128) return fail(VBA_EINVAL, "debug buffer too small");
            fetch_kstamps(reinterpret_cast<unsigned long long*>(out));
            *count = 128;
            return VBA_OK;
        }
#endif
        case VBA_DBG_EST:
        case VBA_DBG_WEIGHT:
        case VBA_DBG_JG: {
            const int64_t per = what == VBA_DBG_EST ? 2 : (what == VBA_DBG_JG ? 12 : 1);
            if (m * per > capacity) return fail(VBA_EINVAL, "debug buffer too small");
            const size_t need = (size_t)m * 15 * 8;
            if (h->dbg_cap < need) {
                if (h->d_dbg) hipFree(h->d_dbg);
                h->d_dbg = nullptr;
                h->dbg_cap = 0;
                HIPCHK(hipMalloc(&h->d_dbg, need));
                h->dbg_cap = need;
            }
            double* est = h->d_dbg;
            double* J = est + 2 * m;
            double* wt = J + 12 * m;
            launch_debug_project(V, window, (int)m, est, J, wt, h->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
            std::vector<double> tmp((size_t)m * per);
            const double* src = what == VBA_DBG_EST ? est : (what == VBA_DBG_JG ? J : wt);
            HIPCHK(hipMemcpy(tmp.data(), src, (size_t)m * per * 8, hipMemcpyDeviceToHost));
            const std::vector<int64_t>& perm = h->perm[window];
            for (int64_t s = 0; s < m; ++s) std::memcpy(out + perm[s] * per, tmp.data() + s * per, per * 8);
            *count = m * per;
            return VBA_OK;
        }
        case VBA_DBG_H: {
            if ((int64_t)n * 36 > capacity) return fail(VBA_EINVAL, "debug buffer too small");
            std::vector<double> tmp((size_t)n * 21);
            HIPCHK(hipMemcpy(tmp.data(), V.Hraw + pb * 21, (size_t)n * 21 * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i)
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) out[(size_t)i * 36 + a * 6 + b] = tmp[(size_t)i * 21 + sym6(a, b)] / wmax;
            *count = (int64_t)n * 36;
            return VBA_OK;
        }
        case VBA_DBG_B: {
            if (int rc = copy(V.braw + pb * 6, (int64_t)n * 6)) return rc;
            for (int64_t k = 0; k < (int64_t)n * 6; ++k) out[k] /= wmax;
            return VBA_OK;
        }
        case VBA_DBG_PHI: return copy(V.Phi + pb * 36, (int64_t)n * 36);
        case VBA_DBG_RPRED: {
            if ((int64_t)(n - 1) * 7 > capacity) return fail(VBA_EINVAL, "debug buffer too small");
            std::vector<double> ro((size_t)n * 6), fa(n);
            HIPCHK(hipMemcpy(ro.data(), V.rorb + pb * 6, (size_t)n * 48, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(fa.data(), V.fatt + pb, (size_t)n * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < n - 1; ++i) {
                for (int r = 0; r < 6; ++r) out[(size_t)i * 7 + r] = ro[(size_t)i * 6 + r];
                out[(size_t)i * 7 + 6] = fa[i];
            }
            *count = (int64_t)(n - 1) * 7;
            return VBA_OK;
        }
        case VBA_DBG_QGRAD: return copy(V.qgrad + pb * 3, (int64_t)n * 3);
        case VBA_DBG_HQ: {
            if ((int64_t)n * 27 > capacity) return fail(VBA_EINVAL, "debug buffer too small");
            std::vector<double> d((size_t)n * 9), u((size_t)n * 9), l((size_t)n * 9);
            HIPCHK(hipMemcpy(d.data(), V.Hd + pb * 9, (size_t)n * 72, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(u.data(), V.Hu + pb * 9, (size_t)n * 72, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(l.data(), V.Hl + pb * 9, (size_t)n * 72, hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < 9; ++k) {
                    out[(size_t)i * 27 + k] = l[(size_t)i * 9 + k];
                    out[(size_t)i * 27 + 9 + k] = d[(size_t)i * 9 + k];
                    out[(size_t)i * 27 + 18 + k] = u[(size_t)i * 9 + k];
                }
            *count = (int64_t)n * 27;
            return VBA_OK;
        }
        case VBA_DBG_BANDS: {
            // latency mode never writes the bands to memory (the chunk kernel forms its blocks in LDS): form them now
            // (VBA_DBG_RAW_BANDS: diagnostic -- fetch what is in memory instead)
            if (!std::getenv("VBA_DBG_RAW_BANDS")) launch_assemble(V, 0, h->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
            if (int rc = copy(V.bands + pb * 243, (int64_t)n * 243)) return rc;
            if (h->last_init) {     // landmark-only phase: the off-diagonal blocks are zero and are not written
                for (int i = 0; i < n; ++i) {
                    std::memset(out + (size_t)i * 243, 0, 81 * 8);
                    std::memset(out + (size_t)i * 243 + 162, 0, 81 * 8);
                }
            }
            return VBA_OK;
        }
        case VBA_DBG_RHS: {
            if (!std::getenv("VBA_DBG_RAW_BANDS")) launch_assemble(V, 0, h->stream);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
            return copy(V.rhs + pb * 9, (int64_t)n * 9);
        }
        case VBA_DBG_DPOSE: return copy(V.dpose + pb * 9, (int64_t)n * 9);
        case VBA_DBG_SCALARS: {
            if (capacity < 8) return fail(VBA_EINVAL, "debug buffer too small");
            StepParams p;
            fill_params(p, h->last_iter, h->last_init);
            out[0] = sc.c_obs; out[1] = wmax; out[2] = sc.init_residual; out[3] = sc.trial_residual;
            out[4] = sc.lam32; out[5] = p.sigma; out[6] = p.alpha; out[7] = (double)sc.n_trials;
            *count = 8;
            return VBA_OK;
        }
        default: return fail(VBA_EINVAL, "unknown debug selector");
    }
}

