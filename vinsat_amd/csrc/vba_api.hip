// vba_api.hip -- C ABI of libvinsat_ba.so (see include/vinsat_ba.h): context, uploads, one BA() step.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>      // types only: the library is resolved at run time (vba_sh_comm_init), never linked

#include "../../include/vinsat_ba.h"
#include "vba_device.h"
#include "vba_launch.h"

using namespace vba;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(VBA_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

struct Arena {
    char* base = nullptr;
    size_t size = 0, used = 0;
    template <class T>
    T* take(size_t count) {
        used = (used + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base + used);
        used += count * sizeof(T);
        return p;
    }
};

}  // namespace

struct vba_context {
    int device = 0;
    int W = 0, n_max = 0;
    int64_t m_max = 0;
    hipStream_t own_stream = nullptr, stream = nullptr, aux_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr;
    Arena arena;
    DevView V{};
    // mutable device pointers (DevView holds const views of some)
    int *d_n = nullptr, *d_m = nullptr, *d_steps = nullptr;
    int *d_long_idx = nullptr, *d_n_long = nullptr, *d_long_off = nullptr;     // long edges of every window (vba_long.hip)
    std::vector<int> n_long;                // ... and how many each window has (host copy; DevView::nblk_long is their maximum)
    // per-observation weights and per-pose normal equations exist per call parity (DevView points at the slot of the call):
    // the accumulation of call c + 1 starts before the accept test of call c is known, whose later trials still read them
    double *wraw2 = nullptr, *Hraw2 = nullptr, *braw2 = nullptr;
    double* dyn2[8] = {};           // xhat, Phi, rorb, fatt, qgrad, Hd, Hu, Hl
    double* d_obs = nullptr;                // observation blocks, [W][obs_stride] (layout: DevView::ox)
    int64_t m_pad = 0;                      // doubles per observation array inside a block
    double *d_intr = nullptr, *d_cumrot = nullptr;
    // uploads go through pinned staging and are asynchronous on the handle's stream (ordered with the kernels that
    // read them); two buffers, so that the host packs window w + 1 while window w is on its way
    double* h_up[2] = {nullptr, nullptr};
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    int up_next = 0;
    WinHead* h_head = nullptr;              // mapped pinned host memory, [W]
    double* h_stage = nullptr;              // pinned staging for vba_set_states: [n_max * 10 + 1]
    double* h_back = nullptr;               // pinned staging for vba_get_states: [n_max * 10] + one WinScalars
    bool back_valid = false;                // h_back holds window 0's states and scalars after the last step (vba_iterate)
    double* S[2] = {nullptr, nullptr};      // the two state buffers [W][n_max][10]; S[par] is the input of the next call
    int par = 0;                            // parity of the next call (WinScalars: what a call hands on lives in the slots of the reader's parity)
    bool need_hist_reset = false;           // a call was abandoned half way: its histograms may be dirty
    hipEvent_t ev_stage = nullptr;          // the last staged copy has left the staging buffer
    std::vector<int> n, m;
    std::vector<char> have_obs, have_win, have_state, have_prior;
    bool reg = false;               // BA_reg semantics (per-pose prior) for the following calls
    double *d_prior_H = nullptr, *d_prior_x = nullptr;
    std::vector<std::vector<int64_t>> perm; // sorted position -> input row
    float last_ms = 0.f;
    bool stepped = false;
    int carry_ok = 0;               // every window's keys / histogram / sum |r| for its current states are on the device: 0 no, 1 with the
                                    // exponent histogram, 2 with the warm histogram (the kind the last trial emitted)
    bool carry_enabled = true;
    bool hist_dirty = false;        // a k_trial<true> has left a warm histogram (digit-0 slot of parity `par`) behind that nobody consumed
    bool fold_enabled = true;       // chained schedule: the first kernel of call c + 1 evaluates the accept test of call c (latency mode)
    int warm_enabled = 1;           // carried keys are selected with the one-pass warm select (vba_set_warm_select; 2: forced misses, test knob)
    int last_iter = 0, last_init = 0;
    int sh_pivot = 0;                       // sharded mode: solver variant of the current call (0 unpivoted, 2 mixed after a failed check)
    bool sh_rode = false, sh_bands_ready = false;   // sharded mode: the dynamics factor rode in the accumulation; bands / rhs are in memory
    // sharded mode with the exchanges issued by the library itself (vba_sh_comm_init / vba_sh_call): RCCL resolved at run time
    struct ShComm {
        void* dl = nullptr;
        ncclComm_t comm = nullptr;
        int nranks = 0, rank = 0;
        ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
        ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
        const char* (*error_string)(ncclResult_t) = nullptr;
        double* buf = nullptr;              // one allocation: abs_local | abs_all | partial_local | partial_all | trial_local | trial_all
        int64_t m_total = 0, m_pad = 0;     // what buf was sized for
        int n = 0;
        int m_local = -1;                   // rows of this rank the +inf padding of abs_local was laid out for
        // carried-keys protocol (vba_sh_run_schedule): exchange buffers that the kernels write in place
        int protocol = 1;                   // 1 = carried keys (default), 0 = the round-3 protocol (every call gathers all keys)
        ncclResult_t (*group_start)() = nullptr;
        ncclResult_t (*group_end)() = nullptr;
        double* buf2 = nullptr;             // sendA[2] | recvA | sendB | recvB
        int lenA = 0, lenB = 0, n2 = 0, nbo2 = 0, nbd2 = 0, cap2 = 0;
        double *sendA[2] = {nullptr, nullptr}, *recvA = nullptr, *sendB = nullptr, *recvB = nullptr;
        bool carried = false;               // recvA holds the exchange of the trial that produced the resident states: the next call may start from it
        int carried_par = 0;                // ... whose parity (the parity of the call that will read it)
        long fallbacks_miss = 0, fallbacks_lm = 0;
        double *abs_local = nullptr, *abs_all = nullptr, *partial_local = nullptr, *partial_all = nullptr, *trial_local = nullptr, *trial_all = nullptr;
    } shc;
    int pack_min = 1 << 30;                 // windows from which three chains share a wavefront: never by default (measured at 1024 / 2048 / 4096
                                            // windows: one wave per window is as fast or faster, 1.52 / 1.96 / 2.70 ms vs 1.52 / 2.06 / 2.78 ms per solve);
                                            // vba_set_solver(h, -3) packs from 3 windows on
    int no_pack = 0;                        // diagnostic: force one window per wavefront in the sequential driver
    int pivot_mode = 0;                     // 0 = fast path with automatic fallback, 1 = always pivot
    int fallbacks = 0;                      // number of solves repeated with pivoting (diagnostic)
    int inline_select = 1;                  // latency mode: warm select inside the accumulation (bin buckets); vba_set_warm_select(h, 3) turns it off
    int chunk_waves = 2;                    // vba_set_chunk_waves
    int cr_levels = 2;                      // cyclic-reduction levels in front of the one-workgroup kernel (VBA_CR_LEVELS / vba_set_cr_levels: 2 or 3)
    int fusion = 15;                        // vba_set_fusion (default: the trial kernel forms the step, the solves form their own blocks, uniform-pass assembly)
    bool fusion_auto = true;                // the mask is the library's own choice (vba_set_fusion not called)
    // the first passes of the last few chained schedules as graphs (vba_run_schedule), each with what it was made for; most recently
    // used first, at most kGraphCache of them (a driver alternates between a handful of schedules: the 20-call loop, its two phases)
    // key: a hash per call's view (the quick reject); views: the bytes of those views, compared exactly on a key match (a 64-bit hash
    // collision would replay another schedule's launches silently; ncalls x sizeof(DevView) of memcmp is ~1 us)
    struct GraphEntry { std::vector<unsigned long long> key; std::vector<unsigned char> views; hipGraphExec_t exec = nullptr; };
    std::vector<GraphEntry> graphs;
    bool graph_broken = false;              // capture or launch failed once: kernel by kernel from then on
    bool graph_enabled = true;              // vba_set_schedule_graph
    long graph_replays = 0, graph_captures = 0;
    int bucket_cap_alloc = 0;               // allocated capacity of a bin bucket (vba_set_bucket_cap lowers the one in use)
    int warm_misses = 0;                    // number of calls whose warm select missed and was repeated with the exact digits (diagnostic)
    double* d_dbg = nullptr;                // lazily allocated scratch for debug fetch
    size_t dbg_cap = 0;
    // Pipelined driver loop (vba_iterate_resident, see iterate_pipelined): the call that was enqueued speculatively behind
    // the one that has just been returned, the chain it belongs to and what has been learnt about the caller's schedule
    struct Spec { bool valid = false; int iter = 0, init = 0; bool reg = false; int c = 0; } spec;
    int chain_par0 = 0;                     // parity of call 0 of the open chain
    int pred_iter[64], pred_init[64];       // what followed a resident call with iter & 63 (-1: not seen yet, -2: nothing resident)
    int prev_res_iter = -1;                 // iter of the previous resident call (for learning), -1: none
    int pipeline = 1;                       // vba_set_pipeline
    bool last_pipelined = false;            // the last call went through iterate_pipelined: a speculated call has reused its scratch
    int spec_hits = 0, spec_discards = 0;   // diagnostics (vba_pipeline_stats)
    struct Watch { const void* live = nullptr; const void* copy = nullptr; size_t bytes = 0; } watch[8];   // vba_set_host_watch
    // The watched buffers are compared by a helper thread of the handle while the calling thread enqueues the speculated call: the
    // comparison of the reference driver's `ii` (400 kB at C3) is ~9 us of memcmp, and a landmark-only call leaves the host no idle
    // time to hide it in (23 us of device work against ~29 us of host work per resident call before this).
    struct WatchWorker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        unsigned long long seq = 0;         // guarded by m: number of the last request
        bool quit = false;                  // guarded by m
        std::atomic<unsigned long long> done_seq{0};    // the request `changed` answers
        bool changed = false;
        bool started = false;
        pid_t owner = 0;                    // the process the helper thread lives in (a forked child inherits `started`, not the thread)
    } ww;
    // vba_set_chain_profile: HIP events at the class boundaries (accumulate | solve | trial) of every call of a chained schedule
    struct ChainProf {
        bool on = false;
        std::vector<hipEvent_t> ev;         // 4 per call: before / behind the accumulation, behind the solve, behind the trial
        double ms[3] = {0.0, 0.0, 0.0};
        int64_t launches[3] = {0, 0, 0};
    } cprof;
    double* h_states_map = nullptr;         // [2][n_max][10] mapped pinned host memory (DevView::host_states), one-window handles
    hipEvent_t ev_first = nullptr;
};

namespace {


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
int settle(vba_handle h, bool boundary = false) {
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

}  // namespace

extern "C" {

static void watch_stop(vba_handle h);
static void watch_quiesce(vba_handle h);
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
static void unpack_scalars(const WinScalars* sc, int par, double* lamda, double* last_hessian, int* n_trials, unsigned* flags) {
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

namespace {

// ---------------------------------------------------------------------------------------------- one BA() call
// The kernels of a call, as the host enqueues them (all asynchronous on the handle's stream):
//
//   front   [k_obs_residual]                     only when the host replaced the states (no carried keys)
//           [select]                             exact digits (2 passes; 3 when digit 0 is not there yet); on carried keys
//                                                ONE warm pass (k_select_warm) -- or, latency mode, nothing: the keys lie
//                                                in per-bin buckets and the accumulation selects in its prologue.  In a
//                                                chained schedule the kernel that starts the call also evaluates the
//                                                accept test of the call in front (fold)
//           k_obs_accumulate (+ dynamics blocks) median finish, weights, per-pose normal equations [+ orbit / attitude factor]
//           [k_assemble]                         only when something reads the bands from memory: batched windows, sharded
//                                                mode, the sequential / always-pivoting solvers
//   trial   [solve]                              full phase: chunk elimination (forming its own blocks in latency mode),
//                                                cyclic reduction of the separators; landmark-only phase: nothing in
//                                                latency mode (the trial kernel solves its 6x6 systems itself)
//           k_trial                              step + retraction (latency mode) + trial residuals + next call's keys
//   decide  [k_decide]                           own launch unless the next call's first kernel folds it
//
// Latency mode, landmark-only call: 2 kernels (accumulate, trial); full call: 6 (+ assembly, chunks, two
// cyclic-reduction kernels).  Call parity p: input states S[p], trial states S[p ^ 1] (see WinScalars).
struct CallSpec {
    bool host_out = false;  // pipelined vba_iterate_resident: trial states and last_hessian also go to mapped host memory
    int iter = 0, initialize = 0;
    int call = -1;          // index inside a chained schedule, -1: stand-alone
    int par = 0;
    int carry = 0;          // the keys of the input states are on the device: 0 no, 1 with their exponent histogram, 2 with a warm one
    int emit = 0;           // leave the next call's keys behind: 0 no, 1 with the exponent histogram, 2 with the warm one
    bool fold = false;      // first kernel evaluates the accept test of call - 1
    bool prof = false;      // serialised schedule with an event between kernel classes
};

struct CallCtx {
    DevView V;
    hipEvent_t after_first = nullptr;   // recorded behind the kernel that starts the call (the folded accept test of the call in front is in it)
    bool fuse_assemble = false;     // first trial's landmark-only solve rides in k_assemble<true> (batched windows)
    bool assembled = false;         // an assembly kernel ran (profile bookkeeping)
    bool bands_ready = false;       // bands / rhs are in memory (the fused landmark-only assembly does not write them)
};

void view_for_call(vba_handle h, DevView& V, const CallSpec& c) {
    V = h->V;
    V.m_total = 0; V.abs_all = nullptr; V.abs_all_count = 0;
    V.reg = h->reg ? 1 : 0;
    V.n_min = *std::min_element(h->n.begin(), h->n.end());
    V.call = c.call;
    V.par = c.par;
    V.states = h->S[c.par];
    V.wraw = h->wraw2 + (size_t)c.par * h->W * h->V.m_max;
    V.Hraw = h->Hraw2 + (size_t)c.par * h->W * h->n_max * 21;
    V.braw = h->braw2 + (size_t)c.par * h->W * h->n_max * 6;
    {
        const size_t wn = (size_t)c.par * h->W * h->n_max;
        V.xhat = h->dyn2[0] + wn * 6; V.Phi = h->dyn2[1] + wn * 36; V.rorb = h->dyn2[2] + wn * 6; V.fatt = h->dyn2[3] + wn;
        V.qgrad = h->dyn2[4] + wn * 3; V.Hd = h->dyn2[5] + wn * 9; V.Hu = h->dyn2[6] + wn * 9; V.Hl = h->dyn2[7] + wn * 9;
    }
    V.states_new = h->S[c.par ^ 1];
    V.states_prev = h->S[c.par];
    if (!c.host_out) V.host_states = nullptr;
    V.emit = c.emit;
    V.carry = c.carry;
    V.fold = c.fold ? 1 : 0;
    V.sel_inline = 0;
    V.median_ready = 0;
    V.redo = 0;
    V.pending_only = 0;
    V.warm_force_miss = h->warm_enabled == 2;
    V.pivot = h->pivot_mode;
    // sequential driver with several windows: four chains per wavefront (k_solve_quad); vba_set_solver(h, -3) asks for the
    // older three-chain packing (equal pose counts only), -2 for one window per wavefront
    V.pack = 0;
    if (V.chunk <= 0 && !h->no_pack && h->W >= 2) {
        V.pack = 2;
        if (h->W >= h->pack_min) {
            V.pack = 1;
            for (int w = 1; w < h->W; ++w) if (h->n[w] != h->n[0]) V.pack = 2;
        }
    }
    fill_params(V.prm, c.iter, c.initialize);
    // who forms the step: latency mode lets the trial kernel do it (landmark-only: 6x6 solve per pose on the unpivoted
    // path; full phase: recovery of the partitioned solve)
    V.fused_trial = 0;
    if (V.lat && (h->fusion & 1)) {     // every trial kernel of such a handle uses the 16-lanes-per-pose geometry
        if (c.initialize) V.fused_trial = h->pivot_mode == 0 ? 1 : 3;
        else V.fused_trial = V.chunk > 0 ? 2 : 3;
        V.nblk_dyn = (V.n_max - 1 + 14) / 15;
    }
    V.fuse_blocks = (h->fusion & 2) ? 1 : 0;
    V.resident = !V.lat ? 0 : (h->fusion & 64) ? 2 : (h->fusion & 32) ? 1 : 0;
    V.chunk_waves = h->chunk_waves;
    V.asm_rows = (h->fusion & 8) ? 1 : 0;
    V.cr_levels = (h->fusion & 16) ? 1 : h->cr_levels;
    V.fuse_walk = ((h->fusion & 4) && !V.lat) ? 1 : 0;
}

// the kernels in front of the first LM trial; ev (profiled variant): events that bracket the kernel classes
int enqueue_front(vba_handle h, CallCtx& C, const CallSpec& c, bool exact_repeat, hipEvent_t* ev) {
    DevView& V = C.V;
    hipStream_t s = h->stream;
    auto mark = [&](int k) { if (ev && ev[k]) (void)hipEventRecord(ev[k], s); };
    const bool init = c.initialize != 0;
    V.sel_inline = 0;       // (a repeat of the front after a missed warm select takes the exact digits and the plain prologue)
    // the dynamics factor depends only on the states: with few windows its blocks ride in the accumulation's grid (no
    // second stream, no cross-stream join), with many it runs beside the observation kernels on a second stream
    const bool ride = !init && !c.prof && V.lat;
    V.dyn_in_acc = ride ? 1 : 0;
    static const bool no_overlap = std::getenv("VBA_NO_OVERLAP") != nullptr;     // diagnostic: dynamics in line on the main stream
    const bool overlap = !init && !c.prof && !ride && !no_overlap;
    auto fork_dynamics = [&]() -> int {
        HIPCHK(hipEventRecord(h->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
        launch_dynamics(V, h->aux_stream);
        HIPCHK(hipEventRecord(h->ev_join, h->aux_stream));
        return VBA_OK;
    };
    // a folding select is what moves the window on to this call (and reads the block sums the previous call's dynamics
    // left): the second stream forks behind it, not in front
    const bool fork_late = overlap && c.fold;
    if (overlap && !fork_late) { if (int rc = fork_dynamics()) return rc; }
    mark(1);
    if (!c.carry) {
        launch_obs_residual(V, nullptr, s);
        mark(2);
        launch_select(V, false, s);
    } else if (c.carry == 2 && !exact_repeat) {
        mark(2);
        // bin buckets (latency mode): the accumulation resolves the histogram and ranks the wanted bin's bucket in its own
        // prologue -- and, in a chained schedule, evaluates the accept test of the call in front there: no select kernel
        V.sel_inline = (V.wbucket && h->inline_select) ? 1 : 0;
        if (!V.sel_inline) launch_select_warm(V, s);
    } else if (c.carry == 1 && !exact_repeat) {     // the trial left digit 0 (exponent histogram) behind: two passes
        mark(2);
        launch_select(V, false, s);
    } else {            // a warm select that missed: the digit-0 slot holds the warm histogram, rebuild it by exponent
        mark(2);
        launch_clear_hist(V, 2, s);
        launch_select(V, true, s);
    }
    if (fork_late) { if (int rc = fork_dynamics()) return rc; }
    V.median_ready = (!V.sel_inline && !V.lat) ? 1 : 0;
    if (V.median_ready) launch_select_finish(V, s);
    mark(3);
    launch_obs_accumulate(V, s);
    if (C.after_first && V.sel_inline) HIPCHK(hipEventRecord(C.after_first, s));
    mark(4);
    if (!init && !overlap && !ride) launch_dynamics(V, s);
    if (overlap) HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
    mark(5);
    C.fuse_assemble = init && h->pivot_mode == 0 && V.fused_trial != 1;
    C.assembled = false;
    C.bands_ready = false;
    const bool need_bands = init ? V.fused_trial != 1 : !solve_forms_blocks(V);
    if (need_bands) {
        launch_assemble(V, C.fuse_assemble, s);
        C.assembled = true;
        C.bands_ready = !C.fuse_assemble;
    }
    mark(6);
    return VBA_OK;
}

// one LM trial: solve (unless the trial kernel or the assembly formed the step) + trial residuals; ev_solve (profiled
// variant): recorded between the two
void enqueue_trial(vba_handle h, CallCtx& C, const CallSpec& c, bool first, hipEvent_t ev_solve = nullptr, int solve_redo = -1) {
    DevView& V = C.V;
    hipStream_t s = h->stream;
    const bool init = c.initialize != 0;
    const int redo_all = V.redo;
    const bool pivoted_round = V.pivot != 0;
    // landmark-only phase: does a solve kernel run, i.e. does anything read the diagonal blocks from memory?  Not in the
    // first trial when the trial kernel or the fused assembly formed the step -- unless some window fell back to the
    // pivoted kernels
    const bool init_solve = init && (V.fused_trial == 1 ? pivoted_round : !(first && C.fuse_assemble));
    if (init_solve && !C.bands_ready) {
        launch_assemble(V, 0, s);
        C.assembled = C.bands_ready = true;
    }
    if (solve_redo >= 0) V.redo = solve_redo;       // which windows the solve kernels of this round take (see step_impl)
    if (init) {
        if (init_solve) launch_solve(V, 1, s);
    } else {
        launch_solve(V, 0, s);
    }
    V.redo = redo_all;
    if (ev_solve) (void)hipEventRecord(ev_solve, s);
    launch_trial(V, s);
}

}  // namespace

// readback >= 0: the states and scalars of that window are copied to the pinned read-back buffer right behind the first
// trial (valid if that trial ends the call: h->back_valid), so that vba_iterate needs one wait instead of two.
static int step_impl(vba_handle h, int iter, int initialize, float* prof, bool emit = true, int readback = -1) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc_settle = settle(h)) return rc_settle;
    if (int rc = ready(h)) return rc;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    CallSpec c;
    c.iter = iter; c.initialize = initialize; c.call = -1; c.par = h->par;
    const int emit_kind = h->warm_enabled ? 2 : 1;
    c.emit = (h->carry_enabled && emit) ? emit_kind : 0;
    c.carry = h->carry_enabled ? h->carry_ok : 0;
    c.prof = prof != nullptr;
    h->carry_ok = 0;
    h->shc.carried = false;         // (an unsharded call on a sharded handle: the gathered exchange of the last sharded trial is stale)
    CallCtx C;
    view_for_call(h, C.V, c);
    DevView& V = C.V;
    if (h->need_hist_reset) {       // an abandoned call may have left counts in any histogram
        DevView Q = V;
        for (int p = 0; p < 2; ++p) { Q.par = p; launch_clear_hist(Q, 1, s); }
        h->need_hist_reset = false;
        h->hist_dirty = false;
    }
    if (!c.carry && h->hist_dirty) launch_clear_hist(V, 0, s);     // the states were replaced after the last trial
    h->hist_dirty = c.emit != 0;
    struct ProfEvents {         // destroyed on every exit path, error returns included
        hipEvent_t e[VBA_NKERNELS + 1] = {};
        ~ProfEvents() { for (hipEvent_t q : e) if (q) (void)hipEventDestroy(q); }
    } pe;
    hipEvent_t* ev = pe.e;
    if (prof) {
        for (int k = 0; k <= VBA_NKERNELS; ++k) HIPCHK(hipEventCreate(&ev[k]));
    }
    auto mark = [&](int k) { if (prof) (void)hipEventRecord(ev[k], s); };
    struct Abandon {            // any error return below leaves a half-run call behind
        vba_handle h; bool armed = true;
        ~Abandon() { if (armed) { h->need_hist_reset = true; h->have_state.assign(h->W, 0); h->carry_ok = 0; } }
    } abandon{h};
    HIPCHK(hipEventRecord(h->ev0, s));
    mark(0);
    if (int rc = enqueue_front(h, C, c, false, prof ? ev : nullptr)) return rc;
    // LM loop (BA_filtering.py:52-77): lamda runs 1e-4 .. 1e4 in decades (at most 9 trials), plus one repeat per window for a
    // pivoted fallback and one for a missed warm select; a loop that is still not done after kMaxTrials means the device
    // never reported an outcome (a fault, a skipped window)
    constexpr int kMaxTrials = 24;
    bool finished = false, first = true;
    int solve_redo = -1;
    for (int trial = 0; trial < kMaxTrials; ++trial) {
        enqueue_trial(h, C, c, first, (first && prof) ? ev[7] : nullptr, solve_redo);
        solve_redo = -1;
        if (first) mark(8);
        launch_decide(V, nullptr, 0, s);
        if (first) {
            mark(9);
            HIPCHK(hipEventRecord(h->ev1, s));
        }
        h->back_valid = false;
        if (readback >= 0) {    // the trial states ARE the result if this trial ends the call
            HIPCHK(hipMemcpyAsync(h->h_back, V.states_new + (size_t)readback * h->n_max * 10, (size_t)h->n[readback] * 80, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(h->h_back + (size_t)h->n_max * 10, V.sc + readback, sizeof(WinScalars), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipGetLastError());
        if (int rc = read_heads(h)) return rc;
        first = false;
        bool all = true, repeat = false, miss = false;
        for (int w = 0; w < h->W; ++w) {
            all = all && head(h, w)->done;
            repeat = repeat || (head(h, w)->flags & 8u);
            miss = miss || (head(h, w)->flags & 32u);
        }
        if (miss) {     // the warm select missed for some window: those repeat the call's front with the exact digits
            for (int w = 0; w < h->W; ++w) if (head(h, w)->flags & 32u) h->h_head[w].flags = 0;
            h->warm_misses++;
            V.redo = 1;
            CallSpec cr = c;
            cr.prof = false;
            if (int rc = enqueue_front(h, C, cr, true, nullptr)) return rc;
            V.redo = 2;             // this round: their first trial, the others' next one
            // ... whose solve the repeating windows skip when the assembly they just ran has formed their first step already
            solve_redo = (c.initialize && C.fuse_assemble) ? 0 : 2;
            continue;
        }
        V.redo = 0;
        if (repeat && V.pivot == 0) {   // a pivot check failed on the fast path: those windows repeat the trial with row pivoting
            V.pivot = 2;
            h->fallbacks++;
            continue;
        }
        if (all) {
            h->back_valid = readback >= 0;      // the copies queued behind this (final) trial hold the result
            finished = true;
            break;
        }
    }
    if (!finished)
        return fail(VBA_ESTATE, "LM loop did not terminate within " + std::to_string(kMaxTrials) + " trials (no outcome reported by the device)");
    abandon.armed = false;
    HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    if (prof) {
        for (int k = 0; k < VBA_NKERNELS; ++k) {
            prof[k] = 0.f;
            (void)hipEventElapsedTime(&prof[k], ev[k], ev[k + 1]);
        }
        if (c.carry) prof[VBA_K_RESIDUAL] = 0.f;        // not launched: the previous trial left the keys behind
        if (c.initialize && (C.fuse_assemble || V.fused_trial == 1)) prof[VBA_K_SOLVE] = 0.f;   // no solve launch: formed inside k_assemble<true> / k_trial
        if (!C.assembled) prof[VBA_K_ASSEMBLE] = 0.f;
        if (c.initialize) prof[VBA_K_DYNAMICS] = 0.f;
    }
    h->par ^= 1;                    // the trial buffer is the next call's input
    h->carry_ok = c.emit;           // (the kind of histogram that came with the keys)
    h->stepped = true;
    h->last_pipelined = false;
    h->prev_res_iter = -1;          // (not a link of the resident loop: nothing to learn from what follows it)
    h->last_iter = iter;
    h->last_init = initialize;
    return VBA_OK;
}

int vba_step(vba_handle h, int iter, int initialize) { return step_impl(h, iter, initialize, nullptr); }

// The first trial of call q.call has been evaluated for the windows that stand at it (stall_at[w] == q.call) but was not
// cleanly accepted by the kernel that was to start the next call (or the call's warm select missed): finish the call the
// ordinary way -- decide, repeat the front with the exact digits where the select missed, further LM trials, the pivoted
// repeat -- until every such window has moved on.  Shared by vba_run_schedule and the pipelined vba_iterate_resident.
static int finish_stalled_call(vba_handle h, const CallSpec& q, const std::vector<int>& stall_at, long& trials) {
    hipStream_t s = h->stream;
    const int sc_call = q.call;
    static const bool trace = std::getenv("VBA_TRACE") != nullptr;
    CallCtx C;
    view_for_call(h, C.V, q);
    DevView& V = C.V;
    // the front of this call has run (for the windows that reached it); what is on the device of it:
    C.fuse_assemble = q.initialize && h->pivot_mode == 0 && V.fused_trial != 1;
    C.assembled = q.initialize ? V.fused_trial != 1 : !solve_forms_blocks(V);
    C.bands_ready = C.assembled && !C.fuse_assemble;
    auto at_call = [&](int w) { return stall_at[w] == sc_call && head(h, w)->call_idx == sc_call; };
    bool any_miss = false;
    for (int w = 0; w < h->W; ++w) any_miss = any_miss || (at_call(w) && (head(h, w)->flags & 32u));
    // (1) the first trial of the windows that got that far has been evaluated but not decided (the decision was left
    //     to the next call's first kernel, which found it not clean): decide it now
    V.pending_only = 1;
    launch_decide(V, nullptr, 0, s);
    V.pending_only = 0;
    // (2) windows whose warm select missed repeat the front with the exact digits and run their first trial
    if (any_miss) {
        for (int w = 0; w < h->W; ++w) if (at_call(w) && (head(h, w)->flags & 32u)) h->h_head[w].flags = 0;
        h->warm_misses++;
        V.redo = 1;
        if (int rc = enqueue_front(h, C, q, true, nullptr)) return rc;
        enqueue_trial(h, C, q, true);
        launch_decide(V, nullptr, 0, s);
        V.redo = 0;
        ++trials;
    }
    HIPCHK(hipGetLastError());
    if (int rc = read_heads(h)) return rc;
    bool finished = false;
    for (int trial = 0; trial <= 24; ++trial) {
        bool repeat = false, all = true;
        for (int w = 0; w < h->W; ++w) {
            if (!at_call(w)) continue;
            all = false;
            repeat = repeat || (head(h, w)->flags & 8u);
        }
        if (all) { finished = true; break; }
        if (trial == 24) break;
        if (repeat && V.pivot == 0) { V.pivot = 2; h->fallbacks++; }
        enqueue_trial(h, C, q, false);
        launch_decide(V, nullptr, 0, s);
        HIPCHK(hipGetLastError());
        if (int rc = read_heads(h)) return rc;
        ++trials;
        if (trace) {
            std::fprintf(stderr, "[vba]   call %d round %d pivot %d:", sc_call, trial, V.pivot);
            for (int w = 0; w < h->W && w < 8; ++w)
                std::fprintf(stderr, " w%d(call %d done %d flags %u ntr %d lam %g)", w, head(h, w)->call_idx, head(h, w)->done, head(h, w)->flags, head(h, w)->n_trials, head(h, w)->lamda);
            std::fprintf(stderr, "\n");
        }
    }
    if (!finished)
        return fail(VBA_ESTATE, "LM loop of call " + std::to_string(sc_call) + " did not terminate (no outcome reported by the device)");
    return VBA_OK;
}

// The 20-call loop of the driver (od_pipe.py:1036-1040) as ONE host call.  The kernels of every call are enqueued
// back to back with a single LM trial each and, on carried keys, without a decide launch between them: the first kernel
// of call c + 1 evaluates the accept test of call c itself.  A window whose first trial is not cleanly accepted (rejected,
// pivot check failed) or whose warm select misses does not advance its device-side call counter, all later kernels skip
// it, and the host finishes that call the ordinary way before re-enqueuing the rest.  Results are identical to ncalls
// vba_step calls.
int vba_run_schedule(vba_handle h, int ncalls, const int* iters, const int* inits, int* trials_total) {
    if (!h || !iters || !inits || ncalls < 1) return fail(VBA_EINVAL, "bad argument");
    if (int rc_settle = settle(h)) return rc_settle;
    if (int rc = ready(h)) return rc;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int par0 = h->par;
    const int emit_kind = h->carry_enabled ? (h->warm_enabled ? 2 : 1) : 0;
    const int carry0 = h->carry_enabled ? h->carry_ok : 0;
    h->carry_ok = 0;
    h->shc.carried = false;
    struct Abandon {
        vba_handle h; bool armed = true;
        ~Abandon() { if (armed) { h->need_hist_reset = true; h->have_state.assign(h->W, 0); h->carry_ok = 0; } }
    } abandon{h};
    auto spec = [&](int c, bool fold) {
        CallSpec q;
        q.iter = iters[c]; q.initialize = inits[c]; q.call = c; q.par = (par0 + c) & 1;
        q.carry = c == 0 ? carry0 : emit_kind;              // every later call starts from a trial of this chain
        q.emit = emit_kind;
        q.fold = fold;
        return q;
    };
    {
        DevView V0;
        view_for_call(h, V0, spec(0, false));
        if (h->need_hist_reset) {
            DevView Q = V0;
            for (int p = 0; p < 2; ++p) { Q.par = p; launch_clear_hist(Q, 1, s); }
            h->need_hist_reset = false;
            h->hist_dirty = false;
        }
        if (!carry0 && h->hist_dirty) launch_clear_hist(V0, 0, s);
        launch_reset_calls(V0, s);
    }
    h->hist_dirty = emit_kind != 0;
    for (int w = 0; w < h->W; ++w) { h->h_head[w].call_idx = 0; h->h_head[w].done = 0; h->h_head[w].flags = 0; }
    long trials = 0;
    int next = 0;
    bool complete = false;
    const bool prof_pass = h->cprof.on;
    if (prof_pass) {
        while ((int)h->cprof.ev.size() < 4 * ncalls) {
            hipEvent_t e = nullptr;
            HIPCHK(hipEventCreate(&e));
            h->cprof.ev.push_back(e);
        }
    }
    // The first pass of a latency-mode handle -- ~70 dependent launches for the driver's 20 calls -- is captured once as a hipGraph and
    // replayed while nothing that goes into its launches has changed: 45.2 -> 42.7 us per call at C3 (the packets of a graph reach the
    // queue in one piece; launched one by one every kernel boundary also pays the runtime's per-launch bookkeeping on the device's
    // clock).  What goes into the launches: the per-call views (every kernel takes its DevView by value: hashed byte for byte), the
    // schedule, and the handful of host-side switches the enqueue functions read.  Stalled calls are finished by the host afterwards
    // exactly as without a graph.  VBA_NO_GRAPH=1 launches kernel by kernel (comparison).
    static const bool no_graph = std::getenv("VBA_NO_GRAPH") != nullptr;
    for (int guard = 0; guard <= ncalls; ++guard) {
        bool capturing = false, replayed = false;
        std::vector<unsigned long long> gkey;
        std::vector<unsigned char> gviews;
        // (latency-mode handles only: with the second stream of the bandwidth mode forked inside it the replay measured 1 .. 2.5 % SLOWER
        // than the launches one by one, 40 .. 1024 windows)
        // (... and not while the chain profile records its events: event records inside a capture fail on this runtime, "invalid resource
        // handle" -- the class times of vba_chain_profile are those of the kernel-by-kernel launches)
        // (... nor with the resident solve of the comparison build, vba_set_fusion bits 5 / 6: its kernels take the epoch of the launch as
        // an argument, which a replay would freeze -- the consumers' flags would read as already set)
        if (!no_graph && h->graph_enabled && guard == 0 && !prof_pass && h->V.lat && !h->graph_broken && next == 0 && (h->fusion & 96) == 0) {
            gkey.reserve(8 + 3 * (size_t)ncalls);
            gviews.resize((size_t)ncalls * sizeof(DevView));
            gkey.push_back((unsigned long long)ncalls); gkey.push_back((unsigned long long)par0); gkey.push_back((unsigned long long)carry0);
            gkey.push_back((unsigned long long)emit_kind); gkey.push_back((unsigned long long)h->pivot_mode);
            gkey.push_back((unsigned long long)h->inline_select | ((unsigned long long)h->fold_enabled << 1));
            gkey.push_back((unsigned long long)(uintptr_t)s);
            for (int c = 0; c < ncalls; ++c) {
                const bool fold = c > 0 && emit_kind == 2 && h->fold_enabled;
                DevView Vc;
                view_for_call(h, Vc, spec(c, fold));
                if (fold) fill_params(Vc.prev, iters[c - 1], inits[c - 1]);
                unsigned long long hsh = 1469598103934665603ull;
                const unsigned char* bytes = reinterpret_cast<const unsigned char*>(&Vc);
                std::memcpy(gviews.data() + (size_t)c * sizeof(DevView), bytes, sizeof(DevView));
                for (size_t o = 0; o + 8 <= sizeof(DevView); o += 8) {
                    unsigned long long wd;
                    std::memcpy(&wd, bytes + o, 8);
                    hsh = (hsh ^ wd) * 1099511628211ull;
                    hsh ^= hsh >> 29;
                }
                gkey.push_back(hsh); gkey.push_back((unsigned long long)iters[c]); gkey.push_back((unsigned long long)inits[c]);
            }
            size_t hit = h->graphs.size();
            for (size_t k = 0; k < h->graphs.size(); ++k)
                if (h->graphs[k].key == gkey && h->graphs[k].views == gviews) { hit = k; break; }
            if (hit < h->graphs.size()) {
                if (hit != 0) std::rotate(h->graphs.begin(), h->graphs.begin() + hit, h->graphs.begin() + hit + 1);     // most recently used first
                if (hipGraphLaunch(h->graphs[0].exec, s) == hipSuccess) { replayed = true; h->graph_replays++; }
                else { (void)hipGetLastError(); h->graph_broken = true; }
            } else {
                if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) capturing = true;
                else { (void)hipGetLastError(); h->graph_broken = true; }
            }
        }
        struct CaptureGuard {       // (an early return between begin and end must not leave the stream capturing)
            hipStream_t s; bool* on;
            ~CaptureGuard() {
                if (*on) {
                    hipGraph_t g = nullptr;
                    (void)hipStreamEndCapture(s, &g);
                    if (g) (void)hipGraphDestroy(g);
                    (void)hipGetLastError();
                }
            }
        } capture_guard{s, &capturing};
        // speculative part: calls next .. ncalls-1, one trial each
        auto enqueue_pass = [&]() -> int {
        for (int c = next; c < ncalls && !replayed; ++c) {
            const bool fold = c > next && emit_kind == 2 && h->fold_enabled;       // call c-1 of this pass left its decision to this call's warm select
            const CallSpec q = spec(c, fold);
            CallCtx C;
            view_for_call(h, C.V, q);
            if (fold) fill_params(C.V.prev, iters[c - 1], inits[c - 1]);
            // chain profile (first pass only): events in front of / behind the accumulation (what runs in front of it --
            // select kernels of the bandwidth mode -- counts as accumulate class: the first event is moved there), behind
            // the solve and behind the trial
            hipEvent_t marks[VBA_NKERNELS + 1] = {};
            hipEvent_t* pe = nullptr;
            if (prof_pass && guard == 0) {
                pe = h->cprof.ev.data() + (size_t)4 * c;
                marks[1] = pe[0];
                marks[6] = pe[1];
            }
            if (int rc = enqueue_front(h, C, q, false, pe ? marks : nullptr)) return rc;
            enqueue_trial(h, C, q, true, pe ? pe[2] : nullptr);
            if (pe) HIPCHK(hipEventRecord(pe[3], s));
            const bool next_folds = c + 1 < ncalls && emit_kind == 2 && h->fold_enabled;
            if (!next_folds) launch_decide(C.V, nullptr, 0, s);
        }
        return VBA_OK;
        };
        if (int rc = enqueue_pass()) return rc;
        if (capturing) {
            // A capture that cannot be ended, instantiated or launched has executed NOTHING (its kernels were only recorded): the
            // handle gives up on graphs (graph_broken: kernel by kernel from then on) and this pass is enqueued again, for real.
            // VBA_GRAPH_FAIL_INJECT = 1 / 2 / 3 pretends that step failed (tests/test_gpu_bench_paths.py).
            static const int inject = std::getenv("VBA_GRAPH_FAIL_INJECT") ? std::atoi(std::getenv("VBA_GRAPH_FAIL_INJECT")) : 0;
            hipGraph_t g = nullptr;
            capturing = false;
            hipGraphExec_t exec = nullptr;
            bool ok = hipStreamEndCapture(s, &g) == hipSuccess && g != nullptr && inject != 1;
            if (ok) ok = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess && inject != 2;
            if (g) (void)hipGraphDestroy(g);
            if (ok && (inject == 3 || hipGraphLaunch(exec, s) != hipSuccess)) ok = false;
            if (!ok) {
                (void)hipGetLastError();
                if (exec) (void)hipGraphExecDestroy(exec);
                h->graph_broken = true;
                if (int rc = enqueue_pass()) return rc;
            } else {
            constexpr size_t kGraphCache = 8;
            if (h->graphs.size() >= kGraphCache) {
                // (the evicted graph may still be executing: the stream is idle here only if the caller made it so -- wait)
                HIPCHK(hipStreamSynchronize(s));
                (void)hipGraphExecDestroy(h->graphs.back().exec);
                h->graphs.pop_back();
            }
            vba_context::GraphEntry ge;
            ge.key = gkey;
            ge.views = std::move(gviews);
            ge.exec = exec;
            h->graphs.insert(h->graphs.begin(), std::move(ge));
            h->graph_captures++;
            }
        }
        HIPCHK(hipGetLastError());
        if (int rc = read_heads(h)) return rc;
        trials += (long)(ncalls - next);
        // After a pass over calls next .. ncalls-1 every window whose counter is below ncalls is stalled AT that call.
        // Every stalled call is finished with the ordinary LM loop -- each one, not only the earliest: a window left at a
        // later call would otherwise run that call again from its start when the chain is re-issued.
        std::vector<int> stalled, stall_at((size_t)h->W);
        for (int w = 0; w < h->W; ++w) {
            const int c = (int)head(h, w)->call_idx;
            stall_at[w] = c;        // a window that the loop below moves on INTO a later stalled call has not run that call's front: it waits for the re-issue
            if (c < ncalls && std::find(stalled.begin(), stalled.end(), c) == stalled.end()) stalled.push_back(c);
        }
        if (prof_pass && guard == 0 && stalled.empty()) {       // every call ran once, in order: its three intervals count
            for (int c = 0; c < ncalls; ++c) {
                const hipEvent_t* pe = h->cprof.ev.data() + (size_t)4 * c;
                for (int k = 0; k < 3; ++k) {
                    float ms = 0.f;
                    if (k == 1 && inits[c]) continue;       // landmark-only call: the step is formed in front of or inside the trial kernel, no solve launch
                    if (hipEventElapsedTime(&ms, pe[k], pe[k + 1]) == hipSuccess) {
                        h->cprof.ms[k] += ms;
                        h->cprof.launches[k]++;
                    }
                }
            }
        }
        if (stalled.empty()) { complete = true; break; }
        std::sort(stalled.begin(), stalled.end());
        static const bool trace = std::getenv("VBA_TRACE") != nullptr;
        if (trace) {
            std::fprintf(stderr, "[vba] pass from call %d:", next);
            for (int w = 0; w < h->W && w < 8; ++w)
                std::fprintf(stderr, " w%d(call %d done %d flags %u ntr %d)", w, head(h, w)->call_idx, head(h, w)->done, head(h, w)->flags, head(h, w)->n_trials);
            std::fprintf(stderr, "\n");
        }
        for (int sc_call : stalled) {
            if (int rc = finish_stalled_call(h, spec(sc_call, false), stall_at, trials)) return rc;
        }
        next = stalled.front() + 1;
        if (next >= ncalls) { complete = true; break; }
    }
    if (!complete) {
        complete = true;
        for (int w = 0; w < h->W; ++w) complete = complete && head(h, w)->call_idx >= ncalls;
        if (!complete) return fail(VBA_ESTATE, "chained schedule did not complete (a window never reached its last call)");
    }
    abandon.armed = false;
    if (trials_total) *trials_total = (int)trials;
    h->par = (par0 + ncalls) & 1;
    h->carry_ok = emit_kind;
    h->stepped = true;
    h->last_pipelined = false;
    h->last_iter = iters[ncalls - 1];
    h->last_init = inits[ncalls - 1];
    return VBA_OK;
}

static int vba_set_schedule_graph(vba_handle h, int on) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc = settle(h)) return rc;
    h->graph_enabled = on != 0;
    return VBA_OK;
}

int vba_schedule_graph_stats(vba_handle h, int* captures, int* replays) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (captures) *captures = (int)h->graph_captures;
    if (replays) *replays = (int)h->graph_replays;
    return VBA_OK;
}

static int vba_set_chain_profile(vba_handle h, int on) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc = settle(h)) return rc;
    h->cprof.on = on != 0;
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

int vba_chain_profile(vba_handle h, double* ms, int64_t* launches, int reset) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    for (int k = 0; k < 3; ++k) {
        if (ms) ms[k] = h->cprof.ms[k];
        if (launches) launches[k] = h->cprof.launches[k];
        if (reset) { h->cprof.ms[k] = 0.0; h->cprof.launches[k] = 0; }
    }
    return VBA_OK;
}

int vba_step_profiled(vba_handle h, int iter, int initialize, float* ms) {
    if (!ms) return fail(VBA_EINVAL, "null ms");
    return step_impl(h, iter, initialize, ms);
}

static int take_back(vba_handle h, double* states_out, double* lamda_out, double* last_hessian, int* n_trials, unsigned* flags) {
    if (h->back_valid) {        // read back together with the step: no second wait
        const WinScalars* sc = reinterpret_cast<const WinScalars*>(h->h_back + (size_t)h->n_max * 10);
        if (states_out) std::memcpy(states_out, h->h_back, (size_t)h->n[0] * 80);
        unpack_scalars(sc, h->par, lamda_out, last_hessian, n_trials, flags);
        return VBA_OK;
    }
    return vba_get_states(h, 0, states_out, lamda_out, last_hessian, n_trials, flags);
}

int vba_iterate(vba_handle h, int iter, int initialize, double lamda_in, const double* states_in, double* states_out,
                double* lamda_out, double* last_hessian, int* n_trials, unsigned* flags) {
    if (int rc = vba_set_states(h, 0, states_in, lamda_in)) return rc;
    // the next call of this kind replaces the states again: nothing to carry over
    if (int rc = step_impl(h, iter, initialize, nullptr, false, 0)) return rc;
    return take_back(h, states_out, lamda_out, last_hessian, n_trials, flags);
}

static bool can_pipeline(vba_handle h);
static bool host_watch_changed(vba_handle h);
static int iterate_pipelined(vba_handle h, int iter, int initialize, double* states_out, double* lamda_out, double* last_hessian,
                             int* n_trials, unsigned* flags);

// vba_iterate as the FIRST call of a driver loop whose following calls will be vba_iterate_resident: the states go up, and the call
// itself is served like a resident one -- returned as soon as its accept test is known, with the next call already enqueued behind
// it (a caller that does not come back with a resident call pays for that speculation: use vba_iterate there).
int vba_iterate_open(vba_handle h, int iter, int initialize, double lamda_in, const double* states_in, double* states_out,
                     double* lamda_out, double* last_hessian, int* n_trials, unsigned* flags) {
    if (int rc = vba_set_states(h, 0, states_in, lamda_in)) return rc;
    if (can_pipeline(h)) return iterate_pipelined(h, iter, initialize, states_out, lamda_out, last_hessian, n_trials, flags);
    const bool watch_changed = host_watch_changed(h);       // (like every resident call: the caller relies on it)
    if (int rc = step_impl(h, iter, initialize, nullptr, false, 0)) return rc;
    if (int rc = take_back(h, states_out, lamda_out, last_hessian, n_trials, flags)) return rc;
    if (flags && watch_changed) *flags |= VBA_FLAG_HOST_CHANGED;
    return VBA_OK;
}

// The driver loop `for iter in range(20): states, ... = BA(iter, states, ...)` (od_pipe.py:1036-1040) hands every call the
// result of the one before, through the host.  Served call by call the device idles while the host unpacks one result and
// enqueues the next call, and the host idles while the device works.  Here the two overlap: behind the call that is being
// returned the NEXT call is enqueued speculatively (what follows iter k is learnt from the caller: k + 1 until told
// otherwise), its first kernel evaluates the accept test of the call in front -- exactly the chained schedule of
// vba_run_schedule, one link at a time -- and the host waits only for that kernel plus a 40 kB copy on a side stream,
// while the rest of the speculated call runs under the caller's feet.  When the caller comes back with the predicted
// arguments the call is already on its way.  A wrong guess costs one call's worth of device time and the carried keys
// (settle); a first trial that is not cleanly accepted sends this call through the ordinary LM loop.  Same bits as
// vba_step: the kernels, their order inside a call and the accept test are those of the chained schedule.
static bool host_watch_changed(vba_handle h) {
    for (const auto& w : h->watch)
        if (w.live && std::memcmp(w.live, w.copy, w.bytes) != 0) return true;
    return false;
}
// ... the same on the handle's helper thread: begin before the enqueues, end once the device has answered.  Small watch lists
// (under 64 kB) are compared in place by watch_end: waking a thread costs more than that.
static size_t host_watch_bytes(vba_handle h) {
    size_t b = 0;
    for (const auto& w : h->watch) if (w.live) b += w.bytes;
    return b;
}
static bool watch_begin(vba_handle h) {
    if (host_watch_bytes(h) < 65536) return false;
    auto& W = h->ww;
    if (W.started && W.owner != getpid()) return false;     // forked child: no helper here, the caller compares in place
    if (!W.started) {
        W.started = true;
        W.owner = getpid();
        W.th = std::thread([h]() {
            auto& Q = h->ww;
            unsigned long long taken = 0;
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(Q.m);
                    Q.cv.wait(lk, [&] { return Q.quit || Q.seq != taken; });
                    if (Q.quit) return;
                    taken = Q.seq;
                }
                Q.changed = host_watch_changed(h);
                Q.done_seq.store(taken, std::memory_order_release);
            }
        });
    }
    {
        std::lock_guard<std::mutex> lk(W.m);
        ++W.seq;
    }
    W.cv.notify_one();
    return true;
}
static bool watch_end(vba_handle h, bool begun) {
    if (!begun) return host_watch_changed(h);
    auto& W = h->ww;
    // (bounded: a helper that does not answer within 20 ms -- a forked child has none, a starved host may park it -- is not waited
    // for; the comparison is then made here, beside it if it still runs: both only read)
    const auto t0 = std::chrono::steady_clock::now();
    const unsigned long long mine = W.seq;      // (written by this thread only)
    for (unsigned spins = 0; W.done_seq.load(std::memory_order_acquire) != mine; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return host_watch_changed(h);
    }
    return W.changed;
}
// the helper has answered every request (a wait that gave up after 20 ms may have left it comparing): before the watch list changes
static void watch_quiesce(vba_handle h) {
    auto& W = h->ww;
    if (!W.started || W.owner != getpid()) return;          // (a forked child has no helper to wait for)
    while (W.done_seq.load(std::memory_order_acquire) != W.seq) std::this_thread::yield();
}
static void watch_stop(vba_handle h) {
    auto& W = h->ww;
    if (!W.started) return;
    if (W.owner != getpid()) {      // forked child: the thread object refers to a thread of the parent -- let go of it, never join
        W.th.detach();
        W.started = false;
        return;
    }
    {
        std::lock_guard<std::mutex> lk(W.m);
        W.quit = true;
    }
    W.cv.notify_one();
    W.th.join();
    W.started = false;
}

static bool can_pipeline(vba_handle h) {
    // (whichever kernel forms the trial states of an unpivoted call -- the trial kernel, or with fusion bit 0 off the fused landmark-only
    // assembly / the recovery of the partitioned solve -- also writes them to mapped host memory)
    return h->pipeline && h->W == 1 && h->h_states_map && h->carry_enabled && h->warm_enabled >= 1 && h->fold_enabled && h->inline_select &&
           h->V.wbucket != nullptr && h->pivot_mode == 0 && h->V.chunk > 0 && h->V.lat;
}

static int iterate_pipelined(vba_handle h, int iter, int initialize, double* states_out, double* lamda_out, double* last_hessian,
                             int* n_trials, unsigned* flags) {
    if (int rc = ready(h)) return rc;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    constexpr int emit_kind = 2;
    initialize = initialize ? 1 : 0;
    // what the caller did after the previous resident call: remember it
    if (h->prev_res_iter >= 0) {
        h->pred_iter[h->prev_res_iter & 63] = iter;
        h->pred_init[h->prev_res_iter & 63] = initialize;
    }
    bool consumed = false;
    if (h->spec.valid) {
        if (h->spec.iter == iter && h->spec.init == initialize && h->spec.reg == h->reg) {
            consumed = true;
            h->spec_hits++;
        } else if (int rc = settle(h)) {
            return rc;
        }
    }
    struct Abandon {
        vba_handle h; bool armed = true;
        ~Abandon() { if (armed) { h->need_hist_reset = true; h->have_state.assign(h->W, 0); h->carry_ok = 0; h->spec.valid = false; h->prev_res_iter = -1; } }
    } abandon{h};
    auto call_spec = [&](int c, int it, int in, int carry, bool fold) {
        CallSpec q;
        q.iter = it; q.initialize = in; q.call = c; q.par = (h->chain_par0 + c) & 1;
        q.carry = carry; q.emit = emit_kind; q.fold = fold;
        q.host_out = true;
        return q;
    };
    bool watch_changed = false;
    int c;                      // index of THIS call in the open chain
    if (consumed) {
        c = h->spec.c;
        h->spec.valid = false;
    } else {                    // open a chain with this call as its call 0
        const int carry0 = h->carry_ok;
        h->carry_ok = 0;
        h->shc.carried = false;
        h->chain_par0 = h->par;
        c = 0;
        const CallSpec q = call_spec(0, iter, initialize, carry0, false);
        CallCtx C;
        view_for_call(h, C.V, q);
        if (h->need_hist_reset) {
            DevView Q = C.V;
            for (int p = 0; p < 2; ++p) { Q.par = p; launch_clear_hist(Q, 1, s); }
            h->need_hist_reset = false;
            h->hist_dirty = false;
        }
        if (!carry0 && h->hist_dirty) launch_clear_hist(C.V, 0, s);
        launch_reset_calls(C.V, s);
        h->h_head[0].call_idx = 0; h->h_head[0].done = 0; h->h_head[0].flags = 0;
        if (int rc = enqueue_front(h, C, q, false, nullptr)) return rc;
        enqueue_trial(h, C, q, true);
    }
    h->hist_dirty = true;
    // the call behind it, speculatively: its first kernel decides this one
    int ni = h->pred_iter[iter & 63], nin = h->pred_init[iter & 63];
    if (ni == -1) { ni = iter + 1; nin = initialize; }
    const bool speculate = ni >= 0;
    const CallSpec qc = call_spec(c, iter, initialize, emit_kind, false);       // (this call, as the stalled path needs it)
    const int par_c = qc.par;
    const bool watching = watch_begin(h);
    struct WatchJoin {          // (an early return must not leave the helper comparing buffers the caller may free)
        vba_handle h; bool begun; bool joined = false;
        bool end() { joined = true; return watch_end(h, begun); }
        ~WatchJoin() { if (begun && !joined) (void)watch_end(h, true); }
    } wj{h, watching};
    if (speculate) {
        const CallSpec qn = call_spec(c + 1, ni, nin, emit_kind, true);
        CallCtx C;
        view_for_call(h, C.V, qn);
        fill_params(C.V.prev, iter, initialize);
        C.after_first = h->ev_first;
        if (int rc = enqueue_front(h, C, qn, false, nullptr)) return rc;
        enqueue_trial(h, C, qn, true);
        // the first kernel of the speculated call has decided this one; the trial states and the outcome are in mapped host
        // memory by then (k_trial, fold_commit): no copy, the rest of the speculated call runs on under the caller's feet
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventSynchronize(h->ev_first));
        watch_changed = wj.end();       // (compared while the device worked)
    } else {                    // nothing resident is expected behind this call: decide it with a launch of its own
        CallCtx C;
        view_for_call(h, C.V, qc);
        launch_decide(C.V, nullptr, 0, s);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
        watch_changed = wj.end();
    }
    h->stepped = true;
    h->last_pipelined = true;
    h->last_iter = iter;
    h->last_init = initialize;
    h->back_valid = false;
    const bool clean = (int)head(h, 0)->call_idx >= c + 1;
    if (clean) {
        h->par = par_c ^ 1;             // the trial buffer of this call is the next call's input
        const volatile WinHead* hd = head(h, 0);
        if (states_out) std::memcpy(states_out, h->h_states_map + (size_t)par_c * h->n_max * 10, (size_t)h->n[0] * 80);
        if (lamda_out) *lamda_out = hd->lamda;
        if (last_hessian) for (int k = 0; k < 81; ++k) last_hessian[k] = hd->last_hessian[k];
        if (n_trials) *n_trials = 1;    // (a clean first trial)
        if (flags) *flags = (hd->flags & 7u) | (watch_changed ? VBA_FLAG_HOST_CHANGED : 0u);
        if (speculate) {
            h->spec.valid = true; h->spec.iter = ni; h->spec.init = nin; h->spec.reg = h->reg; h->spec.c = c + 1;
            h->carry_ok = 0;            // (the keys of the result belong to the speculated call now; settle() keeps the books)
        } else {
            h->carry_ok = emit_kind;
        }
        h->prev_res_iter = iter;
        abandon.armed = false;
        return VBA_OK;
    }
    // Not a clean first trial (rejected, pivot check failed, warm select missed): the speculated call has skipped itself
    // (the window never moved on to it); finish this call the ordinary way.
    HIPCHK(hipStreamSynchronize(s));
    {
        std::vector<int> stall_at(1, c);
        long trials = 0;
        if ((int)head(h, 0)->call_idx != c) return fail(VBA_ESTATE, "pipelined call: the window is at call " + std::to_string(head(h, 0)->call_idx) + ", expected " + std::to_string(c));
        if (int rc = finish_stalled_call(h, qc, stall_at, trials)) return rc;
    }
    h->par = par_c ^ 1;
    h->carry_ok = emit_kind;            // the accepted (or last) trial left the next call's keys behind
    h->prev_res_iter = iter;
    abandon.armed = false;
    if (int rc = vba_get_states(h, 0, states_out, lamda_out, last_hessian, n_trials, flags)) return rc;
    if (flags && watch_changed) *flags |= VBA_FLAG_HOST_CHANGED;
    return VBA_OK;
}

// The next call of a driver loop that hands BA() the states it got back from the previous call: nothing to upload, the
// device already holds them (and the carried keys of the last accepted trial stay usable).
int vba_iterate_resident(vba_handle h, int iter, int initialize, double* states_out, double* lamda_out, double* last_hessian,
                         int* n_trials, unsigned* flags) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (!h->stepped) return fail(VBA_ESTATE, "vba_iterate_resident follows a call that left its result on the device");
    if (can_pipeline(h)) return iterate_pipelined(h, iter, initialize, states_out, lamda_out, last_hessian, n_trials, flags);
    const bool watch_changed = host_watch_changed(h);
    if (int rc = step_impl(h, iter, initialize, nullptr, true, 0)) return rc;
    if (int rc = take_back(h, states_out, lamda_out, last_hessian, n_trials, flags)) return rc;
    if (flags && watch_changed) *flags |= VBA_FLAG_HOST_CHANGED;
    return VBA_OK;
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

// ------------------------------------------------------------------------------------------------ sharded mode
int64_t vba_sh_partial_count(int n) { return 27 * (int64_t)n + 2; }

// the device view of the sharded call in flight: classic kernels throughout (the bands go through memory, the trial reads
// the trial states the recovery wrote, every accept test is its own launch)
static void sharded_view(vba_handle h, DevView& V) {
    CallSpec c;
    c.iter = h->last_iter; c.initialize = h->last_init; c.call = -1; c.par = h->par;
    view_for_call(h, V, c);
    // Since round 3 the pose-chain part of a sharded call uses the latency-mode kernels of the handle (one window): the
    // dynamics factor rides in the accumulation's grid, the chunk elimination forms its own blocks, the trial kernel forms
    // the step (no assembly / recovery launches; the landmark-only phase has no solve launch at all).  What stays classic
    // is everything keyed to the exchanges: keys recomputed per call, exact select over the gathered keys, every accept
    // test a launch of its own on the gathered sums.
    V.fuse_walk = 0;
    V.m_total = h->V.m_total;
}

int vba_sh_stage1(vba_handle h, int iter, int initialize, int64_t m_total, double* d_abs_local) {
    if (!h || !d_abs_local || m_total < 1) return fail(VBA_EINVAL, "bad argument");
    if (int rc_settle = settle(h)) return rc_settle;
    if (h->W != 1) return fail(VBA_EINVAL, "sharded mode uses a single window per handle");
    if (h->reg) return fail(VBA_EINVAL, "sharded mode does not take a prior (vba_set_prior)");
    if (int rc = ready(h)) return rc;
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    h->V.m_total = m_total;
    h->last_iter = iter;
    h->last_init = initialize;
    h->carry_ok = false;
    h->shc.carried = false;
    DevView V;
    sharded_view(h, V);
    if (h->hist_dirty || h->need_hist_reset) {
        DevView Q = V;
        for (int p = 0; p < 2; ++p) { Q.par = p; launch_clear_hist(Q, 1, s); }
        h->hist_dirty = h->need_hist_reset = false;
    }
    launch_obs_residual(V, d_abs_local, s);
    HIPCHK(hipGetLastError());
    return VBA_OK;
}

int vba_sh_stage2(vba_handle h, const double* d_abs_all, int64_t count_all, double* d_partial_local) {
    if (!h || !d_abs_all || !d_partial_local || count_all < 1) return fail(VBA_EINVAL, "bad argument");
    if (h->V.m_total < 1 || count_all < 2 * h->V.m_total) return fail(VBA_ESTATE, "stage1 has not run or count_all < 2*m_total");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    DevView V;
    sharded_view(h, V);
    V.abs_all = d_abs_all;
    V.abs_all_count = count_all;
    launch_select(V, true, s);          // digit 0 over the gathered keys as well
    // the dynamics factor is a function of the states only: its blocks ride in this grid
    h->sh_rode = !h->last_init && V.lat;
    V.dyn_in_acc = h->sh_rode ? 1 : 0;
    launch_obs_accumulate(V, s);
    launch_shard_pack(V, d_partial_local, s);
    HIPCHK(hipGetLastError());
    return VBA_OK;
}

int vba_sh_stage3(vba_handle h, const double* d_partial_all, int ranks, double* d_trial_local) {
    if (!h || !d_trial_local) return fail(VBA_EINVAL, "bad argument");
    if (h->V.m_total < 1) return fail(VBA_ESTATE, "stage1 has not run");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    CallSpec c;
    c.iter = h->last_iter; c.initialize = h->last_init; c.call = -1; c.par = h->par;
    CallCtx C;
    sharded_view(h, C.V);
    DevView& V = C.V;
    const bool init = h->last_init != 0;
    if (d_partial_all) {    // first trial of this call; NULL = another LM trial on the same system
        if (ranks < 1) return fail(VBA_EINVAL, "ranks must be >= 1");
        launch_shard_reduce(V, d_partial_all, ranks, s);
        if (!init && !h->sh_rode) launch_dynamics(V, s);
        // who reads the bands from memory?  Nobody when the trial kernel solves the 6x6 systems itself (landmark-only) or
        // the chunk elimination forms its own blocks (full phase)
        const bool need_bands = init ? V.fused_trial != 1 : !solve_forms_blocks(V);
        if (need_bands) launch_assemble(V, 0, s);
        h->sh_bands_ready = need_bands;
        // every rank holds bit-identical systems (rank-ordered reductions), so the checked unpivoted path and its
        // fallback are taken by all ranks alike: stage4 reports the failed check and the caller's loop repeats stage3
        h->sh_pivot = h->pivot_mode;
    }
    V.pivot = h->sh_pivot;
    C.fuse_assemble = false;
    C.assembled = C.bands_ready = h->sh_bands_ready;
    enqueue_trial(h, C, c, d_partial_all != nullptr);
    h->sh_bands_ready = C.bands_ready;      // (a pivoted repeat of a landmark-only trial assembles the blocks it reads)
    launch_shard_trial_sum(V, d_trial_local, s);
    HIPCHK(hipGetLastError());
    return VBA_OK;
}

int vba_sh_stage4(vba_handle h, const double* d_trial_all, int ranks, int* done) {
    if (!h || !d_trial_all || !done || ranks < 1) return fail(VBA_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    DevView V;
    sharded_view(h, V);
    launch_decide(V, d_trial_all, ranks, h->stream);
    HIPCHK(hipGetLastError());
    if (int rc = read_heads(h)) return rc;
    *done = head(h, 0)->done;
    if (!*done && (head(h, 0)->flags & 8u) && h->sh_pivot == 0) {   // pivot check failed: next stage3 uses the pivoted kernels
        h->sh_pivot = 2;
        h->fallbacks++;
    }
    if (*done) {
        h->stepped = true;
        h->V.m_total = 0;
        h->par ^= 1;            // the trial buffer is the next call's input
    }
    return VBA_OK;
}

// ---- the same protocol with the exchanges issued by the library: RCCL all-gathers on the handle's stream between the stage
// kernels, one host call and (per LM trial) one synchronisation per BA() call.  RCCL is resolved at run time from the path the
// caller names -- the copy the process has loaded already when it also uses torch.distributed -- so the library itself
// carries no link-time dependency on it.
namespace {
void* open_rccl(const char* path) {
    void* dl = dlopen(path, RTLD_NOW | RTLD_NOLOAD);        // the instance the process has loaded already, if any
    if (!dl) dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    return dl;
}
}  // namespace

int vba_sh_unique_id(const char* rccl_path, void* id128) {
    if (!rccl_path || !id128) return fail(VBA_EINVAL, "null argument");
    void* dl = open_rccl(rccl_path);
    if (!dl) return fail(VBA_EINVAL, std::string("cannot open ") + rccl_path + ": " + dlerror());
    auto get_id = reinterpret_cast<ncclResult_t (*)(ncclUniqueId*)>(dlsym(dl, "ncclGetUniqueId"));
    if (!get_id) { dlclose(dl); return fail(VBA_EINVAL, "ncclGetUniqueId not found in the named library"); }
    ncclUniqueId id;
    const ncclResult_t rc = get_id(&id);
    dlclose(dl);
    if (rc != ncclSuccess) return fail(VBA_EHIP, "ncclGetUniqueId failed (" + std::to_string((int)rc) + ")");
    static_assert(sizeof(id) == 128, "unique id size");
    std::memcpy(id128, &id, sizeof(id));
    return VBA_OK;
}

int vba_sh_comm_init(vba_handle h, const char* rccl_path, const void* id128, int nranks, int rank) {
    if (!h || !rccl_path || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(VBA_EINVAL, "bad argument");
    if (int rc_settle = settle(h)) return rc_settle;
    if (h->W != 1) return fail(VBA_EINVAL, "sharded mode uses a single window per handle");
    if (h->shc.comm) return fail(VBA_ESTATE, "the handle has a communicator already");
    HIPCHK(hipSetDevice(h->device));
    auto& S = h->shc;
    S.dl = open_rccl(rccl_path);
    if (!S.dl) return fail(VBA_EINVAL, std::string("cannot open ") + rccl_path + ": " + dlerror());
    auto init_rank = reinterpret_cast<ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>(dlsym(S.dl, "ncclCommInitRank"));
    S.all_gather = reinterpret_cast<decltype(S.all_gather)>(dlsym(S.dl, "ncclAllGather"));
    S.comm_destroy = reinterpret_cast<decltype(S.comm_destroy)>(dlsym(S.dl, "ncclCommDestroy"));
    S.error_string = reinterpret_cast<decltype(S.error_string)>(dlsym(S.dl, "ncclGetErrorString"));
    if (!init_rank || !S.all_gather || !S.comm_destroy || !S.error_string) {
        dlclose(S.dl);
        S = {};
        return fail(VBA_EINVAL, "the named library does not export the RCCL entry points");
    }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t rc = init_rank(&S.comm, nranks, id, rank);       // collective: returns when every rank has joined
    if (rc != ncclSuccess) {
        const std::string why = S.error_string(rc);
        dlclose(S.dl);
        S = {};
        return fail(VBA_EHIP, "ncclCommInitRank failed: " + why);
    }
    S.nranks = nranks;
    S.rank = rank;
    return VBA_OK;
}

int vba_sh_comm_destroy(vba_handle h) {
    if (!h) return VBA_OK;
    auto& S = h->shc;
    if (!S.comm && !S.buf) return VBA_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    if (S.comm) S.comm_destroy(S.comm);
    if (S.buf) hipFree(S.buf);
    if (S.buf2) hipFree(S.buf2);
    if (S.dl) dlclose(S.dl);
    S = {};
    return VBA_OK;
}

namespace {

// exchange buffers of a sharded window (both protocols); m_total rows over all ranks
int sh_ensure_buffers(vba_handle h, int64_t m_total) {
    auto& S = h->shc;
    const int n = h->n[0];
    const int64_t m_pad = (m_total + S.nranks - 1) / S.nranks;     // equal all-gather slots
    if (h->m[0] > m_pad) return fail(VBA_EINVAL, "this rank holds more rows than ceil(m_total / ranks)");
    const int64_t pc = vba_sh_partial_count(n);
    const int64_t R = S.nranks;
    if (S.m_total != m_total || S.n != n) {       // (re)size the exchange buffers; the padding of a slot sorts above every |r|
        HIPCHK(hipStreamSynchronize(h->stream));
        if (S.buf) { HIPCHK(hipFree(S.buf)); S.buf = nullptr; }
        const int64_t total = 2 * m_pad * (1 + R) + pc * (1 + R) + 2 * (1 + R) + 64;
        HIPCHK(hipMalloc((void**)&S.buf, (size_t)total * 8));
        S.abs_local = S.buf;
        S.abs_all = S.abs_local + 2 * m_pad;
        S.partial_local = S.abs_all + 2 * m_pad * R;
        S.partial_all = S.partial_local + pc;
        S.trial_local = S.partial_all + pc * R;
        S.trial_all = S.trial_local + 2;
        S.m_total = m_total; S.m_pad = m_pad; S.n = n;
        S.m_local = -1;
        S.carried = false;
    }
    if (S.m_local != h->m[0]) {     // stage 1 writes 2 * m_local keys: everything behind them must sort above every |r| -- also after
                                    // a re-upload with FEWER rows of this rank than before (the old shard's keys would enter the median)
        HIPCHK(hipStreamSynchronize(h->stream));
        std::vector<double> inf((size_t)(2 * m_pad), INFINITY);
        HIPCHK(hipMemcpy(S.abs_local, inf.data(), inf.size() * 8, hipMemcpyHostToDevice));
        S.m_local = h->m[0];
        S.carried = false;
    }
    // The warm bins must keep the bin of the GLOBAL median short (the gathered buckets of that bin are ranked as one list of at
    // most 1024 keys): their width follows the key count over all ranks, not this rank's share -- 1/256 binade up to 300 000
    // keys, 1/512 up to 600 000, 1/1024 beyond (range [c/2, 2c): a median that moves further between two calls is a miss and
    // takes the exact select, as everywhere).
    if (S.protocol == 1) {
        const int64_t keys = 2 * m_total;
        const int shift = keys <= 300000 ? 44 : (keys <= 600000 ? 43 : 42);
        if (h->V.warm_shift != shift) {
            h->V.warm_shift = shift;
            h->carry_ok = 0;
            S.carried = false;
        }
    }
    // carried-keys protocol: [hist 1024 | part_next nblk_obs | part_trial trial_stride] per call parity, the gathered copy, the
    // bucket slots [count | keys bucket_cap]
    const int nbo = h->V.nblk_obs, cap = h->V.bucket_cap;
    const int lenA = (1024 + nbo + h->V.trial_stride + 3) & ~3, lenB = (cap + 1 + 3) & ~3;      // (16-byte aligned slots)
    if (S.protocol == 1 && cap > 0 && (S.lenA != lenA || S.lenB != lenB || !S.buf2)) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (S.buf2) { HIPCHK(hipFree(S.buf2)); S.buf2 = nullptr; }
        const size_t total = (size_t)lenA * (2 + R) + (size_t)lenB * (1 + R) + 64;
        HIPCHK(hipMalloc((void**)&S.buf2, total * 8));
        HIPCHK(hipMemset(S.buf2, 0, total * 8));
        HIPCHK(hipStreamSynchronize(nullptr));      // (the fill runs on the null stream, the exchanges on the handle's non-blocking one)
        S.sendA[0] = S.buf2;
        S.sendA[1] = S.sendA[0] + lenA;
        S.recvA = S.sendA[1] + lenA;
        S.sendB = S.recvA + (size_t)lenA * R;
        S.recvB = S.sendB + lenB;
        S.lenA = lenA; S.lenB = lenB;
        S.carried = false;
    }
    return VBA_OK;
}

int sh_gather(vba_handle h, const double* src, double* dst, int64_t count) {
    auto& S = h->shc;
    const ncclResult_t rc = S.all_gather(src, dst, (size_t)count, ncclDouble, S.comm, h->stream);
    if (rc != ncclSuccess) return fail(VBA_EHIP, std::string("ncclAllGather failed: ") + S.error_string(rc));
    return VBA_OK;
}

// ---- round-3 protocol: every call gathers all |r| keys (kept for comparison, vba_sh_set_protocol(h, 0))
int sh_call_classic(vba_handle h, int iter, int initialize, int64_t m_total, int* n_trials) {
    auto& S = h->shc;
    const int n = h->n[0];
    const int64_t pc = vba_sh_partial_count(n);
    if (int rc = vba_sh_stage1(h, iter, initialize, m_total, S.abs_local)) return rc;
    if (int rc = sh_gather(h, S.abs_local, S.abs_all, 2 * S.m_pad)) return rc;
    if (int rc = vba_sh_stage2(h, S.abs_all, 2 * S.m_pad * S.nranks, S.partial_local)) return rc;
    if (int rc = sh_gather(h, S.partial_local, S.partial_all, pc)) return rc;
    int trials = 0;
    for (bool first = true;; first = false) {
        if (int rc = vba_sh_stage3(h, first ? S.partial_all : nullptr, S.nranks, S.trial_local)) return rc;
        if (int rc = sh_gather(h, S.trial_local, S.trial_all, 2)) return rc;
        int done = 0;
        if (int rc = vba_sh_stage4(h, S.trial_all, S.nranks, &done)) return rc;
        ++trials;
        if (done) break;
        // lamda runs out after 9 trials (+ one repeat for a pivoted fallback): the device never reported an outcome
        if (trials >= 24) return fail(VBA_ESTATE, "sharded BA call: the LM loop did not terminate within 24 trials");
    }
    if (n_trials) *n_trials = trials;
    return VBA_OK;
}

// ---- carried-keys protocol.  One BA() call of a rank, first trial (everything asynchronous on the handle's stream):
//   front, carried   k_sh_front   [accept test of the call in front on the gathered block sums] + the R warm histograms added up,
//                                 the bin of the global median resolved, this rank's bucket of it -> sendB
//                    all-gather B buckets of that bin (<= 8 kB per rank)
//                    k_obs_accumulate  ranks the gathered buckets in its prologue (exact median), weights, local per-pose sums
//                                 straight into the exchange buffer; the dynamics factor rides in its grid
//   front, classic   (first call on new states; a call whose carried select missed)  residual pass -> all-gather of all keys
//                                 -> exact select -> accumulation -> pack
//                    all-gather C per-pose normal equations (27 n + 2 doubles) [-> rank-ordered reduce; one rank: used in place]
//   solve            the handle's latency-mode kernels (every rank redundantly: bit-identical systems)
//   trial            k_trial: trial residuals of the local rows + next call's keys in bin buckets, warm histogram and block sums,
//                                 the latter two written straight into sendA
//                    all-gather A [histogram | block sums] (~12 kB per rank) -- decided by the NEXT call's k_sh_front
// Every rank enqueues the same collectives whether its window runs a call or skips it (a window that stalls at a call --
// trial not cleanly accepted, select missed -- leaves the rest of the chain untouched on EVERY rank alike, the decisions being
// taken on gathered data), and the host synchronises once per schedule.
struct Sh2 {
    vba_handle h;
    int R;
    const int *iters, *inits;
    int ncalls, par0;

    CallSpec spec(int c, bool carried, bool fold) const {
        CallSpec q;
        q.iter = iters[c]; q.initialize = inits[c]; q.call = c; q.par = (par0 + c) & 1;
        q.carry = carried ? 2 : 0;
        q.emit = 2;
        q.fold = fold;
        return q;
    }
    // the view of call q: exchange buffers where the kernels write anyway
    void view(DevView& V, const CallSpec& q) const {
        auto& S = h->shc;
        view_for_call(h, V, q);
        V.fuse_walk = 0;
        V.m_total = S.m_total;
        V.hist0_ext[0] = reinterpret_cast<unsigned*>(S.sendA[0]);
        V.hist0_ext[1] = reinterpret_cast<unsigned*>(S.sendA[1]);
        V.part_next = S.sendA[q.par ^ 1] + 1024;                     // the trial of this call writes the next call's slot
        V.part_trial = V.part_next + V.nblk_obs;
        V.sel_inline = 0;
        V.pivot = h->sh_pivot;
    }

    int enqueue_call(int c, bool carried, bool fold) {
        auto& S = h->shc;
        hipStream_t s = h->stream;
        const CallSpec q = spec(c, carried, fold);
        const int n = h->n[0];
        const int64_t pc = vba_sh_partial_count(n);
        const bool init = q.initialize != 0;
        CallCtx C;
        view(C.V, q);
        DevView& V = C.V;
        h->sh_pivot = h->pivot_mode;
        V.pivot = h->sh_pivot;
        V.dyn_in_acc = (!init && V.lat) ? 1 : 0;
        if (carried) {
            if (fold) fill_params(V.prev, iters[c - 1], inits[c - 1]);
            if (R > 1) V.wmax_ext = reinterpret_cast<unsigned long long*>(S.partial_local + (size_t)27 * n);     // (the front clears it)
            launch_sh_front(V, S.recvA, R, S.lenA, S.sendB, fold ? 1 : 0, 1, s);
            if (int rc = sh_gather(h, S.sendB, S.recvB, S.lenB)) return rc;
            DevView Va = V;             // the accumulation: gathered buckets in, sums straight into the exchange buffer
            Va.sel_slots = S.recvB; Va.sel_nslots = R; Va.sel_slot_stride = S.lenB;
            Va.Hraw = S.partial_local; Va.braw = S.partial_local + (size_t)21 * n;
            launch_obs_accumulate(Va, s);
            if (int rc = sh_gather(h, S.partial_local, S.partial_all, pc)) return rc;
            if (R > 1) launch_shard_reduce(V, S.partial_all, R, s, 0);
            else { V.Hraw = S.partial_all; V.braw = S.partial_all + (size_t)21 * n; }      // one rank: the gathered copy IS the sum
        } else {
            // no carried keys: residual pass, all keys gathered, exact select
            DevView Q = V;
            for (int p = 0; p < 2; ++p) { Q.par = p; launch_clear_hist(Q, 1, s); }
            V.redo = 2;                 // (a window repeating a call whose carried select missed takes part)
            launch_sh_clear_miss(V, s);
            launch_obs_residual(V, S.abs_local, s);
            if (int rc = sh_gather(h, S.abs_local, S.abs_all, 2 * S.m_pad)) return rc;
            DevView Vs = V;
            Vs.abs_all = S.abs_all; Vs.abs_all_count = 2 * S.m_pad * R;
            launch_select(Vs, true, s);
            launch_obs_accumulate(Vs, s);
            launch_shard_pack(V, S.partial_local, s);
            if (int rc = sh_gather(h, S.partial_local, S.partial_all, pc)) return rc;
            launch_shard_reduce(V, S.partial_all, R, s, 1);
        }
        if (!init && !V.dyn_in_acc) launch_dynamics(V, s);
        const bool need_bands = init ? V.fused_trial != 1 : !solve_forms_blocks(V);
        if (need_bands) launch_assemble(V, 0, s);
        C.fuse_assemble = false;
        C.assembled = C.bands_ready = need_bands;
        enqueue_trial(h, C, q, true);
        if (int rc = sh_gather(h, S.sendA[q.par ^ 1], S.recvA, S.lenA)) return rc;
        return VBA_OK;
    }

    // call c stalled at its first trial (rejected, pivot check failed): the ordinary LM loop, the trial sums gathered per round
    int finish_stalled(int c, bool carried, long& trials) {
        auto& S = h->shc;
        hipStream_t s = h->stream;
        const CallSpec q = spec(c, carried, false);
        CallCtx C;
        view(C.V, q);
        DevView& V = C.V;
        const int n = h->n[0];
        if (R == 1 && carried) { V.Hraw = S.partial_all; V.braw = S.partial_all + (size_t)21 * n; }    // (where the call's sums live, see enqueue_call)
        V.redo = 2;
        const bool init = q.initialize != 0;
        C.fuse_assemble = false;
        C.assembled = C.bands_ready = init ? V.fused_trial != 1 : !solve_forms_blocks(V);
        for (int round = 0; round <= 24; ++round) {
            launch_shard_trial_sum(V, S.trial_local, s);
            if (int rc = sh_gather(h, S.trial_local, S.trial_all, 2)) return rc;
            launch_decide(V, S.trial_all, R, s);
            HIPCHK(hipGetLastError());
            if (int rc = read_heads(h)) return rc;
            if (head(h, 0)->done) {
                // the last trial left the next call's keys, histogram and block sums: exchange them as every trial's are
                if (int rc = sh_gather(h, S.sendA[q.par ^ 1], S.recvA, S.lenA)) return rc;
                S.fallbacks_lm++;
                return VBA_OK;
            }
            if (round == 24) break;
            if ((head(h, 0)->flags & 8u) && h->sh_pivot == 0) { h->sh_pivot = 2; h->fallbacks++; }
            V.pivot = h->sh_pivot;
            enqueue_trial(h, C, q, false);
            ++trials;
        }
        return fail(VBA_ESTATE, "sharded BA call: the LM loop did not terminate within 24 rounds");
    }
};

}  // namespace

int vba_sh_set_protocol(vba_handle h, int carried_keys) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc = settle(h)) return rc;
    h->shc.protocol = carried_keys ? 1 : 0;
    h->shc.carried = false;
    return VBA_OK;
}

int vba_sh_stats(vba_handle h, int64_t* bytes_first_exchange, int64_t* fallbacks_miss, int64_t* fallbacks_lm) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    auto& S = h->shc;
    if (bytes_first_exchange) *bytes_first_exchange = (S.protocol == 1 && S.lenA && h->m_max == S.m_pad) ? (int64_t)S.lenA * 8 : 16 * S.m_pad;
    if (fallbacks_miss) *fallbacks_miss = S.fallbacks_miss;
    if (fallbacks_lm) *fallbacks_lm = S.fallbacks_lm;
    return VBA_OK;
}

int vba_sh_run_schedule(vba_handle h, int ncalls, const int* iters, const int* inits, int64_t m_total, int* trials_total) {
    if (!h || !iters || !inits || ncalls < 1 || m_total < 1) return fail(VBA_EINVAL, "bad argument");
    auto& S = h->shc;
    if (!S.comm) return fail(VBA_ESTATE, "vba_sh_comm_init has not run");
    if (int rc_settle = settle(h)) return rc_settle;
    if (h->W != 1) return fail(VBA_EINVAL, "sharded mode uses a single window per handle");
    if (h->reg) return fail(VBA_EINVAL, "sharded mode does not take a prior (vba_set_prior)");
    if (int rc = ready(h)) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (int rc = sh_ensure_buffers(h, m_total)) return rc;
    // (the exchange buffers are laid out by the handle's geometry -- observation blocks, bucket capacity --, which must be the same on
    // every rank: a handle created for exactly ceil(m_total / ranks) rows; any other takes the round-3 protocol, whose slots are sized
    // by m_total alone)
    // (the carried-keys protocol is written for the trial kernel that forms the step; a handle whose mask was chosen by the library --
    // big single windows get 14 -- takes 15 here)
    if (h->fusion_auto && S.protocol != 0 && !(h->fusion & 1)) h->fusion = 15;
    if (S.protocol == 0 || !h->V.wbucket || !h->V.lat || !(h->fusion & 1) || h->V.chunk <= 0 || h->m_max != S.m_pad) {
        // the round-3 protocol, call by call
        long total = 0;
        for (int c = 0; c < ncalls; ++c) {
            int t = 0;
            if (int rc = sh_call_classic(h, iters[c], inits[c], m_total, &t)) return rc;
            total += t;
        }
        S.carried = false;
        if (trials_total) *trials_total = (int)total;
        return VBA_OK;
    }
    hipStream_t s = h->stream;
    h->V.m_total = m_total;
    Sh2 P{h, S.nranks, iters, inits, ncalls, h->par};
    bool carried0 = S.carried && h->carry_ok == 2 && S.carried_par == h->par;
    h->carry_ok = 0;
    S.carried = false;
    struct Abandon {
        vba_handle h; bool armed = true;
        ~Abandon() { if (armed) { h->need_hist_reset = true; h->have_state.assign(h->W, 0); h->carry_ok = 0; h->V.m_total = 0; } }
    } abandon{h};
    {
        DevView V0;
        P.view(V0, P.spec(0, false, false));
        launch_reset_calls(V0, s);
    }
    h->h_head[0].call_idx = 0; h->h_head[0].done = 0; h->h_head[0].flags = 0;
    long trials = 0;
    int next = 0;
    bool first_carried = carried0;
    for (int guard = 0; guard <= 2 * ncalls + 2; ++guard) {
        for (int c = next; c < ncalls; ++c) {
            const bool carried = c == next ? first_carried : true;
            const bool fold = carried && c > next;
            if (int rc = P.enqueue_call(c, carried, fold)) return rc;
            // a call without carried keys in front of it is decided by a launch of its own (nothing folds it) when the NEXT
            // call's front does not: the next call is always carried, so only the last call of the schedule is left over
        }
        {   // the accept test of the last call: the front kernel with nothing to resolve
            CallSpec q = P.spec(ncalls - 1, true, true);
            q.call = ncalls; q.par = (P.par0 + ncalls) & 1;
            DevView V;
            P.view(V, q);
            fill_params(V.prev, iters[ncalls - 1], inits[ncalls - 1]);
            launch_sh_front(V, S.recvA, P.R, S.lenA, S.sendB, 1, 0, s);
        }
        HIPCHK(hipGetLastError());
        if (int rc = read_heads(h)) return rc;
        const int at = (int)head(h, 0)->call_idx;
        if (at >= ncalls) { trials += (long)(ncalls - next); break; }
        if (head(h, 0)->flags & 32u) {          // the carried select of call `at` missed: that call again, exact select over all keys
            trials += (long)(at - next);
            S.fallbacks_miss++;
            h->warm_misses++;
            h->h_head[0].flags = 0;
            next = at;
            first_carried = false;
            continue;
        }
        trials += (long)(at - next + 1);        // (the stalled call's first trial has run)
        if (int rc = P.finish_stalled(at, at == next ? first_carried : true, trials)) return rc;
        next = at + 1;
        first_carried = true;
        if (next >= ncalls) break;
    }
    if ((int)head(h, 0)->call_idx < ncalls) return fail(VBA_ESTATE, "sharded schedule did not complete (the window never reached its last call)");
    abandon.armed = false;
    h->par = (P.par0 + ncalls) & 1;
    h->carry_ok = 2;
    S.carried = true;
    S.carried_par = h->par;
    h->V.m_total = 0;
    h->stepped = true;
    h->last_pipelined = false;
    h->last_iter = iters[ncalls - 1];
    h->last_init = inits[ncalls - 1];
    if (trials_total) *trials_total = (int)trials;
    return VBA_OK;
}

int vba_sh_call(vba_handle h, int iter, int initialize, int64_t m_total, int* n_trials) {
    return vba_sh_run_schedule(h, 1, &iter, &initialize, m_total, n_trials);
}

}  // extern "C"
