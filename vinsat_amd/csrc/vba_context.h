// vba_context.h -- the host side of libvinsat_ba.so shared by its translation units: the context behind a vba_handle, error
// reporting, and the functions one file defines and another calls.
//   vba_api.hip          context, options, uploads, states, diagnostics (vba_debug_fetch)
//   vba_schedule.hip     the kernels of one BA() call as the host enqueues them: vba_step, vba_run_schedule (chained calls, graph
//                        replay), vba_iterate* (pipelined driver loop, host watch)
//   vba_sharded_api.hip  observation-sharded mode (vba_sh_*)
#pragma once
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

// (everything below is internal to the library: hidden, so that names such as fail / ready / head never meet a host program's)
#pragma GCC visibility push(hidden)
extern thread_local std::string g_err;
int fail(int code, const std::string& msg);
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

// ---- vba_api.hip
void fill_params(StepParams& p, int iter, int initialize);
int check_window(vba_handle h, int window);
int read_heads(vba_handle h);
const volatile WinHead* head(vba_handle h, int w);
hipError_t create_aux_stream(hipStream_t* s);
int settle(vba_handle h, bool boundary = false);
int ready(vba_handle h);
void unpack_scalars(const WinScalars* sc, int par, double* lamda, double* last_hessian, int* n_trials, unsigned* flags);

// ---- vba_schedule.hip
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

void view_for_call(vba_handle h, DevView& V, const CallSpec& c);
int enqueue_front(vba_handle h, CallCtx& C, const CallSpec& c, bool exact_repeat, hipEvent_t* ev);
void enqueue_trial(vba_handle h, CallCtx& C, const CallSpec& c, bool first, hipEvent_t ev_solve = nullptr, int solve_redo = -1);
int step_impl(vba_handle h, int iter, int initialize, float* prof, bool emit = true, int readback = -1);
void watch_stop(vba_handle h);
void watch_quiesce(vba_handle h);
int vba_set_schedule_graph(vba_handle h, int on);      // (options of vba_set_option that live with the schedule)
int vba_set_chain_profile(vba_handle h, int on);
#pragma GCC visibility pop
