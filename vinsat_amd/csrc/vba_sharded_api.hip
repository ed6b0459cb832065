// vba_sharded_api.hip -- observation-sharded multi-GPU operation of a window (vba_sh_*): the stage entry
// points for exchanges dispatched by the caller, and the exchanges issued by the library itself over RCCL (resolved at run time).
#include "vba_context.h"

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

