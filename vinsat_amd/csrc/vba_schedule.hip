// vba_schedule.hip -- one BA() call as the host enqueues it, and the calls of a driver loop: vba_step (call by call), vba_run_schedule
// (the 20-call loop chained on the device, its first pass replayed as a hipGraph), vba_iterate / _open / _resident (states over
// PCIe; the pipelined loop with its speculated next call and the host watch).
#include "vba_context.h"


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
void enqueue_trial(vba_handle h, CallCtx& C, const CallSpec& c, bool first, hipEvent_t ev_solve, int solve_redo) {
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



// readback >= 0: the states and scalars of that window are copied to the pinned read-back buffer right behind the first
// trial (valid if that trial ends the call: h->back_valid), so that vba_iterate needs one wait instead of two.
int step_impl(vba_handle h, int iter, int initialize, float* prof, bool emit, int readback) {
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


int vba_set_schedule_graph(vba_handle h, int on) {
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

int vba_set_chain_profile(vba_handle h, int on) {
    if (!h) return fail(VBA_EINVAL, "null handle");
    if (int rc = settle(h)) return rc;
    h->cprof.on = on != 0;
    return VBA_OK;
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
void watch_quiesce(vba_handle h) {
    auto& W = h->ww;
    if (!W.started || W.owner != getpid()) return;          // (a forked child has no helper to wait for)
    while (W.done_seq.load(std::memory_order_acquire) != W.seq) std::this_thread::yield();
}
void watch_stop(vba_handle h) {
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

