// Context management for libvo_hip.so: device selection, the stream every
// kernel is enqueued on, grow-only device buffers and the last-error string.
#include <cstdlib>

#include "vo_internal.h"

namespace vo {

int fail(vo_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

// Grow-only device buffer: reallocates (after draining the stream) only when
// the request exceeds what is already there.
int ensure(vo_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.bytes && b.p) return VO_OK;
    if (bytes == 0) bytes = 256;
    if (b.p) {
        VO_HIP(c, hipStreamSynchronize(c->stream));
        if (c->side) VO_HIP(c, hipStreamSynchronize(c->side));
        if (c->alt.side) VO_HIP(c, hipStreamSynchronize(c->alt.side));
        if (c->aux) VO_HIP(c, hipStreamSynchronize(c->aux));
        VO_HIP(c, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, VO_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    b.bytes = want;
    ++c->alloc_count;
    return VO_OK;
}

static hipEvent_t take_event(vo_ctx* c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

void t_begin_on(vo_ctx* c, int group, hipStream_t s) {
    if (!((c->timing >> group) & 1u) || c->ev_used > 60000) return;
    hipEvent_t a = take_event(c), b = take_event(c);
    if (!a || !b) return;
    c->ev_pairs[group].emplace_back(a, b);
    (void)hipEventRecord(a, s);
}

void t_begin(vo_ctx* c, int group) { t_begin_on(c, group, c->stream); }

void t_end_on(vo_ctx* c, int group, hipStream_t s) {
    if (!((c->timing >> group) & 1u) || c->ev_pairs[group].empty()) return;
    (void)hipEventRecord(c->ev_pairs[group].back().second, s);
}

void t_end(vo_ctx* c, int group) { t_end_on(c, group, c->stream); }

int join_side(vo_ctx* c) {   // (both tail sets: either may still run a batch's tail)
    if (c->side_busy) {
        VO_HIP(c, hipStreamWaitEvent(c->stream, c->ev_rdone, 0));
        c->side_busy = false;
    }
    if (c->alt.side_busy) {
        VO_HIP(c, hipStreamWaitEvent(c->stream, c->alt.ev_rdone, 0));
        c->alt.side_busy = false;
    }
    return VO_OK;
}

int join_tail(vo_ctx* c) {
    if (c->tail_pending) {
        VO_HIP(c, hipStreamWaitEvent(c->stream, c->ev_tail, 0));
        c->tail_pending = false;
    }
    if (c->alt.tail_pending) {
        VO_HIP(c, hipStreamWaitEvent(c->stream, c->alt.ev_tail, 0));
        c->alt.tail_pending = false;
    }
    return VO_OK;
}

int take_truncation(vo_ctx* c, const char* who) {
    if (!c->flags.p) return VO_OK;
    uint32_t bits = 0;
    VO_HIP(c, hipMemcpy(&bits, c->flags.p, 4, hipMemcpyDeviceToHost));
    if (!bits) return VO_OK;
    VO_HIP(c, hipMemset(c->flags.p, 0, 4));
    return truncation_error(c, who, bits);
}

int truncation_error(vo_ctx* c, const char* who, uint32_t bits) {
    std::string what;
    if (bits & VO_TRUNC_SCORE_LIST) what += " [FAST-score selection: ties at the 2 x quota cut exceed the selection table]";
    if (bits & VO_TRUNC_KEYPOINT_LIST) what += " [Harris selection: ties at the quota cut exceed a level's keypoint table]";
    if (bits & VO_TRUNC_OUTPUT) what += " [output: more keypoints than the caller's capacity]";
    return fail(c, VO_ERR_CAPACITY,
                std::string(who) + ": a keypoint list was truncated since the last report (results of the images "
                                   "concerned are incomplete; vo_pair_result.status names the pairs):" + what);
}

}  // namespace vo

using namespace vo;

extern "C" void vo_ctx_destroy(vo_ctx* c);

static bool create_ticket_events(vo_ctx* c) {
    for (int k = 0; k < VO_TICKET_RING; ++k)
        if (hipEventCreateWithFlags(&c->ev_in[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_out[k], hipEventDisableTiming) != hipSuccess)
            return false;
    return true;
}

// Default of an option: the integer in the named environment variable if it is set and admissible.
static bool option_admits(int option, int v) {
    switch (option) {
        case VO_OPT_FIVE_POINT_LITERAL: return v == 0 || v == 1;
        case VO_OPT_FAST_RAISE: return v >= -1 && v <= 1;
        case VO_OPT_KNN_MODE: return v >= -1 && v <= 3;
        case VO_OPT_KNN_WAVES: return v == 0 || v == 1 || v == 4 || v == 8 || v == 16;
        case VO_OPT_KNN_QUERIES_PER_LANE: return v == 0 || v == 1 || v == 2 || v == 4;
        case VO_OPT_SECOND_CHART: return v == 0 || v == 1;
        case VO_OPT_KNN_TRAIN_SPLIT: return v >= 0 && v <= 64;
        case VO_OPT_SOLVE_VARIANT: return v == 0 || v == 1;
        case VO_OPT_PIPELINE_GRAPH: return v == 0 || v == 1;
        default: return false;
    }
}

static void option_defaults(vo_ctx* c) {
    static const struct { int option; const char* env; int dflt; } table[] = {
        {VO_OPT_FIVE_POINT_LITERAL, "VO_FIVE_POINT_LITERAL", 0}, {VO_OPT_FAST_RAISE, "VO_FAST_RAISE", -1},
        {VO_OPT_KNN_MODE, "VO_KNN_MODE", -1},                     {VO_OPT_KNN_WAVES, "VO_KNN_NW", 0},
        {VO_OPT_KNN_QUERIES_PER_LANE, "VO_KNN_QPL", 0},           {VO_OPT_SECOND_CHART, "VO_SECOND_CHART", 1},
        {VO_OPT_KNN_TRAIN_SPLIT, "VO_KNN_SPLIT", 0},              {VO_OPT_SOLVE_VARIANT, "VO_SOLVE_VARIANT", 0},
        {VO_OPT_PIPELINE_GRAPH, "VO_PIPELINE_GRAPH", 0}};
    for (const auto& t : table) {
        int v = t.dflt;
        if (const char* e = getenv(t.env)) {
            char* end = nullptr;
            const long x = strtol(e, &end, 10);
            if (end != e && option_admits(t.option, (int)x)) v = (int)x;
        }
        c->opt[t.option] = v;
    }
}

extern "C" int vo_ctx_set_option(vo_ctx* c, int option, int value) {
    if (!c) return VO_ERR_INVALID;
    if (option < 0 || option >= VO_OPT_COUNT || !option_admits(option, value))
        return fail(c, VO_ERR_INVALID, "vo_ctx_set_option: option " + std::to_string(option) + " does not take the value " +
                                           std::to_string(value));
    c->opt[option] = value;
    return VO_OK;
}

extern "C" int vo_ctx_get_option(const vo_ctx* c, int option, int* value) {
    if (!c || !value || option < 0 || option >= VO_OPT_COUNT) return VO_ERR_INVALID;
    *value = c->opt[option];
    return VO_OK;
}

extern "C" int vo_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

extern "C" int vo_ctx_create(int device, vo_ctx** out) {
    if (!out) return VO_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VO_ERR_HIP;
    if (device < 0 || device >= n) return VO_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return VO_ERR_HIP;
    vo_ctx* c = new vo_ctx();
    c->device = device;
    option_defaults(c);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_init, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_rdone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->alt.side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->alt.ev_init, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->alt.ev_rdone, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->alt.ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fast0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_blur, hipEventDisableTiming) != hipSuccess || !create_ticket_events(c)) {
        vo_ctx_destroy(c);
        return VO_ERR_HIP;
    }
    if (static_scratch_init(c) != VO_OK) {
        vo_ctx_destroy(c);
        return VO_ERR_HIP;
    }
    *out = c;
    return VO_OK;
}

extern "C" void vo_ctx_destroy(vo_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->alt.side) (void)hipStreamSynchronize(c->alt.side);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
#define X(name) if (c->alt.name.p) (void)hipFree(c->alt.name.p);
    VO_TAIL_BUFS(X)
#undef X
    DevBuf* bufs[] = {&c->fast_thr, &c->fast_redo,
                      &c->tab,      &c->pyr,      &c->blur,    &c->cand,    &c->cand_cnt, &c->kp_x,     &c->kp_y,
                      &c->kp_resp,  &c->sel_keys, &c->sel_resp, &c->sel_cnt, &c->kp_cnt,  &c->flags,   &c->img_status, &c->fast_stats, &c->st_img,   &c->st_kps,   &c->st_desc,
                      &c->st_cnt,   &c->st_q,     &c->st_t,    &c->st_idx,  &c->st_dist,  &c->st_down,  &c->m_top2, &c->k_part,   &c->m_qidx,
                      &c->m_tidx,   &c->m_dist,   &c->m_p1,    &c->m_p2,    &c->m_cnt,    &c->r_state,  &c->r_samples,
                      &c->r_models, &c->r_nmodels, &c->r_models2, &c->r_nmodels2, &c->r_retry, &c->r_roots, &c->r_nroots, &c->r_counts, &c->r_best, &c->r_mask,  &c->r_pts,    &c->r_ptsf, &c->r_res,
                      &c->p_kps,    &c->p_desc,   &c->p_cnt};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    for (auto& e : c->ev_pool)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_init) (void)hipEventDestroy(c->ev_init);
    if (c->ev_rdone) (void)hipEventDestroy(c->ev_rdone);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->alt.ev_init) (void)hipEventDestroy(c->alt.ev_init);
    if (c->alt.ev_rdone) (void)hipEventDestroy(c->alt.ev_rdone);
    if (c->alt.ev_tail) (void)hipEventDestroy(c->alt.ev_tail);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_fast0) (void)hipEventDestroy(c->ev_fast0);
    if (c->ev_blur) (void)hipEventDestroy(c->ev_blur);
    for (int k = 0; k < VO_TICKET_RING; ++k) {
        if (c->ev_in[k]) (void)hipEventDestroy(c->ev_in[k]);
        if (c->ev_out[k]) (void)hipEventDestroy(c->ev_out[k]);
    }
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->alt.side) (void)hipStreamDestroy(c->alt.side);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* vo_last_error(const vo_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" void* vo_ctx_stream(vo_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int vo_sync(vo_ctx* c) {
    if (!c) return VO_ERR_INVALID;
    VO_HIP(c, hipStreamSynchronize(c->stream));
    VO_HIP(c, hipStreamSynchronize(c->side));
    VO_HIP(c, hipStreamSynchronize(c->alt.side));
    VO_HIP(c, hipStreamSynchronize(c->aux));
    c->side_busy = c->alt.side_busy = false;
    c->tail_pending = c->alt.tail_pending = false;
    return take_truncation(c, "vo_sync");
}

extern "C" int vo_set_timing(vo_ctx* c, int groups) {
    if (!c) return VO_ERR_INVALID;
    VO_HIP(c, hipStreamSynchronize(c->stream));
    VO_HIP(c, hipStreamSynchronize(c->side));
    VO_HIP(c, hipStreamSynchronize(c->alt.side));
    VO_HIP(c, hipStreamSynchronize(c->aux));
    c->timing = (unsigned)groups & ((1u << VO_T_COUNT) - 1u);
    for (auto& v : c->ev_pairs) v.clear();
    c->ev_used = 0;
    return VO_OK;
}

extern "C" float vo_kernel_time_ms(vo_ctx* c, int group, int* launches) {
    if (launches) *launches = 0;
    if (!c || group < 0 || group >= VO_T_COUNT) return -1.f;
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipStreamSynchronize(c->side) != hipSuccess ||
        hipStreamSynchronize(c->alt.side) != hipSuccess || hipStreamSynchronize(c->aux) != hipSuccess)
        return -1.f;
    double sum = 0;
    int n = 0;
    for (auto& pr : c->ev_pairs[group]) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            sum += ms;
            ++n;
        }
    }
    if (launches) *launches = n;
    return n ? (float)(sum / n) : -1.f;
}
