// mctsq_dqn_grad: the learner step of DQNModel.update (dqn_per.py:455-470) on an already gathered
// batch.  Host side: carve the caller's workspace, enqueue target forward -> policy forward (keeping
// activations) -> TD loss -> head backward -> encoder backward -> weight-gradient GEMMs -> ordered
// reduction into the flat gradient buffer.  The target forward runs on a helper stream forked from and
// joined to the caller's stream with events; everything else is enqueued on the caller's stream.
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "forward.hpp"
#include "learner.hpp"
#include "per.hpp"
#include "prepare_core.hpp"

using namespace mctsq;

namespace {

struct Carver {
    int64_t pos = 0;                                     // floats
    int64_t take(int64_t n) {
        const int64_t at = pos;
        pos = (pos + n + 63) & ~int64_t(63);            // 256-byte aligned blocks
        return at;
    }
};

struct SlotPlan {                      // float offsets; which fields are used depends on the kind
    int64_t score, stat, pool, x, dsc, p;
    int64_t h, r, z, n, hn;             // gru
    int64_t wp_ih, wp_rz, wp_n, wp_aw, wp_ln;  // gru: per-workgroup partials of gru_bwd_kernel (weight gradients, score Linear, LayerNorm)
    int64_t wt;                         // gru: W_hh as split-bf16 B fragments of the BPTT product (gru_prepare_bwd_kernel)
    int64_t c_h1, c_h2, c_d2, c_d1;     // conv
    int64_t m_x, m_m0, m_d1, m_d0;      // mlp
    // transformer: [l] per layer
    int64_t t_e0, t_hmid[2], t_hout[2], t_qkv[2], t_ctx[2], t_x1[2], t_st1[2], t_x2[2], t_st2[2], t_mean, t_xe;
    int64_t t_w1t[2], t_w2t[2], t_dy2[2], t_dx2[2], t_p2[2], t_dy1[2], t_dx1[2], t_p1[2], t_dqkv[2];
    int64_t t_dhpart, t_wp1[2], t_wp2[2], t_pfin, t_dh0;
    int64_t Tp, T;
    int ntb;
};

struct TrainPlan {
    int64_t Bp;
    int64_t feats_t, q_t, qnext, feats_p, q_p, dq, dfeat, loss_parts;
    int64_t act[4], dact[4];
    int64_t head_prep_bwd;                               // transposed split-bf16 head fragments (-1: generic head)
    SlotPlan slot[3];
    int64_t partials, partial_floats;
    int64_t total;                                       // floats
};

int64_t split_chunk(int64_t K) {                         // K chunk per split: <= 64 splits, >= 64 rows
    int64_t c = (K + 63) / 64;
    if (c < 64) c = 64;
    return (c + 3) & ~int64_t(3);
}
int64_t job_partial_floats(int M, int N, int64_t K) {
    const int64_t kc = split_chunk(K), ns = (K + kc - 1) / kc;
    return ns * ((int64_t)M * N + M);
}

int make_plan(const NetLayout& L, int64_t B, TrainPlan* P) {
    memset(P, 0, sizeof(*P));
    const int64_t Bp = round_up(B > 0 ? B : 1, TILE);
    P->Bp = Bp;
    Carver c;
    P->feats_t = c.take(forward_ws_floats(L, Bp)); P->q_t = c.take(Bp * L.A); P->qnext = c.take(Bp);
    P->feats_p = c.take(forward_ws_floats(L, Bp)); P->q_p = c.take(Bp * L.A);
    P->dq = c.take(Bp * DQ_LD); P->dfeat = c.take(Bp * FEAT);
    P->loss_parts = c.take(2 * (Bp / TILE));                 // one fp64 loss sum per tile (fused head backward)
    for (int i = 0; i + 1 < L.n_fc; ++i) {
        P->act[i] = c.take(Bp * L.fc_out[i]);
        P->dact[i] = c.take(Bp * L.fc_out[i]);
    }
    const bool head_bf16 = L.n_fc == 3 && L.fc_in[0] == FEAT && L.fc_out[0] == 128 && L.fc_in[1] == 128 &&
                           L.fc_out[1] == 128 && L.fc_in[2] == 128 && L.A <= 16;
    P->head_prep_bwd = head_bf16 ? c.take(HEAD_PREP_BWD_BYTES / (int64_t)sizeof(float)) : -1;
    int64_t pf = 0;
    for (int s = 0; s < 3; ++s) {
        const EncOff& e = L.enc[s];
        const int64_t S = e.S, K = S * Bp;
        SlotPlan& g = P->slot[s];
        if (e.kind == MCTSQ_ENC_GRU) {
            g.h = c.take((S + 1) * Bp * E);
            g.r = c.take(K * E); g.z = c.take(K * E); g.n = c.take(K * E); g.hn = c.take(K * E);
            g.score = c.take(K); g.stat = c.take(Bp * 4); g.pool = c.take(Bp * E); g.x = c.take(K * 8);
            g.dsc = c.take(K); g.p = c.take(Bp * E);
            MCTSQ_CHECK(K < ((int64_t)1 << 31), "batch too large");
            const int64_t nwg = Bp / TILE * 2;           // per workgroup of the GRU backward (8- or 16-sample tiles)
            g.wp_ih = c.take(nwg * (G3 * e.F + G3));
            g.wp_rz = c.take(nwg * (128 * E + 128));
            g.wp_n = c.take(nwg * (E * E + E));
            g.wp_aw = c.take(nwg * 65);
            g.wp_ln = c.take(Bp / TILE * 2 * E);
            g.wt = c.take(GRU_PREP_BYTES / (int64_t)sizeof(float));
            pf += 2 * job_partial_floats(E, 0, Bp);
        } else if (e.kind == MCTSQ_ENC_CONV) {
            g.c_h1 = c.take((S + 2) * Bp * E); g.c_h2 = c.take(K * E);
            g.score = c.take(K); g.stat = c.take(Bp * 4); g.pool = c.take(Bp * E); g.x = c.take((S + 2) * Bp * 8);
            g.c_d2 = c.take(K * E); g.c_d1 = c.take(K * E); g.dsc = c.take(K); g.p = c.take(Bp * E);
            pf += 3 * job_partial_floats(E, E, K) + 3 * job_partial_floats(E, e.F, K) + job_partial_floats(1, E, K) +
                  2 * job_partial_floats(E, 0, Bp);
        } else if (e.kind == MCTSQ_ENC_MLP) {
            g.m_x = c.take(Bp * 16); g.m_m0 = c.take(Bp * E); g.m_d1 = c.take(Bp * E); g.m_d0 = c.take(Bp * E);
            pf += job_partial_floats(E, 3 * e.F, Bp) + job_partial_floats(E, E, Bp);
        } else if (e.kind == MCTSQ_ENC_TRANSFORMER) {
            const int64_t Tp = tf_padded_tokens(Bp, e.S), T = Bp * S;
            g.Tp = Tp; g.T = T; g.ntb = tf_bwd_token_blocks(Tp);
            g.t_e0 = c.take(Tp * E); g.t_mean = c.take(Bp * E); g.t_xe = c.take(Tp * 8);
            g.t_pfin = c.take(Bp * E); g.t_dh0 = c.take(Tp * E); g.t_dhpart = c.take(16 * Tp * E);
            for (int l = 0; l < 2; ++l) {
                g.t_hmid[l] = c.take(Tp * E); g.t_hout[l] = c.take(Tp * E); g.t_qkv[l] = c.take(Tp * G3);
                g.t_ctx[l] = c.take(Tp * E); g.t_x1[l] = c.take(Tp * E); g.t_st1[l] = c.take(Tp * 2);
                g.t_x2[l] = c.take(Tp * E); g.t_st2[l] = c.take(Tp * 2);
                g.t_w1t[l] = c.take(2048 * E); g.t_w2t[l] = c.take(2048 * E);
                g.t_dy2[l] = c.take(Tp * E); g.t_dx2[l] = c.take(Tp * E); g.t_p2[l] = c.take(Tp * E);
                g.t_dy1[l] = c.take(Tp * E); g.t_dx1[l] = c.take(Tp * E); g.t_p1[l] = c.take(Tp * E);
                g.t_dqkv[l] = c.take(Tp * G3);
                g.t_wp1[l] = c.take((int64_t)g.ntb * (2048 * E + 2048));
                g.t_wp2[l] = c.take((int64_t)g.ntb * (E * 2048 + E));
                pf += job_partial_floats(G3, E, T) + job_partial_floats(E, E, T) + 5 * job_partial_floats(E, 0, T);
            }
            pf += job_partial_floats(E, e.F, T) + job_partial_floats((int)(S * E), 0, Bp) + 2 * job_partial_floats(E, 0, Bp);
        } else {
            MCTSQ_CHECK(false, "training path: unknown encoder kind %d", e.kind);
        }
    }
    for (int i = 0; i < L.n_fc; ++i) pf += job_partial_floats(L.fc_out[i], L.fc_in[i], Bp);
    P->partial_floats = pf;
    P->partials = c.take(pf);
    P->total = c.pos;
    return 0;
}

struct JobBuilder {
    WJobList lists[4];
    int nlists = 0;
    int64_t part_pos = 0;
    const float* base;                                   // workspace base every operand lives in
    explicit JobBuilder(const float* ws) : base(ws) { memset(lists, 0, sizeof(lists)); }
    WJob* next() {
        if (nlists == 0 || lists[nlists - 1].njobs == MAX_WJOBS) {
            if (nlists >= 4) { set_error("too many weight-gradient jobs"); return nullptr; }
            ++nlists;
        }
        WJobList& jl = lists[nlists - 1];
        return &jl.job[jl.njobs++];
    }
    int add(const float* A, int64_t lda, int a_col0, const float* Bm, int64_t ldb, int b_col0, int M, int N,
            int64_t K, int64_t out_w, int64_t out_ldm, int64_t out_sn, int64_t out_b) {
        WJob* jp = next();
        if (!jp) return -1;
        WJob& j = *jp;
        WJobList& jl = lists[nlists - 1];
        MCTSQ_CHECK(K % 4 == 0 && K < ((int64_t)1 << 31), "weight-gradient contraction length must be a multiple of 4");
        j.a_off = (A - base) + a_col0; j.lda = (int32_t)lda;
        j.b_off = Bm ? (Bm - base) + b_col0 : -1; j.ldb = (int32_t)ldb;
        j.M = M; j.N = Bm ? N : 0; j.K = (int32_t)K;
        j.kchunk = (int32_t)split_chunk(K);
        j.nsplit = (int)((K + j.kchunk - 1) / j.kchunk);
        j.part_off = part_pos;
        part_pos += (int64_t)j.nsplit * ((int64_t)j.M * j.N + j.M);
        j.out_w = Bm ? (int32_t)out_w : -1; j.out_ldm = (int32_t)out_ldm; j.out_sn = (int32_t)out_sn;
        j.out_b = (int32_t)out_b;
        j.mblocks = (M + 63) / 64;
        j.nblocks = j.N > 0 ? (j.N + 63) / 64 : 1;
        j.blk0 = jl.nblocks_total;
        jl.nblocks_total += j.nsplit * j.mblocks * j.nblocks;
        return 0;
    }
    // partial sums [nsplit][M*N + M] already written by another kernel at float offset `part_off`
    // relative to the partial base; only the ordered reduction runs for this job
    int add_external(int64_t part_off, int M, int N, int nsplit, int64_t out_w, int64_t out_ldm, int64_t out_sn,
                     int64_t out_b) {
        WJob* jp = next();
        if (!jp) return -1;
        WJob& j = *jp;
        j.a_off = -1; j.b_off = -1; j.M = M; j.N = N; j.K = 0; j.kchunk = 4; j.nsplit = nsplit;
        j.part_off = part_off; j.out_w = (int32_t)out_w; j.out_ldm = (int32_t)out_ldm; j.out_sn = (int32_t)out_sn;
        j.out_b = (int32_t)out_b;
        j.mblocks = 0; j.nblocks = 1; j.blk0 = lists[nlists - 1].nblocks_total;
        return 0;
    }
};

}  // namespace

namespace {
// The step is a small DAG, not a chain: the target forward (on s') runs beside the policy forward (on s), and after
// the head backward the weight-gradient products that only need the head's deltas, and the backward of the SHORT
// GRU windows with their weight gradients, run beside the backward of the longest window.  The second lane is a
// helper stream forked from / joined to the caller's stream with events (no host synchronisation; the pattern is
// also legal under stream capture).  One helper stream and event set per (device, caller stream), created on first
// use and kept for the life of the process, so calls on different streams never share events; calls on the SAME
// stream are ordered by the caller (one host thread enqueues on a stream at a time).
struct SideStream {
    hipStream_t st = nullptr, st2 = nullptr;               // helper lanes: encoder backward of the short windows; head products
    hipEvent_t fork = nullptr, join = nullptr, head = nullptr, join2 = nullptr, join3 = nullptr;
};
std::mutex g_side_mu;
std::map<std::pair<int, void*>, SideStream> g_side;
int side_stream(hipStream_t caller, SideStream* out) {
    int dev = 0;
    MCTSQ_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_side_mu);
    SideStream& s = g_side[std::make_pair(dev, (void*)caller)];
    if (!s.st) {
        MCTSQ_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
        MCTSQ_HIP(hipStreamCreateWithFlags(&s.st2, hipStreamNonBlocking));
        MCTSQ_HIP(hipEventCreateWithFlags(&s.join3, hipEventDisableTiming));
        MCTSQ_HIP(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        MCTSQ_HIP(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
        MCTSQ_HIP(hipEventCreateWithFlags(&s.head, hipEventDisableTiming));
        MCTSQ_HIP(hipEventCreateWithFlags(&s.join2, hipEventDisableTiming));
    }
    *out = s;
    return 0;
}

bool head_is_bf16(const NetLayout& L) {
    return L.n_fc == 3 && L.fc_in[0] == FEAT && L.fc_out[0] == 128 && L.fc_in[1] == 128 && L.fc_out[1] == 128 &&
           L.fc_in[2] == 128 && L.A <= 16;
}
int n_gru_slots(const NetLayout& L) {
    int n = 0;
    for (int s = 0; s < 3; ++s) n += L.enc[s].kind == MCTSQ_ENC_GRU;
    return n;
}
// the learner's backward fragments: [head (192-128-128-A only)] [W_hh of every GRU slot, slot order]
int64_t prep_bwd_bytes(const NetLayout& L) {
    return (head_is_bf16(L) ? HEAD_PREP_BWD_BYTES : 0) + n_gru_slots(L) * GRU_PREP_BYTES;
}
struct DqnPrep { const void* policy; const void* target; const void* bwd; };
// priority write-back of the step's |td| (dqn_per.py:474), enqueued on the helper lane as soon as the TD errors
// exist (behind the head backward) instead of at the end of the step: it depends on nothing later
struct PerUpdateHook {
    unsigned long long* tree; double* prio; int32_t* stamp; int64_t cap, head, size; const int64_t* idx;
    double alpha; int64_t mass_cap;
    bool grad_zeroed;          // the caller's sampling launch zeroed d_grad (passenger): no fill, no fork for it
};

int prepare_bwd(const NetLayout& L, const float* params, void* bwd, hipStream_t st) {
    char* p = (char*)bwd;
    if (head_is_bf16(L)) {
        if (launch_head_prepare_bwd(params + L.fc_w[0], params + L.fc_w[1], params + L.fc_w[2], L.A, p, st)) return -1;
        p += HEAD_PREP_BWD_BYTES;
    }
    const float* whh[3] = {nullptr, nullptr, nullptr};
    void* wt[3] = {nullptr, nullptr, nullptr};
    int n = 0;
    for (int s = 0; s < 3; ++s)
        if (L.enc[s].kind == MCTSQ_ENC_GRU) { whh[n] = params + L.enc[s].gru.whh; wt[n] = p + n * GRU_PREP_BYTES; ++n; }
    return launch_gru_prepare_bwd(whh, n, wt, st);
}
}  // namespace

namespace {
bool one_launch_prepare(const NetLayout& L) { return n_gru_slots(L) == 3 && head_is_bf16(L); }
void fill_dqn_prep(const NetLayout& L, const float* d_params, const float* d_target_params, void* d_prep_policy,
                   void* d_prep_target, void* d_prep_bwd, DqnPrepArgs* out) {
    DqnPrepArgs& a = *out;
    const float* par[2] = {d_params, d_target_params};
    void* fwd[2] = {d_prep_policy, d_prep_target};
    for (int n = 0; n < 2; ++n) {
        for (int s = 0; s < 3; ++s) a.whh[n][s] = par[n] + L.enc[s].gru.whh;
        for (int l = 0; l < 3; ++l) a.hw[n][l] = par[n] + L.fc_w[l];
        a.gru_fwd[n] = reinterpret_cast<u32x4*>(fwd[n]);
        a.head_fwd[n] = reinterpret_cast<u32x4*>((char*)fwd[n] + 3 * GRU_PREP_BYTES);
    }
    a.head_bwd = reinterpret_cast<u32x4*>(d_prep_bwd);
    for (int s = 0; s < 3; ++s)
        a.gru_bwd[s] = reinterpret_cast<u32x4*>((char*)d_prep_bwd + HEAD_PREP_BWD_BYTES + s * GRU_PREP_BYTES);
    a.A = L.A;
}
}  // namespace

extern "C" int64_t mctsq_dqn_prepared_bwd_bytes(const mctsq_net_desc* net) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    return prep_bwd_bytes(L);
}

extern "C" int mctsq_dqn_prepare(const mctsq_net_desc* net, const float* d_params, const float* d_target_params,
                                 void* d_prep_policy, void* d_prep_target, void* d_prep_bwd, void* stream) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    MCTSQ_CHECK(d_params && d_target_params, "NULL device pointer");
    MCTSQ_CHECK((d_prep_policy && d_prep_target) || prepared_bytes(L) == 0, "NULL prepared-weights pointer");
    MCTSQ_CHECK(d_prep_bwd || prep_bwd_bytes(L) == 0, "NULL backward prepared-weights pointer");
    MCTSQ_CHECK((((uintptr_t)d_prep_policy | (uintptr_t)d_prep_target | (uintptr_t)d_prep_bwd) & 15) == 0,
                "prepared buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (one_launch_prepare(L)) {                             // shipped default: one launch for all of it
        DqnPrepArgs a;
        fill_dqn_prep(L, d_params, d_target_params, d_prep_policy, d_prep_target, d_prep_bwd, &a);
        return launch_dqn_prepare(a, st);
    }
    if (prepared_bytes(L) > 0) {
        if (prepare_impl(L, d_params, d_prep_policy, st)) return -1;
        if (prepare_impl(L, d_target_params, d_prep_target, st)) return -1;
    }
    if (prep_bwd_bytes(L) > 0 && prepare_bwd(L, d_params, d_prep_bwd, st)) return -1;
    return 0;
}

extern "C" int64_t mctsq_dqn_grad_workspace_bytes(const mctsq_net_desc* net, int64_t B) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    if (B < 0) { set_error("negative batch"); return -1; }
    TrainPlan P;
    if (make_plan(L, B, &P)) return -1;
    return P.total * (int64_t)sizeof(float);
}

static int dqn_grad_enqueue(const NetLayout& L, const TrainPlan& P, const SideStream* side, const DqnPrep* pp,
                            const float* d_params, const float* d_target_params, const float* d_sprice,
                            const float* d_sprocess, const float* d_sgas, const float* d_nprice,
                            const float* d_nprocess, const float* d_ngas, const int32_t* d_act, const float* d_rew,
                            const float* d_done, const float* d_isw, float gamma, float loss_scale, int64_t B,
                            float* d_grad, float* d_loss, float* d_abs_td, float* d_qsa, float* d_target,
                            double* d_sumsq_parts, const PerUpdateHook* upd, float* ws, hipStream_t st);

static int dqn_grad_entry(const mctsq_net_desc* net, const DqnPrep* pp, const float* d_params,
                          const float* d_target_params, const float* d_sprice, const float* d_sprocess,
                          const float* d_sgas, const float* d_nprice, const float* d_nprocess, const float* d_ngas,
                          const int32_t* d_act, const float* d_rew, const float* d_done, const float* d_isw,
                          float gamma, float loss_scale, int64_t B, float* d_grad, float* d_loss, float* d_abs_td,
                          float* d_qsa, float* d_target, double* d_sumsq_parts, void* d_ws, int64_t ws_bytes,
                          void* stream, const PerUpdateHook* upd = nullptr) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    MCTSQ_CHECK(B >= 1, "batch must be >= 1");
    MCTSQ_CHECK(d_params && d_target_params && d_sprice && d_sprocess && d_sgas && d_nprice && d_nprocess &&
                    d_ngas && d_act && d_rew && d_done && d_isw && d_grad && d_loss && d_abs_td,
                "NULL device pointer");
    if (pp) {
        MCTSQ_CHECK((pp->policy && pp->target) || prepared_bytes(L) == 0, "NULL prepared-weights pointer");
        MCTSQ_CHECK(pp->bwd || prep_bwd_bytes(L) == 0, "NULL backward prepared-weights pointer");
        MCTSQ_CHECK((((uintptr_t)pp->policy | (uintptr_t)pp->target | (uintptr_t)pp->bwd) & 15) == 0,
                    "prepared buffers must be 16-byte aligned");
    }
    TrainPlan P;
    if (make_plan(L, B, &P)) return -1;
    MCTSQ_CHECK(d_ws && ws_bytes >= P.total * (int64_t)sizeof(float), "workspace too small: %lld < %lld bytes",
                (long long)ws_bytes, (long long)(P.total * (int64_t)sizeof(float)));
    MCTSQ_CHECK(((uintptr_t)d_ws & 15) == 0, "workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SideStream side;
    if (side_stream(st, &side)) return -1;
    // Without prepared fragments the helper lane starts with the backward splits (and the gradient fill): forked here.
    // With prepared fragments its first job waits for the head backward anyway (event `head`), and the fill - unless a
    // passenger of the caller's sampling launch did it - goes to the caller's stream ahead of the forward: no fork (two
    // runtime calls of ~6-8 us each on a host-bound step)
    const bool need_fork = !pp;
    if (need_fork) {
        MCTSQ_HIP(hipEventRecord(side.fork, st));
        MCTSQ_HIP(hipStreamWaitEvent(side.st, side.fork, 0));
    }
    const int rc = dqn_grad_enqueue(L, P, &side, pp, d_params, d_target_params, d_sprice, d_sprocess, d_sgas, d_nprice,
                                    d_nprocess, d_ngas, d_act, d_rew, d_done, d_isw, gamma, loss_scale, B, d_grad,
                                    d_loss, d_abs_td, d_qsa, d_target, d_sumsq_parts, upd, (float*)d_ws, st);
    if (rc < 0) {                      // whatever was enqueued on the helper streams is joined back before returning
        (void)hipEventRecord(side.join2, side.st);
        (void)hipStreamWaitEvent(st, side.join2, 0);
        (void)hipEventRecord(side.join3, side.st2);
        (void)hipStreamWaitEvent(st, side.join3, 0);
    }
    // success: both helper lanes (the priority write-back included) were joined before the reduction
    return rc;
}

extern "C" int mctsq_dqn_grad(const mctsq_net_desc* net, const float* d_params, const float* d_target_params,
                              const float* d_sprice, const float* d_sprocess, const float* d_sgas,
                              const float* d_nprice, const float* d_nprocess, const float* d_ngas,
                              const int32_t* d_act, const float* d_rew, const float* d_done, const float* d_isw,
                              float gamma, float loss_scale, int64_t B, float* d_grad, float* d_loss,
                              float* d_abs_td, float* d_qsa, float* d_target, void* d_ws, int64_t ws_bytes,
                              void* stream) {
    return dqn_grad_entry(net, nullptr, d_params, d_target_params, d_sprice, d_sprocess, d_sgas, d_nprice, d_nprocess,
                          d_ngas, d_act, d_rew, d_done, d_isw, gamma, loss_scale, B, d_grad, d_loss, d_abs_td, d_qsa,
                          d_target, nullptr, d_ws, ws_bytes, stream);
}

extern "C" int64_t mctsq_dqn_grad_sumsq_parts(const mctsq_net_desc* net) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    // one part per 256 reduced entries, rounded up per reduction launch (<= 4 launches): an upper bound
    return (L.total + 255) / 256 + 8 * 160;
}

extern "C" int mctsq_dqn_grad_prepared(const mctsq_net_desc* net, const float* d_params, const float* d_target_params,
                                       const void* d_prep_policy, const void* d_prep_target, const void* d_prep_bwd,
                                       const float* d_sprice, const float* d_sprocess, const float* d_sgas,
                                       const float* d_nprice, const float* d_nprocess, const float* d_ngas,
                                       const int32_t* d_act, const float* d_rew, const float* d_done,
                                       const float* d_isw, float gamma, float loss_scale, int64_t B, float* d_grad,
                                       float* d_loss, float* d_abs_td, float* d_qsa, float* d_target,
                                       double* d_sumsq_parts, int64_t* n_sumsq_parts, void* d_ws, int64_t ws_bytes,
                                       void* stream) {
    DqnPrep pp{d_prep_policy, d_prep_target, d_prep_bwd};
    const int rc = dqn_grad_entry(net, &pp, d_params, d_target_params, d_sprice, d_sprocess, d_sgas, d_nprice, d_nprocess,
                                  d_ngas, d_act, d_rew, d_done, d_isw, gamma, loss_scale, B, d_grad, d_loss, d_abs_td,
                                  d_qsa, d_target, d_sumsq_parts, d_ws, ws_bytes, stream);
    if (rc < 0) return rc;
    if (n_sumsq_parts) *n_sumsq_parts = rc;
    return 0;
}

// returns the number of gradient-norm parts written (>= 0) or < 0 on error
static int dqn_grad_enqueue(const NetLayout& L, const TrainPlan& P, const SideStream* side, const DqnPrep* pp,
                            const float* d_params, const float* d_target_params, const float* d_sprice,
                            const float* d_sprocess, const float* d_sgas, const float* d_nprice,
                            const float* d_nprocess, const float* d_ngas, const int32_t* d_act, const float* d_rew,
                            const float* d_done, const float* d_isw, float gamma, float loss_scale, int64_t B,
                            float* d_grad, float* d_loss, float* d_abs_td, float* d_qsa, float* d_target,
                            double* d_sumsq_parts, const PerUpdateHook* upd, float* ws, hipStream_t st) {
    const int64_t Bp = P.Bp;
    const bool head_bf16 = P.head_prep_bwd >= 0;
    // backward fragments: the caller's (prepared once per parameter change) or split here into the workspace
    const char* bwd = pp ? (const char*)pp->bwd : nullptr;
    const void* head_bwd_prep = nullptr;
    const void* gru_bwd_prep[3] = {nullptr, nullptr, nullptr};
    if (pp) {
        if (head_bf16) { head_bwd_prep = bwd; bwd += HEAD_PREP_BWD_BYTES; }
        for (int s = 0, n = 0; s < 3; ++s)
            if (L.enc[s].kind == MCTSQ_ENC_GRU) gru_bwd_prep[s] = bwd + (n++) * GRU_PREP_BYTES;
    } else {
        if (head_bf16) head_bwd_prep = ws + P.head_prep_bwd;
        for (int s = 0; s < 3; ++s)
            if (L.enc[s].kind == MCTSQ_ENC_GRU) gru_bwd_prep[s] = ws + P.slot[s].wt;
    }

    bool side_used = false;                                  // anything on the first helper lane: joined before the reduction
    // 1. target network on s' -> max_a' Q (dqn_per.py:460-463), on the helper stream (forked by the caller)
    {
        // the flat gradient's alignment pads must read zero; nothing touches it before the ordered reduction (step 7,
        // behind the join), so the fill rides on the helper stream instead of the critical path
        if (!(upd && upd->grad_zeroed)) {
            if (pp) {
                MCTSQ_HIP(hipMemsetAsync(d_grad, 0, (size_t)L.total * sizeof(float), st));
            } else {
                MCTSQ_HIP(hipMemsetAsync(d_grad, 0, (size_t)L.total * sizeof(float), side->st));
                side_used = true;
            }
        }
        if (!pp) {                                           // ... and so do the splits the backward pass multiplies by
            side_used = true;
            if (head_bf16 &&
                launch_head_prepare_bwd(d_params + L.fc_w[0], d_params + L.fc_w[1], d_params + L.fc_w[2], L.A,
                                        ws + P.head_prep_bwd, side->st))
                return -1;
            const float* whh[3] = {nullptr, nullptr, nullptr};
            void* wt[3] = {nullptr, nullptr, nullptr};
            int n = 0;
            for (int s = 0; s < 3; ++s)
                if (L.enc[s].kind == MCTSQ_ENC_GRU) { whh[n] = d_params + L.enc[s].gru.whh; wt[n] = ws + P.slot[s].wt; ++n; }
            if (launch_gru_prepare_bwd(whh, n, wt, side->st)) return -1;
        }
    }
    const float* xn[3] = {d_nprice, d_nprocess, d_ngas};
    const HeadOut out_t{ws + P.q_t, ws + P.qnext, nullptr, nullptr};
    auto target_forward_on_helper = [&]() -> int {
        if (forward_impl(L, d_target_params, pp ? pp->target : nullptr, xn, B, Bp, ws + P.feats_t, out_t, nullptr, nullptr,
                         side->st, nullptr))
            return -1;
        MCTSQ_HIP(hipEventRecord(side->join, side->st));
        return 0;
    };
    // 2. policy network on s, activations kept (dqn_per.py:455-457)
    SlotSave sv[3];
    memset(sv, 0, sizeof(sv));
    for (int s = 0; s < 3; ++s) {
        const SlotPlan& g = P.slot[s];
        SlotSave& v = sv[s];
        v.score = ws + g.score; v.stat = ws + g.stat; v.pool = ws + g.pool; v.x = ws + g.x; v.dsc = ws + g.dsc;
        v.p = ws + g.p;
        v.h = ws + g.h; v.r = ws + g.r; v.z = ws + g.z; v.n = ws + g.n; v.hn = ws + g.hn;
        v.wih = ws + g.wp_ih; v.wrz = ws + g.wp_rz; v.wn = ws + g.wp_n;
        v.aw = ws + g.wp_aw; v.ln = ws + g.wp_ln;
        v.c_h1 = ws + g.c_h1; v.c_h2 = ws + g.c_h2; v.c_d2 = ws + g.c_d2; v.c_d1 = ws + g.c_d1;
        v.m_x = ws + g.m_x; v.m_m0 = ws + g.m_m0; v.m_d1 = ws + g.m_d1; v.m_d0 = ws + g.m_d0;
        if (L.enc[s].kind == MCTSQ_ENC_TRANSFORMER) {
            for (int l = 0; l < 2; ++l) {
                TfLayerBuf& b = v.t_buf[l];
                b.hin = l == 0 ? ws + g.t_e0 : ws + g.t_hout[0];
                b.hmid = ws + g.t_hmid[l]; b.hout = ws + g.t_hout[l]; b.qkv = ws + g.t_qkv[l]; b.ctx = ws + g.t_ctx[l];
                b.x1 = ws + g.t_x1[l]; b.st1 = ws + g.t_st1[l]; b.x2 = ws + g.t_x2[l]; b.st2 = ws + g.t_st2[l];
                // rows [T, Tp) of the FFN backward's token-major inputs are read but never produced
                if (g.Tp > g.T) {
                    MCTSQ_HIP(hipMemsetAsync(b.hmid + g.T * E, 0, (size_t)(g.Tp - g.T) * E * sizeof(float), st));
                    MCTSQ_HIP(hipMemsetAsync(ws + g.t_dx2[l] + g.T * E, 0, (size_t)(g.Tp - g.T) * E * sizeof(float), st));
                }
            }
            v.t_mean = ws + g.t_mean; v.t_xe = ws + g.t_xe;
        }
    }
    float* sv_act[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i + 1 < L.n_fc; ++i) sv_act[i] = ws + P.act[i];
    const float* xs[3] = {d_sprice, d_sprocess, d_sgas};
    {
        // both forwards as ONE launch on the caller's stream when the network has the fused kernel and the weights are
        // prepared; otherwise the target forward runs on the helper stream next to the policy forward
        const HeadOut out{ws + P.q_p, nullptr, nullptr, nullptr};
        const int pr = forward_pair_impl(L, d_params, pp ? pp->policy : nullptr, xs, ws + P.feats_p, out, sv, sv_act,
                                         d_target_params, pp ? pp->target : nullptr, xn, ws + P.feats_t, out_t, B, Bp, st);
        if (pr < 0) return -1;
        if (pr == 1) {
            if (pp) {                                         // the caller skipped the fork (dqn_grad_entry): do it now
                MCTSQ_HIP(hipEventRecord(side->fork, st));
                MCTSQ_HIP(hipStreamWaitEvent(side->st, side->fork, 0));
            }
            side_used = true;
            if (target_forward_on_helper()) return -1;
            if (forward_impl(L, d_params, pp ? pp->policy : nullptr, xs, B, Bp, ws + P.feats_p, out, sv, sv_act, st, nullptr))
                return -1;
            MCTSQ_HIP(hipStreamWaitEvent(st, side->join, 0));
        }
    }
    // 3. TD target, IS-weighted MSE, dLoss/dQ, new priorities (dqn_per.py:464-467, 474): evaluated in the prologue
    //    of the split-bf16 head backward (one launch fewer), or by its own kernel for other head shapes
    TdArgs td;
    memset(&td, 0, sizeof(td));
    td.q = ws + P.q_p; td.qnext = ws + P.qnext; td.act = d_act; td.rew = d_rew; td.done = d_done; td.isw = d_isw;
    td.gamma = gamma; td.grad_scale = loss_scale * 2.0f / (float)B; td.B = B; td.Bp = Bp; td.A = L.A;
    td.loss_norm = (double)loss_scale / (double)B;
    td.dq = ws + P.dq; td.loss = d_loss; td.abs_td = d_abs_td; td.qsa = d_qsa; td.target = d_target;
    double* loss_parts = reinterpret_cast<double*>(ws + P.loss_parts);
    if (!head_bf16 && launch_td_loss(td, st)) return -1;
    // 4. head backward
    {
        HeadBwdArgs h;
        memset(&h, 0, sizeof(h));
        h.n_fc = L.n_fc; h.act = L.act; h.Bp = Bp; h.dq = ws + P.dq; h.dfeat = ws + P.dfeat;
        for (int i = 0; i < L.n_fc; ++i) { h.w[i] = d_params + L.fc_w[i]; h.in[i] = L.fc_in[i]; h.out[i] = L.fc_out[i]; }
        for (int i = 0; i + 1 < L.n_fc; ++i) { h.sv_act[i] = ws + P.act[i]; h.d_act[i] = ws + P.dact[i]; }
        h.wprep_bwd = head_bwd_prep;
        h.fuse_td = head_bf16; h.td = td; h.loss_parts = loss_parts;
        if (launch_head_bwd(h, st)) return -1;
    }
    MCTSQ_HIP(hipEventRecord(side->head, st));

    // 5./6. encoder backward (deltas) and weight gradients (every tensor is one or a few K-contractions), on two
    // lanes: `jbA` holds the products that need only the head's deltas, `jbB` those behind the encoder backward
    JobBuilder jbA(ws), jbB(ws);
    int gru_half[2] = {0, 0};                                 // tile mode of the GRU backward per lane (caller's, helper)
    bool merged = false;                                      // both groups of GRU windows in ONE launch on the caller's stream
    // the longest GRU window stays on the caller's stream, the others go to the helper stream
    int long_slot = -1;
    for (int s = 0; s < 3; ++s)
        if (L.enc[s].kind == MCTSQ_ENC_GRU && (long_slot < 0 || L.enc[s].S > L.enc[long_slot].S)) long_slot = s;
    {
        ConvFwdArgs ca;
        fill_conv_args(ca, L, d_params, nullptr, xs, B, Bp, ws + P.feats_p, sv);
        ConvBwdArgs cb;
        memset(&cb, 0, sizeof(cb));
        cb.nslots = ca.nslots; cb.Bp = Bp; cb.dfeat = ws + P.dfeat;
        for (int i = 0; i < ca.nslots; ++i) cb.slot[i] = ca.slot[i];
        if (launch_conv_bwd(cb, st)) return -1;
        GruFwdArgs ga;
        fill_gru_args(ga, L, d_params, nullptr, xs, B, Bp, ws + P.feats_p, sv);
        GruBwdArgs gbMain, gbSide;
        memset(&gbMain, 0, sizeof(gbMain));
        memset(&gbSide, 0, sizeof(gbSide));
        gbMain.Bp = gbSide.Bp = Bp; gbMain.dfeat = gbSide.dfeat = ws + P.dfeat;
        for (int s = 0, i = 0; s < 3; ++s) {
            if (L.enc[s].kind != MCTSQ_ENC_GRU) continue;
            GruBwdArgs& gb = s == long_slot ? gbMain : gbSide;
            gb.slot[gb.nslots] = ga.slot[i++];
            gb.slot[gb.nslots].wprep_t = gru_bwd_prep[s];
            ++gb.nslots;
        }
        // 8-sample tiles (half the per-step vector work per workgroup, see gru_bwd_kernel) while the launch still fits the
        // chip in one wave of workgroups: for both lanes, else for the long window on the caller's stream only
        {
            const int64_t nt = Bp / TILE;
            if (2 * nt * (gbMain.nslots + gbSide.nslots) <= 256) gbMain.half = gbSide.half = 1;
            else if (nt * (2 * gbMain.nslots + gbSide.nslots) <= 256) gbMain.half = 1;
            gru_half[0] = gbMain.half; gru_half[1] = gbSide.half;
            // The backward workgroups take a SIMD's whole register file each (weight-gradient accumulators), so nothing
            // of the helper lane (the head's weight-gradient products, the priority write-back) shares a CU with them.
            // Where the launch would cover the chip, the short windows of a tile are walked by ONE workgroup (still
            // shorter than the long window's chain) and a quarter of the CUs stays free for that lane.
            if (gbSide.nslots > 1 && gru_bwd_blocks(gbMain) + gru_bwd_blocks(gbSide) > 192) gbSide.serial = 1;
        }
        // One wave of workgroups for both groups together: ONE launch on the caller's stream (and one for their weight
        // gradients below): the groups run side by side either way, and the first helper lane with its launches, its
        // three event operations and its hand-off before the reduction is not needed.  Otherwise: two lanes.
        merged = gbMain.nslots > 0 && gbSide.nslots > 0 && gru_bwd_blocks(gbMain) + gru_bwd_blocks(gbSide) <= 256;
        if (merged) {
            if (launch_gru_bwd_pair(gbMain, gbSide, st)) return -1;
        } else {
            if (gbSide.nslots > 0) {
                MCTSQ_HIP(hipStreamWaitEvent(side->st, side->head, 0));
                if (launch_gru_bwd(gbSide, side->st)) return -1;
                side_used = true;
            }
            if (launch_gru_bwd(gbMain, st)) return -1;
        }
        MlpArgs ma;
        if (fill_mlp_args(ma, L, d_params, xs, B, Bp, ws + P.feats_p, sv)) {
            ma.dfeat = ws + P.dfeat;
            if (launch_mlp_bwd(ma, st)) return -1;
        }
        for (int s = 0; s < 2; ++s) {
            if (L.enc[s].kind != MCTSQ_ENC_TRANSFORMER) continue;
            const SlotPlan& g = P.slot[s];
            TfBwdArgs tb;
            memset(&tb, 0, sizeof(tb));
            fill_tf_args(tb.f, L.enc[s], d_params, xs[s], s, B, Bp, ws + P.feats_p, nullptr, &sv[s]);
            tb.dfeat = ws + P.dfeat; tb.nsb = 64 / L.enc[s].S; tb.ntb = g.ntb;
            for (int l = 0; l < 2; ++l) {
                tb.w1t[l] = ws + g.t_w1t[l]; tb.w2t[l] = ws + g.t_w2t[l];
                tb.dy2[l] = ws + g.t_dy2[l]; tb.dx2[l] = ws + g.t_dx2[l]; tb.p2[l] = ws + g.t_p2[l];
                tb.dy1[l] = ws + g.t_dy1[l]; tb.dx1[l] = ws + g.t_dx1[l]; tb.p1[l] = ws + g.t_p1[l];
                tb.dqkv[l] = ws + g.t_dqkv[l]; tb.wpart1[l] = ws + g.t_wp1[l]; tb.wpart2[l] = ws + g.t_wp2[l];
            }
            tb.dhpart = ws + g.t_dhpart; tb.pfin = ws + g.t_pfin; tb.dh0 = ws + g.t_dh0;
            if (launch_tf_bwd(tb, st)) return -1;
        }
    }
    int64_t part_pos = 0;                                    // the two job lists share one partial buffer
    auto addA = [&](const float* A, int64_t lda, int a_col0, const float* Bm, int64_t ldb, int b_col0, int M, int N,
                    int64_t K, int64_t out_w, int64_t out_ldm, int64_t out_sn, int64_t out_b) {
        jbA.part_pos = part_pos;
        const int rc = jbA.add(A, lda, a_col0, Bm, ldb, b_col0, M, N, K, out_w, out_ldm, out_sn, out_b);
        part_pos = jbA.part_pos;
        return rc;
    };
    auto addB = [&](const float* A, int64_t lda, int a_col0, const float* Bm, int64_t ldb, int b_col0, int M, int N,
                    int64_t K, int64_t out_w, int64_t out_ldm, int64_t out_sn, int64_t out_b) {
        jbB.part_pos = part_pos;
        const int rc = jbB.add(A, lda, a_col0, Bm, ldb, b_col0, M, N, K, out_w, out_ldm, out_sn, out_b);
        part_pos = jbB.part_pos;
        return rc;
    };
    for (int s = 0; s < 3; ++s) {
        const EncOff& e = L.enc[s];
        const SlotPlan& g = P.slot[s];
        const int64_t K = (int64_t)e.S * Bp;
        if (e.kind == MCTSQ_ENC_GRU) {
            // Everything of this encoder arrives as per-workgroup partials of gru_bwd_kernel, reduced here in workgroup
            // order: W_ih [192][F] + b_ih: delta (r|z|n) x x_t;  W_hh rows r,z + b_hh[0:128]: delta (r|z) x h_{t-1};
            // W_hh rows n + b_hh[128:192]: (delta_n * r) x h_{t-1}
            const int nwg = (int)(Bp / TILE) * (gru_half[s == long_slot ? 0 : 1] ? 2 : 1);
            if (jbB.add_external(g.wp_ih - P.partials, G3, e.F, nwg, e.gru.wih, e.F, 1, e.gru.bih)) return -1;
            if (jbB.add_external(g.wp_rz - P.partials, 2 * E, E, nwg, e.gru.whh, E, 1, e.gru.bhh)) return -1;
            if (jbB.add_external(g.wp_n - P.partials, E, E, nwg, e.gru.whh + 2 * E * E, E, 1, e.gru.bhh + 2 * E))
                return -1;
            // attention score Linear(64,1): d(score) x h_t
            if (jbB.add_external(g.wp_aw - P.partials, 1, E, nwg, e.gru.aw, E, 1, e.gru.ab)) return -1;
            // LayerNorm weight: column sums of dfeat * xhat (a launch of its own for the long window's sums sat on the
            // critical path); bias: column sums of dfeat over the batch
            if (jbB.add_external(g.wp_ln - P.partials, E, 0, nwg, -1, 0, 0, e.gru.lnw)) return -1;
            if (addA(ws + P.dfeat, FEAT, s * E, nullptr, 0, 0, E, 0, Bp, -1, 0, 0, e.gru.lnb)) return -1;
        } else if (e.kind == MCTSQ_ENC_CONV) {
            // conv weights [out][in][3]: one contraction per tap against the time-shifted input
            // (zero blocks at both ends of the saved images implement the padding)
            for (int tap = 0; tap < 3; ++tap) {
                if (addB(ws + g.c_d2, E, 0, ws + g.c_h1 + (int64_t)tap * Bp * E, E, 0, E, E, K, e.conv.w2 + tap, 3 * E,
                         3, tap == 0 ? e.conv.b2 : -1)) return -1;
                if (addB(ws + g.c_d1, E, 0, ws + g.x + (int64_t)tap * Bp * 8, 8, 0, E, e.F, K, e.conv.w1 + tap,
                         3 * e.F, 3, tap == 0 ? e.conv.b1 : -1)) return -1;
            }
            if (addB(ws + g.dsc, 1, 0, ws + g.c_h2, E, 0, 1, E, K, e.conv.aw, E, 1, e.conv.ab)) return -1;
            if (addB(ws + g.p, E, 0, nullptr, 0, 0, E, 0, Bp, -1, 0, 0, e.conv.lnw)) return -1;
            if (addA(ws + P.dfeat, FEAT, s * E, nullptr, 0, 0, E, 0, Bp, -1, 0, 0, e.conv.lnb)) return -1;
        } else if (e.kind == MCTSQ_ENC_TRANSFORMER) {
            const int64_t T = g.T;
            const TfOff& o = e.tfm;
            for (int l = 0; l < 2; ++l) {
                const TfLayerOff& t = o.layer[l];
                const float* hin = l == 0 ? ws + g.t_e0 : ws + g.t_hout[0];
                if (addB(ws + g.t_dqkv[l], G3, 0, hin, E, 0, G3, E, T, t.inw, E, 1, t.inb)) return -1;
                if (addB(ws + g.t_dx1[l], E, 0, ws + g.t_ctx[l], E, 0, E, E, T, t.ow, E, 1, t.ob)) return -1;
                if (jbB.add_external(g.t_wp1[l] - P.partials, 2048, E, g.ntb, t.l1w, E, 1, t.l1b)) return -1;
                if (jbB.add_external(g.t_wp2[l] - P.partials, E, 2048, g.ntb, t.l2w, 2048, 1, -1)) return -1;
                if (addB(ws + g.t_dx2[l], E, 0, nullptr, 0, 0, E, 0, T, -1, 0, 0, t.l2b)) return -1;
                if (addB(ws + g.t_p1[l], E, 0, nullptr, 0, 0, E, 0, T, -1, 0, 0, t.n1w)) return -1;
                if (addB(ws + g.t_dy1[l], E, 0, nullptr, 0, 0, E, 0, T, -1, 0, 0, t.n1b)) return -1;
                if (addB(ws + g.t_p2[l], E, 0, nullptr, 0, 0, E, 0, T, -1, 0, 0, t.n2w)) return -1;
                if (addB(ws + g.t_dy2[l], E, 0, nullptr, 0, 0, E, 0, T, -1, 0, 0, t.n2b)) return -1;
            }
            // embedding Linear(F,64) and the positional table rows 0..S-1 (sum over samples)
            if (addB(ws + g.t_dh0, E, 0, ws + g.t_xe, 8, 0, E, e.F, T, o.ew, e.F, 1, o.eb)) return -1;
            if (addB(ws + g.t_dh0, (int64_t)e.S * E, 0, nullptr, 0, 0, e.S * E, 0, Bp, -1, 0, 0, o.pos)) return -1;
            if (addB(ws + g.t_pfin, E, 0, nullptr, 0, 0, E, 0, Bp, -1, 0, 0, o.lnw)) return -1;
            if (addA(ws + P.dfeat, FEAT, s * E, nullptr, 0, 0, E, 0, Bp, -1, 0, 0, o.lnb)) return -1;
        } else {    // mlp
            if (addB(ws + g.m_d0, E, 0, ws + g.m_x, 16, 0, E, 3 * e.F, Bp, e.mlp.w0, 3 * e.F, 1, e.mlp.b0)) return -1;
            if (addB(ws + g.m_d1, E, 0, ws + g.m_m0, E, 0, E, E, Bp, e.mlp.w1, E, 1, e.mlp.b1)) return -1;
        }
    }
    for (int i = 0; i < L.n_fc; ++i) {
        const bool last = i == L.n_fc - 1;
        const float* delta = last ? ws + P.dq : ws + P.dact[i];
        const int64_t ldd = last ? DQ_LD : L.fc_out[i];
        const float* in = i == 0 ? ws + P.feats_p : ws + P.act[i - 1];
        if (addA(delta, ldd, 0, in, L.fc_in[i], 0, L.fc_out[i], L.fc_in[i], Bp, L.fc_w[i], L.fc_in[i], 1, L.fc_b[i]))
            return -1;
    }
    MCTSQ_CHECK(part_pos <= P.partial_floats, "internal: partial buffer plan mismatch");
    // second helper lane: what needs nothing but the head backward - the head's weight-gradient products and bias /
    // LayerNorm-bias sums (15 us of latency-bound split-K launches), then the priority write-back.  On the first helper
    // lane they sat behind the short windows' backward and made that lane the longest of the three.
    MCTSQ_HIP(hipStreamWaitEvent(side->st2, side->head, 0));
    for (int i = 0; i < jbA.nlists; ++i)
        if (launch_wgrad(jbA.lists[i], ws, ws + P.partials, side->st2)) return -1;
    if (upd) {                                                // ONE event covers both jobs of the lane (every runtime call
        // is ~6-8 us of host time); the write-back is a single workgroup, so behind the products it is the only thing
        // of this lane that still runs when the weight-gradient launch of the caller's stream needs the CUs
        if (launch_per_update(upd->tree, upd->prio, upd->stamp, upd->cap, upd->head, upd->size, upd->idx, d_abs_td, B,
                              upd->alpha, upd->mass_cap, side->st2))
            return -1;
    }
    MCTSQ_HIP(hipEventRecord(side->join3, side->st2));
    // first helper lane: the short windows' backward, unless both groups went to the caller's stream together
    if (side_used) MCTSQ_HIP(hipEventRecord(side->join2, side->st));
    // caller's stream: everything else behind the encoder backward
    for (int i = 0; i < jbB.nlists; ++i)
        if (launch_wgrad(jbB.lists[i], ws, ws + P.partials, st)) return -1;
    if (side_used) MCTSQ_HIP(hipStreamWaitEvent(st, side->join2, 0));
    MCTSQ_HIP(hipStreamWaitEvent(st, side->join3, 0));
    // 7. ordered reduction of all partials into the flat gradient (+ the pieces of its norm, + the loss): the two
    // lanes' job lists are reduced by the same launches (40 jobs each)
    WJobList rl;
    memset(&rl, 0, sizeof(rl));
    int nparts = 0;
    bool first = true;
    auto flush = [&]() {
        if (rl.njobs == 0) return 0;
        ReduceExtras ex;
        memset(&ex, 0, sizeof(ex));
        if (d_sumsq_parts) ex.sumsq_parts = d_sumsq_parts + nparts;
        if (first && head_bf16) { ex.loss_parts = loss_parts; ex.n_loss_parts = (int)(Bp / TILE); ex.loss_norm = td.loss_norm; ex.loss_out = d_loss; }
        if (launch_wgrad_reduce(rl, ws + P.partials, d_grad, &ex, st)) return -1;
        nparts += wgrad_reduce_blocks(rl);
        first = false;
        rl.njobs = 0;
        return 0;
    };
    JobBuilder* both[2] = {&jbA, &jbB};
    for (int b = 0; b < 2; ++b)
        for (int i = 0; i < both[b]->nlists; ++i)
            for (int k = 0; k < both[b]->lists[i].njobs; ++k) {
                rl.job[rl.njobs++] = both[b]->lists[i].job[k];
                if (rl.njobs == MAX_WJOBS && flush()) return -1;
            }
    if (flush()) return -1;
    return d_sumsq_parts ? nparts : 0;
}

// ------------------------------------------------------------------------------------------------------------
// DQNModel.update() (dqn_per.py:401-476) as ONE call: sample -> batch assembly -> gradient -> clip / Adam / Polyak
// -> weight fragments for the next step -> priority write-back.  Same kernels and results as the separate entry
// points called in that order; what it saves is host time (one binding call and no per-step buffer management
// instead of five calls), which at the training batch size is what bounds the step otherwise.
namespace {
struct StepPlan {
    int64_t idx, phys, w, abs_td, sprice, sprocess, sgas, nprice, nprocess, ngas, act, rew, done, sumsq, grad_ws;
    int64_t targets, w64, draw, wmax;                        // data-parallel form: shard selection / raw weights
    int64_t total;                                           // bytes
};
int64_t take(int64_t& pos, int64_t bytes) {
    const int64_t at = pos;
    pos = (pos + bytes + 255) & ~int64_t(255);
    return at;
}
int make_step_plan(const mctsq_net_desc* net, const NetLayout& L, int64_t B, StepPlan* S) {
    int64_t pos = 0;
    S->idx = take(pos, B * 8); S->phys = take(pos, B * 8); S->w = take(pos, B * 4); S->abs_td = take(pos, B * 4);
    S->sprice = take(pos, B * L.enc[0].S * L.enc[0].Fd * 4); S->sprocess = take(pos, B * L.enc[1].S * L.enc[1].Fd * 4);
    S->sgas = take(pos, B * L.enc[2].S * L.enc[2].Fd * 4);
    S->nprice = take(pos, B * L.enc[0].S * L.enc[0].Fd * 4); S->nprocess = take(pos, B * L.enc[1].S * L.enc[1].Fd * 4);
    S->ngas = take(pos, B * L.enc[2].S * L.enc[2].Fd * 4);
    S->act = take(pos, B * 4); S->rew = take(pos, B * 4); S->done = take(pos, B * 4);
    S->targets = take(pos, B * 8); S->w64 = take(pos, B * 8); S->draw = take(pos, B * 4); S->wmax = take(pos, 8);
    const int64_t np = mctsq_dqn_grad_sumsq_parts(net);
    if (np < 0) return -1;
    S->sumsq = take(pos, np * 8);
    const int64_t gw = mctsq_dqn_grad_workspace_bytes(net, B);
    if (gw < 0) return -1;
    S->grad_ws = take(pos, gw);
    S->total = pos;
    return 0;
}
}  // namespace

extern "C" int64_t mctsq_per_dqn_step_workspace_bytes(const mctsq_net_desc* net, int64_t B) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    if (B < 1) { set_error("batch must be >= 1"); return -1; }
    StepPlan S;
    if (make_step_plan(net, L, B, &S)) return -1;
    return S.total;
}

extern "C" int mctsq_per_dqn_step(const mctsq_net_desc* net, const mctsq_step_args* a, void* stream) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    MCTSQ_CHECK(a != nullptr, "step arguments are NULL");
    MCTSQ_CHECK(a->B >= 1, "batch must be >= 1");
    MCTSQ_CHECK(L.enc[0].S == 25 && L.enc[0].Fd == 1 && L.enc[1].S == 10 && L.enc[1].Fd == 6 && L.enc[2].S == 3 &&
                    L.enc[2].Fd == 2, "the replay ring stores 25x1 / 10x6 / 3x2 windows (MCTSQ_STATE_ROW)");
    MCTSQ_CHECK(a->d_u && a->d_ws && a->d_loss && a->d_grad && a->d_m && a->d_v && a->d_params && a->d_target_params,
                "NULL device pointer");
    StepPlan S;
    if (make_step_plan(net, L, a->B, &S)) return -1;
    MCTSQ_CHECK(a->ws_bytes >= S.total, "workspace too small: %lld < %lld bytes", (long long)a->ws_bytes,
                (long long)S.total);
    MCTSQ_CHECK(((uintptr_t)a->d_ws & 255) == 0, "workspace must be 256-byte aligned");
    char* ws = (char*)a->d_ws;
    int64_t* d_idx = (int64_t*)(ws + S.idx); int64_t* d_phys = (int64_t*)(ws + S.phys);
    float* d_w = (float*)(ws + S.w); float* d_abs_td = (float*)(ws + S.abs_td);
    float* sp = (float*)(ws + S.sprice); float* sc = (float*)(ws + S.sprocess); float* sg = (float*)(ws + S.sgas);
    float* np_ = (float*)(ws + S.nprice); float* nc = (float*)(ws + S.nprocess); float* ng = (float*)(ws + S.ngas);
    int32_t* act = (int32_t*)(ws + S.act); float* rew = (float*)(ws + S.rew); float* done = (float*)(ws + S.done);
    double* sumsq = (double*)(ws + S.sumsq);
    // Deferred re-split (args->resplit_first): the fragments of the parameters as the previous step left them are
    // re-split by extra workgroups of the sampling launch (neither job needs the other), or, for networks without the
    // one-launch prepare, by their launches ahead of it.  (On a helper lane under the sampling it cost more than it
    // hid: the two cross-queue hand-offs took the step from 0.167 to 0.180 ms.)
    const bool resplit_first = a->resplit_first != 0;
    DqnPrepArgs pa;
    SamplePassengers extra{nullptr, a->d_grad, L.total};     // the gradient fill always rides along
    if (resplit_first) {
        MCTSQ_CHECK((a->d_prep_policy && a->d_prep_target) || prepared_bytes(L) == 0, "NULL prepared-weights pointer");
        MCTSQ_CHECK(a->d_prep_bwd || prep_bwd_bytes(L) == 0, "NULL backward prepared-weights pointer");
        MCTSQ_CHECK((((uintptr_t)a->d_prep_policy | (uintptr_t)a->d_prep_target | (uintptr_t)a->d_prep_bwd) & 15) == 0,
                    "prepared buffers must be 16-byte aligned");
        if (one_launch_prepare(L)) {
            fill_dqn_prep(L, a->d_params, a->d_target_params, a->d_prep_policy, a->d_prep_target, a->d_prep_bwd, &pa);
            extra.prep = &pa;
        } else if (mctsq_dqn_prepare(net, a->d_params, a->d_target_params, a->d_prep_policy, a->d_prep_target,
                                     a->d_prep_bwd, stream)) {
            return -1;
        }
    }
    // sample + IS weights + batch assembly (dqn_per.py:412-452)
    int rc = per_sample_batch_impl(a->d_tree, a->d_prio, a->capacity, a->head, a->size, a->d_u, a->B, a->alpha, a->beta,
                                   a->mass_capacity, d_idx, d_phys, d_w, a->d_wmax, a->d_states, a->d_next_states,
                                   a->d_actions, a->d_rewards, a->d_dones, sp, sc, sg, np_, nc, ng, act, rew, done,
                                   a->d_per_ws, a->per_ws_bytes, &extra, stream);
    if (rc) return rc;
    // loss, backward (:455-470); the priority write-back (:474) rides on the helper lane behind the TD errors
    MCTSQ_CHECK(a->d_tree && a->d_prio && a->d_stamp, "NULL device pointer");
    MCTSQ_CHECK(a->B < ((int64_t)1 << 31), "update batch too large");
    PerUpdateHook upd{(unsigned long long*)a->d_tree, a->d_prio, a->d_stamp, a->capacity, a->head, a->size, d_idx,
                      a->alpha, a->mass_capacity, true};
    DqnPrep pp{a->d_prep_policy, a->d_prep_target, a->d_prep_bwd};
    rc = dqn_grad_entry(net, &pp, a->d_params, a->d_target_params, sp, sc, sg, np_, nc, ng, act, rew, done, d_w,
                        (float)a->gamma, 1.0f, a->B, a->d_grad, a->d_loss, d_abs_td, nullptr, nullptr, sumsq,
                        ws + S.grad_ws, S.total - S.grad_ws, stream, &upd);
    if (rc < 0) return rc;
    const int64_t nparts = rc;
    // clip, Adam (:471-472) and, when tau >= 0, the target sync of mctsq_config_mcts_per.py:181 (dqn_per.py:510-516)
    rc = mctsq_clip_adam_polyak_ex(a->d_params, a->d_grad, a->d_m, a->d_v, a->d_target_params, L.total, a->adam_step,
                                   a->lr, a->beta1, a->beta2, a->eps, a->max_norm, a->tau, a->d_gnorm, sumsq, nparts,
                                   nullptr, 0, stream);
    if (rc) return rc;
    if (!resplit_first)
        rc = mctsq_dqn_prepare(net, a->d_params, a->d_target_params, a->d_prep_policy, a->d_prep_target, a->d_prep_bwd,
                               stream);
    // (the priority write-back ran first on the second helper lane, which the reduction joined)
    if (rc) return rc;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// The data-parallel form of the step (include/mctsq_hip.h): everything up to the flat gradient / everything behind the
// gradient all-reduce.  Same kernels as the one-call form; the shard selection is one more (single-workgroup) launch,
// the importance weights stay raw until the finish call has the global maximum.
namespace {
int check_dp_args(const NetLayout& L, const mctsq_step_args* a) {
    MCTSQ_CHECK(a != nullptr, "step arguments are NULL");
    MCTSQ_CHECK(L.enc[0].S == 25 && L.enc[0].Fd == 1 && L.enc[1].S == 10 && L.enc[1].Fd == 6 && L.enc[2].S == 3 &&
                    L.enc[2].Fd == 2, "the replay ring stores 25x1 / 10x6 / 3x2 windows (MCTSQ_STATE_ROW)");
    MCTSQ_CHECK(a->world >= 1 && a->world <= 64 && a->rank >= 0 && a->rank < a->world, "bad shard geometry (world %lld, rank %lld)",
                (long long)a->world, (long long)a->rank);
    MCTSQ_CHECK(a->B >= 1 && a->B < ((int64_t)1 << 31), "global draw count must be >= 1");
    MCTSQ_CHECK(a->bcap >= 16 && a->bcap % 16 == 0 && a->bcap <= 16384, "bcap must be a multiple of 16 in [16, 16384]");
    MCTSQ_CHECK(a->d_ws && a->d_loss && a->d_grad && a->d_m && a->d_v && a->d_params && a->d_target_params && a->d_exchange,
                "NULL device pointer");
    MCTSQ_CHECK(a->d_loss == a->d_grad + L.total && a->d_exchange == a->d_loss + 16,
                "d_grad | d_loss | d_exchange must be one contiguous block (total, 16, 2 * world floats)");
    MCTSQ_CHECK(((uintptr_t)a->d_ws & 255) == 0, "workspace must be 256-byte aligned");
    return 0;
}
}  // namespace

extern "C" int mctsq_per_dqn_step_begin(const mctsq_net_desc* net, const mctsq_step_args* a, void* stream) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    if (check_dp_args(L, a)) return -1;
    MCTSQ_CHECK(a->d_u && a->d_shard_mass && a->d_shard_size && a->d_count && a->d_tree && a->d_prio && a->d_stamp,
                "NULL device pointer");
    MCTSQ_CHECK(a->capacity >= 1 && a->size >= 0 && a->size <= a->capacity && a->head >= 0 && a->head < a->capacity &&
                    (a->size == a->capacity || a->head == 0), "bad ring geometry");
    MCTSQ_CHECK(a->mass_capacity >= a->capacity && a->mass_capacity <= (int64_t)1 << 36, "mass_capacity out of range");
    MCTSQ_CHECK(a->d_per_ws && a->per_ws_bytes >= PER_WS_CTRL_BYTES && ((uintptr_t)a->d_per_ws & 15) == 0,
                "sampling workspace too small or misaligned");
    const int64_t R = a->bcap;
    StepPlan S;
    if (make_step_plan(net, L, R, &S)) return -1;
    MCTSQ_CHECK(a->ws_bytes >= S.total, "workspace too small: %lld < %lld bytes", (long long)a->ws_bytes, (long long)S.total);
    char* ws = (char*)a->d_ws;
    hipStream_t st = (hipStream_t)stream;
    int64_t* d_idx = (int64_t*)(ws + S.idx); int64_t* d_phys = (int64_t*)(ws + S.phys);
    float* d_w = (float*)(ws + S.w); float* d_abs_td = (float*)(ws + S.abs_td);
    float* sp = (float*)(ws + S.sprice); float* sc = (float*)(ws + S.sprocess); float* sg = (float*)(ws + S.sgas);
    float* np_ = (float*)(ws + S.nprice); float* nc = (float*)(ws + S.nprocess); float* ng = (float*)(ws + S.ngas);
    int32_t* act = (int32_t*)(ws + S.act); float* rew = (float*)(ws + S.rew); float* done = (float*)(ws + S.done);
    unsigned long long* targets = (unsigned long long*)(ws + S.targets);
    double* w64 = (double*)(ws + S.w64);
    int32_t* draw = a->d_draw ? a->d_draw : (int32_t*)(ws + S.draw);
    double* wmax = a->d_wmax ? a->d_wmax : (double*)(ws + S.wmax);
    const bool resplit_first = a->resplit_first != 0;
    DqnPrepArgs pa;
    SamplePassengers extra{nullptr, a->d_grad, L.total};     // the gradient fill rides along (not the exchange slots:
    if (resplit_first) {                                     // the sampling launch's last workgroup writes all of them)
        MCTSQ_CHECK((a->d_prep_policy && a->d_prep_target) || prepared_bytes(L) == 0, "NULL prepared-weights pointer");
        MCTSQ_CHECK(a->d_prep_bwd || prep_bwd_bytes(L) == 0, "NULL backward prepared-weights pointer");
        MCTSQ_CHECK((((uintptr_t)a->d_prep_policy | (uintptr_t)a->d_prep_target | (uintptr_t)a->d_prep_bwd) & 15) == 0,
                    "prepared buffers must be 16-byte aligned");
        if (one_launch_prepare(L)) {
            fill_dqn_prep(L, a->d_params, a->d_target_params, a->d_prep_policy, a->d_prep_target, a->d_prep_bwd, &pa);
            extra.prep = &pa;
        } else if (mctsq_dqn_prepare(net, a->d_params, a->d_target_params, a->d_prep_policy, a->d_prep_target,
                                     a->d_prep_bwd, stream)) {
            return -1;
        }
    }
    // this rank's share of the global draws, compacted (draw order kept); rows behind d_count[0] are padding
    if (launch_per_shard_select(a->d_u, a->B, (const unsigned long long*)a->d_shard_mass, (int)a->world, (int)a->rank, R,
                                targets, draw, a->d_count, st))
        return -1;
    PerGatherArgs ga;
    memset(&ga, 0, sizeof(ga));
    MCTSQ_CHECK(a->d_states && a->d_next_states && a->d_actions && a->d_rewards && a->d_dones, "NULL device pointer");
    ga.states = a->d_states; ga.next_states = a->d_next_states; ga.actions = a->d_actions; ga.rewards = a->d_rewards;
    ga.dones = a->d_dones; ga.phys = d_phys; ga.B = R;
    ga.sprice = sp; ga.sprocess = sc; ga.sgas = sg; ga.nprice = np_; ga.nprocess = nc; ga.ngas = ng;
    ga.act = act; ga.rew = rew; ga.done = done;
    const ShardWeightOut raw{d_w, a->d_exchange, (int)a->world, (int)a->rank};
    if (launch_per_sample_shard((const unsigned long long*)a->d_tree, a->d_prio, a->capacity, a->head, a->size, targets,
                                a->d_count, R, a->alpha, a->beta, a->mass_capacity,
                                (const unsigned long long*)a->d_shard_mass, a->d_shard_size, (int)a->world, d_idx, d_phys,
                                w64, wmax, a->d_per_ws, &ga, st, &extra, &raw))
        return -1;
    PerUpdateHook upd{(unsigned long long*)a->d_tree, a->d_prio, a->d_stamp, a->capacity, a->head, a->size, d_idx,
                      a->alpha, a->mass_capacity, true};
    DqnPrep pp{a->d_prep_policy, a->d_prep_target, a->d_prep_bwd};
    const int rc = dqn_grad_entry(net, &pp, a->d_params, a->d_target_params, sp, sc, sg, np_, nc, ng, act, rew, done, d_w,
                                  (float)a->gamma, (float)((double)R / (double)a->B), R, a->d_grad, a->d_loss, d_abs_td,
                                  nullptr, nullptr, nullptr, ws + S.grad_ws, S.total - S.grad_ws, stream, &upd);
    return rc < 0 ? rc : 0;
}

extern "C" int mctsq_per_dqn_step_finish(const mctsq_net_desc* net, const mctsq_step_args* a, void* stream) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    if (check_dp_args(L, a)) return -1;
    StepPlan S;
    if (make_step_plan(net, L, a->bcap, &S)) return -1;
    MCTSQ_CHECK(a->ws_bytes >= S.total, "workspace too small: %lld < %lld bytes", (long long)a->ws_bytes, (long long)S.total);
    // the norm is that of the SUMMED gradient: recomputed here (redundantly per block for the small networks, from
    // partials in the step workspace otherwise); the reduction's own pieces describe this rank's share only
    int rc = clip_adam_polyak_impl(a->d_params, a->d_grad, a->d_m, a->d_v, a->d_target_params, L.total, a->adam_step, a->lr,
                                   a->beta1, a->beta2, a->eps, a->max_norm, a->tau, a->d_gnorm, nullptr, 0,
                                   (char*)a->d_ws + S.sumsq, mctsq_optim_workspace_bytes(), a->d_exchange, (int)a->world,
                                   a->d_grad, a->d_loss, stream);
    if (rc) return rc;
    if (!a->resplit_first)
        rc = mctsq_dqn_prepare(net, a->d_params, a->d_target_params, a->d_prep_policy, a->d_prep_target, a->d_prep_bwd,
                               stream);
    return rc;
}
