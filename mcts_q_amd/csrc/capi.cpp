// extern "C" surface of libmctsq_hip.so (see include/mctsq_hip.h).  Host-side argument checks
// live here: every kernel below assumes the shapes validated in this file.
#include <string.h>

#include "kernels.hpp"
#include "per.hpp"

using namespace mctsq;

#include "forward.hpp"

namespace mctsq {

static void fill_gru_slot(GruSlot& g, const EncOff& e, const float* params, const float* x, int slot,
                          const SlotSave* sv) {
    memset(&g, 0, sizeof(g));
    g.x = x;
    g.wih = params + e.gru.wih; g.whh = params + e.gru.whh;
    g.bih = params + e.gru.bih; g.bhh = params + e.gru.bhh;
    g.aw = params + e.gru.aw;   g.ab = params + e.gru.ab;
    g.lnw = params + e.gru.lnw; g.lnb = params + e.gru.lnb;
    g.S = e.S; g.Fd = e.Fd; g.F = e.F; g.tf = e.tf; g.out_col = slot * E;
    if (sv) {
        g.sv_h = sv->h; g.sv_r = sv->r; g.sv_z = sv->z; g.sv_n = sv->n; g.sv_hn = sv->hn;
        g.sv_score = sv->score; g.sv_stat = sv->stat; g.sv_pool = sv->pool; g.sv_x = sv->x;
        g.bw_wih = sv->wih; g.bw_wrz = sv->wrz; g.bw_wn = sv->wn; g.bw_dsc = sv->dsc; g.bw_p = sv->p; g.bw_aw = sv->aw; g.bw_ln = sv->ln;
    }
}

void fill_gru_args(GruFwdArgs& ga, const NetLayout& L, const float* params, const void* prep,
                   const float* const x[3], int64_t B, int64_t Bp, float* feats, const SlotSave* sv) {
    memset(&ga, 0, sizeof(ga));
    ga.B = B; ga.Bp = Bp; ga.feats = feats;
    for (int s = 0; s < 3; ++s)
        if (L.enc[s].kind == MCTSQ_ENC_GRU) {
            fill_gru_slot(ga.slot[ga.nslots], L.enc[s], params, x[s], s, sv ? &sv[s] : nullptr);
            ga.slot[ga.nslots].wprep = prep ? (const char*)prep + ga.nslots * GRU_PREP_BYTES : nullptr;
            ++ga.nslots;
        }
}

void fill_conv_args(ConvFwdArgs& ca, const NetLayout& L, const float* params, const void* prep,
                    const float* const x[3], int64_t B, int64_t Bp, float* feats, const SlotSave* sv) {
    memset(&ca, 0, sizeof(ca));
    ca.B = B; ca.Bp = Bp; ca.feats = feats;
    for (int s = 0; s < 2; ++s) {
        const EncOff& e = L.enc[s];
        if (e.kind != MCTSQ_ENC_CONV) continue;
        ConvSlot& c = ca.slot[ca.nslots++];
        c.x = x[s];
        c.w1 = params + e.conv.w1; c.b1 = params + e.conv.b1; c.w2 = params + e.conv.w2; c.b2 = params + e.conv.b2;
        c.aw = params + e.conv.aw; c.ab = params + e.conv.ab; c.lnw = params + e.conv.lnw; c.lnb = params + e.conv.lnb;
        c.S = e.S; c.Fd = e.Fd; c.F = e.F; c.tf = e.tf; c.out_col = s * E;
        c.wprep = prep ? (const char*)prep + gru_prepared_bytes(L) + (ca.nslots - 1) * CONV_PREP_BYTES : nullptr;
        if (sv) {
            const SlotSave& v = sv[s];
            c.sv_h1 = v.c_h1; c.sv_h2 = v.c_h2; c.sv_score = v.score; c.sv_stat = v.stat; c.sv_pool = v.pool;
            c.sv_x = v.x; c.bw_d2 = v.c_d2; c.bw_d1 = v.c_d1; c.bw_dsc = v.dsc; c.bw_p = v.p;
        }
    }
}

bool fill_mlp_args(MlpArgs& ma, const NetLayout& L, const float* params, const float* const x[3], int64_t B,
                   int64_t Bp, float* feats, const SlotSave* sv) {
    memset(&ma, 0, sizeof(ma));
    const EncOff& e = L.enc[2];
    if (e.kind != MCTSQ_ENC_MLP) return false;
    ma.x = x[2];
    ma.w0 = params + e.mlp.w0; ma.b0 = params + e.mlp.b0; ma.w1 = params + e.mlp.w1; ma.b1 = params + e.mlp.b1;
    ma.Fd = e.Fd; ma.F = e.F; ma.tf = e.tf; ma.out_col = 2 * E; ma.B = B; ma.Bp = Bp; ma.feats = feats;
    if (sv) { ma.sv_x = sv[2].m_x; ma.sv_m0 = sv[2].m_m0; ma.bw_d1 = sv[2].m_d1; ma.bw_d0 = sv[2].m_d0; }
    return true;
}

int64_t forward_ws_floats(const NetLayout& L, int64_t Bp) {
    int64_t n = Bp * FEAT;
    for (int s = 0; s < 2; ++s)
        if (L.enc[s].kind == MCTSQ_ENC_TRANSFORMER) n += 2 * tf_padded_tokens(Bp, L.enc[s].S) * E;
    return n + prepared_bytes(L) / (int64_t)sizeof(float);
}

// layout of the prepared buffer: the GRU slots' fragments first (slot order), then the conv slots', then 2 layers
// per transformer slot
int64_t gru_prepared_bytes(const NetLayout& L) {
    int n = 0;
    for (int s = 0; s < 3; ++s) n += L.enc[s].kind == MCTSQ_ENC_GRU;
    return n * GRU_PREP_BYTES;
}
static int64_t conv_prepared_bytes(const NetLayout& L) {
    int n = 0;
    for (int s = 0; s < 2; ++s) n += L.enc[s].kind == MCTSQ_ENC_CONV;
    return n * CONV_PREP_BYTES;
}
static int64_t tf_prepared_bytes(const NetLayout& L) {
    int n = 0;
    for (int s = 0; s < 3; ++s) n += L.enc[s].kind == MCTSQ_ENC_TRANSFORMER;
    return 2 * n * TF_LAYER_PREP_BYTES;
}
// the head shape the split-bf16 head serves (fused kernel and head_bf16_fwd_kernel; head_core.hpp)
static bool head_bf16_layout(const NetLayout& L) {
    return L.n_fc == 3 && L.fc_in[0] == FEAT && L.fc_out[0] == 128 && L.fc_in[1] == 128 && L.fc_out[1] == 128 &&
           L.fc_in[2] == 128 && L.A <= 16;
}
// ... then the head's fragments when the fused kernel serves the network
int64_t prepared_bytes(const NetLayout& L) {
    return gru_prepared_bytes(L) + conv_prepared_bytes(L) + tf_prepared_bytes(L) + (head_bf16_layout(L) ? HEAD_PREP_BYTES : 0);
}

int prepare_impl(const NetLayout& L, const float* params, void* prep, hipStream_t st) {
    const float* whh[3] = {nullptr, nullptr, nullptr};
    int n = 0;
    for (int s = 0; s < 3; ++s)
        if (L.enc[s].kind == MCTSQ_ENC_GRU) whh[n++] = params + L.enc[s].gru.whh;
    if (launch_gru_prepare(whh, n, prep, st)) return -1;
    char* p = (char*)prep + gru_prepared_bytes(L);
    {
        const float* w2[2] = {nullptr, nullptr};
        int nc = 0;
        for (int s = 0; s < 2; ++s)
            if (L.enc[s].kind == MCTSQ_ENC_CONV) w2[nc++] = params + L.enc[s].conv.w2;
        if (launch_conv_prepare(w2, nc, p, st)) return -1;
        p += conv_prepared_bytes(L);
    }
    for (int s = 0; s < 3; ++s) {
        if (L.enc[s].kind != MCTSQ_ENC_TRANSFORMER) continue;
        for (int l = 0; l < 2; ++l) {
            const TfLayerOff& o = L.enc[s].tfm.layer[l];
            TfLayerW w{};
            w.inw = params + o.inw; w.ow = params + o.ow; w.l1w = params + o.l1w; w.l1b = params + o.l1b;
            w.l2w = params + o.l2w; w.n1w = params + o.n1w; w.n1b = params + o.n1b;
            if (launch_tf_prepare(w, p, st)) return -1;
            p += TF_LAYER_PREP_BYTES;
        }
    }
    if (head_bf16_layout(L) &&
        launch_head_prepare(params + L.fc_w[0], params + L.fc_w[1], params + L.fc_w[2], L.A, p, st))
        return -1;
    return 0;
}

void fill_tf_args(TfArgs& t, const EncOff& e, const float* params, const float* x, int slot, int64_t B,
                  int64_t Bp, float* feats, float* hbuf, const SlotSave* sv) {
    memset(&t, 0, sizeof(t));
    t.x = x;
    t.pos = params + e.tfm.pos; t.ew = params + e.tfm.ew; t.eb = params + e.tfm.eb;
    for (int l = 0; l < 2; ++l) {
        const TfLayerOff& o = e.tfm.layer[l];
        TfLayerW& w = t.layer[l];
        w.inw = params + o.inw; w.inb = params + o.inb; w.ow = params + o.ow; w.ob = params + o.ob;
        w.l1w = params + o.l1w; w.l1b = params + o.l1b; w.l2w = params + o.l2w; w.l2b = params + o.l2b;
        w.n1w = params + o.n1w; w.n1b = params + o.n1b; w.n2w = params + o.n2w; w.n2b = params + o.n2b;
    }
    t.lnw = params + e.tfm.lnw; t.lnb = params + e.tfm.lnb;
    t.S = e.S; t.Fd = e.Fd; t.F = e.F; t.tf = e.tf; t.out_col = slot * E; t.NS = 64 / e.S;
    t.B = B; t.Bvalid = B; t.Bp = Bp; t.T = B * e.S; t.Tp = tf_padded_tokens(Bp, e.S);
    float *h0 = hbuf, *h1 = hbuf + t.Tp * E;                  // inference: ping-pong
    for (int l = 0; l < 2; ++l) { t.buf[l].hin = h0; t.buf[l].hmid = h1; t.buf[l].hout = h0; }
    t.feats = feats;
    if (sv) {                       // training: every stage keeps its output; padded samples are processed too
        t.buf[0] = sv->t_buf[0]; t.buf[1] = sv->t_buf[1];
        t.sv_mean = sv->t_mean; t.sv_xe = sv->t_xe;
        t.B = Bp; t.T = Bp * e.S;
    }
}

// TripleEncoderDQN.forward (dqn_per.py:232-236) + leaf epilogue.  `sv` != NULL keeps the
// activations the backward pass needs (training) and always takes the per-encoder kernels.
// head arguments of one forward (shared by forward_impl and forward_pair_impl)
static void fill_head_args(HeadArgs& ha, const NetLayout& L, const float* params, const void* prep, int64_t B, int64_t Bp,
                           float* feats, const HeadOut& out, const SlotSave* sv, float* const* sv_act) {
    memset(&ha, 0, sizeof(ha));
    ha.feats = feats; ha.n_fc = L.n_fc; ha.act = L.act; ha.A = L.A; ha.B = B; ha.Bp = Bp;
    for (int i = 0; i < L.n_fc; ++i) {
        ha.w[i] = params + L.fc_w[i]; ha.b[i] = params + L.fc_b[i];
        ha.in[i] = L.fc_in[i]; ha.out[i] = L.fc_out[i];
    }
    ha.q = out.q; ha.vmax = out.vmax; ha.argmax = out.argmax; ha.prior = out.prior;
    if (sv != nullptr)
        for (int i = 0; i + 1 < L.n_fc; ++i) ha.sv_act[i] = sv_act[i];
    if (head_bf16_layout(L) && prep)
        ha.wprep = (const char*)prep + gru_prepared_bytes(L) + conv_prepared_bytes(L) + tf_prepared_bytes(L);
}

// Policy forward on s (records kept) and target forward on s' as ONE launch.  Returns 1 (nothing enqueued) when the
// network is not the fused gru/gru/gru shape or a prepared-weights buffer is missing: the caller then issues the two
// forwards separately.
int forward_pair_impl(const NetLayout& L, const float* params, const void* prep, const float* const x[3], float* ws_p,
                      const HeadOut& out_p, const SlotSave* sv, float* const* sv_act, const float* tparams,
                      const void* tprep, const float* const xt[3], float* ws_t, const HeadOut& out_t, int64_t B,
                      int64_t Bp, hipStream_t st) {
    if (!prep || !tprep || !sv) return 1;
    GruFwdArgs g0, g1;
    fill_gru_args(g0, L, params, prep, x, B, Bp, ws_p, sv);
    fill_gru_args(g1, L, tparams, tprep, xt, B, Bp, ws_t, nullptr);
    HeadArgs h0, h1;
    fill_head_args(h0, L, params, prep, B, Bp, ws_p, out_p, sv, sv_act);
    fill_head_args(h1, L, tparams, tprep, B, Bp, ws_t, out_t, nullptr, nullptr);
    if (!fused_gru_head_applicable(g0, h0) || !fused_gru_head_applicable(g1, h1)) return 1;
    g1.feats = nullptr;
    return launch_gru3_head_fwd_pair(g0, h0, g1, h1, st) ? -1 : 0;
}

int forward_impl(const NetLayout& L, const float* params, const void* prep, const float* const x[3], int64_t B,
                 int64_t Bp, float* ws, const HeadOut& out, const SlotSave* sv, float* const* sv_act, hipStream_t st,
                 void* ev_encoders_done) {
    float* feats = ws;
    if (!prep && prepared_bytes(L) > 0) {                     // the workspace's tail holds the split weights
        void* own = ws + forward_ws_floats(L, Bp) - prepared_bytes(L) / (int64_t)sizeof(float);
        if (prepare_impl(L, params, own, st)) return -1;
        prep = own;
    }
    GruFwdArgs ga;
    fill_gru_args(ga, L, params, prep, x, B, Bp, feats, sv);
    ConvFwdArgs ca;
    fill_conv_args(ca, L, params, prep, x, B, Bp, feats, sv);
    MlpArgs ma;
    const bool has_mlp = fill_mlp_args(ma, L, params, x, B, Bp, feats, sv);
    const bool has_tf = L.enc[0].kind == MCTSQ_ENC_TRANSFORMER || L.enc[1].kind == MCTSQ_ENC_TRANSFORMER;
    (void)has_tf;
    HeadArgs ha;
    fill_head_args(ha, L, params, prep, B, Bp, feats, out, sv, sv_act);
    const bool save = sv != nullptr;

    if (fused_gru_head_applicable(ga, ha)) {                 // shipped default: one launch
        // (measured at the training batch, 64 tiles: the per-window kernels + head kernel take 38 + 9 us against 53
        // for this kernel's activation-saving variant, which spills registers at two waves per SIMD, but policy and
        // target forward then oversubscribe the CUs and the step as a whole gets slower: 0.221 vs 0.213 ms)
        if (!save) ga.feats = nullptr;                       // training keeps the feature rows for the head backward
        if (launch_gru3_head_fwd(ga, ha, save, st)) return -1;
        if (ev_encoders_done) MCTSQ_HIP(hipEventRecord((hipEvent_t)ev_encoders_done, st));
        return 0;
    }
    {
        float* hbuf = ws + Bp * FEAT;
        const char* tprep = (const char*)prep + gru_prepared_bytes(L) + conv_prepared_bytes(L);
        TfArgs t[2];
        int nt = 0;
        for (int s = 0; s < 2; ++s) {
            if (L.enc[s].kind != MCTSQ_ENC_TRANSFORMER) continue;
            fill_tf_args(t[nt], L.enc[s], params, x[s], s, B, Bp, feats, hbuf, sv ? &sv[s] : nullptr);
            for (int l = 0; l < 2; ++l) { t[nt].ffn_prep[l] = tprep; tprep += TF_LAYER_PREP_BYTES; }
            hbuf += 2 * t[nt].Tp * E;
            ++nt;
        }
        if (nt == 2 ? launch_tf_fwd_pair(t[0], t[1], st) : (nt == 1 ? launch_tf_fwd(t[0], st) : 0)) return -1;
    }
    // longest kernels first; the gas/EUA MLP rides in the conv encoders' launch when there is one
    const bool mlp_rides = has_mlp && ca.nslots > 0 && B > 0;
    if (launch_conv_fwd(ca, save, st, mlp_rides ? &ma : nullptr)) return -1;
    if (launch_gru_fwd(ga, save, st)) return -1;
    if (has_mlp && !mlp_rides && launch_mlp_fwd(ma, save, st)) return -1;
    if (ev_encoders_done) MCTSQ_HIP(hipEventRecord((hipEvent_t)ev_encoders_done, st));
    return launch_head_fwd(ha, st);
}

}  // namespace mctsq

extern "C" {

int mctsq_abi_version(void) { return MCTSQ_ABI_VERSION; }
const char* mctsq_last_error(void) { return mctsq::last_error(); }

int mctsq_param_layout(const mctsq_net_desc* net, int64_t* offs, int64_t* sizes, int cap, int64_t* total) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    for (int i = 0; i < L.n_tensors && i < cap; ++i) {
        if (offs) offs[i] = L.t_off[i];
        if (sizes) sizes[i] = L.t_size[i];
    }
    if (total) *total = L.total;
    return L.n_tensors;
}

int mctsq_time_features(int mode, int seq_len, float* out) {
    MCTSQ_CHECK(mode == MCTSQ_TF_MINUTE || mode == MCTSQ_TF_GAS_HOURS, "bad time feature mode %d", mode);
    MCTSQ_CHECK(seq_len >= 1 && seq_len <= MAX_SEQ && out, "bad seq_len / out");
    host_time_features(mode, seq_len, out);
    return 0;
}

int64_t mctsq_q_forward_workspace_bytes(const mctsq_net_desc* net, int64_t B) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    if (B < 0) { set_error("negative batch"); return -1; }
    const int64_t Bp = round_up(B > 0 ? B : 1, TILE);
    return forward_ws_floats(L, Bp) * (int64_t)sizeof(float);
}

int mctsq_q_forward(const mctsq_net_desc* net, const float* d_params, const float* d_price,
                    const float* d_process, const float* d_gas, int64_t B, float* d_q, float* d_vmax,
                    int32_t* d_argmax, float* d_prior, void* d_ws, int64_t ws_bytes, void* stream) {
    return mctsq_q_forward_ex(net, d_params, d_price, d_process, d_gas, B, d_q, d_vmax, d_argmax, d_prior, d_ws,
                              ws_bytes, stream, nullptr);
}

int mctsq_q_forward_ex(const mctsq_net_desc* net, const float* d_params, const float* d_price,
                       const float* d_process, const float* d_gas, int64_t B, float* d_q, float* d_vmax,
                       int32_t* d_argmax, float* d_prior, void* d_ws, int64_t ws_bytes, void* stream,
                       void* ev_encoders_done) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_params && d_price && d_process && d_gas && d_q, "NULL device pointer");
    const int64_t Bp = round_up(B, TILE);
    MCTSQ_CHECK(d_ws && ws_bytes >= forward_ws_floats(L, Bp) * (int64_t)sizeof(float),
                "workspace too small (%lld bytes)", (long long)ws_bytes);
    MCTSQ_CHECK(((uintptr_t)d_ws & 15) == 0, "workspace must be 16-byte aligned");
    const float* x[3] = {d_price, d_process, d_gas};
    HeadOut out{d_q, d_vmax, d_argmax, d_prior};
    return forward_impl(L, d_params, nullptr, x, B, Bp, (float*)d_ws, out, nullptr, nullptr, (hipStream_t)stream,
                        ev_encoders_done);
}

int64_t mctsq_q_prepared_bytes(const mctsq_net_desc* net) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    return prepared_bytes(L);
}

int mctsq_q_prepare(const mctsq_net_desc* net, const float* d_params, void* d_prep, int64_t prep_bytes, void* stream) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    if (prepared_bytes(L) == 0) return 0;
    MCTSQ_CHECK(d_params && d_prep && prep_bytes >= prepared_bytes(L), "NULL pointer or prepared buffer too small");
    MCTSQ_CHECK(((uintptr_t)d_prep & 15) == 0, "prepared buffer must be 16-byte aligned");
    return prepare_impl(L, d_params, d_prep, (hipStream_t)stream);
}

int mctsq_q_forward_prepared(const mctsq_net_desc* net, const float* d_params, const void* d_prep,
                             const float* d_price, const float* d_process, const float* d_gas, int64_t B, float* d_q,
                             float* d_vmax, int32_t* d_argmax, float* d_prior, void* d_ws, int64_t ws_bytes,
                             void* stream, void* ev_encoders_done) {
    NetLayout L;
    if (build_layout(net, &L)) return -1;
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_params && d_price && d_process && d_gas && d_q, "NULL device pointer");
    MCTSQ_CHECK(d_prep || prepared_bytes(L) == 0, "NULL prepared-weights pointer");
    MCTSQ_CHECK(((uintptr_t)d_prep & 15) == 0, "prepared buffer must be 16-byte aligned");
    const int64_t Bp = round_up(B, TILE);
    MCTSQ_CHECK(d_ws && ws_bytes >= forward_ws_floats(L, Bp) * (int64_t)sizeof(float),
                "workspace too small (%lld bytes)", (long long)ws_bytes);
    MCTSQ_CHECK(((uintptr_t)d_ws & 15) == 0, "workspace must be 16-byte aligned");
    const float* x[3] = {d_price, d_process, d_gas};
    HeadOut out{d_q, d_vmax, d_argmax, d_prior};
    return forward_impl(L, d_params, d_prep, x, B, Bp, (float*)d_ws, out, nullptr, nullptr, (hipStream_t)stream,
                        ev_encoders_done);
}

// ------------------------------------------------------------------------------- PER
int64_t mctsq_per_tree_words(int64_t capacity) {
    PerTree T;
    if (per_tree_desc(capacity, &T)) return -1;
    return T.words;
}
int mctsq_per_mass_shift(int64_t capacity) { return per_mass_shift(capacity); }

int64_t mctsq_per_workspace_bytes(int64_t capacity, int64_t B) {
    (void)capacity;
    // control words + one raw fp64 weight per draw; never less than the 1024 partials of mctsq_per_refresh_max
    const int64_t a = PER_WS_CTRL_BYTES + (B > 0 ? B : 1) * 8, b = PER_WS_CTRL_BYTES + 1024 * 8;
    return a > b ? a : b;
}

static int check_ring(int64_t capacity, int64_t head, int64_t size) {
    MCTSQ_CHECK(capacity >= 1, "capacity must be >= 1");
    MCTSQ_CHECK(size >= 0 && size <= capacity, "size %lld outside [0, capacity]", (long long)size);
    MCTSQ_CHECK(head >= 0 && head < capacity, "head %lld outside [0, capacity)", (long long)head);
    MCTSQ_CHECK(size == capacity || head == 0, "head must be 0 until the ring is full");
    return 0;
}

static int check_mass_cap(int64_t capacity, int64_t mass_capacity) {
    MCTSQ_CHECK(mass_capacity >= capacity && mass_capacity <= (int64_t)1 << 36,
                "mass_capacity %lld must be >= capacity and <= 2^36", (long long)mass_capacity);
    return 0;
}

int mctsq_per_push(uint64_t* d_tree, double* d_prio, const double* d_maxprio, int64_t capacity, int64_t head,
                   int64_t size, int64_t K, double alpha, int64_t mass_capacity, void* stream) {
    if (check_ring(capacity, head, size) || check_mass_cap(capacity, mass_capacity)) return -1;
    MCTSQ_CHECK(d_tree && d_prio && d_maxprio, "NULL device pointer");
    MCTSQ_CHECK(K >= 0 && K <= capacity, "push count %lld outside [0, capacity]", (long long)K);
    if (K == 0) return 0;
    return launch_per_push((unsigned long long*)d_tree, d_prio, d_maxprio, capacity, head, size, K, alpha,
                           mass_capacity, (hipStream_t)stream);
}

int mctsq_per_refresh_max(const double* d_prio, int64_t capacity, int64_t head, int64_t size, double* d_maxprio,
                          void* d_ws, int64_t ws_bytes, void* stream) {
    if (check_ring(capacity, head, size)) return -1;
    MCTSQ_CHECK(d_prio && d_maxprio && d_ws && ws_bytes >= PER_WS_CTRL_BYTES + 1024 * 8,
                "NULL pointer or workspace smaller than mctsq_per_workspace_bytes(capacity, 1)");
    // live entries are physical [0, size) (ring not full) or all of [0, capacity); the partials sit behind the
    // control words of the sampling workspace, which this call leaves alone
    return launch_per_refresh_max(d_prio, size, d_maxprio, (double*)((char*)d_ws + PER_WS_CTRL_BYTES), 1024,
                                  (hipStream_t)stream);
}

static int fill_gather(PerGatherArgs& a, const float* d_states, const float* d_next_states, const int32_t* d_actions,
                       const float* d_rewards, const float* d_dones, const int64_t* d_phys, int64_t B, float* d_sprice,
                       float* d_sprocess, float* d_sgas, float* d_nprice, float* d_nprocess, float* d_ngas,
                       int32_t* d_act, float* d_rew, float* d_done) {
    MCTSQ_CHECK(d_states && d_next_states && d_actions && d_rewards && d_dones && d_phys && d_sprice &&
                    d_sprocess && d_sgas && d_nprice && d_nprocess && d_ngas && d_act && d_rew && d_done,
                "NULL device pointer");
    a.states = d_states; a.next_states = d_next_states; a.actions = d_actions; a.rewards = d_rewards;
    a.dones = d_dones; a.phys = d_phys; a.B = B;
    a.sprice = d_sprice; a.sprocess = d_sprocess; a.sgas = d_sgas;
    a.nprice = d_nprice; a.nprocess = d_nprocess; a.ngas = d_ngas;
    a.act = d_act; a.rew = d_rew; a.done = d_done;
    return 0;
}

static int check_sample_ws(const void* d_ws, int64_t ws_bytes, int64_t B) {
    MCTSQ_CHECK(d_ws && ws_bytes >= PER_WS_CTRL_BYTES + B * 8, "workspace too small");
    MCTSQ_CHECK(((uintptr_t)d_ws & 15) == 0, "workspace must be 16-byte aligned");
    return 0;
}

int mctsq_per_sample(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head, int64_t size,
                     const double* d_u, int64_t B, double alpha, double beta, int64_t mass_capacity,
                     int64_t* d_idx, int64_t* d_phys, float* d_w, double* d_wmax, void* d_ws, int64_t ws_bytes,
                     void* stream) {
    if (check_ring(capacity, head, size) || check_mass_cap(capacity, mass_capacity)) return -1;
    MCTSQ_CHECK(size >= 1, "cannot sample from an empty buffer");
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_tree && d_prio && d_u && d_idx && d_phys && d_w, "NULL device pointer");
    if (check_sample_ws(d_ws, ws_bytes, B)) return -1;
    return launch_per_sample((const unsigned long long*)d_tree, d_prio, capacity, head, size, d_u, B, alpha,
                             beta, mass_capacity, d_idx, d_phys, d_w, d_wmax, d_ws, nullptr, (hipStream_t)stream);
}

int mctsq_per_sample_batch(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head, int64_t size,
                           const double* d_u, int64_t B, double alpha, double beta, int64_t mass_capacity,
                           int64_t* d_idx, int64_t* d_phys, float* d_w, double* d_wmax, const float* d_states,
                           const float* d_next_states, const int32_t* d_actions, const float* d_rewards,
                           const float* d_dones, float* d_sprice, float* d_sprocess, float* d_sgas, float* d_nprice,
                           float* d_nprocess, float* d_ngas, int32_t* d_act, float* d_rew, float* d_done, void* d_ws,
                           int64_t ws_bytes, void* stream) {
    return mctsq::per_sample_batch_impl(d_tree, d_prio, capacity, head, size, d_u, B, alpha, beta, mass_capacity, d_idx,
                                        d_phys, d_w, d_wmax, d_states, d_next_states, d_actions, d_rewards, d_dones,
                                        d_sprice, d_sprocess, d_sgas, d_nprice, d_nprocess, d_ngas, d_act, d_rew, d_done,
                                        d_ws, ws_bytes, nullptr, stream);
}

// extra != NULL: the launch also carries the learner step's passengers (per.hpp: SamplePassengers)
extern "C++" int mctsq::per_sample_batch_impl(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head,
                                 int64_t size, const double* d_u, int64_t B, double alpha, double beta,
                                 int64_t mass_capacity, int64_t* d_idx, int64_t* d_phys, float* d_w, double* d_wmax,
                                 const float* d_states, const float* d_next_states, const int32_t* d_actions,
                                 const float* d_rewards, const float* d_dones, float* d_sprice, float* d_sprocess,
                                 float* d_sgas, float* d_nprice, float* d_nprocess, float* d_ngas, int32_t* d_act,
                                 float* d_rew, float* d_done, void* d_ws, int64_t ws_bytes,
                                 const SamplePassengers* extra, void* stream) {
    if (check_ring(capacity, head, size) || check_mass_cap(capacity, mass_capacity)) return -1;
    MCTSQ_CHECK(size >= 1, "cannot sample from an empty buffer");
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_tree && d_prio && d_u && d_idx && d_phys && d_w, "NULL device pointer");
    if (check_sample_ws(d_ws, ws_bytes, B)) return -1;
    PerGatherArgs ga;
    if (fill_gather(ga, d_states, d_next_states, d_actions, d_rewards, d_dones, d_phys, B, d_sprice, d_sprocess, d_sgas,
                    d_nprice, d_nprocess, d_ngas, d_act, d_rew, d_done)) return -1;
    return launch_per_sample((const unsigned long long*)d_tree, d_prio, capacity, head, size, d_u, B, alpha,
                             beta, mass_capacity, d_idx, d_phys, d_w, d_wmax, d_ws, &ga, (hipStream_t)stream, extra);
}

int mctsq_per_shard_select(const double* d_u, int64_t B, const uint64_t* d_shard_mass, int world, int rank,
                           int64_t bcap, uint64_t* d_targets, int32_t* d_draw, int32_t* d_count, void* stream) {
    MCTSQ_CHECK(B >= 0 && B < ((int64_t)1 << 31) && world >= 1 && world <= 64 && rank >= 0 && rank < world,
                "bad shard geometry");
    MCTSQ_CHECK(bcap >= 1, "per-rank draw capacity must be >= 1");
    MCTSQ_CHECK((B == 0 || d_u) && d_shard_mass && d_targets && d_draw && d_count, "NULL device pointer");
    return launch_per_shard_select(d_u, B, (const unsigned long long*)d_shard_mass, world, rank, bcap,
                                   (unsigned long long*)d_targets, d_draw, d_count, (hipStream_t)stream);
}

int mctsq_per_sample_shard(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head, int64_t size,
                           const uint64_t* d_targets, const int32_t* d_count, int64_t bcap, double alpha, double beta,
                           int64_t mass_capacity, const uint64_t* d_shard_mass, const int64_t* d_shard_size, int world,
                           int64_t* d_idx, int64_t* d_phys, double* d_w64, double* d_wmax, const float* d_states,
                           const float* d_next_states, const int32_t* d_actions, const float* d_rewards,
                           const float* d_dones, float* d_sprice, float* d_sprocess, float* d_sgas, float* d_nprice,
                           float* d_nprocess, float* d_ngas, int32_t* d_act, float* d_rew, float* d_done, void* d_ws,
                           int64_t ws_bytes, void* stream) {
    if (check_ring(capacity, head, size) || check_mass_cap(capacity, mass_capacity)) return -1;
    MCTSQ_CHECK(bcap >= 1 && world >= 1 && world <= 64, "bad shard geometry");
    MCTSQ_CHECK(d_tree && d_prio && d_targets && d_count && d_shard_mass && d_shard_size && d_idx && d_phys && d_w64 &&
                    d_wmax, "NULL device pointer");
    if (check_sample_ws(d_ws, ws_bytes, 0)) return -1;
    PerGatherArgs ga;
    if (fill_gather(ga, d_states, d_next_states, d_actions, d_rewards, d_dones, d_phys, bcap, d_sprice, d_sprocess,
                    d_sgas, d_nprice, d_nprocess, d_ngas, d_act, d_rew, d_done)) return -1;
    return launch_per_sample_shard((const unsigned long long*)d_tree, d_prio, capacity, head, size,
                                   (const unsigned long long*)d_targets, d_count, bcap, alpha, beta, mass_capacity,
                                   (const unsigned long long*)d_shard_mass, d_shard_size, world, d_idx, d_phys, d_w64,
                                   d_wmax, d_ws, &ga, (hipStream_t)stream);
}

int mctsq_per_total_mass(const uint64_t* d_tree, int64_t capacity, int64_t size, uint64_t* d_mass, void* stream) {
    MCTSQ_CHECK(d_tree && d_mass, "NULL device pointer");
    MCTSQ_CHECK(size >= 0 && size <= capacity, "size %lld outside [0, capacity]", (long long)size);
    return launch_per_total_mass((const unsigned long long*)d_tree, capacity, size, (unsigned long long*)d_mass,
                                 (hipStream_t)stream);
}

int mctsq_per_shard_targets(const double* d_u, int64_t B, const uint64_t* d_shard_mass, int world, int rank,
                            uint64_t* d_targets, int32_t* d_mine, void* stream) {
    MCTSQ_CHECK(B >= 0 && world >= 1 && world <= 64 && rank >= 0 && rank < world, "bad shard geometry");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_u && d_shard_mass && d_targets && d_mine, "NULL device pointer");
    return launch_per_shard_targets(d_u, B, (const unsigned long long*)d_shard_mass, world, rank,
                                    (unsigned long long*)d_targets, d_mine, (hipStream_t)stream);
}

int mctsq_per_sample_at(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head, int64_t size,
                        const uint64_t* d_targets, int64_t B, double alpha, double beta, int64_t mass_capacity,
                        double n_total, double mass_total, int64_t* d_idx, int64_t* d_phys, double* d_w64,
                        double* d_wmax, void* stream) {
    if (check_ring(capacity, head, size) || check_mass_cap(capacity, mass_capacity)) return -1;
    MCTSQ_CHECK(size >= 1, "cannot sample from an empty buffer");
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_tree && d_prio && d_targets && d_idx && d_phys && d_w64, "NULL device pointer");
    return launch_per_sample_at((const unsigned long long*)d_tree, d_prio, capacity, head, size,
                                (const unsigned long long*)d_targets, B, alpha, beta, mass_capacity, n_total,
                                mass_total, d_idx, d_phys, d_w64, d_wmax, (hipStream_t)stream);
}

int mctsq_per_normalize_weights(const double* d_w64, int64_t B, const double* d_wmax, float* d_w, void* stream) {
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_w64 && d_wmax && d_w, "NULL device pointer");
    return launch_per_normalize(d_w64, B, d_wmax, d_w, (hipStream_t)stream);
}

int mctsq_per_update(uint64_t* d_tree, double* d_prio, int32_t* d_stamp, int64_t capacity, int64_t head,
                     int64_t size, const int64_t* d_idx, const float* d_newprio, int64_t B, double alpha,
                     int64_t mass_capacity, void* stream) {
    if (check_ring(capacity, head, size) || check_mass_cap(capacity, mass_capacity)) return -1;
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_tree && d_prio && d_stamp && d_idx && d_newprio, "NULL device pointer");
    MCTSQ_CHECK(B < ((int64_t)1 << 31), "update batch too large");
    return launch_per_update((unsigned long long*)d_tree, d_prio, d_stamp, capacity, head, size, d_idx, d_newprio, B,
                             alpha, mass_capacity, (hipStream_t)stream);
}

int mctsq_per_gather(const float* d_states, const float* d_next_states, const int32_t* d_actions,
                     const float* d_rewards, const float* d_dones, const int64_t* d_phys, int64_t B,
                     float* d_sprice, float* d_sprocess, float* d_sgas, float* d_nprice, float* d_nprocess,
                     float* d_ngas, int32_t* d_act, float* d_rew, float* d_done, void* stream) {
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    PerGatherArgs a;
    if (fill_gather(a, d_states, d_next_states, d_actions, d_rewards, d_dones, d_phys, B, d_sprice, d_sprocess, d_sgas,
                    d_nprice, d_nprocess, d_ngas, d_act, d_rew, d_done)) return -1;
    return launch_per_gather(a, (hipStream_t)stream);
}

}  // extern "C"


// ------------------------------------------------------------------------------- one encoder alone
// ConvAttentionEnc / GRUAttentionEnc / TransformerEnc / GasEUAEncoder as stand-alone modules (dqn_per.py:21-130).
// The encoder is placed in a slot of an otherwise minimal network layout (slot 0, or slot 2 for the gas/EUA MLP
// whose Linear(3 F) is only defined there); its own tensors are then a contiguous, 16-float aligned run of that
// layout, so a buffer that holds just them is addressed as `params - base`.  Only that slot's kernels are launched.
namespace {
struct SoloEnc {
    NetLayout L;
    int slot;
    int first, count;          // tensors of the slot inside L.t_off / L.t_size
    int64_t base, total;       // floats
};
int solo_layout(const mctsq_encoder_desc* enc, SoloEnc* S) {
    MCTSQ_CHECK(enc != nullptr, "encoder descriptor is NULL");
    mctsq_net_desc nd;
    memset(&nd, 0, sizeof(nd));
    const mctsq_encoder_desc dummy = {MCTSQ_ENC_GRU, 1, 1, MCTSQ_TF_NONE};
    S->slot = enc->kind == MCTSQ_ENC_MLP ? 2 : 0;
    for (int s = 0; s < 3; ++s) nd.enc[s] = s == S->slot ? *enc : dummy;
    nd.embed_dim = E; nd.hidden_layers = 1; nd.hidden_units = 16; nd.action_dim = 1; nd.activation = MCTSQ_ACT_RELU;
    if (build_layout(&nd, &S->L)) return -1;
    const int per_kind = enc->kind == MCTSQ_ENC_TRANSFORMER ? 29 : enc->kind == MCTSQ_ENC_MLP ? 4 : 8;
    S->first = S->slot == 0 ? 0 : 16;                     // two dummy GRU slots of 8 tensors precede slot 2
    S->count = per_kind;
    S->base = S->L.t_off[S->first];
    S->total = S->L.t_off[S->first + per_kind] - S->base; // the next tensor (dummy slot / head) starts 16-aligned
    return 0;
}
int64_t solo_prep_bytes(int kind) {
    return kind == MCTSQ_ENC_GRU ? GRU_PREP_BYTES : kind == MCTSQ_ENC_CONV ? CONV_PREP_BYTES
         : kind == MCTSQ_ENC_TRANSFORMER ? 2 * TF_LAYER_PREP_BYTES : 0;
}
int64_t solo_ws_floats(const SoloEnc& S, int64_t Bp) {
    const EncOff& e = S.L.enc[S.slot];
    int64_t n = Bp * FEAT;
    if (e.kind == MCTSQ_ENC_TRANSFORMER) n += 2 * tf_padded_tokens(Bp, e.S) * E;
    return round_up(n, 64) + solo_prep_bytes(e.kind) / (int64_t)sizeof(float);
}
}  // namespace

extern "C" {

int mctsq_encoder_param_layout(const mctsq_encoder_desc* enc, int64_t* offs, int64_t* sizes, int cap, int64_t* total) {
    SoloEnc S;
    if (solo_layout(enc, &S)) return -1;
    for (int i = 0; i < S.count && i < cap; ++i) {
        if (offs) offs[i] = S.L.t_off[S.first + i] - S.base;
        if (sizes) sizes[i] = S.L.t_size[S.first + i];
    }
    if (total) *total = S.total;
    return S.count;
}

int64_t mctsq_encoder_workspace_bytes(const mctsq_encoder_desc* enc, int64_t B) {
    SoloEnc S;
    if (solo_layout(enc, &S)) return -1;
    if (B < 0) { set_error("negative batch"); return -1; }
    return solo_ws_floats(S, round_up(B > 0 ? B : 1, TILE)) * (int64_t)sizeof(float);
}

int mctsq_encoder_forward(const mctsq_encoder_desc* enc, const float* d_enc_params, const float* d_x, int64_t B,
                          float* d_out, void* d_ws, int64_t ws_bytes, void* stream) {
    SoloEnc S;
    if (solo_layout(enc, &S)) return -1;
    MCTSQ_CHECK(B >= 0, "negative batch");
    if (B == 0) return 0;
    MCTSQ_CHECK(d_enc_params && d_x && d_out, "NULL device pointer");
    const int64_t Bp = round_up(B, TILE);
    MCTSQ_CHECK(d_ws && ws_bytes >= solo_ws_floats(S, Bp) * (int64_t)sizeof(float), "workspace too small (%lld bytes)",
                (long long)ws_bytes);
    MCTSQ_CHECK(((uintptr_t)d_ws & 15) == 0 && ((uintptr_t)d_enc_params & 15) == 0,
                "workspace and parameters must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const NetLayout& L = S.L;
    const EncOff& e = L.enc[S.slot];
    const float* params = d_enc_params - S.base;
    float* ws = (float*)d_ws;
    float* feats = ws;
    float* hbuf = ws + Bp * FEAT;
    char* prep = (char*)(ws + solo_ws_floats(S, Bp)) - solo_prep_bytes(e.kind);
    const float* x[3] = {nullptr, nullptr, nullptr};
    x[S.slot] = d_x;
    if (e.kind == MCTSQ_ENC_GRU) {
        const float* whh[3] = {params + e.gru.whh, nullptr, nullptr};
        if (launch_gru_prepare(whh, 1, prep, st)) return -1;
        GruFwdArgs ga;
        fill_gru_args(ga, L, params, nullptr, x, B, Bp, feats, nullptr);   // fills all three GRU slots ...
        ga.nslots = 1;                                                     // ... of which only the first is real
        ga.slot[0].wprep = prep;
        if (launch_gru_fwd(ga, false, st)) return -1;
    } else if (e.kind == MCTSQ_ENC_CONV) {
        const float* w2[2] = {params + e.conv.w2, nullptr};
        if (launch_conv_prepare(w2, 1, prep, st)) return -1;
        ConvFwdArgs ca;
        fill_conv_args(ca, L, params, nullptr, x, B, Bp, feats, nullptr);
        ca.slot[0].wprep = prep;
        if (launch_conv_fwd(ca, false, st)) return -1;
    } else if (e.kind == MCTSQ_ENC_TRANSFORMER) {
        TfArgs t;
        fill_tf_args(t, e, params, d_x, 0, B, Bp, feats, hbuf, nullptr);
        for (int l = 0; l < 2; ++l) {
            char* p = prep + l * TF_LAYER_PREP_BYTES;
            if (launch_tf_prepare(t.layer[l], p, st)) return -1;
            t.ffn_prep[l] = p;
        }
        if (launch_tf_fwd(t, st)) return -1;
    } else {
        MlpArgs ma;
        if (!fill_mlp_args(ma, L, params, x, B, Bp, feats, nullptr)) { set_error("internal: mlp slot"); return -1; }
        if (launch_mlp_fwd(ma, false, st)) return -1;
    }
    MCTSQ_HIP(hipMemcpy2DAsync(d_out, E * sizeof(float), feats + S.slot * E, FEAT * sizeof(float), E * sizeof(float),
                               (size_t)B, hipMemcpyDeviceToDevice, st));
    return 0;
}

}  // extern "C"
